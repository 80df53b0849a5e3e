#!/usr/bin/env python3
"""Executes INTEGRATION.md: the reference's OWN tutorial script (examples/plot_scale/svat_tutorial/svat.py, unmodified, through its
click command line `-b hip`) runs on the reference's own driver (`RogerSetup.setup / step / run`, roger/roger.py) with the four edits
of INTEGRATION.md sections 1-4 applied IN MEMORY, its process routines bound to the C ABI's entry points, and must reproduce the
golden trajectory the NumPy backend produced for the same script (tests/golden/svat_tutorial.npz).

Build container only: needs /root/reference, which never travels and which this script does not modify (nothing of it is copied
either: the edits are applied to the module source at import time, or as attribute assignments afterwards).  Without a GPU the
entry points are served by the tests' device double (tests/oracle_context.OracleContext, the oracle behind the ABI's method names);
with `--device gpu` on a box that has both the reference and a GPU they are `roger_amd._native.Context`'s.

    python tests/integration/reference_hip_backend.py [--days N]        (N: compare the first N days only, default 365)

What is applied, by section of INTEGRATION.md:
  1  roger/backend.py:3 BACKENDS + get_backend_module; roger/core/operators.py:150-186 third branch (the NumPy operators: user hooks
     and user-defined kernels run on host mirrors); roger/cli/roger_run_base.py:6 `-b hip`.
  2  the ctypes stub: here `roger_amd._native` (or the double) with the same method names.
  3  roger/state.py: RogerState.initialize_variables creates the device context for the local chunk; RogerVariables.__setattr__
     marks variables host-dirty; they are uploaded before the next native call, what a native call wrote is downloaded after it.
  4  roger/routines.py / roger/roger.py: the routines `RogerSetup.setup` and `RogerSetup.step` call (roger.py:311-318, 396-457)
     are replaced one to one by entry points (table in INTEGRATION.md section 4); `calc_parameters_surface_kernel`, which the
     script's `set_parameters` hook calls itself, likewise.  `step()` itself runs as it stands.

Third-party modules the reference imports and this image lacks are replaced by stand-ins without model arithmetic: loguru (a logger
that swallows every call), h5netcdf (an in-memory container: what `write_forcing` stores is what the script's hooks read back),
cftime.date2num (hours since an origin, for the Time axis nobody reads)."""
import argparse
import importlib.abc
import importlib.util
import os
import re
import runpy
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
REPO = os.path.dirname(TESTS)
REFERENCE = "/root/reference"
TUTORIAL = os.path.join(REFERENCE, "examples", "plot_scale", "svat_tutorial", "svat.py")
for p in (REPO, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)


# ------------------------------------------------------------------------------------------------------------------------------
# stand-ins for absent third-party modules (no model arithmetic)
# ------------------------------------------------------------------------------------------------------------------------------
class _MemVar:
    def __init__(self, name, dims, shape, dtype):
        self.name, self.dimensions, self.attrs = name, tuple(dims), {}
        self.data = np.zeros(shape, dtype=dtype)

    shape = property(lambda self: self.data.shape)
    dtype = property(lambda self: self.data.dtype)

    def __setitem__(self, key, val):
        self.data[key] = val

    def __getitem__(self, key):
        return self.data[key]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.data, dtype=dtype)

    def __len__(self):
        return len(self.data)


class _MemFile:
    """h5netcdf.File, in memory, keyed by path: the calls roger/tools/setup.py:469-640 (write) and the setup scripts (read) make."""
    STORE = {}

    def __init__(self, path, mode="r", **kw):
        self.path = str(path)
        if mode == "w":
            self.STORE[self.path] = dict(attrs={}, dims={}, vars={})
        if self.path not in self.STORE:
            raise FileNotFoundError(self.path)
        self._rec = self.STORE[self.path]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    attrs = property(lambda self: self._rec["attrs"])
    variables = property(lambda self: self._rec["vars"])

    @property
    def dimensions(self):
        return self._rec["dims"]

    @dimensions.setter
    def dimensions(self, d):
        self._rec["dims"].update(d)

    def create_variable(self, name, dims, dtype, data=None, **kw):
        v = _MemVar(name, dims, tuple(int(self._rec["dims"][d]) for d in dims), np.dtype(dtype))
        if data is not None:
            v.data[...] = data
        self._rec["vars"][name] = v
        return v


def install_standins():
    class _Null:
        def __getattr__(self, name):
            def f(*a, **k):
                return self
            return f

    lg = types.ModuleType("loguru")
    lg.logger = _Null()
    sys.modules["loguru"] = lg
    h5 = types.ModuleType("h5netcdf")
    h5.File = _MemFile
    sys.modules["h5netcdf"] = h5
    cf = types.ModuleType("cftime")

    def date2num(dates, units, calendar="standard"):
        import pandas as pd

        origin = pd.Timestamp(units.split("since", 1)[1].strip())
        return np.array([(pd.Timestamp(d) - origin).total_seconds() / 3600.0 for d in dates])

    cf.date2num = date2num
    sys.modules["cftime"] = cf


# ------------------------------------------------------------------------------------------------------------------------------
# section 1: source edits applied at import time
# ------------------------------------------------------------------------------------------------------------------------------
SOURCE_EDITS = {
    "roger.backend": [
        ('BACKENDS = ("numpy", "jax")', 'BACKENDS = ("numpy", "jax", "hip")'),
        ('    elif backend_name == "numpy":\n        import numpy as backend_module',
         '    elif backend_name == "numpy":\n        import numpy as backend_module\n\n'
         '    elif backend_name == "hip":\n        import numpy as backend_module   # user hooks work on host mirrors; the kernels run natively'),
    ],
    "roger.core.operators": [
        ('if runtime_settings.backend == "numpy":', 'if runtime_settings.backend in ("numpy", "hip"):   # hip: the NumPy operators on the host mirrors'),
    ],
    "roger.core.utilities": [   # :162-166, the same choice for one helper
        ('if runtime_settings.backend == "numpy":\n    linear_regression = linear_regression_numpy',
         'if runtime_settings.backend in ("numpy", "hip"):\n    linear_regression = linear_regression_numpy'),
    ],
    "roger.cli.roger_run_base": [
        ('type=click.Choice(["numpy", "jax"])', 'type=click.Choice(["numpy", "jax", "hip"])'),
    ],
}


class _EditingLoader(importlib.abc.SourceLoader):
    def __init__(self, fullname, path):
        self.fullname, self.path = fullname, path

    def get_filename(self, fullname):
        return self.path

    def get_data(self, path):
        src = open(path, "rb").read()
        if path == self.path:
            txt = src.decode()
            for old, new in SOURCE_EDITS[self.fullname]:
                if txt.count(old) != 1:
                    raise RuntimeError(f"{self.fullname}: the text of the edit is not found exactly once: {old!r}")
                txt = txt.replace(old, new)
            return txt.encode()
        return src

    def source_to_code(self, data, path, *, _optimize=-1):
        return compile(data, path, "exec", dont_inherit=True, optimize=_optimize)

    def path_stats(self, path):
        raise OSError   # never use cached bytecode of the unedited source

    def set_data(self, path, data):
        pass            # and never write bytecode into the reference tree


class _EditingFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname not in SOURCE_EDITS:
            return None
        file = os.path.join(REFERENCE, *fullname.split(".")) + ".py"
        return importlib.util.spec_from_file_location(fullname, file, loader=_EditingLoader(fullname, file))


# ------------------------------------------------------------------------------------------------------------------------------
# sections 2-4: the binding
# ------------------------------------------------------------------------------------------------------------------------------
SCALARS = ("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")
SCALARS2 = ("event_id", "year", "month", "doy")


def written_planes():
    """Planes each per-routine entry point writes: the generated store sets of the device code (roger_amd/csrc/rh_sets.inc)."""
    txt = open(os.path.join(REPO, "roger_amd", "csrc", "rh_sets.inc")).read()
    sets = {m.group(1): re.findall(r"X\((\w+)\)", m.group(2)) for m in re.finditer(r"#define RH_SET_STORE_(\w+)\(X\)(.*)", txt)}
    return sets


class Binding:
    def __init__(self, device):
        self.device = device
        self.dirty = set()
        self.ctx = None
        self.stores = written_planes()
        self.native_calls = 0

    # -- section 3 ------------------------------------------------------------------------------------------------------------
    def create_context(self, state):
        from roger import runtime_settings as rs

        s = state.settings
        nx, ny = s.nx // rs.num_proc[0], s.ny // rs.num_proc[1]
        consts = {k: getattr(s, k) for k in ("r_mp", "l_sc", "sf", "ta_fm", "rmax", "hpi", "end_event", "atol", "rtol", "transp_water_stress",
                                               "clay_min", "clay_max", "theta_rew_min", "theta_rew_max", "rew_min", "rew_max", "z_evap_max",
                                               "zroot_to_zsoil_max", "a_bc", "b_bc")}
        if self.device == "gpu":
            from roger_amd import _native

            self.ctx = _native.Context(nx, ny, **consts)
        else:
            import oracle_context

            self.ctx = oracle_context.OracleContext(nx, ny, **{k: v for k, v in consts.items()})
        self.planes = {nm: is_int for nm, is_int in self.ctx.planes}
        self.all_dirty(state)

    def all_dirty(self, state):
        active = state.variables.active_vars()
        self.active = {nm for nm in self.planes if (nm[:-3] if nm.endswith("_m1") else nm) in active}   # (oneD planes are not, in SVAT)
        self.dirty = {nm[:-3] if nm.endswith("_m1") else nm for nm in self.active}

    def plane_of(self, vs, name, level):
        a = np.asarray(getattr(vs, name))
        return a[2:-2, 2:-2, level] if a.ndim == 3 else a[2:-2, 2:-2]

    def to_device(self, state):
        vs = state.variables
        for var in sorted(self.dirty):
            for nm, lvl in ((var, 1), (var + "_m1", 0)):
                if nm in self.active and (lvl == 1 or np.asarray(getattr(vs, var)).ndim == 3):
                    self.ctx.upload(nm, np.ascontiguousarray(self.plane_of(vs, var, lvl)))
        self.dirty.clear()
        s = self.ctx.get_scalars()
        for k in SCALARS:
            setattr(s, k, float(np.asarray(getattr(vs, k))) if k == "dt" else int(np.asarray(getattr(vs, k))))
        for k in SCALARS2:
            a = np.asarray(getattr(vs, k))
            getattr(s, k)[0], getattr(s, k)[1] = int(a[0]), int(a[1])
        self.ctx.set_scalars(s)

    def from_device(self, state, planes):
        vs = state.variables
        with vs.unlock():
            by_var = {}
            for nm in planes:
                if nm not in self.active:
                    continue
                var, lvl = (nm[:-3], 0) if nm.endswith("_m1") else (nm, 1)
                by_var.setdefault(var, []).append((nm, lvl))
            for var, lst in by_var.items():
                a = np.array(getattr(vs, var))
                for nm, lvl in lst:
                    vals = self.ctx.download(nm).reshape(a.shape[0] - 4, a.shape[1] - 4)
                    if a.ndim == 3:
                        a[2:-2, 2:-2, lvl] = vals
                    else:
                        a[2:-2, 2:-2] = vals
                setattr(vs, var, a)
                self.dirty.discard(var)
            s = self.ctx.get_scalars()
            for k in SCALARS:
                setattr(vs, k, getattr(s, k))
            for k in SCALARS2:
                setattr(vs, k, np.array([getattr(s, k)[0], getattr(s, k)[1]]))
            for k in SCALARS + SCALARS2:
                self.dirty.discard(k)
        return s

    # -- section 4 ------------------------------------------------------------------------------------------------------------
    def native(self, state, entries, store_sets):
        self.to_device(state)
        for e in entries:
            self.ctx.call(e)
            self.native_calls += 1
        planes = []
        for st in store_sets:
            planes += self.stores[st]
        return self.from_device(state, planes)


def bind(device):
    """State now; the routine dispatch once the reference has imported roger.core (which locks the runtime settings the script's
    command line sets first, roger/core/__init__.py:42-44): RogerSetup.__init__ does that import (roger/roger.py:36-45)."""
    import roger
    from roger import state as rstate

    B = Binding(device)
    orig_setup_init = roger.RogerSetup.__init__

    def setup_init(self, *a, **k):
        orig_setup_init(self, *a, **k)
        if not getattr(B, "routines_bound", False):
            bind_routines(B)
            B.routines_bound = True

    roger.RogerSetup.__init__ = setup_init

    # section 3: the arena is created with the variables; assignments mark variables host-dirty
    orig_init = rstate.RogerState.initialize_variables

    def initialize_variables(self):
        orig_init(self)
        B.create_context(self)

    rstate.RogerState.initialize_variables = initialize_variables
    orig_setattr = rstate.RogerVariables.__setattr__

    def tracking_setattr(self, key, val):
        if not key.startswith("_"):
            B.dirty.add(key)
        return orig_setattr(self, key, val)

    rstate.RogerVariables.__setattr__ = tracking_setattr
    return B, roger


def bind_routines(B):
    from roger import KernelOutput, roger_routine
    from roger.core import (adaptive_time_stepping, capillary_rise, evapotranspiration, infiltration, interception, numerics, root_zone, snow,
                            soil, subsoil, subsurface_runoff, surface)

    # section 4: routine -> entry point (the table of INTEGRATION.md)
    def routine(entries, store_sets, before=None):
        @roger_routine
        def hip_routine(state):
            if before:
                before(state)
            B.native(state, entries, store_sets)
        return hip_routine

    noop = roger_routine(lambda state: None)

    def hand_over_forcing(state):   # what set_forcing assigned to vs.prec_day / ta_day / pet_day (svat.py:283-309)
        vs = state.variables
        day = [np.asarray(getattr(vs, k))[2:-2, 2:-2, :].reshape(-1, 144) for k in ("prec_day", "ta_day", "pet_day")]
        if all((d == d[:1]).all() for d in day):
            B.ctx.set_forcing_day(*[np.ascontiguousarray(d[0]) for d in day])
        else:
            B.ctx.set_forcing_day(*[np.ascontiguousarray(d) for d in day])

    def luts(state):
        vs = state.variables
        B.ctx.set_luts(np.asarray(vs.lut_ilu), np.asarray(vs.lut_gc), np.asarray(vs.lut_gcm), np.asarray(vs.lut_rdlu))

    adaptive_time_stepping.adaptive_time_stepping = routine(["rh_adaptive_dt"], ["rt_select_prec_ta", "rt_select_pet"], before=hand_over_forcing)
    interception.calculate_interception = routine(["rh_interception"], ["rt_interception"])
    evapotranspiration.calculate_evapotranspiration = routine(["rh_evapotranspiration"], ["rt_evapotranspiration"])
    snow.calculate_snow = routine(["rh_snow"], ["rt_snow"])
    infiltration.calculate_infiltration = routine(["rh_infiltration"], ["rt_infiltration"])
    subsurface_runoff.calculate_subsurface_runoff = routine(["rh_subsurface_runoff"], ["rt_subsurface_runoff"])
    capillary_rise.calculate_capillary_rise = routine(["rh_capillary_rise"], ["rt_capillary_rise"])
    # surface / root zone / subsoil / soil storages and numerics.calc_storage are ONE entry point
    surface.calculate_surface = noop
    root_zone.calculate_root_zone = noop
    subsoil.calculate_subsoil = noop
    soil.calculate_soil = noop
    numerics.calc_storage = routine(["rh_storage"], ["rt_storage"])

    def sanity_check(state):   # numerics.sanity_check (:727-1391) + calculate_num_error (:715-724): one entry point, the flag comes back
        s = B.native(state, ["rh_num_error"], ["rt_num_error"])
        return bool(s.sanity_ok)

    numerics.sanity_check = sanity_check
    numerics.calculate_num_error = noop
    surface.calculate_parameters = routine(["rh_topo", "rh_params_surface"], ["rt_topo", "rt_params_surface"], before=luts)
    soil.calculate_parameters = routine(["rh_params_soil"], ["rt_params_soil"])
    surface.calculate_initial_conditions = noop
    soil.calculate_initial_conditions = routine(["rh_initial_conditions"], ["rt_initial_conditions"])

    def calc_parameters_surface_kernel(state):   # called by the script's set_parameters hook: vs.update(calc_parameters_surface_kernel(state))
        B.native(state, ["rh_params_surface"], ["rt_params_surface"])
        return KernelOutput()

    surface.calc_parameters_surface_kernel = calc_parameters_surface_kernel


# ------------------------------------------------------------------------------------------------------------------------------
def run(days=365, device="cpu", verbose=True):
    from golden_util import compare, load_case

    install_standins()
    sys.meta_path.insert(0, _EditingFinder())
    sys.path.insert(0, REFERENCE)
    os.environ["ROGER_DISKLESS_MODE"] = "1"   # no output files: the reference's own runtime switch (roger/runtime.py:121, :133)
    B, roger = bind(device)

    g, names, _ = load_case("svat_tutorial")
    scal_ref = g["scal"]
    nsteps_ref = int(g["nsteps"])
    stop_time = days * 86400
    record = dict(scal=[], snaps={}, steps=0)
    SC = ("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")
    SC2 = ("event_id", "year", "month", "doy")

    class _Enough(Exception):
        pass

    orig_step = roger.RogerSetup.__dict__["step"]   # the routine object itself (a descriptor)

    def recording_step(self, state):   # instrumentation around the reference's step(), not a change of it
        orig_step.__get__(self, type(self))(state)
        vs = state.variables
        row = [float(np.asarray(getattr(vs, k))) for k in SC]
        for k in SC2:
            a = np.asarray(getattr(vs, k))
            row += [float(a[0]), float(a[1])]
        record["scal"].append(row)
        record["steps"] += 1
        key = f"s{record['steps']:05d}"
        if key in g.files:
            snap = []
            for nm in names:
                var, lvl = (nm[:-3], 0) if nm.endswith("_m1") else (nm, 1)
                try:
                    a = np.asarray(getattr(vs, var))
                    snap.append(np.asarray(a[2:-2, 2:-2, lvl] if a.ndim == 3 else a[2:-2, 2:-2], dtype=np.float64).ravel())
                except RuntimeError:
                    snap.append(np.zeros(1))
            record["snaps"][key] = np.stack(snap)
        if float(np.asarray(vs.time)) >= stop_time:
            raise _Enough

    roger.RogerSetup.step = recording_step
    argv = sys.argv
    sys.argv = [TUTORIAL, "-b", "hip", "--float-type", "float64"]
    try:
        runpy.run_path(TUTORIAL, run_name="__main__")
    except _Enough:
        pass
    except SystemExit as e:
        if e.code not in (0, None):
            raise
    finally:
        sys.argv = argv
    from roger import runtime_settings

    assert runtime_settings.backend == "hip", runtime_settings.backend
    n = record["steps"]
    assert n > 0 and B.native_calls >= 9 * n, (n, B.native_calls)
    got = np.array(record["scal"])
    np.testing.assert_array_equal(got, scal_ref[:n], err_msg="time-stepping scalars differ from the NumPy backend's run of the same script")
    for key, snap in sorted(record["snaps"].items()):
        compare(snap, g[key], names, what=f"reference driver + hip backend, {key}")
    if days >= 365:
        assert n == nsteps_ref, (n, nsteps_ref)
    if verbose:
        print(f"ok: {TUTORIAL} ran unmodified under backend='hip' ({'GPU' if device == 'gpu' else 'device double'}): {n} steps, "
              f"{B.native_calls} native calls, scalars identical at every step, {len(record['snaps'])} state snapshots within the golden tolerance")
    return n


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--days", type=int, default=365)
    ap.add_argument("--device", choices=("cpu", "gpu"), default="cpu")
    a = ap.parse_args()
    if not os.path.isdir(REFERENCE):
        raise SystemExit("this script runs in the build container only: /root/reference is absent")
    run(a.days, a.device)
