"""Output diagnostics of the hip backend: the reference's `state.diagnostics` interface (roger/diagnostics/api.py, base.py,
rate.py, collect.py, average.py) on top of the device-side accumulators (include/roger_hip.h, rh_diag_*).

A setup script configures them in `set_diagnostics` exactly as with the reference,

    diagnostics = state.diagnostics
    diagnostics["rate"].output_variables = ["prec", "aet", "q_ss"]
    diagnostics["rate"].output_frequency = 24 * 60 * 60
    diagnostics["rate"].sampling_frequency = 1
    diagnostics["collect"].output_variables = ["S_rz", "S_ss"] ...

and gets `<identifier>.rate.nc`, `<identifier>.collect.nc`, `<identifier>.average.nc` with the reference's layout
(roger/io_tools/netcdf.py): dimensions x, y and the unlimited Time, variables transposed to (Time, y, x), `Time` in days
with `time_origin`, record 0 holding the initial values, one record per output interval.  What differs: the accumulation
runs on the device (`rate += var` after every step is a 24-byte-per-cell kernel, nothing is downloaded between outputs);
one output interval -- a day, an hour or ten minutes -- serves all diagnostics (the accumulators are indexed by the
interval of a step's start); the offline transport model,
whose step is a day, is read after every step instead (`output_transport`); the files are netCDF-4 like the reference's
(dimension scales, unlimited Time, one chunk per record), written by `roger_amd.nc4lite` because neither h5py nor h5netcdf is part
of this environment; long names and units come from a short table here, not from the reference's variable registry.
"""
import datetime
import os

import numpy as np

from . import runtime_settings as rs

DAY = 24 * 60 * 60

# units of what setup scripts usually write (fluxes are per time step of the model, as in the reference)
_UNITS = {
    **{k: "mm/dt" for k in ("prec", "pet", "aet", "evap_soil", "transp", "inf_mat_rz", "inf_mp_rz", "inf_sc_rz", "inf_ss",
                            "inf_mat", "inf_mp", "inf_sc", "q_rz", "q_ss", "q_sur", "q_hof", "q_sof", "q_sub", "q_sub_rz",
                            "q_sub_ss", "cpr_rz", "q_snow", "int_prec", "evap_int", "evap_dep", "rain_ground")},
    **{k: "mm" for k in ("S", "S_s", "S_rz", "S_ss", "S_sur", "S_snow", "S_dep", "S_int_top", "S_int_ground", "swe", "swe_top",
                         "S_pwp_rz", "S_pwp_ss", "S_sat_rz", "S_sat_ss", "z_sat", "z0", "dS", "dS_num_error")},
    **{k: "-" for k in ("theta", "theta_rz", "theta_ss", "ground_cover", "k_stress_transp", "k_stress_evap")},
    "ta": "degC", "z_root": "mm", "z_soil": "mm",
}


class RogerDiagnostic:
    """Attribute interface of roger/diagnostics/base.py."""

    name = None
    sampling_frequency = 0.0
    output_frequency = 0.0
    base_output_path = None
    native = False

    def __init__(self, state):
        self.output_variables = []
        self.output_path = "{identifier}." + self.name + ".nc"
        self._records = {}     # variable -> list of (ny, nx) arrays
        self._times = []       # days
        self._path = None

    @property
    def active(self):
        return bool(self.output_variables) and bool(self.output_frequency or self.sampling_frequency)

    def get_output_file_name(self, state):
        from . import runtime_state

        name = self.output_path.format(identifier=state.settings.identifier)
        if runtime_state.proc_num > 1:   # one file per rank with the rank's own block of columns (the reference writes one
            name = name[:-3] + f".{runtime_state.proc_rank:04d}.nc"   # file through parallel HDF5, which is not available here)
        return os.path.join(self.base_output_path, name) if self.base_output_path else name


class Rate(RogerDiagnostic):
    """Sum of every step of the output interval (roger/diagnostics/rate.py:66-84)."""

    name, native = "rate", True


class Collect(RogerDiagnostic):
    """Value at the end of the output interval (roger/diagnostics/collect.py)."""

    name, native = "collect", True


class Average(RogerDiagnostic):
    """Mean over the steps of the output interval (roger/diagnostics/average.py): the day's sum divided by rh_diag_steps."""

    name, native = "average", True


def _unsupported(diag_name):
    return type(diag_name.title().replace("_", ""), (RogerDiagnostic,), {"name": diag_name})


def create_default_diagnostics(state):
    """roger/diagnostics/api.py:4-20: the same keys; rate, collect and average are native."""
    kinds = [Average, _unsupported("snapshot"), Collect, _unsupported("constant"), Rate, _unsupported("minimum"),
             _unsupported("maximum"), _unsupported("tracer_monitor"), _unsupported("water_monitor")]
    return {k.name: k(state) for k in kinds}


def _interior(state, name):
    """(ny, nx) array of time level tau -- the transposed layout the reference writes (netcdf.py:121-142)."""
    a = np.asarray(getattr(state.variables, name))[2:-2, 2:-2]
    if a.ndim == 3:
        a = a[:, :, 1]
    return np.ascontiguousarray(a.T, dtype=np.float64)


def initialize(state):
    """roger/diagnostics/api.py:23-30 + base.initialize_output: register the variables with the device-side accumulators
    and write record 0 (the initial values, as the reference does at the end of setup())."""
    diags = state.diagnostics
    for d in diags.values():
        if d.active and not d.native:
            raise NotImplementedError(f'diagnostic "{d.name}" is outside the hip backend (native: rate, collect, average)')
    active = [d for d in diags.values() if d.active]
    state._diag_active = active
    if not active:
        return
    if state.settings.enable_offline_transport:
        return _initialize_transport(state, active)
    ctx = state.backend_context
    freqs = {int(d.output_frequency) for d in active}
    if len(freqs) != 1 or not freqs <= {DAY, 60 * 60, 10 * 60}:
        raise NotImplementedError(f"output frequencies {sorted(freqs)}: the device-side accumulators serve ONE output interval "
                                  "of a day, an hour or ten minutes (the step classes) for all diagnostics")
    state._diag_interval = freqs.pop()
    for d in active:
        for v in d.output_variables:
            meta = state.var_meta.get(v)
            if meta is None or meta.plane is None or meta.dtype is not None:
                raise NotImplementedError(f'diagnostic "{d.name}": {v!r} is not a float64 (x, y) variable of the device arena')
    rate = list(dict.fromkeys(diags["rate"].output_variables + diags["average"].output_variables))
    collect = list(dict.fromkeys(diags["collect"].output_variables))
    nint = int(state.settings.runlen // state._diag_interval) + 2
    per_slot = (len(rate) + len(collect)) * ctx.n * 8
    state._diag_slots = max(2, min(nint, int((1 << 31) // max(per_slot, 1))))   # at most 2 GiB of resident intervals
    state.variables.flush_to_device()
    ctx.diag_configure(rate=rate, collect=collect, n_slots=state._diag_slots)
    if state._diag_interval != DAY:
        ctx.diag_set_interval(state._diag_interval)
    state._diag_written_day = 0      # index of the last output interval that was looked at
    for d in active:
        d._records = {v: [_interior(state, v)] for v in d.output_variables}   # record 0: initial values
        d._times = [0.0]
        d._path = None if rs.diskless_mode else d.get_output_file_name(state)
        if d._path and os.path.isfile(d._path) and not getattr(rs, "force_overwrite", False):
            raise IOError(f'output file {d._path} for diagnostic "{d.name}" exists (change the output path, enable the '
                          "force_overwrite runtime setting or delete it)")
        _write(state, d)


# ---- offline transport: one step = one day = one output interval, nothing to accumulate ----------------------------
_AGED = ("sa_rz", "sa_ss", "msa_rz", "msa_ss")


def _initialize_transport(state, active):
    """The transport model writes after every (daily) step once the warm-up is done (write_output,
    roger/core/transport.py:3399-3418): rate, collect and average of one step are that step's values.  They are read from
    the device after the step.  The reference writes BEFORE it ages the storages, the native step includes the ageing:
    the age-resolved storages themselves (sa_rz, sa_ss, msa_rz, msa_ss) are therefore not offered; sa_s / msa_s, the
    distributions, concentrations and age statistics are."""
    for d in active:
        if d.output_frequency != DAY:
            raise NotImplementedError(f'diagnostic "{d.name}": output_frequency {d.output_frequency} s (the transport step is a day)')
        for v in d.output_variables:
            meta = state.var_meta.get(v)
            if meta is None or meta.dims is None or meta.dims[:2] != ("x", "y"):
                raise NotImplementedError(f'diagnostic "{d.name}": {v!r} is not an (x, y, ...) variable of the transport model')
            if v in _AGED:
                raise NotImplementedError(f'diagnostic "{d.name}": {v!r} would be written after the ageing, the reference writes it '
                                          "before (use sa_s / msa_s, or read vs.<name> from the setup script)")
    state._diag_transport = True
    for d in active:
        d._records, d._times = {v: [] for v in d.output_variables}, []
        d._path = None if rs.diskless_mode else d.get_output_file_name(state)
        if d._path and os.path.isfile(d._path) and not getattr(rs, "force_overwrite", False):
            raise IOError(f'output file {d._path} for diagnostic "{d.name}" exists (change the output path, enable the '
                          "force_overwrite runtime setting or delete it)")


def _transport_record(state, name):
    """(y, x) or (ages, y, x) of time level tau: the reference's transposed layout."""
    a = np.asarray(getattr(state.variables, name))[2:-2, 2:-2]
    meta = state.var_meta[name]
    if "timesteps" in meta.dims:
        a = a[:, :, 1]
    return np.ascontiguousarray(a.T, dtype=np.float64)


def output_transport(state):
    """After the warm-up (record 0: initial values, roger/roger.py:515-521) and after every step of the run proper."""
    if not getattr(state, "_diag_transport", False) or not state.settings.warmup_done:
        return
    for d in state._diag_active:
        for v in d.output_variables:
            d._records[v].append(_transport_record(state, v))
        d._times.append(float(state.variables.time) / DAY)
        small = sum(len(r) * r[0].nbytes for r in d._records.values()) < (32 << 20)
        if small or len(d._times) % 30 == 0:
            _write(state, d)


def output(state, final=False):
    """roger/diagnostics/api.py:47-70 for daily output: every completed day that has not been written yet becomes a
    record.  Called after each step of run() (one completed day at most) and after run_device() (all of them)."""
    active = getattr(state, "_diag_active", None)
    if not active or getattr(state, "_diag_transport", False):
        return
    ctx = state.backend_context
    iv = state._diag_interval
    now_days = int(state.variables.time // iv)        # completed output intervals
    first = state._diag_written_day + 1
    if now_days < first:
        return
    if now_days - first + 1 > state._diag_slots:
        raise RuntimeError(f"{now_days - first + 1} output intervals completed since the last output but only {state._diag_slots} "
                           "are resident on the device: call run_device() in shorter pieces")
    for k in range(first, now_days + 1):
        slot = (k - 1) % state._diag_slots
        t0, t1 = (k - 1) * iv, k * iv
        if iv != DAY:   # an interval inside a longer step was never started: the reference writes at the END of that step
            t0_slot, t1 = ctx.diag_slot_times(slot)
            if t0_slot != t0:
                continue
        steps = None
        for d in active:
            for v in d.output_variables:
                a = ctx.diag_download(v, slot).reshape(state.settings.nx // rs.num_proc[0], state.settings.ny // rs.num_proc[1])
                if d.name == "average":
                    steps = steps or ctx.diag_steps(slot)
                    a = a / steps
                d._records[v].append(np.ascontiguousarray(a.T))
            d._times.append(t1 / DAY)
    state._diag_written_day = now_days
    for d in active:
        small = sum(len(r) * r[0].nbytes for r in d._records.values()) < (32 << 20)
        if final or small or len(d._times) % 30 == 0:
            _write(state, d)


def _write(state, d):
    """The whole file from the records held in memory, as netCDF-4 in the reference's layout (roger/io_tools/netcdf.py:22-72, 121-165):
    dimensions x, y and the unlimited Time as dimension scales, variables (Time, y, x) chunked by record with `_FillValue`, the global
    attributes of `initialize_file`.  Written by roger_amd.nc4lite (h5netcdf is not part of this image)."""
    if not d._path:
        return
    from . import nc4lite

    vs, settings = state.variables, state.settings
    os.makedirs(os.path.dirname(os.path.abspath(d._path)), exist_ok=True)
    x, y = np.asarray(vs.x)[2:-2], np.asarray(vs.y)[2:-2]
    dims = {"x": len(x), "y": len(y), "Time": None}
    variables = {
        "x": (("x",), x, {"long_name": "x", "units": "m"}),
        "y": (("y",), y, {"long_name": "y", "units": "m"}),
        "Time": (("Time",), np.asarray(d._times, dtype=np.float64), {"long_name": "Time", "units": "days", "time_origin": str(settings.time_origin)}),
    }
    for name in d.output_variables:
        rec = d._records[name]
        extra = ()
        if rec and rec[0].ndim == 3:   # (ages | nages, y, x)
            dim = "ages" if rec[0].shape[0] == settings.ages else "nages"
            dims.setdefault(dim, rec[0].shape[0])
            extra = (dim,)
        data = np.stack(rec) if rec else np.zeros((0,) + tuple(dims[k] for k in extra) + (len(y), len(x)))
        variables[name] = (("Time",) + extra + ("y", "x"), np.asarray(data, dtype=np.float64),
                           {"_FillValue": np.float64(-9999.0), "long_name": name, "units": _UNITS.get(name, "")})
    nc4lite.write(d._path, dims, variables, {
        "date_created": datetime.datetime.today().isoformat(), "roger_version": "roger_amd (hip backend)",
        "comment": ("First timestep (t=0) contains initial values. Simulations start are written from second timestep "
                    "(t=1) to last timestep (t=N)."),
        "setup_identifier": str(settings.identifier)})


def close(state):
    """Write what is still held in memory (end of run)."""
    for d in getattr(state, "_diag_active", None) or ():
        _write(state, d)
