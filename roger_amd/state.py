"""State containers mirroring roger/state.py: KernelOutput, RogerSettings, RogerVariables,
RogerState -- with the arrays living in the device arena.

`vs.<name>` returns a host array of the reference's shape (ghost frame included, read-only);
assigning validates shape and dtype exactly like the reference (state.py:240-251).  Behind the
attribute bag sits lazy two-way synchronisation: an assignment marks the variable host-dirty and
it is uploaded before the next native call; a native call marks what it writes device-newer and
those variables are downloaded on first access.  Ownership: the backend owns the arena, this
object owns the host mirrors.
"""
import contextlib
from collections import defaultdict, namedtuple

import numpy as np

from . import _native
from . import variables as var_mod
from .settings import SETTINGS
from .timer import Timer


def make_namedtuple(**kwargs):
    return namedtuple("KernelOutput", list(kwargs.keys()))(*kwargs.values())


KernelOutput = make_namedtuple


class DeviceResult:
    """What a natively implemented @roger_kernel returns instead of host arrays: the names it
    updated in the arena.  `vs.update(DeviceResult)` only marks them as newer on the device."""

    def __init__(self, names=()):
        self._fields = tuple(names)

    def __iter__(self):
        return iter(())

    def __repr__(self):
        return f"DeviceResult({list(self._fields)})"


class Lockable:
    __locked__ = True

    @contextlib.contextmanager
    def unlock(self):
        prev = self.__locked__
        try:
            object.__setattr__(self, "__locked__", False)
            yield
        finally:
            object.__setattr__(self, "__locked__", prev)

    @contextlib.contextmanager
    def lock(self):
        prev = self.__locked__
        try:
            object.__setattr__(self, "__locked__", True)
            yield
        finally:
            object.__setattr__(self, "__locked__", prev)

    def _check_lock(self, key):
        if not key.startswith("_") and self.__locked__:
            clsname = self.__class__.__qualname__
            raise RuntimeError(
                f"{clsname} is locked to modifications. If you know what you are doing, "
                f'you can unlock it via the "{clsname}.unlock()" context manager.')


class RogerSettings(Lockable):
    """roger/state.py:170-186"""

    def __init__(self, settings_meta=SETTINGS):
        object.__setattr__(self, "__metadata__", settings_meta)
        for k, meta in settings_meta.items():
            object.__setattr__(self, k, meta.type(meta.default))

    def __setattr__(self, key, val):
        self._check_lock(key)
        if key.startswith("_"):
            return object.__setattr__(self, key, val)
        if key not in self.__metadata__:
            raise AttributeError(f"Unknown attribute {key}")
        object.__setattr__(self, key, self.__metadata__[key].type(val))

    def fields(self):
        return self.__metadata__.keys()

    def update(self, other=None, **new):
        for k, v in dict(other or {}, **new).items():
            setattr(self, k, v)
        return self


class HostScalars:
    """Stand-in for the arena context in offline-transport runs, where the time-stepping scalars (itt, time, dt,
    dt_secs) are plain host values (the transport step has no device-side bookkeeping)."""

    def __init__(self):
        self._s = _native.RhScalars()
        self._s.dt, self._s.dt_secs, self._s.sanity_ok = 1.0, 3600, 1

    def get_scalars(self):
        return self._s

    def set_scalars(self, s):
        self._s = s

    def sync(self):
        pass

    def close(self):
        pass


class RogerVariables(Lockable):
    """roger/state.py:189-257 over a `_native.Context` (SVAT / oneD) or a `_native.SasContext` (offline
    transport)."""

    def __init__(self, var_meta, dimensions, ctx, sas_ctx=None):
        d = object.__setattr__
        d(self, "__metadata__", var_meta)
        d(self, "__dimensions__", dimensions)
        d(self, "_ctx", ctx)
        d(self, "_sas", sas_ctx)
        d(self, "_host", {})
        d(self, "_host_dirty", set())
        # freshness of the host mirrors.  A native call that may have written ANY plane (a whole step) bumps `_epoch` -- O(1) per step
        # instead of a set of ~200 names --: a plane variable whose mirror was filled in an earlier epoch is newer on the device.
        # `_device_newer` names the variables a call marked individually.
        d(self, "_device_newer", set())
        d(self, "_epoch", 0)
        d(self, "_mirror_epoch", {})
        d(self, "_plane_vars", frozenset(k for k, v in var_meta.items() if v.plane is not None))
        d(self, "_probe", None)            # roger_amd/hooks.py: a recording session (hook recognition), or None
        d(self, "_scalars", None)          # cached rh_scalars (valid until the next native call)
        d(self, "_scalars_dirty", False)
        d(self, "_forcing_dirty", False)
        for key, var in var_meta.items():
            if var.dims is None:
                if key in var_mod.DEVICE_SCALARS:
                    continue
                self._host[key] = (var.dtype or np.float64)(var.initial or 0)
            elif key in var_mod.DEVICE_SCALARS2:
                continue
            else:
                self._host[key] = var_mod.allocate(dimensions, var.dims, dtype=var.dtype or np.float64,
                                                   fill=var.initial if var.initial is not None else 0)
        # the arena already holds the registry's initial values (rh_create)

    # -- scalars on the device -------------------------------------------------------------
    def _get_scalars(self):
        if self._scalars is None:
            object.__setattr__(self, "_scalars", self._ctx.get_scalars())
        return self._scalars

    # -- attribute protocol ------------------------------------------------------------------
    def _is_device_newer(self, key):
        return key in self._device_newer or (key in self._plane_vars and self._mirror_epoch.get(key, 0) != self._epoch)

    def __getattr__(self, key):
        meta = object.__getattribute__(self, "__metadata__")
        if key not in meta:
            raise AttributeError(key)
        probe = self._probe
        if probe is not None:              # a hook is being probed: reads are recorded, the probe's values stand in for the scalars
            probe.reads.add(key)
            if key in probe.writes:
                return probe.writes[key]
            if key in probe.values:
                return probe.values[key]
        if key in var_mod.DEVICE_SCALARS:
            v = getattr(self._get_scalars(), key)
            return float(v) if key == "dt" else int(v)
        if key in var_mod.DEVICE_SCALARS2:
            a = np.array(list(getattr(self._get_scalars(), key)), dtype=np.int64)
            a.flags.writeable = False
            return a
        if self._is_device_newer(key):
            self._download(key)
        return self._host[key]

    def __setattr__(self, key, val):
        self._check_lock(key)
        if key.startswith("_"):
            return object.__setattr__(self, key, val)
        if key not in self.__metadata__:
            raise AttributeError(f"Unknown attribute {key}")
        if self._probe is not None:        # recorded, not applied
            self._probe.writes[key] = val
            return
        var = self.__metadata__[key]
        if key in var_mod.DEVICE_SCALARS:
            s = self._get_scalars()
            setattr(s, key, float(val) if key == "dt" else int(val))
            object.__setattr__(self, "_scalars_dirty", True)
            return
        if key in var_mod.DEVICE_SCALARS2:
            arr = np.asarray(val, dtype=np.int64)
            if arr.shape != (2,):
                raise ValueError(f"Got unexpected shape for variable {key} (expected: (2,), got: {arr.shape})")
            s = self._get_scalars()
            getattr(s, key)[0], getattr(s, key)[1] = int(arr[0]), int(arr[1])
            object.__setattr__(self, "_scalars_dirty", True)
            return
        dtype = var.dtype or np.float64
        if var.dims is None:
            self._host[key] = dtype(val)
            return
        arr = np.asarray(val, dtype=dtype)
        expected = var_mod.get_shape(self.__dimensions__, var.dims)
        if arr.shape != expected:
            raise ValueError(f"Got unexpected shape for variable {key} (expected: {expected}, got: {arr.shape})")
        if arr.flags.writeable:
            arr = arr.copy()
            arr.flags.writeable = False
        self._host[key] = arr
        self._device_newer.discard(key)
        self._mirror_epoch[key] = self._epoch
        if var.plane is not None or var.sas is not None:
            self._host_dirty.add(key)
        elif key in ("prec_day", "ta_day", "pet_day"):
            object.__setattr__(self, "_forcing_dirty", True)

    def fields(self):
        return self.__metadata__.keys()

    def __contains__(self, key):
        return key in self.__metadata__

    def update(self, other=None, **new_fields):
        """roger/state.py:70-90; a DeviceResult only moves the freshness marker."""
        if isinstance(other, DeviceResult):
            if self._probe is not None:
                return self
            self._device_newer.update(other._fields)
            for k in other._fields:
                self._host_dirty.discard(k)
            return self
        if other is not None:
            if hasattr(other, "_fields"):
                new_fields = dict(zip(other._fields, other))
            elif isinstance(other, dict):
                new_fields = other
            else:
                raise TypeError(f"Cannot update from {type(other)} type")
        for key in new_fields:
            if key not in self.__metadata__:
                raise AttributeError(f"unknown attribute {key}")
        for key, val in new_fields.items():
            setattr(self, key, val)
        return self

    # -- synchronisation ---------------------------------------------------------------------
    def _levels(self, key):
        return (("_m1", 0), ("", 1)) if len(self.__metadata__[key].dims) == 3 else (("", None),)

    def _download(self, key):
        host = self._host[key].copy()
        nxl, nyl = host.shape[0] - 4, host.shape[1] - 4
        var = self.__metadata__[key]
        if var.sas is not None:
            dev = self._sas.download(var.sas)
            if var.sas in _native.DAILY_INPUTS:
                dev = dev[0]
            dev = dev.reshape((nxl, nyl) + dev.shape[1:]).astype(host.dtype)
            if "timesteps" in var.dims:
                # the device holds time level tau; after the step tau and taum1 agree (after_substep_iso,
                # roger/core/transport.py:3421-3521)
                host[2:-2, 2:-2, 0] = dev
                host[2:-2, 2:-2, 1] = dev
            else:
                host[2:-2, 2:-2] = dev
            host.flags.writeable = False
            self._host[key] = host
            self._device_newer.discard(key)
            self._mirror_epoch[key] = self._epoch
            return
        for suffix, lvl in self._levels(key):
            flat = self._ctx.download(key + suffix).reshape(nxl, nyl)
            if lvl is None:
                host[2:-2, 2:-2] = flat.astype(host.dtype)
            else:
                host[2:-2, 2:-2, lvl] = flat.astype(host.dtype)
        host.flags.writeable = False
        self._host[key] = host
        self._device_newer.discard(key)
        self._mirror_epoch[key] = self._epoch

    def flush_to_device(self):
        """Upload everything assigned on the host since the last native call."""
        if not (self._host_dirty or self._scalars_dirty or self._forcing_dirty):
            return
        for key in sorted(self._host_dirty):
            host = self._host[key]
            var = self.__metadata__[key]
            if var.sas is not None:
                a = host[2:-2, 2:-2]
                if "timesteps" in var.dims:
                    a = a[:, :, 1]
                a = np.ascontiguousarray(a).reshape((a.shape[0] * a.shape[1],) + a.shape[2:])
                self._sas.upload(var.sas, a[None, :] if var.sas in _native.DAILY_INPUTS else a)
                continue
            for suffix, lvl in self._levels(key):
                self._ctx.upload(key + suffix, host[2:-2, 2:-2] if lvl is None else host[2:-2, 2:-2, lvl])
        self._host_dirty.clear()
        if self._scalars_dirty:
            self._ctx.set_scalars(self._scalars)
            object.__setattr__(self, "_scalars_dirty", False)
        if self._forcing_dirty:
            days = [self._host[k][2:-2, 2:-2, :] for k in ("prec_day", "ta_day", "pet_day")]
            first = [np.array(d[0, 0]) for d in days]
            # one series for all columns: known from the strides when the hook assigned a broadcast value (operators.update),
            # compared otherwise (a hook that filled the interior column by column with the same series)
            if all((d.strides[0] == 0 and d.strides[1] == 0) or bool((np.isnan(d) & np.isnan(f) | (d == f)).all())
                   for d, f in zip(days, first)):
                self._ctx.set_forcing_day(*first)  # one station series broadcast to all cells
            else:
                self._ctx.set_forcing_day(*[np.ascontiguousarray(d).reshape(-1, 144) for d in days])
            object.__setattr__(self, "_forcing_dirty", False)

    def mark_device_newer(self, names=None, scalars=None):
        """After a native call: `names` (or every arena variable) changed on the device.  `scalars`: the rh_scalars the call
        read back itself (rh_svat_step_scalars), cached until the next native call."""
        if names is None:
            object.__setattr__(self, "_epoch", self._epoch + 1)
        else:
            self._device_newer.update(k for k in names if k in self.__metadata__)
        object.__setattr__(self, "_scalars", scalars)


class RogerState:
    """roger/state.py:331-420"""

    def __init__(self, var_meta=None, setting_meta=SETTINGS, dimensions=var_mod.DIM_TO_SHAPE_VAR):
        self._var_meta = var_meta
        self._variables = None
        self._settings = RogerSettings(setting_meta)
        self._dimensions = dimensions
        self._ctx = None
        self._diagnostics = {}
        self.timers = defaultdict(Timer)
        self.profile_timers = defaultdict(Timer)

    def _manifest_dimensions(self):
        out = {}
        for dim, target in self._dimensions.items():
            out[dim] = int(getattr(self._settings, target) if isinstance(target, str) else target)
        return out

    @property
    def dimensions(self):
        return self._manifest_dimensions()

    def initialize_variables(self):
        from . import runtime_settings as rs

        if self._variables is not None:
            raise RuntimeError("Variables are already initialized.")
        s = self._settings
        px, py = rs.num_proc
        import os

        device = int(os.environ.get("LOCAL_RANK", "0"))
        if s.enable_offline_transport:
            # offline oxygen-18 / bromide transport: the state lives in a SAS context (include/roger_hip_sas.h); the reference's
            # variables tt_*, mtt_*, TT_*, sa_s, msa_s are exposed, hence keep_distributions
            if self._var_meta is None:
                self._var_meta = var_mod.build_transport_variables(s)
            tracer = next(t for t in ("oxygen18", "deuterium", "bromide", "chloride", "virtualtracer") if getattr(s, "enable_" + t))
            iso = (s.VSMOW_conc2H, s.d2H_min, s.d2H_max) if s.enable_deuterium else (s.VSMOW_conc18O, s.d18O_min, s.d18O_max)
            self._sas_ctx = _native.SasContext(
                (s.nx // px) * (s.ny // py), s.ages, s.sas_solver_substeps, device=device, forcing_days=1,
                age_statistics=s.enable_age_statistics, keep_distributions=True, tracer=tracer,
                solver=s.sas_solver or "deterministic", vsmow=iso[0], d18O_min=iso[1], d18O_max=iso[2])
            self._ctx = HostScalars()
            self._variables = RogerVariables(self._var_meta, self._manifest_dimensions(), self._ctx, self._sas_ctx)
            return
        if self._var_meta is None:
            self._var_meta = var_mod.build_variables(s)
        consts = {k: getattr(s, k) for k in (
            "pi", "r_mp", "l_sc", "sf", "ta_fm", "rmax", "transp_water_stress", "atol", "rtol", "clay_min", "clay_max",
            "theta_rew_min", "theta_rew_max", "rew_min", "rew_max", "z_evap_max", "zroot_to_zsoil_max", "a_bc", "b_bc",
            "end_event", "hpi")}
        consts["enable_lateral_flow"] = int(bool(s.enable_lateral_flow))   # oneD model
        consts["dx"] = float(s.dx)
        consts["enable_routing_1D"] = int(bool(s.enable_routing_1D))   # D8 routing of surface and subsurface runoff
        consts["dy"] = float(s.dy)
        self._ctx = _native.Context(s.nx // px, s.ny // py, device=device, **consts)
        self._variables = RogerVariables(self._var_meta, self._manifest_dimensions(), self._ctx)

    @property
    def var_meta(self):
        return self._var_meta

    @property
    def variables(self):
        if self._variables is None:
            raise RuntimeError("Variables have not been initialized yet.")
        return self._variables

    @property
    def settings(self):
        return self._settings

    @property
    def diagnostics(self):
        return self._diagnostics

    @property
    def var_meta(self):
        """roger/state.py: the variable registry (name -> Variable)."""
        return self._var_meta

    @property
    def backend_context(self):
        return self._ctx

    @property
    def sas_context(self):
        """The `_native.SasContext` of an offline-transport run (None otherwise)."""
        return getattr(self, "_sas_ctx", None)
