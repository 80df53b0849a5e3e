"""ctypes binding of libroger_hip.so (the C ABI declared in include/roger_hip.h).

This module is the only place the host package touches the native library.  There is no CPU
fallback: if the library or a GPU is missing, creating a context raises.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROGER_HIP_LIB", os.path.join(PKG, "libroger_hip.so"))  # override: kernel experiments


ABI_VERSION = 4   # include/roger_hip.h: RH_ABI_VERSION


class RhConfig(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("device", C.c_int32), ("enable_lateral_flow", C.c_int32)] + [
        (k, C.c_double) for k in (
            "pi", "r_mp", "l_sc", "sf", "ta_fm", "rmax", "transp_water_stress", "atol", "rtol", "clay_min",
            "clay_max", "theta_rew_min", "theta_rew_max", "rew_min", "rew_max", "z_evap_max", "zroot_to_zsoil_max",
            "a_bc", "b_bc")
    ] + [("end_event", C.c_int64), ("hpi", C.c_int64), ("dx", C.c_double), ("placement_probes", C.c_int32), ("enable_routing_1D", C.c_int32),
         ("dy", C.c_double)]


class RhScalars(C.Structure):
    _fields_ = [
        ("itt", C.c_int64), ("time", C.c_int64), ("dt_secs", C.c_int64), ("itt_day", C.c_int64),
        ("itt_forc", C.c_int64), ("time_event0", C.c_int64), ("event_id_counter", C.c_int64),
        ("event_id", C.c_int64 * 2), ("year", C.c_int64 * 2), ("month", C.c_int64 * 2), ("doy", C.c_int64 * 2),
        ("dt", C.c_double), ("sanity_ok", C.c_int64),
    ]


class NativeError(RuntimeError):
    pass


_lib = None

_ENTRY_POINTS = (
    "rh_topo", "rh_params_surface", "rh_params_soil", "rh_initial_conditions", "rh_adaptive_dt", "rh_interception",
    "rh_evapotranspiration", "rh_snow", "rh_infiltration", "rh_subsurface_runoff", "rh_capillary_rise", "rh_storage",
    "rh_num_error", "rh_after_timestep", "rh_step_phase1", "rh_step_phase2", "rh_sync", "rh_hooks_phase", "rh_step_core", "rh_params_lateral",
    "rh_step_summary", "rh_adaptive_dt_finish", "rh_surface_routing", "rh_subsurface_routing",
)


class _Missing:
    def __init__(self, name):
        self.name = name

    def __call__(self, *a, **k):
        raise NativeError(f"{self.name} is not exported by {LIB_PATH}")


class _Tolerant:
    def __init__(self, lib):
        self.__dict__["_lib"] = lib
        self.__dict__["_missing"] = {}

    def __getattr__(self, name):
        try:
            return getattr(self._lib, name)
        except AttributeError:
            return self._missing.setdefault(name, _Missing(name))


def load():
    """Load the shared library and declare every symbol of roger_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: build it with `python -m roger_amd.build` (there is no CPU fallback)")
    # PyTorch-ROCm bundles its own HIP runtime (libamdhip64).  Two HIP runtimes in one process do
    # not work ("No HIP GPUs are available" from whichever initialises second), so when torch is
    # installed let it load its copy first; libroger_hip.so then binds to the same one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    if os.environ.get("RH_OLD_VARIANT"):   # A/B against a library built from an older commit (tools/ab_variants.sh): entry points
        lib = _Tolerant(lib)               # it lacks are declared on a stand-in that raises when called
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    lib.rh_abi_version.restype = i32
    # the structures below are laid out for ONE version of include/roger_hip.h: a library built from another one would read past
    # (or short of) the caller's rh_config / rh_sas_config.  RH_OLD_VARIANT=1 (A/B against a library of an older commit) accepts
    # an older library, whose structures are prefixes of today's.
    got = int(lib.rh_abi_version())
    if got != ABI_VERSION and not (os.environ.get("RH_OLD_VARIANT") and got < ABI_VERSION):
        raise NativeError(f"{LIB_PATH} implements ABI version {got}, this binding is written for version {ABI_VERSION} "
                          "(include/roger_hip.h: RH_ABI_VERSION): rebuild with `python -m roger_amd.build --force`")
    lib.rh_default_config.argtypes = [C.POINTER(RhConfig)]
    lib.rh_default_config.restype = None
    lib.rh_create.argtypes = [C.POINTER(RhConfig), C.POINTER(vp)]
    lib.rh_destroy.argtypes = [vp]
    lib.rh_destroy.restype = None
    lib.rh_last_error.argtypes = [vp]
    lib.rh_last_error.restype = C.c_char_p
    lib.rh_set_stream.argtypes = [vp, vp]
    lib.rh_num_planes.restype = i32
    lib.rh_plane_name.argtypes = [i32]
    lib.rh_plane_name.restype = C.c_char_p
    lib.rh_plane_is_int.argtypes = [i32]
    lib.rh_plane_index.argtypes = [C.c_char_p]
    lib.rh_num_cells.argtypes = [vp]
    lib.rh_num_cells.restype = i64
    lib.rh_upload.argtypes = [vp, i32, vp, C.c_size_t]
    lib.rh_download.argtypes = [vp, i32, vp, C.c_size_t]
    lib.rh_plane_device_ptr.argtypes = [vp, i32]
    lib.rh_plane_device_ptr.restype = vp
    lib.rh_set_scalars.argtypes = [vp, C.POINTER(RhScalars)]
    lib.rh_get_scalars.argtypes = [vp, C.POINTER(RhScalars)]
    lib.rh_set_luts.argtypes = [vp, vp, vp, vp, vp]
    lib.rh_set_forcing_day.argtypes = [vp, vp, vp, vp, i32]
    lib.rh_set_lut_mlms.argtypes = [vp, vp, i64]
    for name in _ENTRY_POINTS:
        getattr(lib, name).argtypes = [vp]
    lib.rh_step_phase3.argtypes = [vp, i32]
    lib.rh_step_finish.argtypes = [vp, i32]
    lib.rh_placement_report.argtypes = [vp, C.POINTER(C.c_double), i32]
    lib.rh_step_mode.argtypes = [vp]
    lib.rh_plane_is_pure_output.argtypes = [i32, i32]
    lib.rh_set_time_limit.argtypes = [vp, i64]
    lib.rh_sparse_steps.argtypes = [vp]
    lib.rh_sparse_steps.restype = i64
    lib.rh_set_forcing_stations.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]
    lib.rh_diag_upload.argtypes = [vp, i32, i32, vp, C.c_size_t]
    lib.rh_diag_set_slot_state.argtypes = [vp, i32, i64, i64, i64]
    lib.rh_comm_unique_id.argtypes = [vp]
    lib.rh_comm_init.argtypes = [vp, vp, i32, i32]
    lib.rh_set_comm.argtypes = [vp, vp]
    lib.rh_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.rh_run_steps_dist.argtypes = [vp, i64]
    lib.rh_step_routed.argtypes = [vp, i32]
    lib.rh_planes_held.argtypes = [vp]
    lib.rh_route_out.argtypes = [vp, i32]
    lib.rh_route_in.argtypes = [vp, i32]
    lib.rh_route_get_edges.argtypes = [vp, i32, vp, vp]
    lib.rh_route_get_static_edges.argtypes = [vp, vp, vp, vp, vp]
    lib.rh_route_set_halo.argtypes = [vp, i32, vp, vp, vp]
    lib.rh_diag_set_interval.argtypes = [vp, i64]
    lib.rh_diag_slot_times.argtypes = [vp, i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.rh_step_summary_expand.argtypes = [vp, vp]
    lib.rh_step_finish_compress.argtypes = [vp, i32, vp]
    lib.rh_set_forcing_weights.argtypes = [vp, vp, vp, vp]
    lib.rh_diag_configure.argtypes = [vp, vp, i32, vp, i32, i32]
    lib.rh_diag_download.argtypes = [vp, i32, i32, vp, C.c_size_t]
    lib.rh_diag_steps.argtypes = [vp, i32, C.POINTER(C.c_int64)]
    lib.rh_diag_device_ptr.argtypes = [vp, i32, i32]
    lib.rh_diag_device_ptr.restype = vp
    lib.rh_svat_step.argtypes = [vp, i32]
    lib.rh_svat_step_scalars.argtypes = [vp, i32, C.POINTER(RhScalars)]
    lib.rh_param_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rh_run_steps.argtypes = [vp, i64]
    lib.rh_set_forcing_series.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64]
    lib.rh_predicates_expand.argtypes = [vp, i32, vp]
    lib.rh_predicates_compress.argtypes = [vp, i32, vp]
    lib.rh_calibrate_copy.argtypes = [vp, i32, i32, i32]
    lib.rh_selftest_pow.argtypes = [vp, vp, vp, i64]
    lib.rh_selftest_window_sum.argtypes = [vp, vp, i64, vp]
    lib.rh_debug_swap_arenas.argtypes = [vp, vp]
    lib.rh_predicate_words.argtypes = [vp]
    lib.rh_predicate_words.restype = vp
    lib.rh_timing_summary.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.rh_timing_detail.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]
    lib.rh_enable_timing.argtypes = [vp, i32]
    _declare_sas(lib)
    _lib = lib
    return lib


class RhSasConfig(C.Structure):
    _fields_ = [("n_cells", C.c_int64), ("ages", C.c_int32), ("substeps", C.c_int32), ("device", C.c_int32),
                ("forcing_days", C.c_int32), ("age_statistics", C.c_int32), ("keep_distributions", C.c_int32),
                ("vsmow", C.c_double), ("d18O_min", C.c_double), ("d18O_max", C.c_double),
                ("tracer", C.c_int32), ("solver", C.c_int32)]


# the planes of the routing (settings.enable_routing_1D): the last ones of include/rh_fields.def, held by routing contexts only
ROUTING_PLANES = ("flow_dir_topo", "outer_boundary", "k_st", "q_sur_out", "q_sur_in", "q_sub_out", "q_sub_in", "q_sub_in_rz", "q_sub_in_ss")

# RH_SAS_TRACER_*.  Deuterium runs the isotope kernels of oxygen-18 with its own constants in the vsmow / d18O_min /
# d18O_max fields of rh_sas_config (roger/core/transport.py:315-340, roger/settings.py:79-81)
SAS_TRACERS = {"oxygen18": 0, "deuterium": 0, "bromide": 1, "chloride": 2, "virtualtracer": 3}
# RH_SAS_SOLVER_*: settings.sas_solver
SAS_SOLVERS = {"deterministic": 0, "Euler": 1, "RK4": 2}
DEUTERIUM_DEFAULTS = {"vsmow": 155.76e-6, "d18O_min": -160.0, "d18O_max": 0.0}


def _declare_sas(lib):
    """include/roger_hip_sas.h"""
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    lib.rh_sas_default_config.argtypes = [C.POINTER(RhSasConfig)]
    lib.rh_sas_default_config.restype = None
    lib.rh_sas_create.argtypes = [C.POINTER(RhSasConfig), C.POINTER(vp)]
    lib.rh_sas_destroy.argtypes = [vp]
    lib.rh_sas_destroy.restype = None
    lib.rh_sas_last_error.argtypes = [vp]
    lib.rh_sas_last_error.restype = C.c_char_p
    lib.rh_sas_set_stream.argtypes = [vp, vp]
    lib.rh_sas_sync.argtypes = [vp]
    lib.rh_sas_num_arrays.restype = i32
    lib.rh_sas_array_name.argtypes = [i32]
    lib.rh_sas_array_name.restype = C.c_char_p
    lib.rh_sas_array_index.argtypes = [C.c_char_p]
    lib.rh_sas_array_elems.argtypes = [vp, i32]
    lib.rh_sas_array_elems.restype = i64
    lib.rh_sas_array_is_int.argtypes = [i32]
    lib.rh_sas_upload.argtypes = [vp, i32, vp, C.c_size_t]
    lib.rh_sas_download.argtypes = [vp, i32, vp, C.c_size_t]
    lib.rh_sas_upload_cells.argtypes = [vp, i32, i64, i64, vp, C.c_size_t]
    lib.rh_sas_download_cells.argtypes = [vp, i32, i64, i64, vp, C.c_size_t]
    lib.rh_sas_set_daily_from_device.argtypes = [vp, i32, i64, vp]
    lib.rh_sas_array_device_ptr.argtypes = [vp, i32]
    lib.rh_sas_array_device_ptr.restype = vp
    lib.rh_sas_stages.argtypes = [vp, i64, i32]
    lib.rh_sas_step.argtypes = [vp, i64]
    lib.rh_sas_run_days.argtypes = [vp, i64, i64]
    lib.rh_sas_selftest_pow.argtypes = [vp, vp, vp, i64]
    lib.rh_sas_selftest_div.argtypes = [vp, vp, vp, i64]
    lib.rh_sas_enable_timing.argtypes = [vp, i32]
    lib.rh_sas_timing_summary.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]


SAS_DECLARED_SYMBOLS = (
    "rh_sas_default_config", "rh_sas_create", "rh_sas_destroy", "rh_sas_last_error", "rh_sas_set_stream",
    "rh_sas_sync", "rh_sas_num_arrays", "rh_sas_array_name", "rh_sas_array_index", "rh_sas_array_elems",
    "rh_sas_array_is_int", "rh_sas_upload", "rh_sas_download", "rh_sas_upload_cells", "rh_sas_download_cells", "rh_sas_set_daily_from_device", "rh_sas_array_device_ptr", "rh_sas_stages",
    "rh_sas_step", "rh_sas_run_days", "rh_sas_enable_timing", "rh_sas_timing_summary", "rh_sas_selftest_pow", "rh_sas_selftest_div",
)

# stage bits of rh_sas_stages (include/roger_hip_sas.h)
SAS_STAGES = dict(INF_RZ=1, EVAP=2, TRANSP=4, Q_RZ=8, INF_SS=16, Q_SS=32, CPR=64, STORAGE=128, AGEING=256, ALL=511,
                  RESCALE=512)


class SasContext:
    """One SAS / oxygen-18 transport problem on the device (rh_sas_ctx).  Thin, 1:1 with the C ABI."""

    def __init__(self, n_cells, ages, substeps=1, device=0, forcing_days=1, age_statistics=False,
                 keep_distributions=False, tracer="oxygen18", solver="deterministic", **settings):
        lib = load()
        cfg = RhSasConfig()
        lib.rh_sas_default_config(C.byref(cfg))
        if tracer not in SAS_TRACERS:
            raise ValueError(f"tracer {tracer!r}: the hip backend transports {sorted(SAS_TRACERS)}")
        cfg.tracer = SAS_TRACERS[tracer]
        self.tracer = tracer
        if solver not in SAS_SOLVERS:
            raise ValueError(f"solver {solver!r}: the hip backend implements {sorted(SAS_SOLVERS)}")
        cfg.solver = SAS_SOLVERS[solver]
        self.solver = solver
        if tracer == "deuterium":
            settings = {**DEUTERIUM_DEFAULTS, **settings}
        cfg.n_cells, cfg.ages, cfg.substeps, cfg.device = int(n_cells), int(ages), int(substeps), int(device)
        cfg.forcing_days = int(forcing_days)
        cfg.age_statistics, cfg.keep_distributions = int(bool(age_statistics)), int(bool(keep_distributions))
        for k, v in settings.items():
            if k not in ("vsmow", "d18O_min", "d18O_max"):
                raise AttributeError(f"unknown setting {k}")
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.rh_sas_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise NativeError(f"rh_sas_create failed ({rc}): {lib.rh_sas_last_error(None).decode()}")
        self._h, self._lib, self.cfg = h, lib, cfg
        self.n, self.ages, self.substeps, self.forcing_days = cfg.n_cells, cfg.ages, cfg.substeps, cfg.forcing_days
        self.names = [lib.rh_sas_array_name(i).decode() for i in range(lib.rh_sas_num_arrays())]
        self._index = {nm: i for i, nm in enumerate(self.names)}

    def _check(self, rc, what):
        if rc != 0:
            raise NativeError(f"{what} failed ({rc}): {self._lib.rh_sas_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rh_sas_destroy(self._h)
            self._h = None

    __del__ = close

    def index(self, name):
        try:
            return self._index[name]
        except KeyError:
            raise KeyError(f"unknown SAS array {name!r}") from None

    def shape(self, name):
        """Logical shape of an array held by this context."""
        i = self.index(name)
        elems = self._lib.rh_sas_array_elems(self._h, i)
        if elems == 0:
            raise NativeError(f"array {name} is not held by this context (age_statistics / keep_distributions / tracer)")
        if name in DAILY_INPUTS:
            return (self.forcing_days, self.n)
        if elems == self.n:
            return (self.n,)
        return (self.n, elems // self.n)

    def dtype(self, name):
        return np.int32 if self._lib.rh_sas_array_is_int(self.index(name)) else np.float64

    def upload(self, name, host):
        a = np.ascontiguousarray(host, dtype=self.dtype(name))
        if a.shape != self.shape(name):
            raise ValueError(f"{name}: shape {a.shape}, expected {self.shape(name)}")
        self._check(self._lib.rh_sas_upload(self._h, self.index(name), a.ctypes.data_as(C.c_void_p), a.nbytes),
                    f"rh_sas_upload({name})")

    def download(self, name):
        a = np.empty(self.shape(name), dtype=self.dtype(name))
        self._check(self._lib.rh_sas_download(self._h, self.index(name), a.ctypes.data_as(C.c_void_p), a.nbytes),
                    f"rh_sas_download({name})")
        return a

    def upload_cells(self, name, first_cell, host):
        """Rows [first_cell, first_cell + len(host)) of a per-cell array."""
        a = np.ascontiguousarray(host, dtype=self.dtype(name))
        if a.shape[1:] != self.shape(name)[1:]:
            raise ValueError(f"{name}: row shape {a.shape[1:]}, expected {self.shape(name)[1:]}")
        self._check(self._lib.rh_sas_upload_cells(self._h, self.index(name), int(first_cell), a.shape[0],
                                                  a.ctypes.data_as(C.c_void_p), a.nbytes), f"rh_sas_upload_cells({name})")

    def download_cells(self, name, first_cell, n_cells):
        a = np.empty((int(n_cells),) + self.shape(name)[1:], dtype=self.dtype(name))
        self._check(self._lib.rh_sas_download_cells(self._h, self.index(name), int(first_cell), a.shape[0],
                                                    a.ctypes.data_as(C.c_void_p), a.nbytes), f"rh_sas_download_cells({name})")
        return a

    def set_daily_from_device(self, name, day_row, dev_ptr):
        """Row `day_row` of a daily input from n float64 on the device (e.g. Context.diag_device_ptr)."""
        self._check(self._lib.rh_sas_set_daily_from_device(self._h, self.index(name), int(day_row), C.c_void_p(dev_ptr)),
                    f"rh_sas_set_daily_from_device({name})")

    def device_ptr(self, name):
        return self._lib.rh_sas_array_device_ptr(self._h, self.index(name))

    def stages(self, day, mask):
        self._check(self._lib.rh_sas_stages(self._h, int(day), int(mask)), "rh_sas_stages")

    def step(self, day):
        self._check(self._lib.rh_sas_step(self._h, int(day)), "rh_sas_step")

    def run_days(self, day0, ndays):
        self._check(self._lib.rh_sas_run_days(self._h, int(day0), int(ndays)), "rh_sas_run_days")

    def sync(self):
        self._check(self._lib.rh_sas_sync(self._h), "rh_sas_sync")

    def set_stream(self, stream):
        self._check(self._lib.rh_sas_set_stream(self._h, C.c_void_p(stream)), "rh_sas_set_stream")

    def enable_timing(self, on=True):
        self._check(self._lib.rh_sas_enable_timing(self._h, int(on)), "rh_sas_enable_timing")

    def timing_summary(self):
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._lib.rh_sas_timing_summary(self._h, C.byref(ms), C.byref(cnt)), "rh_sas_timing_summary")
        return ms.value, cnt.value


def sas_selftest_div(a, d):
    """a / d by the SAS kernel's hoisted-reciprocal division (rh_sas_selftest_div)."""
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.float64)
    d = np.ascontiguousarray(np.broadcast_to(d, a.shape), dtype=np.float64)
    out = np.empty_like(a)
    rc = lib.rh_sas_selftest_div(a.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.size)
    if rc != 0:
        raise NativeError(f"rh_sas_selftest_div failed ({rc})")
    return out


def sas_selftest_pow(x, k):
    """x ** k by the SAS kernel's own routine (rh_sas_selftest_pow)."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    k = np.ascontiguousarray(np.broadcast_to(k, x.shape), dtype=np.float64)
    out = np.empty_like(x)
    rc = lib.rh_sas_selftest_pow(x.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size)
    if rc != 0:
        raise NativeError(f"rh_sas_selftest_pow failed ({rc})")
    return out


DAILY_INPUTS = ("inf_mat_rz", "inf_pf_rz", "inf_pf_ss", "evap_soil", "transp", "q_rz", "q_ss", "cpr_rz", "C_in")


DECLARED_SYMBOLS = (
    "rh_default_config", "rh_create", "rh_destroy", "rh_last_error", "rh_abi_version", "rh_set_stream", "rh_sync",
    "rh_num_planes", "rh_plane_name", "rh_plane_is_int", "rh_plane_index", "rh_num_cells", "rh_upload",
    "rh_download", "rh_plane_device_ptr", "rh_set_scalars", "rh_get_scalars", "rh_set_luts", "rh_set_forcing_day",
    "rh_topo", "rh_params_surface", "rh_params_soil", "rh_initial_conditions", "rh_adaptive_dt", "rh_interception",
    "rh_evapotranspiration", "rh_snow", "rh_infiltration", "rh_subsurface_runoff", "rh_capillary_rise",
    "rh_storage", "rh_num_error", "rh_after_timestep", "rh_step_phase1", "rh_step_phase2", "rh_step_phase3",
    "rh_svat_step", "rh_svat_step_scalars", "rh_param_stats", "rh_predicate_words", "rh_timing_summary", "rh_timing_detail", "rh_enable_timing", "rh_set_forcing_series",
    "rh_hooks_phase", "rh_run_steps", "rh_predicates_expand", "rh_predicates_compress", "rh_step_core", "rh_calibrate_copy", "rh_debug_swap_arenas", "rh_selftest_pow", "rh_selftest_window_sum", "rh_set_lut_mlms", "rh_params_lateral",
    "rh_step_summary", "rh_step_finish", "rh_diag_set_interval", "rh_diag_slot_times", "rh_placement_report", "rh_step_summary_expand", "rh_step_finish_compress", "rh_diag_configure", "rh_diag_download", "rh_diag_device_ptr", "rh_diag_steps",
    "rh_set_forcing_weights", "rh_adaptive_dt_finish", "rh_diag_upload", "rh_diag_set_slot_state", "rh_set_forcing_stations", "rh_step_mode", "rh_comm_unique_id", "rh_comm_init", "rh_set_comm", "rh_comm_info", "rh_plane_is_pure_output", "rh_sparse_steps", "rh_set_time_limit", "rh_run_steps_dist",
    "rh_surface_routing", "rh_subsurface_routing", "rh_step_routed", "rh_planes_held", "rh_route_out", "rh_route_in", "rh_route_get_edges", "rh_route_get_static_edges", "rh_route_set_halo",
)


def selftest_window_sum(v144, itd):
    """np.sum over the series masked to [itd, itd + 6) as the kernels form it: (by their function, by its general path) per window start."""
    lib = load()
    v = np.ascontiguousarray(v144, dtype=np.float64)
    t = np.ascontiguousarray(itd, dtype=np.int64)
    if v.size != 144:
        raise ValueError("selftest_window_sum: a series of 144 slots")
    out = np.empty(2 * t.size, dtype=np.float64)
    rc = lib.rh_selftest_window_sum(v.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), t.size, out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise NativeError(f"rh_selftest_window_sum failed ({rc})")
    return out[0::2], out[1::2]


def selftest_pow(x, y):
    """x ** y by the kernels' own power function (rh_pow.h) on the device."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(x)
    rc = lib.rh_selftest_pow(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size)
    if rc != 0:
        raise NativeError(f"rh_selftest_pow failed ({rc})")
    return out


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator (ncclGetUniqueId): created on rank 0, handed to every rank's `comm_init`."""
    lib = load()
    buf = C.create_string_buffer(128)
    if lib.rh_comm_unique_id(buf) != 0:
        raise NativeError(f"rh_comm_unique_id failed: {lib.rh_last_error(None).decode()}")
    return buf.raw


def plane_table():
    lib = load()
    n = lib.rh_num_planes()
    return [(lib.rh_plane_name(i).decode(), bool(lib.rh_plane_is_int(i))) for i in range(n)]


class Context:
    """One device arena + stream (rh_ctx).  Thin, 1:1 with the C ABI."""

    def __init__(self, nx, ny, device=0, **settings):
        lib = load()
        cfg = RhConfig()
        lib.rh_default_config(C.byref(cfg))
        cfg.nx, cfg.ny, cfg.device = int(nx), int(ny), int(device)
        for k, v in settings.items():
            if not hasattr(cfg, k):
                raise AttributeError(f"unknown setting {k}")
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.rh_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise NativeError(f"rh_create failed ({rc}): {lib.rh_last_error(None).decode()}")
        self._h = h
        self._lib = lib
        self.nx, self.ny, self.n = int(nx), int(ny), int(nx) * int(ny)
        self.planes = plane_table()
        self.index = {nm: i for i, (nm, _) in enumerate(self.planes)}
        self.planes_held = int(lib.rh_planes_held(h))   # (the routing's planes exist in routing contexts only)
        self.lateral = bool(cfg.enable_lateral_flow)
        self.routing = bool(cfg.enable_routing_1D)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise NativeError(f"{what} failed ({rc}): {self._lib.rh_last_error(self._h).decode()}")

    # -- planes -----------------------------------------------------------------------------
    def dtype_of(self, name):
        return np.int32 if self.planes[self.index[name]][1] else np.float64

    def upload(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=self.dtype_of(name)).reshape(-1)
        if a.size != self.n:
            raise ValueError(f"{name}: expected {self.n} values, got {a.size}")
        self._check(self._lib.rh_upload(self._h, self.index[name], a.ctypes.data_as(C.c_void_p), a.nbytes),
                    f"rh_upload({name})")

    def download(self, name):
        out = np.empty(self.n, dtype=self.dtype_of(name))
        self._check(self._lib.rh_download(self._h, self.index[name], out.ctypes.data_as(C.c_void_p), out.nbytes),
                    f"rh_download({name})")
        return out

    def device_ptr(self, name):
        return self._lib.rh_plane_device_ptr(self._h, self.index[name])

    # -- scalars / tables / forcing -----------------------------------------------------------
    def set_scalars(self, s):
        self._check(self._lib.rh_set_scalars(self._h, C.byref(s)), "rh_set_scalars")

    def get_scalars(self):
        s = RhScalars()
        self._check(self._lib.rh_get_scalars(self._h, C.byref(s)), "rh_get_scalars")
        return s

    def set_luts(self, ilu, gc, gcm, rdlu):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (ilu, gc, gcm, rdlu)]
        for a, shape in zip(arrs, ((25, 13), (25, 13), (25, 2), (25, 7))):
            if a.shape != shape:
                raise ValueError(f"look-up table has shape {a.shape}, expected {shape}")
        self._check(self._lib.rh_set_luts(self._h, *[a.ctypes.data_as(C.c_void_p) for a in arrs]), "rh_set_luts")

    def set_lut_mlms(self, mlms):
        a = np.ascontiguousarray(mlms, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != 9:
            raise ValueError(f"lut_mlms has shape {a.shape}, expected (n_slope, 9)")
        self._check(self._lib.rh_set_lut_mlms(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0]), "rh_set_lut_mlms")

    def set_forcing_day(self, prec_day, ta_day, pet_day):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (prec_day, ta_day, pet_day)]
        per_cell = arrs[0].ndim > 1
        want = (self.n, 144) if per_cell else (144,)
        for a in arrs:
            if a.reshape(want).shape != want:
                raise ValueError("forcing must be (144,) or (n_cells, 144)")
        self._check(self._lib.rh_set_forcing_day(self._h, *[a.ctypes.data_as(C.c_void_p) for a in arrs],
                                                 int(per_cell)), "rh_set_forcing_day")

    def set_forcing_series(self, F):
        """F: dict with PREC, TA, PET (float64) and YEAR, MONTH, DOY (int64) 10-minute vectors."""
        f = [np.ascontiguousarray(F[k], dtype=np.float64) for k in ("PREC", "TA", "PET")]
        c = [np.ascontiguousarray(F[k], dtype=np.int64) for k in ("YEAR", "MONTH", "DOY")]
        n = f[0].size
        if any(a.size != n for a in f + c):
            raise ValueError("forcing vectors differ in length")
        self._check(self._lib.rh_set_forcing_series(self._h, *[a.ctypes.data_as(C.c_void_p) for a in f + c], n),
                    "rh_set_forcing_series")

    def set_forcing_stations(self, F, station_index):
        """F: PREC, TA, PET as (n_stations, nitt_forc) float64 and YEAR, MONTH, DOY (nitt_forc,) int64; station_index: per cell the row
        of its station (< 0: none).  settings.enable_distributed_input (roger/bmimodels/svat_dist/svat_dist.py:274-310)."""
        f = [np.ascontiguousarray(F[k], dtype=np.float64) for k in ("PREC", "TA", "PET")]
        c = [np.ascontiguousarray(F[k], dtype=np.int64) for k in ("YEAR", "MONTH", "DOY")]
        ns, nitt = f[0].shape
        if any(a.shape != (ns, nitt) for a in f) or any(a.shape != (nitt,) for a in c):
            raise ValueError("station forcing: PREC / TA / PET must be (n_stations, nitt_forc), the calendar (nitt_forc,)")
        idx = np.ascontiguousarray(station_index, dtype=np.int32).reshape(-1)
        if idx.size != self.n or idx.max() >= ns:
            raise ValueError("station_index: one row index per cell, below the number of stations")
        self._check(self._lib.rh_set_forcing_stations(self._h, *[a.ctypes.data_as(C.c_void_p) for a in f + c], nitt, ns,
                                                      idx.ctypes.data_as(C.c_void_p)), "rh_set_forcing_stations")

    def run_steps(self, nsteps):
        self._check(self._lib.rh_run_steps(self._h, int(nsteps)), "rh_run_steps")

    # -- routing (settings.enable_routing_1D) -----------------------------------------------------
    def step_routed(self, monthly=False):
        self._check(self._lib.rh_step_routed(self._h, -1 if monthly is None else int(bool(monthly))), "rh_step_routed")

    def route_out(self, which):
        self._check(self._lib.rh_route_out(self._h, int(which)), "rh_route_out")

    def route_in(self, which):
        self._check(self._lib.rh_route_in(self._h, int(which)), "rh_route_in")

    def route_edges(self, which):
        """(lo, hi): q_out of the rank's edge columns x = 0 and x = nx - 1 (what the x-neighbours' halos take)."""
        lo, hi = np.empty(self.ny), np.empty(self.ny)
        self._check(self._lib.rh_route_get_edges(self._h, int(which), lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p)),
                    "rh_route_get_edges")
        return lo, hi

    def route_static_edges(self):
        """(flow_dir_lo, flow_dir_hi, mask_lo, mask_hi) of the rank's edge columns."""
        a = [np.empty(self.ny, dtype=np.int32) for _ in range(4)]
        self._check(self._lib.rh_route_get_static_edges(self._h, *[x.ctypes.data_as(C.c_void_p) for x in a]), "rh_route_get_static_edges")
        return tuple(a)

    def route_set_halo(self, side, q=None, flow_dir=None, mask=None):
        keep = [None if x is None else np.ascontiguousarray(x, dtype=dt) for x, dt in ((q, np.float64), (flow_dir, np.int32), (mask, np.int32))]
        for x in keep:
            if x is not None and x.size != self.ny:
                raise ValueError(f"a halo column has ny = {self.ny} values, got {x.size}")
        self._check(self._lib.rh_route_set_halo(self._h, int(side), *[None if x is None else x.ctypes.data_as(C.c_void_p) for x in keep]),
                    "rh_route_set_halo")

    # -- routines -----------------------------------------------------------------------------
    def call(self, entry):
        self._check(getattr(self._lib, entry)(self._h), entry)

    def step(self, monthly=False):
        self._check(self._lib.rh_svat_step(self._h, int(bool(monthly))), "rh_svat_step")

    def step_scalars(self, monthly=False):
        """rh_svat_step + the read-back of the scalars in ONE native call (a driver that keeps its hooks on the host needs vs.time
        before the next step); returns the RhScalars after the step."""
        s = RhScalars()
        rc = self._lib.rh_svat_step_scalars(self._h, 1 if monthly else 0, C.byref(s))
        if rc != 0:
            self._check(rc, "rh_svat_step_scalars")
        return s

    def step_phase3(self, monthly=False):
        # monthly: False/True, or -1 to use the month-change flag computed on the device
        self._check(self._lib.rh_step_phase3(self._h, int(monthly)), "rh_step_phase3")

    def set_forcing_weights(self, prec_weight=None, ta_offset=None, pet_weight=None):
        """Per-cell weights on the resident series (prec * w, ta + offset, pet * w); None x 3 clears them."""
        if prec_weight is None and ta_offset is None and pet_weight is None:
            self._check(self._lib.rh_set_forcing_weights(self._h, None, None, None), "rh_set_forcing_weights")
            return
        arrs = [np.ascontiguousarray(a, dtype=np.float64).reshape(-1) for a in (prec_weight, ta_offset, pet_weight)]
        if any(a.size != self.n for a in arrs):
            raise ValueError("weights must have one value per cell")
        self._check(self._lib.rh_set_forcing_weights(self._h, *[a.ctypes.data_as(C.c_void_p) for a in arrs]),
                    "rh_set_forcing_weights")

    def diag_configure(self, rate=(), collect=(), n_slots=1):
        """Device-side daily accumulators: `rate` variables are summed per day, `collect` variables keep their
        end-of-day value (time level tau); n_slots days stay resident."""
        self._diag_names = list(rate) + list(collect)
        ids = lambda names: (C.c_int * max(1, len(names)))(*[self.index[n] for n in names])  # noqa: E731
        r, c = ids(list(rate)), ids(list(collect))
        self._check(self._lib.rh_diag_configure(self._h, r, len(rate), c, len(collect), int(n_slots)), "rh_diag_configure")

    def diag_download(self, name, slot):
        a = np.empty(self.n, dtype=np.float64)
        self._check(self._lib.rh_diag_download(self._h, self._diag_names.index(name), int(slot),
                                               a.ctypes.data_as(C.c_void_p), a.nbytes), "rh_diag_download")
        return a

    def diag_steps(self, slot):
        """Steps accumulated in a day slot (divide a rate variable by it for the "average" diagnostic)."""
        n = C.c_int64()
        self._check(self._lib.rh_diag_steps(self._h, int(slot), C.byref(n)), "rh_diag_steps")
        return n.value

    def diag_set_interval(self, seconds):
        self._check(self._lib.rh_diag_set_interval(self._h, C.c_int64(int(seconds))), "rh_diag_set_interval")

    def diag_slot_times(self, slot):
        """(start time of the interval's first step or -1, end time of its last step) of a slot."""
        t0, t1 = C.c_int64(), C.c_int64()
        self._check(self._lib.rh_diag_slot_times(self._h, int(slot), C.byref(t0), C.byref(t1)), "rh_diag_slot_times")
        return t0.value, t1.value

    def diag_upload(self, name, slot, values):
        a = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        self._check(self._lib.rh_diag_upload(self._h, self._diag_names.index(name), int(slot), a.ctypes.data_as(C.c_void_p), a.nbytes),
                    "rh_diag_upload")

    def diag_set_slot_state(self, slot, steps, t_start, t_end):
        self._check(self._lib.rh_diag_set_slot_state(self._h, int(slot), int(steps), int(t_start), int(t_end)), "rh_diag_set_slot_state")

    def diag_device_ptr(self, name, slot):
        return self._lib.rh_diag_device_ptr(self._h, self._diag_names.index(name), int(slot))

    def pure_output_planes(self):
        """Names of the planes the fused step of this context's model only produces (not stored by the steps of an rh_run_steps call
        that another step follows)."""
        model = 2 if self.routing else int(self.lateral)
        return [nm for nm, p in self.index.items() if self._lib.rh_plane_is_pure_output(model, p) == 1]

    def set_time_limit(self, t_end):
        """No step begins at or beyond model time t_end (None / negative: no limit): rh_run_steps(n) then runs at most n steps."""
        self._check(self._lib.rh_set_time_limit(self._h, -1 if t_end is None else int(t_end)), "rh_set_time_limit")

    def param_stats(self):
        """(fraction of the wavefronts whose derived parameters are not loaded, bytes per column of parameter loads that are one element
        per wave) -- how the lazy variants of the fused kernel will read the parameter planes as they are now."""
        d, u = C.c_double(), C.c_double()
        self._check(self._lib.rh_param_stats(self._h, C.byref(d), C.byref(u)), "rh_param_stats")
        return d.value, u.value

    def sparse_steps(self):
        return int(self._lib.rh_sparse_steps(self._h))

    def step_mode(self):
        """(lazy, tail) of the last fused step: it deferred the tau -> taum1 copies / its last wavefront formed the next step's control part."""
        m = self._lib.rh_step_mode(self._h)
        return bool(m & 1), bool(m & 2)

    # -- multi-GPU stepping without the host in the loop (RCCL from C) -------------------------------------------------
    def comm_init(self, unique_id, nranks, rank):
        """Every rank, collectively: an RCCL communicator on this context's device from rank 0's `comm_unique_id()` bytes."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.rh_comm_init(self._h, buf, int(nranks), int(rank)), "rh_comm_init")

    def comm_init_torch(self, group=None):
        """The same through an initialised torch.distributed process group: rank 0 creates the id, broadcast to the others."""
        import torch.distributed as dist

        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        box = [comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self.comm_init(box[0], world, rank)

    def comm_info(self):
        """(ranks, this rank) of the communicator the context holds, as RCCL reports them (ncclCommCount, ncclCommUserRank)."""
        n, r = C.c_int(), C.c_int()
        self._check(self._lib.rh_comm_info(self._h, C.byref(n), C.byref(r)), "rh_comm_info")
        return n.value, r.value

    def run_steps_dist(self, nsteps):
        self._check(self._lib.rh_run_steps_dist(self._h, int(nsteps)), "rh_run_steps_dist")

    def placement_report(self):
        """Streaming-kernel time (ms) on every candidate arena rh_create probed, the chosen one first ([]: probing off)."""
        buf = (C.c_double * 32)()
        n = self._lib.rh_placement_report(self._h, buf, 32)
        return [float(buf[k]) for k in range(min(n, 32))]

    def step_finish(self, monthly=-1):
        self._check(self._lib.rh_step_finish(self._h, int(monthly)), "rh_step_finish")

    def step_summary_expand(self, dev_dst64):
        """rh_step_summary with the summary word also written as 64 int32 (0 / 1) to the device pointer."""
        self._check(self._lib.rh_step_summary_expand(self._h, C.c_void_p(dev_dst64)), "rh_step_summary_expand")

    def step_finish_compress(self, dev_src64, monthly=-1):
        """rh_step_finish from the all-reduced 64 int32 at the device pointer."""
        self._check(self._lib.rh_step_finish_compress(self._h, int(monthly), C.c_void_p(dev_src64)), "rh_step_finish_compress")

    def sync(self):
        self.call("rh_sync")

    def set_stream(self, stream_handle):
        self._check(self._lib.rh_set_stream(self._h, C.c_void_p(stream_handle)), "rh_set_stream")

    def predicates_expand(self, word, dev_ptr):
        self._check(self._lib.rh_predicates_expand(self._h, int(word), C.c_void_p(dev_ptr)), "rh_predicates_expand")

    def predicates_compress(self, word, dev_ptr):
        self._check(self._lib.rh_predicates_compress(self._h, int(word), C.c_void_p(dev_ptr)),
                    "rh_predicates_compress")

    def calibrate_copy(self, src_plane0, dst_plane0, nplanes):
        self._check(self._lib.rh_calibrate_copy(self._h, src_plane0, dst_plane0, nplanes), "rh_calibrate_copy")

    def predicate_words_ptr(self):
        return self._lib.rh_predicate_words(self._h)

    def enable_timing(self, on=True):
        self._check(self._lib.rh_enable_timing(self._h, int(on)), "rh_enable_timing")

    def timing_summary(self):
        """(total kernel ms, launches) of the fused kernel since enable_timing(True)."""
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._lib.rh_timing_summary(self._h, C.byref(ms), C.byref(cnt)), "rh_timing_summary")
        return float(ms.value), int(cnt.value)

    def timing_detail(self, cap=65536):
        """(kernel ms, dt_secs) per timed launch since enable_timing(True): the time-step class of every step."""
        ms = np.zeros(cap, dtype=np.float64)
        dts = np.zeros(cap, dtype=np.int32)
        cnt = C.c_int64()
        self._check(self._lib.rh_timing_detail(self._h, ms.ctypes.data_as(C.POINTER(C.c_double)),
                                               dts.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(cnt)), "rh_timing_detail")
        k = min(int(cnt.value), cap)
        return ms[:k], dts[:k]
