"""ctypes binding of oracle/libsvat_oracle.so -- test infrastructure only.

The oracle is the CPU restatement of the reference's SVAT step; tests use it as
the checker for the HIP path.  Nothing under roger_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libsvat_oracle.so")

SCALARS = ("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")
SCALARS2 = ("event_id", "year", "month", "doy")


class OcScalars(C.Structure):
    _fields_ = [
        ("itt", C.c_int64), ("time", C.c_int64), ("dt_secs", C.c_int64), ("itt_day", C.c_int64),
        ("itt_forc", C.c_int64), ("time_event0", C.c_int64), ("event_id_counter", C.c_int64),
        ("event_id", C.c_int64 * 2), ("year", C.c_int64 * 2), ("month", C.c_int64 * 2), ("doy", C.c_int64 * 2),
        ("dt", C.c_double), ("sanity_ok", C.c_int64),
    ]


class OcSettings(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "pi", "r_mp", "l_sc", "sf", "ta_fm", "rmax", "transp_water_stress", "atol", "rtol", "clay_min", "clay_max",
        "theta_rew_min", "theta_rew_max", "rew_min", "rew_max", "z_evap_max", "zroot_to_zsoil_max", "a_bc", "b_bc",
    )] + [("end_event", C.c_int64), ("hpi", C.c_int64), ("enable_lateral_flow", C.c_int64), ("dx", C.c_double),
         ("enable_routing_1D", C.c_int64), ("dy", C.c_double), ("nx", C.c_int64), ("ny", C.c_int64)]


def default_settings():
    """roger/settings.py:52-122 defaults."""
    return OcSettings(
        pi=3.14159265358979323846264338327950588, r_mp=2.5, l_sc=10000, sf=3, ta_fm=0, rmax=30,
        transp_water_stress=0.75, atol=1e-2, rtol=1e-2, clay_min=0.01, clay_max=0.71, theta_rew_min=0.02,
        theta_rew_max=0.24, rew_min=2, rew_max=12, z_evap_max=150, zroot_to_zsoil_max=0.7, a_bc=2, b_bc=2,
        end_event=21600, hpi=5, enable_lateral_flow=0, dx=1.0, enable_routing_1D=0, dy=1.0, nx=0, ny=0,
    )


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oc_plane_name.restype = C.c_char_p
        _lib.oc_np_sum.restype = C.c_double
        _lib.oc_np_sum.argtypes = [C.c_void_p, C.c_int64]
    return _lib


def plane_names():
    L = lib()
    return [L.oc_plane_name(i).decode() for i in range(L.oc_nplanes())]


def plane_is_int():
    L = lib()
    return [bool(L.oc_plane_is_int(i)) for i in range(L.oc_nplanes())]


class OracleState:
    """SoA planes (one contiguous array of n cells per plane) + scalars + settings."""

    def __init__(self, n):
        self.n = int(n)
        self.names = plane_names()
        self.is_int = plane_is_int()
        self.planes = {}
        for nm, ii in zip(self.names, self.is_int):
            self.planes[nm] = np.zeros(self.n, dtype=np.int32 if ii else np.float64)
        self.scal = OcScalars()
        self.settings = default_settings()
        self.luts = {}
        self._refresh_ptrs()

    def _refresh_ptrs(self):
        arr = (C.c_void_p * len(self.names))()
        for i, nm in enumerate(self.names):
            arr[i] = self.planes[nm].ctypes.data
        self._ptrs = arr

    # -- loading from golden snapshots -------------------------------------------------
    def load_snapshot(self, snap, names=None):
        """`names`: the plane list of the golden file the snapshot comes from (files written before a field was added to the
        oracle hold fewer planes); snapshot() then returns that list's planes, in its order."""
        if names is not None:
            self._view = [nm for nm in names]
        names = list(names) if names is not None else self.names
        for row, nm in zip(snap, names):
            if nm in self.planes:
                self.planes[nm][:] = row.astype(self.planes[nm].dtype)

    def snapshot(self, names=None):
        names = names if names is not None else getattr(self, "_view", self.names)
        return np.stack([self.planes[nm].astype(np.float64) for nm in names])

    def load_scalars(self, row):
        i = 0
        for k in SCALARS:
            setattr(self.scal, k, float(row[i]) if k == "dt" else int(row[i]))
            i += 1
        for k in SCALARS2:
            getattr(self.scal, k)[0] = int(row[i])
            getattr(self.scal, k)[1] = int(row[i + 1])
            i += 2

    def scalars_row(self):
        row = [float(getattr(self.scal, k)) for k in SCALARS]
        for k in SCALARS2:
            row += [float(getattr(self.scal, k)[0]), float(getattr(self.scal, k)[1])]
        return np.array(row)

    def set_luts(self, ilu, gc, gcm, rdlu):
        self.luts = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in
                     dict(ilu=ilu, gc=gc, gcm=gcm, rdlu=rdlu).items()}

    def _lut_args(self):
        return [self.luts[k].ctypes.data_as(C.c_void_p) for k in ("ilu", "gc", "gcm", "rdlu")]

    # -- routines -----------------------------------------------------------------------
    def _forc(self, prec_day, ta_day, pet_day):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (prec_day, ta_day, pet_day)]
        stride = 0 if arrs[0].ndim == 1 else 144
        self._keep = arrs
        return [a.ctypes.data_as(C.c_void_p) for a in arrs] + [C.c_int64(stride)]

    def adaptive_dt(self, prec_day, ta_day, pet_day):
        lib().oc_adaptive_dt(self._ptrs, C.c_int64(self.n), *self._forc(prec_day, ta_day, pet_day),
                             C.byref(self.scal), C.byref(self.settings))

    def interception(self):
        lib().oc_interception(self._ptrs, C.c_int64(self.n), C.byref(self.settings))

    def evapotranspiration(self):
        lib().oc_evapotranspiration(self._ptrs, C.c_int64(self.n), C.byref(self.settings))

    def snow(self):
        lib().oc_snow(self._ptrs, C.c_int64(self.n), C.byref(self.scal), C.byref(self.settings))

    def infiltration(self):
        lib().oc_infiltration(self._ptrs, C.c_int64(self.n), C.byref(self.scal), C.byref(self.settings))

    def subsurface_runoff(self):
        lib().oc_subsurface_runoff(self._ptrs, C.c_int64(self.n), C.byref(self.scal), C.byref(self.settings))

    def surface_routing(self):
        """surface_runoff.calculate_surface_runoff (settings.enable_routing_1D)"""
        lib().oc_surface_routing(self._ptrs, C.c_int64(self.n), C.byref(self.scal), C.byref(self.settings))

    def subsurface_routing(self):
        """the routing part of subsurface_runoff.calculate_subsurface_runoff"""
        lib().oc_subsurface_routing(self._ptrs, C.c_int64(self.n), C.byref(self.settings))

    def capillary_rise(self):
        lib().oc_capillary_rise(self._ptrs, C.c_int64(self.n), C.byref(self.scal))

    def storage(self):
        lib().oc_storage(self._ptrs, C.c_int64(self.n), C.byref(self.scal))

    def num_error(self):
        return bool(lib().oc_num_error(self._ptrs, C.c_int64(self.n), C.byref(self.scal), C.byref(self.settings)))

    def after_timestep(self):
        lib().oc_after_timestep(self._ptrs, C.c_int64(self.n), C.byref(self.scal), C.byref(self.settings))

    def params_surface(self):
        lib().oc_params_surface(self._ptrs, C.c_int64(self.n), C.byref(self.scal), *self._lut_args())

    def topo(self):
        lib().oc_topo(self._ptrs, C.c_int64(self.n))

    def params_soil(self):
        lib().oc_params_soil(self._ptrs, C.c_int64(self.n), C.byref(self.settings), *self._lut_args())

    def params_lateral(self, mlms):
        self._mlms = np.ascontiguousarray(mlms, dtype=np.float64)
        lib().oc_params_lateral(self._ptrs, C.c_int64(self.n), self._mlms.ctypes.data_as(C.c_void_p),
                                C.c_int64(self._mlms.shape[0]))

    def initial_conditions(self):
        lib().oc_initial_conditions(self._ptrs, C.c_int64(self.n))

    def step(self, prec_day, ta_day, pet_day, monthly=False):
        return bool(lib().oc_step(self._ptrs, C.c_int64(self.n), *self._forc(prec_day, ta_day, pet_day),
                                  C.byref(self.scal), C.byref(self.settings), C.c_int(int(monthly)),
                                  *self._lut_args()))


class ForcingDriver:
    """Host-side `set_forcing` / `set_parameters` hooks of the benchmark setup
    (benchmarks/SVAT_benchmark.py:105-110,151-171) for the oracle."""

    def __init__(self, forcing, weights=None, stations=None):
        self.F = forcing
        self.day = None
        self.weights = weights   # dict(prec_weight, ta_offset, pet_weight) per column: the distributed catchment setups' set_forcing
        self.stations = stations   # golden_util.load_stations: several series, a row index per column (svat_dist.py:274-310)

    def day_slice(self, i):
        day = tuple(self.F[k][i:i + 144].copy() for k in ("PREC", "TA", "PET"))
        if self.stations is not None:
            st, idx = self.stations, self.stations["station_index"]
            rows = [np.where(idx[:, None] >= 0, st[k][np.maximum(idx, 0), i:i + 144], 0.0) for k in ("PREC", "TA", "PET")]
            w = self.weights or dict(prec_weight=np.ones(idx.size), ta_offset=np.zeros(idx.size), pet_weight=np.ones(idx.size))
            return (rows[0] * w["prec_weight"][:, None], rows[1] + w["ta_offset"][:, None], rows[2] * w["pet_weight"][:, None])
        if self.weights is not None:   # examples/catchment_scale/eberbaechle/svat_distributed/svat.py:276-296
            w = self.weights
            day = (day[0][None, :] * w["prec_weight"][:, None], day[1][None, :] + w["ta_offset"][:, None],
                   day[2][None, :] * w["pet_weight"][:, None])
        return day

    def before_step(self, st):
        """Returns (prec_day, ta_day, pet_day, monthly)."""
        s = st.scal
        if s.time % 86400 == 0:
            s.itt_day = 0
            i = s.itt_forc
            s.year[1] = int(self.F["YEAR"][i])
            s.month[1] = int(self.F["MONTH"][i])
            s.doy[1] = int(self.F["DOY"][i])
            self.day = self.day_slice(i)
            s.itt_forc = i + 144
        monthly = (s.month[1] != s.month[0]) and (s.itt > 1)
        return (*self.day, monthly)
