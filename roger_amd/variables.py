"""Variable registry of the SVAT path, mirroring roger/variables.py (names, dims, dtypes, initial
values, 2-cell ghost frame on x and y)."""
import numpy as np

from . import _native


class Variable:
    def __init__(self, name, dims, units="", long_description="", dtype=None, initial=None, plane=None, sas=None):
        self.name, self.dims, self.units, self.long_description = name, dims, units, long_description
        self.dtype, self.initial = dtype, initial
        self.plane = plane  # name of the device plane backing it (None: host-only variable)
        self.sas = sas      # name of the array of the SAS context backing it (offline transport)
        self.active = True


CATCH_GRID = ("x", "y")
TIMESTEPS = ("timesteps",)
GHOST_DIMENSIONS = ("x", "y")

# roger/variables.py:101-131
DIM_TO_SHAPE_VAR = {
    "x": "nx", "y": "ny", "t_forc": "nitt_forc", "timesteps": 2, "timesteps_day": 6 * 24,
    "t": "nitt", "ages": "ages", "nages": "nages", "n_sas_params": "nsas", "n_stations": "nstations",
    "n_lu": 25, "n_sealing": 101, "n_slope": 10000, "n_params2": 2, "n_params7": 7, "n_params9": 9, "n_params13": 13,
}


def get_shape(dimensions, grid, include_ghosts=True, local=True):
    """roger/variables.py:152-186: sizes of `grid` dims; x and y are the *local* chunk + 4 ghosts."""
    from . import runtime_settings

    if grid is None:
        return ()
    px, py = runtime_settings.num_proc
    sizes = dict(dimensions)
    if local:
        if "x" in sizes:
            sizes["x"] = sizes["x"] // px
        if "y" in sizes:
            sizes["y"] = sizes["y"] // py
    if include_ghosts:
        for d in GHOST_DIMENSIONS:
            if d in sizes:
                sizes[d] += 4
    shape = []
    for d in grid:
        if isinstance(d, int):
            shape.append(d)
        elif d in sizes:
            shape.append(sizes[d])
        else:
            raise ValueError(f"unrecognized dimension {d}")
    return tuple(shape)


def allocate(dimensions, grid, dtype=None, include_ghosts=True, local=True, fill=0):
    """roger/variables.py:6423-6435: host array of the reference's shape (read-only, like there)."""
    if dtype is None:
        dtype = np.float64
    out = np.full(get_shape(dimensions, grid, include_ghosts=include_ghosts, local=local), fill, dtype=dtype)
    out.flags.writeable = False
    return out


_INITIAL = {"maskCatch": True, "ta": 15, "z_gw": 1000, "c_int": 1.0, "c_root": 1.0}
_BOOL = {"maskCatch", "maskRiver", "maskLake"}


def _arena_variables(routing=True):
    out = {}
    table = _native.plane_table()
    names = [n for n, _ in table]
    for name, is_int in table:
        if name.endswith("_m1") and name[:-3] in names:
            continue
        if not routing and name in _native.ROUTING_PLANES:   # active with settings.enable_routing_1D only, as in roger/variables.py
            continue
        two = (name + "_m1") in names
        dims = CATCH_GRID + TIMESTEPS if two else CATCH_GRID
        dtype = np.bool_ if name in _BOOL else (np.int32 if is_int else None)
        out[name] = Variable(name, dims, dtype=dtype, initial=_INITIAL.get(name), plane=name)
    return out


def build_variables(settings=None):
    V = {}
    i64 = np.int64
    # scalars (roger/variables.py:189-330); the time-stepping ones live in the device's rh_scalars
    for name, initial in (("tau", 1), ("taup1", 2), ("taum1", 0), ("time", 0), ("itt", 0), ("itt_day", 0),
                          ("itt_forc", 0), ("time_event0", 0), ("dt_secs", 3600), ("event_id_counter", 1),
                          ("time_for_diag", 0)):
        V[name] = Variable(name, None, dtype=i64, initial=initial)
    V["dt"] = Variable("dt", None, initial=1)
    for name, initial in (("year", 1900), ("month", 1), ("doy", 1), ("event_id", 0)):
        V[name] = Variable(name, TIMESTEPS, dtype=i64, initial=initial)
    V["x"] = Variable("x", ("x",))
    V["y"] = Variable("y", ("y",))
    V["lut_ilu"] = Variable("lut_ilu", ("n_lu", "n_params13"))
    V["lut_gc"] = Variable("lut_gc", ("n_lu", "n_params13"))
    V["lut_gcm"] = Variable("lut_gcm", ("n_lu", "n_params2"))
    V["lut_is"] = Variable("lut_is", ("n_sealing", "n_params2"))
    V["lut_rdlu"] = Variable("lut_rdlu", ("n_lu", "n_params7"))
    V["lut_mlms"] = Variable("lut_mlms", ("n_slope", "n_params9"))
    for name in ("PREC", "TA", "PET"):
        V[name] = Variable(name, ("t_forc",))
    for name in ("YEAR", "MONTH", "DOY"):
        V[name] = Variable(name, ("t_forc",), dtype=i64)
    for name in ("prec_day", "ta_day", "pet_day"):
        V[name] = Variable(name, CATCH_GRID + ("timesteps_day",))
    # per-cell weights of the station forcing (roger/variables.py:938, 3543, 4153) and, with settings.enable_distributed_input, the
    # stations themselves (:882-916, 3522, 4138, 6383-6402); host-side: the user's set_forcing hook reads them, and
    # RogerSetup.enable_device_hooks hands them to the device (rh_set_forcing_weights, rh_set_forcing_stations)
    V["prec_weight"] = Variable("prec_weight", CATCH_GRID, initial=1.0)
    V["ta_offset"] = Variable("ta_offset", CATCH_GRID, initial=0.0)
    V["pet_weight"] = Variable("pet_weight", CATCH_GRID, initial=1.0)
    if settings is not None and settings.enable_distributed_input:
        V["station_ids"] = Variable("station_ids", ("n_stations",), dtype=i64)
        V["station_id"] = Variable("station_id", CATCH_GRID, dtype=i64)
        for name in ("PREC_DIST", "TA_DIST", "PET_DIST"):
            V[name] = Variable(name, ("n_stations", "t_forc"))
    # catchment boundary of the routing setups (oneD_distributed_routing_tutorial/oneD.py:133-197): the user's set_topography derives
    # outer_boundary (a plane of the arena) from it on the host
    routing = settings is None or bool(settings.enable_routing_1D)
    if routing:
        V["inner_boundary"] = Variable("inner_boundary", CATCH_GRID, dtype=np.int32)
    V.update(_arena_variables(routing))
    return V


SAS_FLUXES = ("evap_soil", "transp", "q_rz", "q_ss", "cpr_rz")
SAS_STAT_TARGETS = (("transp", "tt"), ("q_ss", "tt"), ("rz", "rt"), ("ss", "rt"), ("s", "rt"))


def build_transport_variables(settings):
    """Registry of the offline oxygen-18 transport model (roger/variables.py, the variables that are `active`
    for enable_offline_transport & enable_oxygen18 and that the deterministic path touches).  Variables with
    `sas=` live in the SAS context on the device; the others are host-only (time series read by the user's
    set_forcing hook, storages, snow signal)."""
    V = {}
    i64 = np.int64
    for name, initial in (("tau", 1), ("taup1", 2), ("taum1", 0), ("time", 0), ("itt", 0), ("dt_secs", 3600),
                          ("time_for_diag", 0), ("itt_substep", 0)):
        V[name] = Variable(name, None, dtype=i64, initial=initial)
    V["dt"] = Variable("dt", None, initial=1)
    V["x"] = Variable("x", ("x",))
    V["y"] = Variable("y", ("y",))
    V["ages"] = Variable("ages", ("ages",))
    V["nages"] = Variable("nages", ("nages",))
    AG, NAG = ("ages",), ("nages",)
    V["maskCatch"] = Variable("maskCatch", CATCH_GRID, dtype=np.bool_, initial=True, sas="maskCatch")
    for st in ("rz", "ss"):
        V[f"sa_{st}"] = Variable(f"sa_{st}", CATCH_GRID + TIMESTEPS + AG, sas=f"sa_{st}")
        V[f"msa_{st}"] = Variable(f"msa_{st}", CATCH_GRID + TIMESTEPS + AG, sas=f"msa_{st}")
    V["sa_s"] = Variable("sa_s", CATCH_GRID + TIMESTEPS + AG, sas="sa_s")
    V["msa_s"] = Variable("msa_s", CATCH_GRID + TIMESTEPS + AG, sas="msa_s")
    for st in ("rz", "ss", "s"):
        V[f"SA_{st}"] = Variable(f"SA_{st}", CATCH_GRID + TIMESTEPS + NAG)       # host-only diagnostic
        V[f"csa_{st}"] = Variable(f"csa_{st}", CATCH_GRID + TIMESTEPS + AG)      # host-only diagnostic
        V[f"C_{st}"] = Variable(f"C_{st}", CATCH_GRID + TIMESTEPS, sas=f"C_{st}")
        V[f"C_iso_{st}"] = Variable(f"C_iso_{st}", CATCH_GRID + TIMESTEPS, sas=f"C_iso_{st}")
        V[f"S_{st}"] = Variable(f"S_{st}", CATCH_GRID + TIMESTEPS)
    for f in SAS_FLUXES:
        V[f] = Variable(f, CATCH_GRID, sas=f)
        V[f"sas_params_{f}"] = Variable(f"sas_params_{f}", CATCH_GRID + ("n_sas_params",), sas=f"sas_params_{f}")
        V[f"tt_{f}"] = Variable(f"tt_{f}", CATCH_GRID + AG, sas=f"tt_{f}")
        V[f"mtt_{f}"] = Variable(f"mtt_{f}", CATCH_GRID + AG, sas=f"mtt_{f}")
        V[f"TT_{f}"] = Variable(f"TT_{f}", CATCH_GRID + NAG, sas=f"TT_{f}")
        V[f"C_{f}"] = Variable(f"C_{f}", CATCH_GRID, sas=f"C_{f}")
        V[f"C_iso_{f}"] = Variable(f"C_iso_{f}", CATCH_GRID, sas=f"C_iso_{f}")
    for f in ("inf_mat_rz", "inf_pf_rz", "inf_pf_ss"):
        V[f] = Variable(f, CATCH_GRID, sas=f)
        V[f"C_{f}"] = Variable(f"C_{f}", CATCH_GRID, sas=f"C_{f}")
        V[f"C_iso_{f}"] = Variable(f"C_iso_{f}", CATCH_GRID, sas=f"C_iso_{f}")
    V["C_in"] = Variable("C_in", CATCH_GRID, sas="C_in")
    V["C_iso_in"] = Variable("C_iso_in", CATCH_GRID)
    for name in ("prec", "S_snow", "C_snow", "C_iso_snow"):
        V[name] = Variable(name, CATCH_GRID + TIMESTEPS)
    for name in ("S_pwp_rz", "S_pwp_ss", "S_sat_rz", "S_sat_ss"):
        V[name] = Variable(name, CATCH_GRID)
    for name in ("S_rz_init", "S_ss_init"):
        V[name] = Variable(name, CATCH_GRID, sas=name)
    for name in ("PREC_DIST_DAILY", "INF_MAT_RZ", "INF_PF_RZ", "INF_PF_SS", "TRANSP", "EVAP_SOIL", "CPR_RZ", "Q_RZ", "Q_SS",
                 "S_RZ", "S_SS", "S_S", "S_SNOW", "C_IN", "C_ISO_IN"):
        V[name] = Variable(name, CATCH_GRID + ("t",))
    if settings.enable_age_statistics:
        for w, p in SAS_STAT_TARGETS:
            for q in ("10", "25", "50", "75", "90", "avg"):
                V[f"{p}{q}_{w}"] = Variable(f"{p}{q}_{w}", CATCH_GRID, sas=f"{p}{q}_{w}")
    if settings.enable_bromide or settings.enable_chloride or settings.enable_virtualtracer:
        # the anion kernels' variables (roger/variables.py:534, 2114-2122, 2474, 2536, 4071, 4324, 4666, 5377-5405, 5602)
        V["S_sat_rz"] = Variable("S_sat_rz", CATCH_GRID, sas="S_sat_rz")
        V["lu_id"] = Variable("lu_id", CATCH_GRID, dtype=i64, sas="lu_id")
        for name in ("alpha_transp", "alpha_q"):
            V[name] = Variable(name, CATCH_GRID, initial=1, sas=name)
        for f in SAS_FLUXES + ("inf_mat_rz", "inf_pf_rz", "inf_pf_ss"):
            V[f"M_{f}"] = Variable(f"M_{f}", CATCH_GRID, sas=f"M_{f}")
        for st in ("rz", "ss", "s"):
            V[f"M_{st}"] = Variable(f"M_{st}", CATCH_GRID + TIMESTEPS, sas=f"M_{st}")
        V["M_in"] = Variable("M_in", CATCH_GRID)
    return V


DEVICE_SCALARS = ("time", "itt", "itt_day", "itt_forc", "time_event0", "dt_secs", "event_id_counter", "dt")
DEVICE_SCALARS2 = ("event_id", "year", "month", "doy")
