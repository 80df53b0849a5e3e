"""Variable registry of the SVAT path, mirroring roger/variables.py (names, dims, dtypes, initial
values, 2-cell ghost frame on x and y)."""
import numpy as np

from . import _native


class Variable:
    def __init__(self, name, dims, units="", long_description="", dtype=None, initial=None, plane=None):
        self.name, self.dims, self.units, self.long_description = name, dims, units, long_description
        self.dtype, self.initial = dtype, initial
        self.plane = plane  # name of the device plane backing it (None: host-only variable)
        self.active = True


CATCH_GRID = ("x", "y")
TIMESTEPS = ("timesteps",)
GHOST_DIMENSIONS = ("x", "y")

# roger/variables.py:101-131
DIM_TO_SHAPE_VAR = {
    "x": "nx", "y": "ny", "t_forc": "nitt_forc", "timesteps": 2, "timesteps_day": 6 * 24,
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


def _arena_variables():
    out = {}
    table = _native.plane_table()
    names = [n for n, _ in table]
    for name, is_int in table:
        if name.endswith("_m1") and name[:-3] in names:
            continue
        two = (name + "_m1") in names
        dims = CATCH_GRID + TIMESTEPS if two else CATCH_GRID
        dtype = np.bool_ if name in _BOOL else (np.int32 if is_int else None)
        out[name] = Variable(name, dims, dtype=dtype, initial=_INITIAL.get(name), plane=name)
    return out


def build_variables():
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
    V.update(_arena_variables())
    return V


DEVICE_SCALARS = ("time", "itt", "itt_day", "itt_forc", "time_event0", "dt_secs", "event_id_counter", "dt")
DEVICE_SCALARS2 = ("event_id", "year", "month", "doy")
