"""Which of a setup script's per-step hooks does the device's control part perform itself?  Decided by BEHAVIOUR.

Every script the reference ships defines its own `set_forcing`, `set_parameters` and `after_timestep`
(benchmarks/SVAT_benchmark.py:105-110, 152-181; roger/models/svat/svat.py:113-185): inheritance from the ready-made model
classes says nothing about them.  What the device-side control part of the fused step does (roger_hip.hip: hooks_set_forcing,
ctrl_wave, the monthly pipeline of k_step, the rotation inside the kernel) is exactly

    read_data, set_boundary_conditions   nothing
    set_forcing       at midnight: itt_day = 0, year / month / doy [tau] from YEAR / MONTH / DOY [itt_forc], the day's 144 slots of
                      PREC / TA / PET (of the cell's station, times / plus the cell's weights) into prec_day / ta_day / pet_day,
                      itt_forc += 144; otherwise nothing
    set_parameters    calc_parameters_surface_kernel iff month[tau] != month[taum1] and itt > 1
    after_timestep    after_timestep_kernel, unconditionally

A hook is run against the state in RECORDING mode (`RogerVariables._probe`): reads are logged and the scalars it branches on are
the probe's, assignments are logged and NOT applied, natively implemented kernels are logged and NOT run.  A hook is the device's
own if, at every probe point, it reads nothing outside the variables the stock hook may look at, calls exactly the kernels above
and assigns exactly the values above (bit for bit).  Anything else -- a hook that assigns a variable of its own, calls a host
kernel, reads a storage -- keeps the hook on the host (tests/test_host_package.py, tests/test_hip_host_package.py).

The probes CALL a hook a few times outside the time loop (at the first run() / step()), so only a hook that cannot have effects
outside the state is probed at all: in front of the probes stands a static screen of the hook's code object -- every global and
attribute name it mentions (its own and those of nested code) must be one the stock hook of that kind could mention (`update`,
`at`, the variables above ...).  A hook that appends to a list, logs, reads a file or calls anything else of its own mentions a name
outside that set and is the host's without having been run.  `RogerSetup.recognise_hooks = False` or RH_NO_HOOK_RECOGNITION=1
switches the recognition off; hooks inherited from the ready-made model classes are marked and never probed.
"""
import os

import numpy as np

from . import variables as var_mod
from .state import DeviceResult

DAY = 86400
SLOTS = 144
LEVELS = ("tau", "taum1", "taup1")
CALENDAR = (("year", "YEAR"), ("month", "MONTH"), ("doy", "DOY"))
DAY_ARRAYS = (("prec_day", "PREC", "prec_weight"), ("ta_day", "TA", "ta_offset"), ("pet_day", "PET", "pet_weight"))
FORCING_READS = frozenset(
    LEVELS + ("time", "itt_forc", "itt_day", "itt", "year", "month", "doy", "prec_day", "ta_day", "pet_day", "YEAR", "MONTH", "DOY",
              "PREC", "TA", "PET", "prec_weight", "ta_offset", "pet_weight", "station_id", "station_ids", "PREC_DIST", "TA_DIST", "PET_DIST"))
FORCING_WRITES = frozenset(("itt_day", "itt_forc", "year", "month", "doy", "prec_day", "ta_day", "pet_day"))


# names a hook of each kind may mention (code.co_names: globals and attributes); anything else and the hook is not even probed
_COMMON_NAMES = frozenset(("variables", "settings") + LEVELS)
_OPERATOR_NAMES = frozenset(("update", "at", "npx", "np", "numpy", "newaxis", "where", "zeros", "shape", "enumerate", "getattr", "setattr", "slice",
                             "range", "int", "len"))
STATIC_NAMES = {
    "read_data": _COMMON_NAMES,
    "set_boundary_conditions": _COMMON_NAMES,
    "set_parameters": _COMMON_NAMES | {"month", "itt", "update", "calc_parameters_surface_kernel"},
    "after_timestep": _COMMON_NAMES | {"update", "after_timestep_kernel"},
    "set_forcing": _COMMON_NAMES | _OPERATOR_NAMES | FORCING_READS,
}


def _code_names(code):
    names = set(code.co_names)
    for c in code.co_consts:
        if hasattr(c, "co_names"):
            names |= _code_names(c)
    return names


def statically_plain(hook, kind):
    """The hook's code mentions nothing but what a stock hook of its kind mentions (it may then be probed without side effects)."""
    fn = getattr(hook, "__wrapped__", hook)
    code = getattr(fn, "__code__", None)
    return code is not None and _code_names(code) <= STATIC_NAMES[kind]


class Probe:
    """One recording session: `values` stand in for the variables the hook branches on."""

    def __init__(self, values):
        self.values = dict(values)
        self.reads = set()
        self.writes = {}
        self.calls = []


def record_kernel_call(state, name, native_entry):
    """Called by the @roger_kernel wrapper / run_native while a probe is active: the kernel is logged, not run."""
    state._variables._probe.calls.append(native_entry or ("host:" + name))
    return DeviceResult(())


def _scalar_values(vs):
    """The device scalars as they are now (one read-back), as probe values: a probed hook never touches the device."""
    s = vs._get_scalars()
    vals = {k: (float(getattr(s, k)) if k == "dt" else int(getattr(s, k))) for k in var_mod.DEVICE_SCALARS}
    for k in var_mod.DEVICE_SCALARS2:
        a = np.array(list(getattr(s, k)), dtype=np.int64)
        a.flags.writeable = False
        vals[k] = a
    return vals


def _run(hook, setup, state, values):
    """The undecorated hook against the recording state; None if it raised (a hook that cannot run out of turn is the host's)."""
    vs = state.variables
    probe = Probe(values)
    fn = getattr(hook, "__wrapped__", hook)   # (RogerSetup.read_data is a plain method)
    object.__setattr__(vs, "_probe", probe)
    try:
        with vs.unlock():
            fn(setup, state)
    except Exception:
        return None
    finally:
        object.__setattr__(vs, "_probe", None)
    return probe


def _int2(a, b):
    v = np.array([a, b], dtype=np.int64)
    v.flags.writeable = False
    return v


def _is_noop(setup, state, hook, base):
    for time in (0, 600, DAY):
        p = _run(hook, setup, state, dict(base, time=time))
        if p is None or p.reads - set(LEVELS) or p.writes or p.calls:
            return False
    return True


def _is_stock_set_parameters(setup, state, hook, base):
    allowed = set(LEVELS) | {"month", "itt"}
    for month, itt, expect in (((3, 4), 7, True), ((12, 1), 2, True), ((4, 4), 7, False), ((3, 4), 1, False), ((3, 4), 0, False)):
        p = _run(hook, setup, state, dict(base, month=_int2(*month), itt=itt))
        if p is None or p.reads - allowed or p.writes:
            return False
        if p.calls != (["rh_params_surface"] if expect else []):
            return False
    return True


def _is_stock_after_timestep(setup, state, hook, base):
    for itt in (1, 5):
        p = _run(hook, setup, state, dict(base, itt=itt))
        if p is None or p.reads - set(LEVELS) or p.writes or p.calls != ["rh_after_timestep"]:
            return False
    return True


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.array_equal(a, b, equal_nan=a.dtype.kind == "f"))


def _same_day(got, want):
    """got: the interior (nx, ny, 144) of what the hook assigned; want: (1, 1, 144) or (nx, ny, 144).  One series broadcast to all
    cells (zero strides, core/operators.update) is compared as one series."""
    if want.shape[:2] == (1, 1) and got.strides[0] == 0 and got.strides[1] == 0:
        return _same(got[0, 0], want[0, 0])
    return _same(got, np.broadcast_to(want, got.shape))


def _expected_day(vs, settings, series_name, weight_name, first, add):
    """What the device-side hook stages for a day, on the interior: the cell's station series (one series, or row station index of
    (n_stations, nitt) -- zeros for a cell without a station), times / plus the cell's weight."""
    sl = slice(first, first + SLOTS)
    w = np.asarray(getattr(vs, weight_name))[2:-2, 2:-2, np.newaxis]
    if w.size and bool((w == w.flat[0]).all()):
        w = w[:1, :1]                      # one weight for all cells: the day stays one (1, 1, 144) series
    if settings.enable_distributed_input:
        ids = np.asarray(vs.station_ids)
        cell = np.asarray(vs.station_id)[2:-2, 2:-2]
        rows = np.asarray(getattr(vs, series_name + "_DIST"))
        day = np.zeros(cell.shape + (SLOTS,))
        for i, sid in enumerate(ids):
            day = np.where((cell == sid)[:, :, np.newaxis], rows[i, sl][np.newaxis, np.newaxis, :], day)
    else:
        day = np.asarray(getattr(vs, series_name))[np.newaxis, np.newaxis, sl]
    return day + w if add else day * w


def _is_stock_set_forcing(setup, state, hook, base):
    vs, settings = state.variables, state.settings
    nitt = int(settings.nitt_forc)
    if nitt < SLOTS:
        return False
    # between two midnights: nothing
    for time in (600, DAY + 3600):
        p = _run(hook, setup, state, dict(base, time=time, itt_forc=SLOTS))
        if p is None or p.reads - FORCING_READS or p.writes or p.calls:
            return False
    # at midnight: the stock hook's assignments, value by value, on the first, the second and the last whole day of the series
    days = sorted({0, min(1, nitt // SLOTS - 1), nitt // SLOTS - 1})
    for d in days:
        first = d * SLOTS
        cal = {k: _int2(1900 + d, 1901 + d) for k, _ in CALENDAR}
        p = _run(hook, setup, state, dict(base, time=d * DAY, itt_forc=first, itt_day=5, **cal))
        if p is None or p.reads - FORCING_READS or p.calls or set(p.writes) != FORCING_WRITES:
            return False
        if int(p.writes["itt_day"]) != 0 or int(p.writes["itt_forc"]) != first + SLOTS:
            return False
        for k, series in CALENDAR:
            if not _same(p.writes[k], _int2(1900 + d, int(np.asarray(getattr(vs, series))[first]))):
                return False
        for k, series, weight in DAY_ARRAYS:
            got = np.asarray(p.writes[k])
            if got.ndim != 3 or got.shape[2] != SLOTS:
                return False
            want = _expected_day(vs, settings, series, weight, first, add=(k == "ta_day"))
            if not _same_day(got[2:-2, 2:-2, :], want):
                return False
    return True


_CHECKS = {
    "read_data": _is_noop,
    "set_boundary_conditions": _is_noop,
    "set_forcing": _is_stock_set_forcing,
    "set_parameters": _is_stock_set_parameters,
    "after_timestep": _is_stock_after_timestep,
}


def classify(setup, names):
    """{hook name: True if the device's control part performs this hook itself}."""
    state = setup.state
    out = {}
    enabled = getattr(setup, "recognise_hooks", True) and not os.environ.get("RH_NO_HOOK_RECOGNITION")
    base = None
    for name in names:
        hook = getattr(type(setup), name, None)
        if getattr(hook, "device_equivalent", False):   # the ready-made model classes' own hooks (roger_amd/models)
            out[name] = True
            continue
        if not enabled or hook is None or state.settings.enable_offline_transport:
            out[name] = False
            continue
        if not statically_plain(hook, name):
            out[name] = False
            continue
        if base is None:
            state.variables.flush_to_device()
            base = _scalar_values(state.variables)
        out[name] = bool(_CHECKS[name](setup, state, hook, base))
    return out
