"""Checkpoint / restart of a model state on the hip backend: roger/restart.py's `write_restart` / `read_restart` in the reference's
file layout -- an HDF5 file whose group "core" holds one dataset per variable under the variable's name, in the reference's GLOBAL
array shape (2-cell ghost frame included, trailing `timesteps` axis for the two-level variables, 0-d datasets for the scalars;
roger/restart.py:32-67, 129-174), so that states can go back and forth between the two implementations.

* What is written: group "core" = every `write_to_restart` variable of the reference that this path holds
  (`REFERENCE_RESTART_VARIABLES`, read off roger/variables.py) and nothing else -- the reference's reader walks every dataset of
  "core" and fails on a name that is not on its list (roger/restart.py:15-16); group "hip_core" = every other variable of the arena:
  the reference's own list does not carry the whole state of a step (`prec`, `ta`, the event bookkeeping of the wetting fronts ... are
  missing from it), with the rest of the planes a restarted run continues BIT FOR BIT (tests/test_hip_restart.py).  Exchanged with
  the reference in both directions: tests/test_restart_interchange.py.
* The device-side output accumulators (rh_diag_*) go into the group "hip_diag": a restart in the middle of an output interval keeps
  the partial sums.
* Several ranks (num_proc = (N, 1)): the slabs are gathered to rank 0, which writes ONE file with the global arrays like the
  reference's MPI-IO write (roger/restart.py:43-64); every rank reads its own chunk, ghost frame included (`get_chunk_slices(...,
  include_overlap=True)`, :20-24).
* Container: h5py where it is installed; otherwise `roger_amd.h5lite`, a small classic-HDF5 writer / reader (contiguous datasets;
  it reads the chunked + gzip files h5py writes for the reference).

Driven like the reference: `settings.restart_input_filename` is read at the end of `setup()` (roger/roger.py:324-326),
`settings.restart_output_filename` (Python format syntax over the variables and settings, :156-158) is written every
`settings.restart_frequency` seconds at the start of a step (:385-386) and, with `settings.write_restart`, at the end of `run()`
(:578-579); `runtime_settings.diskless_mode` suppresses writing (:134)."""
import os

import numpy as np

from . import distributed, h5lite, logger, runtime_settings as rs, runtime_state as rst

# roger/variables.py: the variables with write_to_restart=True that exist on this path (the others belong to modules that are out of
# scope: crops, film flow, groundwater, routing, transport)
REFERENCE_RESTART_VARIABLES = (
    "maskCatch", "maskRiver", "maskLake", "lu_id", "sealing", "z0", "slope", "slope_per", "S_dep_tot", "ground_cover", "lai",
    "throughfall_coeff_top", "throughfall_coeff_ground", "basal_evap_coeff", "k_stress_evap", "evap_coeff", "basal_transp_coeff",
    "k_stress_transp", "transp_coeff", "S_int_top_tot", "S_int_ground_tot", "S", "S_snow", "swe", "swe_top", "swe_top_tot", "swe_ground",
    "S_dep", "S_sur", "S_int_top", "S_int_ground", "z_soil", "c_root", "clay", "sand", "lmpv", "lmpv_non_sat", "z_sc", "z_sc_non_sat",
    "dmpv", "mp_drain_area", "dmph", "theta_ufc", "theta_ac", "theta_pwp", "theta_sat", "theta_fc", "theta_27", "theta_4", "theta_6",
    "S_ufc_s", "S_ac_s", "S_pwp_s", "S_sat_s", "S_fc_s", "wfs", "ks", "ks_ss", "ha", "lambda_bc", "m_bc", "n_salv", "z_sc_max", "kf",
    "theta", "theta_d", "theta_d_fp", "theta_d_rel", "theta_d_rel_t0", "theta_d_t0", "theta_d_t1", "S_s", "S_fp_s", "S_lp_s", "S_zsat",
    "k", "h", "y_mp", "y_sc", "z_wf", "z_wf_t0", "z_wf_t1", "z_wf_fc", "z_sat_layer_1", "z_sat_layer_2", "z_sat_layer_3", "z_sat_layer_4",
    "z_sat_layer_5", "z_sat_layer_6", "z_sat_layer_7", "z_sat_layer_8", "z_sat", "z_root", "z_evap", "tew", "rew", "de", "k_rz", "h_rz",
    "theta_rz", "theta_fp_rz", "theta_lp_rz", "S_rz", "S_fp_rz", "S_lp_rz", "S_pwp_rz", "S_fc_rz", "S_sat_rz", "S_ufc_rz", "S_ac_rz",
    "k_ss", "h_ss", "theta_ss", "theta_fp_ss", "theta_lp_ss", "S_ss", "S_fp_ss", "S_lp_ss", "S_zsat_ss", "S_pwp_ss", "S_fc_ss",
    "S_sat_ss", "S_ufc_ss", "S_ac_ss", "z_gw", "no_wf", "pi_gr", "pi_m", "t_sat", "Fs", "Fs_t0", "inf_mat_pot_event_csum",
    "inf_mat_event_csum", "inf_mp_event_csum", "inf_sc_event_csum", "evap_int", "prec_event_csum", "t_event_csum", "c_int",
    # scalars (0-d datasets)
    "time", "time_event0", "itt", "itt_day", "itt_forc", "year", "month", "doy", "dt", "dt_secs", "event_id", "event_id_counter",
    # the forcing series and the grid (the reference restarts them too)
    "PREC", "TA", "PET", "YEAR", "MONTH", "DOY", "x", "y",
)
# ... and the rest of the reference's list for these models: variables of modules outside this path (crops, film flow, radiation) and
# registry constants.  The reference's reader insists on finding EVERY variable of its list (roger/restart.py:93-97), so a file that
# the reference is to accept carries them with the values the reference itself writes for an SVAT / oneD run: its initial values
# (read off a restart file of the reference, tests/golden/restart/reference_restart.h5).  name: (dims, dtype, value)
REFERENCE_ONLY_RESTART_VARIABLES = {
    **{k: (("x", "y"), "float64", 0.0) for k in (
        "Fs_t1", "c1_mak", "c2_mak", "dmpv_crop", "frac_fp", "frac_lp", "lmpv_crop", "t_sat_t0", "t_sat_t1", "theta_ac_ss", "theta_d_rel_t1",
        "theta_eff", "theta_fp", "theta_lp", "theta_pc", "theta_water_stress", "water_stress_fp")},
    "zroot_to_zsoil_max": (("x", "y"), "float64", 0.75),
    "maskUrban": (("x", "y"), "bool", 0),
    "RS": (("t_forc",), "float64", 0.0),
    **{k: ((), "int64", v) for k, v in dict(itt_cr=0, itt_substep=0, tau=1, taum1=0, taup1=2, tau_event=0, time_for_diag=86400).items()},
    "z": (("z",), "float64", 0.0),
}
# on the reference's list only with settings.enable_lateral_flow (oneD model)
_LATERAL_ONLY = ("dmph",) + tuple(f"z_sat_layer_{i}" for i in range(1, 9))
_NOT_STATE = ("tau", "taup1", "taum1")   # constants of the registry
MORE_GROUP = "hip_core"   # the variables of the arena that are not on the reference's restart list
# inputs that set_forcing_setup puts back at setup(), not state: the station series and the calendar (at 10^6 columns the series of a
# distributed setup and the per-cell day arrays below are gigabytes)
_FORCING_SERIES = ("PREC_DIST", "TA_DIST", "PET_DIST")   # (the single station's series are on the reference's own list: kept)
# the day's 144 slots per cell: set_forcing re-derives them from itt_forc at midnight, so they are state only in the middle of a day
_FORCING_DAY = ("prec_day", "ta_day", "pet_day")


def _h5py():
    try:
        import h5py

        return h5py
    except ImportError:
        return None


def _write_file(path, groups):
    h5py = _h5py()
    if h5py is None:
        return h5lite.write(path, groups)
    with h5py.File(path, "w") as f:
        for gname, dsets in groups.items():
            g = f.require_group(gname)
            for k, a in dsets.items():
                a = np.asarray(a)
                g.create_dataset(k, data=a.astype(np.int8) if a.dtype == np.bool_ else a)


def _read_file(path):
    h5py = _h5py()
    if h5py is None:
        return h5lite.read(path)
    with h5py.File(path, "r") as f:
        return {gname: {k: np.asarray(v) for k, v in g.items()} for gname, g in f.items()}


def _restart_filename(state, template):
    statedict = {}
    for key in state.var_meta:
        try:
            v = np.asarray(getattr(state.variables, key))
        except Exception:   # noqa: BLE001
            continue
        if v.ndim == 0:
            statedict[key] = v.item()
    statedict.update({k: getattr(state.settings, k) for k in state.settings.__metadata__})
    return template.format(**statedict)


def _gather_global(state, key, local):
    """The global array (ghost frame included) of an (x, y, ...) variable from the ranks' slabs, on rank 0; None elsewhere."""
    import torch.distributed as dist

    world, rank = rst.proc_num, rst.proc_rank
    parts = [None] * world if rank == 0 else None
    dist.gather_object(np.asarray(local), parts, dst=0)
    if rank != 0:
        return None
    nx, ny = state.settings.nx, state.settings.ny
    out = np.zeros((nx + 4, ny + 4) + local.shape[2:], dtype=local.dtype)
    for r, part in enumerate(parts):
        (gx, gy), (lx, ly) = distributed.get_chunk_slices(nx, ny, rs.num_proc, r, include_overlap=True)
        out[gx, gy] = part[lx, ly]
    return out


def collect(state):
    """{group: {name: array}} of everything a restart file holds."""
    vs = state.variables
    core = {}
    multi = rst.proc_num > 1
    # The day's 3 x (x, y, 144) forcing arrays are left out at midnight ONLY where the script's set_forcing is known to re-derive them
    # there (the stock hook, or one recognised as doing the same: roger_amd/hooks.py); a set_forcing of the script's own that does
    # not must find the arrays it left.
    midnight = int(np.asarray(vs.time)) % 86400 == 0 and bool(getattr(state, "_stock_set_forcing", False))
    more = {}
    for key, var in state.var_meta.items():
        if key in _NOT_STATE or key in _FORCING_SERIES or (midnight and key in _FORCING_DAY):
            continue
        val = np.asarray(getattr(vs, key))
        if multi and var.dims is not None and tuple(var.dims[:2]) == ("x", "y"):
            val = _gather_global(state, key, val)
        # the reference's reader walks EVERY dataset of "core" and looks it up in its own restart list (roger/restart.py:15-16:
        # a KeyError for anything else), so "core" holds exactly that list; the rest of the arena goes into a group of its own,
        # which the reference never opens
        on_list = key in REFERENCE_RESTART_VARIABLES and (state.settings.enable_lateral_flow or key not in _LATERAL_ONLY)
        (core if on_list else more)[key] = val
    sizes = dict(x=state.settings.nx + 4, y=state.settings.ny + 4, t_forc=state.settings.nitt_forc, z=1)
    for key, (dims, dtype, value) in REFERENCE_ONLY_RESTART_VARIABLES.items():
        core.setdefault(key, np.full(tuple(sizes[d] for d in dims), value, dtype=dtype))
    groups = {"core": core, MORE_GROUP: more}
    # one group per diagnostic of the reference that has restart variables (roger/restart.py:100-126 opens every one of them): the
    # counters of the monitors and of the average diagnostic, as the reference writes them when those diagnostics are not in use
    groups.update({"average": {"average_nitts": np.float64(0.0)}, "snapshot": {},
                   "tracer_monitor": {"C_s": np.float64(0.0), "M_S": np.float64(0.0)},
                   "water_monitor": {"S_s": np.float64(0.0), "S_sur": np.float64(0.0)}})
    ctx = state.backend_context
    active = getattr(state, "_diag_active", None)
    if active and not getattr(state, "_diag_transport", False) and not multi and hasattr(ctx, "diag_slot_times"):
        # the accumulators of the output interval the run is in (roger_amd/diagnostics.py; completed intervals were written out)
        iv = state._diag_interval
        slot = int(int(np.asarray(vs.time)) // iv) % state._diag_slots
        names = list(dict.fromkeys(v for d in active for v in d.output_variables))
        t0, t1 = ctx.diag_slot_times(slot)
        groups["hip_diag"] = {nm: ctx.diag_download(nm, slot) for nm in names}
        groups["hip_diag"]["slot_state"] = np.array([slot, ctx.diag_steps(slot), t0, t1, state._diag_written_day], dtype=np.int64)
    return groups


def write_restart(state, force=False, filename=None):
    """roger/restart.py:129-174.  Returns the file name (None if nothing was due)."""
    vs, settings = state.variables, state.settings
    if rs.diskless_mode and filename is None:
        return None
    template = filename or settings.restart_output_filename
    if not template:
        return None
    write_now = force or filename is not None or (
        settings.restart_frequency > 0 and int(np.asarray(vs.itt)) > 0 and int(np.asarray(vs.time)) % settings.restart_frequency == 0)
    if not write_now:
        return None
    fname = _restart_filename(state, str(template))
    groups = collect(state)
    if rst.proc_rank == 0:
        logger.info(f"Writing restart file {fname}")
        _write_file(fname, groups)
    return fname


def read_restart(state, filename=None):
    """roger/restart.py:70-126: assigns every variable found in the group "core" (shape- and dtype-checked by RogerVariables like any
    assignment); the device copies are refreshed before the next native call."""
    settings = state.settings
    template = filename or settings.restart_input_filename
    if not template:
        return None
    if rs.force_overwrite:
        raise RuntimeError("To prevent data loss, force_overwrite cannot be used in restart runs")
    fname = _restart_filename(state, str(template))
    if not os.path.isfile(fname):
        raise IOError(f"restart file {fname} not found")
    logger.info(f"Reading restart data from {fname}")
    groups = _read_file(fname)
    if "core" not in groups:
        raise RuntimeError(f"{fname} has no group 'core': not a RoGeR restart file")
    core = dict(groups["core"])
    core.update(groups.get(MORE_GROUP, {}))   # a file of the reference has "core" only: the rest keeps what setup() left
    vs = state.variables
    missing = [k for k in REFERENCE_RESTART_VARIABLES if k in state.var_meta and k not in core and k not in ("dmph",) + tuple(
        f"z_sat_layer_{i}" for i in range(1, 9)) + ("slope", "slope_per")]
    if missing:
        raise RuntimeError(f"No restart data found for variable {missing[0]} in {fname}")
    nx, ny = settings.nx, settings.ny
    (gx, gy), (lx, ly) = distributed.get_chunk_slices(nx, ny, rs.num_proc, rst.proc_rank, include_overlap=True)
    with vs.unlock():
        for key, val in core.items():
            if key not in state.var_meta or key in _NOT_STATE:
                continue   # a variable of a module this path does not have
            var = state.var_meta[key]
            val = np.asarray(val)
            if var.dims is None:
                setattr(vs, key, val.item() if val.ndim == 0 else val)
                continue
            if tuple(var.dims[:2]) == ("x", "y") and rst.proc_num > 1:
                local = np.array(getattr(vs, key))
                local[lx, ly] = val[gx, gy]
                val = local
            if np.asarray(getattr(vs, key)).dtype == np.bool_:
                val = val.astype(np.bool_)
            setattr(vs, key, val)
    active = getattr(state, "_diag_active", None)
    if active and "hip_diag" in groups and rst.proc_num == 1 and not getattr(state, "_diag_transport", False):
        ctx = state.backend_context
        slot, steps, t0, t1, written = (int(v) for v in groups["hip_diag"]["slot_state"])
        for nm in dict.fromkeys(v for d in active for v in d.output_variables):
            if nm in groups["hip_diag"]:
                ctx.diag_upload(nm, slot, groups["hip_diag"][nm])
        ctx.diag_set_slot_state(slot, steps, t0, t1)
        state._diag_written_day = written
    return fname
