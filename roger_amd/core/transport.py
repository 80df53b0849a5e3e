"""Offline transport on the hip backend, mirroring the entry points of roger/core/transport.py that a setup
script and `RogerSetup.step` use: `delta_to_conc`, `conc_to_delta` (host helpers for the user hooks) and
`calculate_storage_selection` (the native SAS step).
"""
import numpy as np

from .. import _native
from ..routines import roger_routine
from ..variables import SAS_FLUXES, SAS_STAT_TARGETS


def _isotope_constants(settings):
    if settings.enable_deuterium:
        return settings.VSMOW_conc2H, settings.d2H_min, settings.d2H_max
    return settings.VSMOW_conc18O, settings.d18O_min, settings.d18O_max


def delta_to_conc(state, delta_iso):
    """roger/core/transport.py:315-325 (oxygen-18, deuterium)."""
    v = _isotope_constants(state.settings)[0]
    delta_iso = np.asarray(delta_iso, dtype=np.float64)
    return v * (delta_iso / 1000. + 1.) / (1. + (delta_iso / 1000. + 1.) * v)


def conc_to_delta(state, conc):
    """roger/core/transport.py:328-340 (oxygen-18, deuterium)."""
    v, lo, hi = _isotope_constants(state.settings)
    conc = np.asarray(conc, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = 1000. * (conc / (v * (1. - conc)) - 1.)
        return np.where((d < lo) | (d > hi), np.nan, d)


def _written_by_step(settings):
    if settings.enable_bromide or settings.enable_chloride or settings.enable_virtualtracer:   # the anion kernels: solute masses instead of delta values, no signal of the soil evaporation
        names = ["sa_rz", "msa_rz", "sa_ss", "msa_ss", "sa_s", "msa_s", "C_rz", "C_ss", "C_s", "M_rz", "M_ss", "M_s"]
        for f in SAS_FLUXES:
            names += [f"tt_{f}", f"TT_{f}"] + ([f"mtt_{f}", f"C_{f}", f"M_{f}"] if f != "evap_soil" or settings.enable_virtualtracer else [])
        for f in ("inf_mat_rz", "inf_pf_rz", "inf_pf_ss"):
            names += [f"C_{f}", f"M_{f}"]
        if settings.enable_age_statistics:
            names += [f"{p}{q}_{w}" for w, p in SAS_STAT_TARGETS for q in ("10", "25", "50", "75", "90", "avg")]
        return names
    names = ["sa_rz", "msa_rz", "sa_ss", "msa_ss", "sa_s", "msa_s", "C_rz", "C_ss", "C_s", "C_iso_rz", "C_iso_ss", "C_iso_s"]
    for f in SAS_FLUXES:
        names += [f"tt_{f}", f"mtt_{f}", f"TT_{f}", f"C_{f}", f"C_iso_{f}"]
    for f in ("inf_mat_rz", "inf_pf_rz", "inf_pf_ss"):
        names += [f"C_{f}", f"C_iso_{f}"]
    if settings.enable_age_statistics:
        names += [f"{p}{q}_{w}" for w, p in SAS_STAT_TARGETS for q in ("10", "25", "50", "75", "90", "avg")]
    return names


@roger_routine
def calculate_storage_selection(state):
    """roger/core/transport.py:3136 for `(enable_oxygen18, enable_deuterium, enable_bromide or enable_chloride) and sas_solver == "deterministic"`:
    svat_transport_model_deterministic (:949-991) as one native launch (rh_sas_step).  What the user's
    set_forcing hook assigned (vs.inf_mat_rz, ..., vs.C_in) is uploaded first; results stay on the device until a
    `vs.<name>` is read.

    `sas_solver == "Euler"` / `"RK4"` (:3220-3304, isotopes and anions): the loop over the sub-steps -- svat_transport_model_euler / _rk4, the storages, the age
    statistics, the ageing at the end of the day -- is one native launch as well; the model time advances here, by the sub-steps'
    `int(dt_secs / substeps)` (the reference ages the water when that reaches a full day, so the sub-steps must divide the day).
    The variables hold the values after the last sub-step (the reference's per-sub-step `write_output` is not reproduced)."""
    vs = state.variables
    settings = state.settings
    sas = state.sas_context
    if sas is None:
        raise RuntimeError("calculate_storage_selection needs settings.enable_offline_transport")
    vs.flush_to_device()
    if settings.sas_solver in ("Euler", "RK4"):
        dt_secs = int(np.asarray(vs.dt_secs))
        if dt_secs % settings.sas_solver_substeps or dt_secs != 24 * 60 * 60:
            raise NotImplementedError("the explicit solvers on the hip backend need daily steps that the sub-steps divide evenly "
                                      f"(dt_secs = {dt_secs}, sas_solver_substeps = {settings.sas_solver_substeps})")
        with vs.unlock():
            vs.time = vs.time + dt_secs
    sas.step(0)
    vs.mark_device_newer(_written_by_step(state.settings))


SAS_STAGES = _native.SAS_STAGES
