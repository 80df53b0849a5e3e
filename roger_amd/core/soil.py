"""roger/core/soil.py counterparts used by RogerSetup.setup()."""
from ..routines import roger_routine, run_native
from . import numerics


@roger_routine
def calculate_parameters(state):
    """roger/core/soil.py:727-739"""
    numerics.validate_parameters_soil(state)
    run_native(state, "rh_params_soil")
    if state.settings.enable_lateral_flow:
        run_native(state, "rh_params_lateral")   # calc_parameters_lateral_flow_kernel, soil.py:560-641


@roger_routine
def calculate_initial_conditions(state):
    """roger/core/soil.py:995-1010 (+ surface.calc_initial_conditions_surface_kernel)"""
    numerics.validate_initial_conditions_soil(state)
    run_native(state, "rh_initial_conditions")


def rescale_SA(state):
    """roger/core/soil.py:1650-1671: rescale_sa_msa_iso_soil_kernel (oxygen-18) or the bromide branch of
    rescale_sa_msa_anion_soil_kernel as one native launch (rh_sas_stages with RH_SAS_RESCALE)."""
    from .. import _native

    sas = state.sas_context
    if sas is None:
        raise RuntimeError("rescale_SA needs settings.enable_offline_transport")
    vs = state.variables
    vs.flush_to_device()
    sas.stages(0, _native.SAS_STAGES["RESCALE"])
    vs.mark_device_newer(["sa_rz", "sa_ss", "msa_rz", "msa_ss", "sa_s", "msa_s", "C_rz", "C_ss", "C_s", "C_iso_rz", "C_iso_ss", "C_iso_s",
                          "M_rz", "M_ss", "M_s"])
