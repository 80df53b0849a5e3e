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
