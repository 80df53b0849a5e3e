"""roger/core/surface.py counterparts that setup scripts import."""
from ..routines import roger_kernel, roger_routine, run_native


@roger_kernel
def calc_parameters_surface_kernel(state):
    """Land-use x month surface parameters (roger/core/surface.py:74-343); runs rh_params_surface."""
    raise AssertionError("replaced by the native entry point (routines.NATIVE_KERNELS)")


@roger_routine
def calculate_parameters(state):
    """roger/core/surface.py:385-395"""
    run_native(state, "rh_topo")
    run_native(state, "rh_params_surface")


@roger_routine
def calculate_initial_conditions(state):
    """roger/core/surface.py:417-427 -- fused with the soil initial conditions (rh_initial_conditions)."""
    pass
