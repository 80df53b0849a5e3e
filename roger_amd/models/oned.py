"""`ONEDSetup`: the 1D model (roger/models/oneD/oneD.py): SVAT plus lateral subsurface runoff in
eight 200 mm layers (`settings.enable_lateral_flow`).  Forcing in memory, as in SVATSetup."""
from .. import roger_kernel, roger_routine
from ..core.operators import at, update
from .svat import DEFAULT_PARAMETERS, LOOK_UP_TABLES, SVATSetup

# the 1D model's defaults (roger/models/oneD/oneD.py:130-160): the SVAT table with a slope, horizontal macropores and longer vertical ones
ONED_PARAMETERS = dict(DEFAULT_PARAMETERS, slope=0.05, dmph=50, lmpv=1000)


class ONEDSetup(SVATSetup):
    """A 1D model."""

    parameters = ONED_PARAMETERS
    look_up_tables = dict(LOOK_UP_TABLES, lut_mlms="ARR_MLMS")
    identifier = "ONED"

    @roger_routine
    def set_settings(self, state):
        SVATSetup.set_settings.__wrapped__(self, state)
        state.settings.enable_lateral_flow = True

    @roger_routine
    def set_parameters_setup(self, state):
        SVATSetup.set_parameters_setup.__wrapped__(self, state)
        vs = state.variables
        vs.slope_per = update(vs.slope_per, at[2:-2, 2:-2], vs.slope[2:-2, 2:-2] * 100)   # slope in per cent: the row of lut_mlms

    @roger_routine
    def after_timestep(self, state):
        vs = state.variables
        vs.update(after_timestep_kernel(state))


ONEDSetup.after_timestep.device_equivalent = True   # the oneD rotation is the fused oneD kernel's (see models/svat.py)


@roger_kernel
def after_timestep_kernel(state):
    """models/oneD/oneD.py: tau -> taum1 rotation (no snapping of the pore storages); native."""
    raise AssertionError("replaced by the native entry point (routines.NATIVE_KERNELS)")
