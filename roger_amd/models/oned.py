"""`ONEDSetup`: the 1D model (roger/models/oneD/oneD.py): SVAT plus lateral subsurface runoff in
eight 200 mm layers (`settings.enable_lateral_flow`).  Forcing in memory, as in SVATSetup."""
from .. import lookuptables as lut
from .. import roger_kernel, roger_routine
from ..core.operators import at, update
from .svat import SVATSetup


class ONEDSetup(SVATSetup):
    """A 1D model."""

    @roger_routine
    def set_settings(self, state):
        SVATSetup.set_settings.__wrapped__(self, state)
        settings = state.settings
        settings.identifier = "ONED"
        settings.enable_lateral_flow = True

    @roger_routine
    def set_look_up_tables(self, state):
        SVATSetup.set_look_up_tables.__wrapped__(self, state)
        vs = state.variables
        vs.lut_mlms = update(vs.lut_mlms, at[:, :], lut.ARR_MLMS)

    @roger_routine
    def set_parameters_setup(self, state):
        vs = state.variables
        # roger/models/oneD/oneD.py:130-160
        vs.lu_id = update(vs.lu_id, at[2:-2, 2:-2], 8)
        vs.sealing = update(vs.sealing, at[2:-2, 2:-2], 0)
        vs.slope = update(vs.slope, at[2:-2, 2:-2], 0.05)
        vs.slope_per = update(vs.slope_per, at[2:-2, 2:-2], vs.slope[2:-2, 2:-2] * 100)
        vs.S_dep_tot = update(vs.S_dep_tot, at[2:-2, 2:-2], 0)
        vs.z_soil = update(vs.z_soil, at[2:-2, 2:-2], 2000)
        vs.dmpv = update(vs.dmpv, at[2:-2, 2:-2], 50)
        vs.dmph = update(vs.dmph, at[2:-2, 2:-2], 50)
        vs.lmpv = update(vs.lmpv, at[2:-2, 2:-2], 1000)
        vs.theta_ac = update(vs.theta_ac, at[2:-2, 2:-2], 0.1)
        vs.theta_ufc = update(vs.theta_ufc, at[2:-2, 2:-2], 0.1)
        vs.theta_pwp = update(vs.theta_pwp, at[2:-2, 2:-2], 0.2)
        vs.ks = update(vs.ks, at[2:-2, 2:-2], 5)
        vs.kf = update(vs.kf, at[2:-2, 2:-2], 2500)

    @roger_routine
    def after_timestep(self, state):
        vs = state.variables
        vs.update(after_timestep_kernel(state))


@roger_kernel
def after_timestep_kernel(state):
    """models/oneD/oneD.py: tau -> taum1 rotation (no snapping of the pore storages); native."""
    raise AssertionError("replaced by the native entry point (routines.NATIVE_KERNELS)")
