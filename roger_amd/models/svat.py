"""`SVATSetup`: the ready-made SVAT model class (roger/models/svat/svat.py:12-185) with the forcing
held in memory instead of being read from `input/forcing.nc` (netCDF I/O is outside the hot path).

Subclass it and override `set_settings` / `set_parameters_setup` / `set_initial_conditions`, or
pass `forcing=` (dict with PREC, TA, PET, YEAR, MONTH, DOY at 10-minute resolution, e.g. from
roger_amd.forcing) and `nx`, `ny`, `ndays`.
"""
from .. import KernelOutput, RogerSetup, roger_kernel, roger_routine  # noqa: F401
from .. import lookuptables as lut
from ..core.operators import at, numpy as npx, update
from ..core.surface import calc_parameters_surface_kernel
from ..variables import allocate


class SVATSetup(RogerSetup):
    """A SVAT model."""

    def __init__(self, forcing=None, nx=1, ny=1, ndays=None, override=None):
        super().__init__(override=override)
        self._forcing = forcing
        self._nx, self._ny = nx, ny
        self._ndays = ndays if ndays is not None else (len(forcing["PREC"]) // 144 if forcing is not None else 0)

    @roger_routine
    def set_settings(self, state):
        settings = state.settings
        settings.identifier = "SVAT"
        settings.output_frequency = 86400
        settings.nx, settings.ny = self._nx, self._ny
        settings.runlen = 24 * 60 * 60 * self._ndays
        settings.nitt_forc = len(self._forcing["PREC"])
        settings.dx = 1
        settings.dy = 1
        settings.x_origin = 0.0
        settings.y_origin = 0.0
        settings.enable_groundwater_boundary = False
        settings.enable_macropore_lower_boundary_condition = False
        settings.enable_adaptive_time_stepping = True

    @roger_routine
    def read_data(self, state):
        pass

    @roger_routine
    def set_grid(self, state):
        vs = state.variables
        settings = state.settings
        dx = allocate(state.dimensions, ("x",))
        dx = update(dx, at[:], settings.dx)
        dy = allocate(state.dimensions, ("y",))
        dy = update(dy, at[:], settings.dy)
        vs.x = update(vs.x, at[3:-2], settings.x_origin + npx.cumsum(dx[3:-2]))
        vs.y = update(vs.y, at[3:-2], settings.y_origin + npx.cumsum(dy[3:-2]))

    @roger_routine
    def set_look_up_tables(self, state):
        vs = state.variables
        vs.lut_ilu = update(vs.lut_ilu, at[:, :], lut.ARR_ILU)
        vs.lut_gc = update(vs.lut_gc, at[:, :], lut.ARR_GC)
        vs.lut_gcm = update(vs.lut_gcm, at[:, :], lut.ARR_GCM)
        vs.lut_rdlu = update(vs.lut_rdlu, at[:, :], lut.ARR_RDLU)

    @roger_routine
    def set_topography(self, state):
        pass

    @roger_routine
    def set_parameters_setup(self, state):
        vs = state.variables
        vs.lu_id = update(vs.lu_id, at[2:-2, 2:-2], 8)
        vs.sealing = update(vs.sealing, at[2:-2, 2:-2], 0)
        vs.S_dep_tot = update(vs.S_dep_tot, at[2:-2, 2:-2], 0)
        vs.z_soil = update(vs.z_soil, at[2:-2, 2:-2], 2000)
        vs.dmpv = update(vs.dmpv, at[2:-2, 2:-2], 50)
        vs.lmpv = update(vs.lmpv, at[2:-2, 2:-2], 500)
        vs.theta_ac = update(vs.theta_ac, at[2:-2, 2:-2], 0.1)
        vs.theta_ufc = update(vs.theta_ufc, at[2:-2, 2:-2], 0.1)
        vs.theta_pwp = update(vs.theta_pwp, at[2:-2, 2:-2], 0.2)
        vs.ks = update(vs.ks, at[2:-2, 2:-2], 5)
        vs.kf = update(vs.kf, at[2:-2, 2:-2], 2500)

    @roger_routine
    def set_parameters(self, state):
        vs = state.variables
        if (vs.month[vs.tau] != vs.month[vs.taum1]) & (vs.itt > 1):
            vs.update(calc_parameters_surface_kernel(state))

    @roger_routine
    def set_initial_conditions_setup(self, state):
        pass

    @roger_routine
    def set_initial_conditions(self, state):
        vs = state.variables
        vs.theta_rz = update(vs.theta_rz, at[2:-2, 2:-2, : vs.taup1], 0.3)
        vs.theta_ss = update(vs.theta_ss, at[2:-2, 2:-2, : vs.taup1], 0.3)

    @roger_routine
    def set_boundary_conditions_setup(self, state):
        pass

    @roger_routine
    def set_boundary_conditions(self, state):
        pass

    @roger_routine
    def set_forcing_setup(self, state):
        vs = state.variables
        F = self._forcing
        for k in ("PREC", "TA", "PET", "YEAR", "MONTH", "DOY"):
            setattr(vs, k, update(getattr(vs, k), at[:], F[k]))

    @roger_routine
    def set_forcing(self, state):
        vs = state.variables
        condt = vs.time % (24 * 60 * 60) == 0
        if condt:
            vs.itt_day = 0
            vs.year = update(vs.year, at[1], vs.YEAR[vs.itt_forc])
            vs.month = update(vs.month, at[1], vs.MONTH[vs.itt_forc])
            vs.doy = update(vs.doy, at[1], vs.DOY[vs.itt_forc])
            sl = slice(vs.itt_forc, vs.itt_forc + 6 * 24)
            vs.prec_day = update(vs.prec_day, at[:, :, :], vs.PREC[npx.newaxis, npx.newaxis, sl])
            vs.ta_day = update(vs.ta_day, at[:, :, :], vs.TA[npx.newaxis, npx.newaxis, sl])
            vs.pet_day = update(vs.pet_day, at[:, :, :], vs.PET[npx.newaxis, npx.newaxis, sl])
            vs.itt_forc = vs.itt_forc + 6 * 24

    @roger_routine
    def set_diagnostics(self, state):
        pass

    @roger_routine
    def after_timestep(self, state):
        vs = state.variables
        vs.update(after_timestep_kernel(state))


@roger_kernel
def after_timestep_kernel(state):
    """tau -> taum1 rotation of the prognostic variables (roger/models/svat/svat.py:187-384);
    replaced by the native rh_after_timestep."""
    raise AssertionError("replaced by the native entry point (routines.NATIVE_KERNELS)")
