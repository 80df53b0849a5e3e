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


# default parameter table of the ready-made model (the values of roger/models/svat/svat.py:93-117; a dict instead of one assignment per
# parameter: subclasses and callers override entries, see `parameters=`)
DEFAULT_PARAMETERS = dict(lu_id=8, sealing=0, S_dep_tot=0, z_soil=2000, dmpv=50, lmpv=500, theta_ac=0.1, theta_ufc=0.1, theta_pwp=0.2,
                          ks=5, kf=2500)
DEFAULT_INITIAL_THETA = dict(theta_rz=0.3, theta_ss=0.3)
LOOK_UP_TABLES = dict(lut_ilu="ARR_ILU", lut_gc="ARR_GC", lut_gcm="ARR_GCM", lut_rdlu="ARR_RDLU")
FORCING_SERIES = ("PREC", "TA", "PET", "YEAR", "MONTH", "DOY")
DAY = 24 * 60 * 60
SLOTS = 6 * 24


def _fill_interior(vs, table, levels=False):
    """vs.<name>[2:-2, 2:-2(, :taup1)] = value for every entry of a parameter table (scalars or (nx, ny) arrays)."""
    for name, value in table.items():
        where = at[2:-2, 2:-2, : vs.taup1] if levels else at[2:-2, 2:-2]
        setattr(vs, name, update(getattr(vs, name), where, value))


class SVATSetup(RogerSetup):
    """A SVAT model: uniform default parameters (`DEFAULT_PARAMETERS`), forcing series in memory, the benchmark's forcing hooks
    (one station broadcast to all cells; monthly surface parameters)."""

    parameters = DEFAULT_PARAMETERS
    initial_theta = DEFAULT_INITIAL_THETA
    look_up_tables = LOOK_UP_TABLES
    identifier = "SVAT"

    def __init__(self, forcing=None, nx=1, ny=1, ndays=None, override=None, parameters=None):
        super().__init__(override=override)
        self._forcing = forcing
        self._nx, self._ny = nx, ny
        self._ndays = ndays if ndays is not None else (len(forcing["PREC"]) // SLOTS if forcing is not None else 0)
        if parameters:
            self.parameters = dict(self.parameters, **parameters)

    @roger_routine
    def set_settings(self, state):
        settings = state.settings
        for name, value in dict(identifier=self.identifier, output_frequency=DAY, nx=self._nx, ny=self._ny, runlen=DAY * self._ndays,
                                nitt_forc=len(self._forcing["PREC"]), dx=1, dy=1, x_origin=0.0, y_origin=0.0,
                                enable_groundwater_boundary=False, enable_macropore_lower_boundary_condition=False,
                                enable_adaptive_time_stepping=True).items():
            setattr(settings, name, value)

    @roger_routine
    def read_data(self, state):
        pass

    @roger_routine
    def set_grid(self, state):
        vs, settings = state.variables, state.settings
        for axis, origin, spacing in (("x", settings.x_origin, settings.dx), ("y", settings.y_origin, settings.dy)):
            d = update(allocate(state.dimensions, (axis,)), at[:], spacing)
            setattr(vs, axis, update(getattr(vs, axis), at[3:-2], origin + npx.cumsum(d[3:-2])))   # distance from the origin

    @roger_routine
    def set_look_up_tables(self, state):
        vs = state.variables
        for name, table in self.look_up_tables.items():
            setattr(vs, name, update(getattr(vs, name), at[:, :], getattr(lut, table)))

    @roger_routine
    def set_topography(self, state):
        pass

    @roger_routine
    def set_parameters_setup(self, state):
        _fill_interior(state.variables, self.parameters)

    @roger_routine
    def set_parameters(self, state):
        vs = state.variables
        if (vs.month[vs.tau] != vs.month[vs.taum1]) & (vs.itt > 1):    # a new month: the land-use dependent surface parameters
            vs.update(calc_parameters_surface_kernel(state))

    @roger_routine
    def set_initial_conditions_setup(self, state):
        pass

    @roger_routine
    def set_initial_conditions(self, state):
        _fill_interior(state.variables, self.initial_theta, levels=True)

    @roger_routine
    def set_boundary_conditions_setup(self, state):
        pass

    @roger_routine
    def set_boundary_conditions(self, state):
        pass

    @roger_routine
    def set_forcing_setup(self, state):
        vs = state.variables
        for name in FORCING_SERIES:
            setattr(vs, name, update(getattr(vs, name), at[:], self._forcing[name]))

    @roger_routine
    def set_forcing(self, state):
        vs = state.variables
        if vs.time % DAY:
            return
        first = vs.itt_forc                       # midnight: the calendar entry and the next 144 ten-minute slots
        vs.itt_day = 0
        for name, series in (("year", vs.YEAR), ("month", vs.MONTH), ("doy", vs.DOY)):
            setattr(vs, name, update(getattr(vs, name), at[1], series[first]))
        for name, series in (("prec_day", vs.PREC), ("ta_day", vs.TA), ("pet_day", vs.PET)):
            setattr(vs, name, update(getattr(vs, name), at[:, :, :], series[npx.newaxis, npx.newaxis, first:first + SLOTS]))
        vs.itt_forc = first + SLOTS

    @roger_routine
    def set_diagnostics(self, state):
        pass

    @roger_routine
    def after_timestep(self, state):
        vs = state.variables
        vs.update(after_timestep_kernel(state))


# The per-step hooks above are what the device-side control part performs itself (roger_hip.hip: hooks_set_forcing / ctrl_wave, the
# monthly pipeline of k_step, the rotation inside the fused kernel): RogerSetup.run() advances on the device as long as a setup
# script does not override them (roger_amd/roger.py: device_run_possible).
for _hook in (SVATSetup.read_data, SVATSetup.set_boundary_conditions, SVATSetup.set_forcing, SVATSetup.set_parameters, SVATSetup.after_timestep):
    _hook.device_equivalent = True


@roger_kernel
def after_timestep_kernel(state):
    """tau -> taum1 rotation of the prognostic variables (roger/models/svat/svat.py:187-384);
    replaced by the native rh_after_timestep."""
    raise AssertionError("replaced by the native entry point (routines.NATIVE_KERNELS)")
