"""Model settings: the names, types and defaults of roger/settings.py that the SVAT / oneD / offline-transport path uses
(setup scripts assign them in `set_settings`), and the scope check of the native backend."""
from collections import namedtuple

Setting = namedtuple("Setting", ("default", "type", "group"))


def optional(type_):
    """None stays None, anything else is converted."""
    return lambda arg: arg if arg is None else type_(arg)


PI = 3.14159265358979323846264338327950588

# name: (default, type); grouped by what they are for
_GROUPS = {
    "run": {   # grid, run length, clock
        "identifier": ("UNNAMED", str), "nx": (1, int), "ny": (1, int), "nz": (1, int), "dx": (1, int),
        "dy": (1, int), "dz": (1, int), "nitt": (1, int), "nitt_forc": (1, int), "ages": (1, int), "nages": (2, int),
        "nsas": (8, int), "nstations": (2, int), "runlen": (0.0, float), "runlen_warmup": (0.0, float), "x_origin": (0, float),
        "y_origin": (0, float), "time_origin": ("1900-01-01 00:00:00", str), "output_frequency": (0.0, float),
    },
    "const": {   # constants the kernels use (device copy: rh_config)
        "pi": (PI, float), "r_mp": (2.5, float), "l_sc": (10000, float), "sf": (3, float), "ta_fm": (0, float),
        "rmax": (30, float), "end_event": (21600, int), "hpi": (5, int), "a_bc": (2, int), "b_bc": (2, int),
        "clay_min": (0.01, float), "clay_max": (0.71, float), "theta_rew_min": (0.02, float),
        "theta_rew_max": (0.24, float), "zroot_to_zsoil_max": (0.7, float), "rew_min": (2, float),
        "rew_max": (12, float), "z_evap_max": (150, float), "transp_water_stress": (0.75, float),
        "atol": (1e-2, float), "rtol": (1e-2, float),
    },
    "tracer": {   # offline transport: isotope constants, tracer switches, SAS solver
        "VSMOW_conc18O": (2005.2e-6, float), "d18O_min": (-20, float), "d18O_max": (0, float),
        "VSMOW_conc2H": (155.76e-6, float), "d2H_min": (-160, float), "d2H_max": (0, float),
        "enable_bromide": (False, bool), "enable_chloride": (False, bool), "enable_oxygen18": (False, bool),
        "enable_deuterium": (False, bool), "enable_nitrate": (False, bool), "enable_virtualtracer": (False, bool),
        "tm_structure": ("UNNAMED", str), "enable_age_statistics": (False, bool),
        "sas_solver": (None, optional(str)), "sas_solver_substeps": (1, int),
        "h": (1.0, float),   # temporal increment of the explicit SAS solvers as a fraction of the time step (roger/settings.py:121)
    },
    "switch": {   # process switches (what is native: check_setting_conflicts)
        "enable_distributed_input": (False, bool), "enable_film_flow": (False, bool),
        "enable_lateral_flow": (False, bool), "enable_crop_phenology": (False, bool),
        "enable_net_irrigation": (False, bool), "enable_soil_compaction": (False, bool),
        "enable_offline_transport": (False, bool), "enable_groundwater_boundary": (False, bool),
        "enable_groundwater": (False, bool), "enable_routing_1D": (False, bool), "enable_routing_2D": (False, bool),
        "enable_macropore_lower_boundary_condition": (False, bool), "enable_adaptive_time_stepping": (False, bool),
        "warmup_done": (False, bool),
        # roger/settings.py:117-133
        "write_restart": (False, bool), "restart_input_filename": (None, optional(str)),
        "restart_output_filename": ("{identifier}_{itt:0>4d}.restart.h5", optional(str)), "restart_frequency": (0, float),
    },
}
SETTINGS = {name: Setting(default, type_, group) for group, table in _GROUPS.items() for name, (default, type_) in table.items()}

_UNSUPPORTED_SWITCHES = (
    "enable_film_flow", "enable_crop_phenology",
    "enable_net_irrigation", "enable_soil_compaction", "enable_groundwater_boundary",
    "enable_groundwater", "enable_routing_2D", "enable_macropore_lower_boundary_condition",
    "enable_nitrate",
)


def check_setting_conflicts(settings):
    """roger/settings.py:139-171 (conflict checks) + the scope of the native backend."""
    for name in _UNSUPPORTED_SWITCHES:
        if getattr(settings, name):
            raise NotImplementedError(
                f"settings.{name}=True is outside the hot path of the hip backend (SURVEY.md section 8: out of scope)")
    if settings.enable_routing_1D and not settings.enable_lateral_flow:
        raise NotImplementedError("settings.enable_routing_1D routes the lateral subsurface runoff: it needs settings.enable_lateral_flow "
                                  "(examples/hillslope_scale/oneD_distributed_routing_tutorial/oneD.py:82-83)")
    if settings.enable_offline_transport:
        # the native transport path: oxygen-18, deuterium, bromide, chloride or the virtual tracer with any of the three SAS solvers
        # 
        if sum(int(getattr(settings, k)) for k in ("enable_oxygen18", "enable_deuterium", "enable_bromide", "enable_chloride",
                                                   "enable_virtualtracer")) != 1:
            raise NotImplementedError("offline transport on the hip backend is implemented for oxygen-18, deuterium, bromide, chloride "
                                      "and the virtual tracer (exactly one of settings.enable_oxygen18 / enable_deuterium / "
                                      "enable_bromide / enable_chloride / enable_virtualtracer must be True)")
        if settings.sas_solver not in ("deterministic", "Euler", "RK4"):
            raise NotImplementedError(f'settings.sas_solver={settings.sas_solver!r}: the SAS solvers are "deterministic" and the '
                                      'explicit "Euler" and "RK4" (roger/settings.py:119)')
        if settings.sas_solver in ("Euler", "RK4"):
            # benchmarks/SVATOXYGEN18_benchmark.py:30-31: the increment of the numerical solver is the length of a sub-step
            if abs(settings.h * settings.sas_solver_substeps - 1) > 1e-12:
                raise ValueError("settings.h must be 1 / settings.sas_solver_substeps for the explicit solvers")
        if settings.nages != settings.ages + 1:
            raise ValueError("settings.nages must be settings.ages + 1")
        return
    if not settings.enable_adaptive_time_stepping:
        raise NotImplementedError("the hip backend implements the adaptive time stepping of the SVAT benchmarks "
                                  "(settings.enable_adaptive_time_stepping must be True)")
