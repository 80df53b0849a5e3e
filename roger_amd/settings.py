"""Model settings of the SVAT path, mirroring roger/settings.py (same names, defaults, types)."""
from collections import namedtuple

Setting = namedtuple("setting", ("default", "type", "description"))


def optional(type_):
    def wrapped(arg):
        return arg if arg is None else type_(arg)

    return wrapped


PI = 3.14159265358979323846264338327950588

SETTINGS = {
    "identifier": Setting("UNNAMED", str, "Identifier of the current simulation"),
    "nx": Setting(1, int, "Grid points in zonal (x) direction"),
    "ny": Setting(1, int, "Grid points in meridional (y) direction"),
    "nz": Setting(1, int, "Grid points in vertical (z) direction"),
    "dx": Setting(1, int, "Zonal grid spacing"),
    "dy": Setting(1, int, "Meridional grid spacing"),
    "dz": Setting(1, int, "Vertical grid spacing"),
    "nitt": Setting(1, int, "Number of total iterations"),
    "nitt_forc": Setting(1, int, "Number of total iterations of forcing"),
    "ages": Setting(1, int, "Number of water ages"),
    "nages": Setting(2, int, "Number of water ages to calculate cumulated distributions"),
    "nsas": Setting(8, int, "Number of entries per grid cell containing SAS parameters"),
    "runlen": Setting(0.0, float, "Length of simulation in seconds"),
    "runlen_warmup": Setting(0.0, float, "Length of warmup simulation in seconds"),
    "x_origin": Setting(0, float, "Grid origin in x-direction"),
    "y_origin": Setting(0, float, "Grid origin in y-direction"),
    "time_origin": Setting("1900-01-01 00:00:00", str, "time origin"),
    "output_frequency": Setting(0.0, float, "Time between output"),
    # physical constants used by the kernels (roger/settings.py:52-80)
    "pi": Setting(PI, float, "Pi"),
    "r_mp": Setting(2.5, float, "Macropore radius in mm"),
    "l_sc": Setting(10000, float, "Total length of shrinkage cracks in mm/m^2"),
    "sf": Setting(3, float, "Degree-day factor in -"),
    "ta_fm": Setting(0, float, "freeze-melt threshold in degC"),
    "rmax": Setting(30, float, "Retention capacity of liquid water in snow cover in %"),
    "end_event": Setting(21600, int, "Time after which no rainfall/snow melt occurs in seconds"),
    "hpi": Setting(5, int, "threshold for classification of heavy rainfall event in mm/10min"),
    "a_bc": Setting(2, int, "a parameter for Brooks-Corey"),
    "b_bc": Setting(2, int, "b parameter for Brooks-Corey"),
    "clay_min": Setting(0.01, float, "minimum clay content of soil"),
    "clay_max": Setting(0.71, float, "maximum clay content of soil"),
    "theta_rew_min": Setting(0.02, float, "minimum soil water content at permanent wilting point in -"),
    "theta_rew_max": Setting(0.24, float, "maximum soil water content at permanent wilting point in -"),
    "zroot_to_zsoil_max": Setting(0.7, float, "maximum ratio of root zone depth to soil depth in -"),
    "rew_min": Setting(2, float, "minimum readily evaporable water in mm"),
    "rew_max": Setting(12, float, "maximum readily evaporable water in mm"),
    "z_evap_max": Setting(150, float, "maximum soil evaporation depth in mm"),
    "transp_water_stress": Setting(0.75, float, "fraction of fine pore storage in -"),
    "atol": Setting(1e-2, float, "absolute tolerance of solutions"),
    "rtol": Setting(1e-2, float, "relative tolerance of solutions"),
    # offline transport (roger/settings.py:76-78, 102-120)
    "VSMOW_conc18O": Setting(2005.2e-6, float, "oxygen-18 abundancy ratios according to VSMOW in -"),
    "d18O_min": Setting(-20, float, "potentially lowest oxygen-18 value in per mille"),
    "d18O_max": Setting(0, float, "potentially greatest oxygen-18 value in per mille"),
    "VSMOW_conc2H": Setting(155.76e-6, float, "deuterium abundancy ratios according to VSMOW in -"),
    "d2H_min": Setting(-160, float, "potentially lowest deuterium value in per mille"),
    "d2H_max": Setting(0, float, "potentially greatest deuterium value in per mille"),
    "enable_bromide": Setting(False, bool, "enable bromide"),
    "enable_chloride": Setting(False, bool, "enable enable_chloride"),
    "enable_oxygen18": Setting(False, bool, "enable oxygen-18"),
    "enable_deuterium": Setting(False, bool, "enable deuterium"),
    "enable_nitrate": Setting(False, bool, "enable nitrate"),
    "enable_virtualtracer": Setting(False, bool, "enable virtual tracer"),
    "tm_structure": Setting("UNNAMED", str, "transport model structure"),
    "enable_age_statistics": Setting(False, bool, "enable calculation of age statistics"),
    "sas_solver": Setting(None, optional(str), "numerical solver scheme for StorAge selection"),
    "sas_solver_substeps": Setting(1, int, "substeps to solver for StorAge selection numerically"),
    # process switches: only the SVAT path is implemented natively; the others must stay off
    "enable_distributed_input": Setting(False, bool, "enable distributed input"),
    "enable_film_flow": Setting(False, bool, "enable film flow process"),
    "enable_lateral_flow": Setting(False, bool, "enable lateral flow"),
    "enable_crop_phenology": Setting(False, bool, "enable crop phenology"),
    "enable_net_irrigation": Setting(False, bool, "enable net crop irrigation"),
    "enable_soil_compaction": Setting(False, bool, "enable soil compaction"),
    "enable_offline_transport": Setting(False, bool, "enable offline transport"),
    "enable_groundwater_boundary": Setting(False, bool, "enable groundwater boundary"),
    "enable_groundwater": Setting(False, bool, "enable groundwater"),
    "enable_routing_1D": Setting(False, bool, "enable unidirectional routing"),
    "enable_routing_2D": Setting(False, bool, "enable bidirectional routing"),
    "enable_macropore_lower_boundary_condition": Setting(False, bool, "enable lower boundary condition of macropores"),
    "enable_adaptive_time_stepping": Setting(False, bool, "enable_adaptive_time_stepping"),
    "warmup_done": Setting(False, bool, "True if after model warmup"),
    "restart_input_filename": Setting(None, optional(str), "File name of restart input."),
}

_UNSUPPORTED_SWITCHES = (
    "enable_distributed_input", "enable_film_flow", "enable_crop_phenology",
    "enable_net_irrigation", "enable_soil_compaction", "enable_groundwater_boundary",
    "enable_groundwater", "enable_routing_1D", "enable_routing_2D", "enable_macropore_lower_boundary_condition",
    "enable_nitrate", "enable_virtualtracer",
)


def check_setting_conflicts(settings):
    """roger/settings.py:139-171 (conflict checks) + the scope of the native backend."""
    for name in _UNSUPPORTED_SWITCHES:
        if getattr(settings, name):
            raise NotImplementedError(
                f"settings.{name}=True is outside the hot path of the hip backend (SURVEY.md section 8: out of scope)")
    if settings.enable_offline_transport:
        # the native transport path: oxygen-18, deuterium, bromide or chloride with the deterministic SAS solver
        # 
        if sum(int(getattr(settings, k)) for k in ("enable_oxygen18", "enable_deuterium", "enable_bromide", "enable_chloride")) != 1:
            raise NotImplementedError("offline transport on the hip backend is implemented for oxygen-18, deuterium, bromide and "
                                      "chloride (exactly one of settings.enable_oxygen18 / enable_deuterium / enable_bromide / "
                                      "enable_chloride must be True)")
        if settings.sas_solver != "deterministic":
            raise NotImplementedError(f'settings.sas_solver={settings.sas_solver!r}: the hip backend implements the '
                                      '"deterministic" SAS solver (Euler / RK4 are out of scope, SURVEY.md section 8)')
        if settings.nages != settings.ages + 1:
            raise ValueError("settings.nages must be settings.ages + 1")
        return
    if not settings.enable_adaptive_time_stepping:
        raise NotImplementedError("the hip backend implements the adaptive time stepping of the SVAT benchmarks "
                                  "(settings.enable_adaptive_time_stepping must be True)")
