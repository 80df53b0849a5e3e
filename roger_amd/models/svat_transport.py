"""The ready-made offline transport model (oxygen-18, deuterium, bromide, chloride) on top of a SVAT run, in the style of
roger/models/svat_oxygen18 and benchmarks/SVATOXYGEN18_benchmark.py, with its inputs held in memory instead of `SVAT.nc` /
`forcing_tracer.nc`:

    svat = read_svat_output("output/SVAT.rate.nc", "output/SVAT.collect.nc")    # what a SVAT run of this package wrote
    svat["d18O_prec"] = ...                                                     # the tracer signal of the precipitation
    model = make_transport_model("roger_amd", svat, sas_params, ages, substeps, ndays, age_statistics=True)
    model.setup(); model.warmup(); model.run()

It is written once against the reference's operator surface and parametrised by the package that provides it, because it
is also the setup of the SAS golden vectors:

    make_transport_model("roger", ...)      -> the reference (tests/golden/make_golden_sas.py, build container only)
    make_transport_model("roger_amd", ...)  -> the hip backend

It is a `RogerSetup` subclass in the style of benchmarks/SVATOXYGEN18_benchmark.py: power-law (or mixed) SAS
parameters per flux, initial conditions spread evenly over the age classes at -10 permil, daily fluxes and storages
of a SVAT run as input, the snow signal mixed in `set_forcing`.
"""
import importlib

import numpy as np

SVAT_RATES = ("prec", "inf_mat_rz", "inf_mp_rz", "inf_sc_rz", "inf_ss", "transp", "evap_soil", "cpr_rz", "q_rz", "q_ss")
SVAT_STORAGES = ("S_rz", "S_ss", "S_snow", "S_pwp_rz", "S_pwp_ss", "S_sat_rz", "S_sat_ss")


def read_svat_output(rate_nc, collect_nc):
    """The daily flux sums and end-of-day storages a SVAT run wrote through `state.diagnostics` (roger_amd/diagnostics.py:
    `<identifier>.rate.nc`, `<identifier>.collect.nc`) as the `svat` dict of `make_transport_model`: arrays (nx, ny, days + 1),
    record 0 = initial values; the reference's own netCDF-4 output files are read as well -- the variables the reference's transport models read from SVAT.nc
    (roger/models/svat_oxygen18/svat_oxygen18.py:148-163, 231-241, 452-473)."""
    from scipy.io import netcdf_file

    from .. import h5lite

    out = {}
    for path, names in ((rate_nc, SVAT_RATES), (collect_nc, SVAT_STORAGES)):
        if h5lite.is_hdf5(str(path)):   # netCDF-4, as the reference itself writes its output (h5netcdf): same variables, same (Time, y, x)
            try:
                import h5py

                with h5py.File(str(path), "r") as f:
                    have = {k: np.asarray(f[k]) for k in names if k in f}
            except ImportError:
                have = {k: v for k, v in h5lite.read_root(str(path)).items() if k in names}
            for k, v in have.items():
                out[k] = np.ascontiguousarray(np.asarray(v, dtype=np.float64).transpose(2, 1, 0))
            continue
        with netcdf_file(str(path), "r", mmap=False) as f:
            for k in names:
                if k in f.variables:
                    out[k] = np.ascontiguousarray(np.asarray(f.variables[k][:], dtype=np.float64).transpose(2, 1, 0))   # (Time, y, x) -> (x, y, Time)
    missing = [k for k in SVAT_RATES + SVAT_STORAGES if k not in out]
    if missing:
        raise KeyError(f"the SVAT output lacks {missing}: add them to the rate / collect diagnostics of the SVAT run")
    for k in ("S_pwp_rz", "S_pwp_ss", "S_sat_rz", "S_sat_ss"):   # parameters of the soil, constant in time
        out[k] = np.ascontiguousarray(out[k][:, :, 0])
    return out


def make_transport_model(pkg, svat, sas, ages, substeps, ndays, age_statistics, warmup_days=0, tracer="oxygen18", extra=None,
                         solver="deterministic"):
    top = importlib.import_module(pkg)
    RogerSetup, roger_kernel, roger_routine, KernelOutput = top.RogerSetup, top.roger_kernel, top.roger_routine, top.KernelOutput
    allocate = importlib.import_module(pkg + ".variables").allocate
    ops = importlib.import_module(pkg + ".core.operators")
    npx, update, at = ops.numpy, ops.update, ops.at
    tr = importlib.import_module(pkg + ".core.transport")
    delta_to_conc, conc_to_delta = tr.delta_to_conc, tr.conc_to_delta

    nx, ny = svat["S_rz"].shape[:2]
    virtual = tracer == "virtualtracer"   # as chloride, and the soil evaporation takes the tracer along
    chloride = tracer in ("chloride", "virtualtracer")
    deuterium = tracer == "deuterium"   # the isotope kernels with the constants of 2H (roger/core/transport.py:315-340)
    d0 = -70 if deuterium else -10      # initial signal of the soil water in permil
    bromide = tracer in ("bromide", "chloride", "virtualtracer")   # anion transport (mass based): models/svat_bromide, deterministic solver
    extra = extra or {}

    class GoldenSAS(RogerSetup):
        @roger_routine
        def set_settings(self, state):
            s = state.settings
            s.identifier = "GoldenSAS"
            s.sas_solver = solver
            s.sas_solver_substeps = substeps
            if solver in ("RK4", "Euler"):   # benchmarks/SVATOXYGEN18_benchmark.py:30-31
                s.h = 1 / s.sas_solver_substeps
            s.nx, s.ny = nx, ny
            s.runlen = 24 * 60 * 60 * ndays
            s.runlen_warmup = 24 * 60 * 60 * warmup_days
            s.nitt = ndays + 1
            s.ages = ages
            s.nages = ages + 1
            s.dx = 1
            s.dy = 1
            s.x_origin = 0.0
            s.y_origin = 0.0
            s.time_origin = "01-01-2022"
            s.enable_offline_transport = True
            s.enable_oxygen18 = not bromide and not deuterium
            s.enable_deuterium = deuterium
            s.enable_bromide = bromide and not chloride
            s.enable_chloride = chloride and not virtual
            s.enable_virtualtracer = virtual
            s.tm_structure = "power"
            s.enable_age_statistics = bool(age_statistics)

        @roger_routine
        def read_data(self, state):
            pass

        @roger_routine
        def set_grid(self, state):
            vs = state.variables
            settings = state.settings
            vs.dt_secs = 60 * 60 * 24
            vs.dt = 24.0
            vs.ages = update(vs.ages, at[:], npx.arange(1, settings.nages))
            vs.nages = update(vs.nages, at[:], npx.arange(settings.nages))
            vs.x = update(vs.x, at[3:-2], npx.cumsum(npx.ones(vs.x[3:-2].shape)))
            vs.y = update(vs.y, at[3:-2], npx.cumsum(npx.ones(vs.y[3:-2].shape)))

        @roger_routine
        def set_look_up_tables(self, state):
            pass

        @roger_routine
        def set_topography(self, state):
            pass

        @roger_routine
        def set_parameters_setup(self, state):
            vs = state.variables
            for k in ("S_pwp_rz", "S_pwp_ss", "S_sat_rz", "S_sat_ss"):
                setattr(vs, k, update(getattr(vs, k), at[2:-2, 2:-2], svat[k]))
            for flux, arr in sas.items():   # (nx, ny, 8): [code, k, ...]
                name = f"sas_params_{flux}"
                setattr(vs, name, update(getattr(vs, name), at[2:-2, 2:-2, :], arr))
            if bromide:   # models/svat_bromide/svat_bromide.py:193-195; lu_id feeds the crop-uptake switch of the transpiration kernel
                for k in ("alpha_transp", "alpha_q", "lu_id"):
                    setattr(vs, k, update(getattr(vs, k), at[2:-2, 2:-2], extra[k]))

        @roger_routine
        def set_parameters(self, state):
            pass

        @roger_routine
        def set_initial_conditions_setup(self, state):
            vs = state.variables
            for k, src in (("S_snow", "S_snow"), ("S_rz", "S_rz"), ("S_ss", "S_ss")):
                setattr(vs, k, update(getattr(vs, k), at[2:-2, 2:-2, : vs.taup1], svat[src][:, :, 0, None]))
            vs.S_s = update(vs.S_s, at[2:-2, 2:-2, : vs.taup1],
                            vs.S_rz[2:-2, 2:-2, : vs.taup1] + vs.S_ss[2:-2, 2:-2, : vs.taup1])
            vs.S_rz_init = update(vs.S_rz_init, at[2:-2, 2:-2], vs.S_rz[2:-2, 2:-2, 0])
            vs.S_ss_init = update(vs.S_ss_init, at[2:-2, 2:-2], vs.S_ss[2:-2, 2:-2, 0])

        @roger_routine
        def set_initial_conditions(self, state):
            # benchmarks/SVATOXYGEN18_benchmark.py:172-340
            vs = state.variables
            settings = state.settings
            arr0 = allocate(state.dimensions, ("x", "y"))
            for sa, S in (("sa_rz", vs.S_rz), ("sa_ss", vs.S_ss)):
                setattr(vs, sa, update(
                    getattr(vs, sa), at[2:-2, 2:-2, : vs.taup1, 1:],
                    npx.diff(npx.linspace(arr0[2:-2, 2:-2], S[2:-2, 2:-2, vs.tau], settings.ages, axis=-1), axis=-1)[
                        :, :, npx.newaxis, :]))
            vs.SA_rz = update(vs.SA_rz, at[2:-2, 2:-2, :, 1:], npx.cumsum(vs.sa_rz[2:-2, 2:-2, :, :], axis=-1))
            vs.SA_ss = update(vs.SA_ss, at[2:-2, 2:-2, :, 1:], npx.cumsum(vs.sa_rz[2:-2, 2:-2, :, :], axis=-1))
            vs.sa_s = update(vs.sa_s, at[2:-2, 2:-2, :, :], vs.sa_rz[2:-2, 2:-2, :, :] + vs.sa_ss[2:-2, 2:-2, :, :])
            vs.SA_s = update(vs.SA_s, at[2:-2, 2:-2, :, 1:], npx.cumsum(vs.sa_s[2:-2, 2:-2, :, :], axis=-1))
            if bromide:   # no bromide in the soil at the start (models/svat_bromide/svat_bromide.py:230-260): msa = 0
                if chloride:   # a background concentration (mg/l) that differs between root zone and subsoil
                    vs.msa_rz = update(vs.msa_rz, at[2:-2, 2:-2, :, :], vs.sa_rz[2:-2, 2:-2, :, :] * extra["C_init_rz"])
                    vs.msa_ss = update(vs.msa_ss, at[2:-2, 2:-2, :, :], vs.sa_ss[2:-2, 2:-2, :, :] * extra["C_init_ss"])
                return
            vs.C_iso_snow = update(vs.C_iso_snow, at[2:-2, 2:-2, : vs.taup1], npx.nan)
            vs.C_iso_rz = update(vs.C_iso_rz, at[2:-2, 2:-2, : vs.taup1], d0)
            vs.C_iso_ss = update(vs.C_iso_ss, at[2:-2, 2:-2, : vs.taup1], d0)
            for C, Ciso, msa in (("C_rz", vs.C_iso_rz, "msa_rz"), ("C_ss", vs.C_iso_ss, "msa_ss")):
                setattr(vs, C, update(getattr(vs, C), at[2:-2, 2:-2, : vs.taup1],
                                      delta_to_conc(state, Ciso[2:-2, 2:-2, vs.tau, npx.newaxis])))
                setattr(vs, msa, update(getattr(vs, msa), at[2:-2, 2:-2, : vs.taup1, :],
                                        getattr(vs, C)[2:-2, 2:-2, : vs.taup1, npx.newaxis]))
                setattr(vs, msa, update(getattr(vs, msa), at[2:-2, 2:-2, : vs.taup1, 0], 0))
            tot = vs.sa_rz[2:-2, 2:-2, :, :] + vs.sa_ss[2:-2, 2:-2, :, :]
            vs.msa_s = update(vs.msa_s, at[2:-2, 2:-2, :, :], npx.where(
                tot > 0,
                vs.msa_rz[2:-2, 2:-2, :, :] * (vs.sa_rz[2:-2, 2:-2, :, :] / tot)
                + vs.msa_ss[2:-2, 2:-2, :, :] * (vs.sa_ss[2:-2, 2:-2, :, :] / tot), 0))
            vs.msa_s = update(vs.msa_s, at[2:-2, 2:-2, : vs.taup1, 0], 0)

        @roger_routine
        def set_boundary_conditions_setup(self, state):
            pass

        @roger_routine
        def set_boundary_conditions(self, state):
            pass

        @roger_routine
        def set_forcing_setup(self, state):
            vs = state.variables
            vs.PREC_DIST_DAILY = update(vs.PREC_DIST_DAILY, at[2:-2, 2:-2, :], svat["prec"])
            vs.INF_MAT_RZ = update(vs.INF_MAT_RZ, at[2:-2, 2:-2, :], svat["inf_mat_rz"])
            vs.INF_PF_RZ = update(vs.INF_PF_RZ, at[2:-2, 2:-2, :], svat["inf_mp_rz"] + svat["inf_sc_rz"])
            vs.INF_PF_SS = update(vs.INF_PF_SS, at[2:-2, 2:-2, :], svat["inf_ss"])
            vs.TRANSP = update(vs.TRANSP, at[2:-2, 2:-2, :], svat["transp"])
            vs.EVAP_SOIL = update(vs.EVAP_SOIL, at[2:-2, 2:-2, :], svat["evap_soil"])
            vs.CPR_RZ = update(vs.CPR_RZ, at[2:-2, 2:-2, :], svat["cpr_rz"])
            vs.Q_RZ = update(vs.Q_RZ, at[2:-2, 2:-2, :], svat["q_rz"])
            vs.Q_SS = update(vs.Q_SS, at[2:-2, 2:-2, :], svat["q_ss"])
            vs.S_RZ = update(vs.S_RZ, at[2:-2, 2:-2, :], svat["S_rz"])
            vs.S_SS = update(vs.S_SS, at[2:-2, 2:-2, :], svat["S_ss"])
            vs.S_S = update(vs.S_S, at[2:-2, 2:-2, :], vs.S_RZ[2:-2, 2:-2, :] + vs.S_SS[2:-2, 2:-2, :])
            vs.S_SNOW = update(vs.S_SNOW, at[2:-2, 2:-2, :], svat["S_snow"])
            if bromide:   # bromide concentration of the input in mg/l, a pulse on selected days (extra["C_IN"]: (nx, ny, t))
                if not virtual:   # (the reference has no C_IN series for the virtual tracer: set_forcing assigns C_in itself)
                    vs.C_IN = update(vs.C_IN, at[2:-2, 2:-2, :], extra["C_IN"])
                return
            vs.C_ISO_IN = update(vs.C_ISO_IN, at[2:-2, 2:-2, 0], npx.nan)
            # deuterium: the same series on the meteoric water line, d2H = 8 * d18O + 10
            d_prec = svat["d18O_prec"] * 8 + 10 if deuterium else svat["d18O_prec"]
            vs.C_ISO_IN = update(vs.C_ISO_IN, at[2:-2, 2:-2, 1:], d_prec[None, None, 1:])
            vs.C_IN = update(vs.C_IN, at[2:-2, 2:-2, :], delta_to_conc(state, vs.C_ISO_IN)[2:-2, 2:-2, :])

        @roger_routine
        def set_forcing(self, state):
            # benchmarks/SVATOXYGEN18_benchmark.py:384-437
            vs = state.variables
            i = vs.itt
            vs.prec = update(vs.prec, at[2:-2, 2:-2, vs.tau], vs.PREC_DIST_DAILY[2:-2, 2:-2, i])
            for dst, src in (("inf_mat_rz", "INF_MAT_RZ"), ("inf_pf_rz", "INF_PF_RZ"), ("inf_pf_ss", "INF_PF_SS"),
                             ("transp", "TRANSP"), ("evap_soil", "EVAP_SOIL"), ("cpr_rz", "CPR_RZ"), ("q_rz", "Q_RZ"),
                             ("q_ss", "Q_SS")):
                setattr(vs, dst, update(getattr(vs, dst), at[2:-2, 2:-2], getattr(vs, src)[2:-2, 2:-2, i]))
            vs.S_rz = update(vs.S_rz, at[2:-2, 2:-2, vs.tau], vs.S_RZ[2:-2, 2:-2, i])
            vs.S_ss = update(vs.S_ss, at[2:-2, 2:-2, vs.tau], vs.S_SS[2:-2, 2:-2, i])
            vs.S_s = update(vs.S_s, at[2:-2, 2:-2, vs.tau], vs.S_rz[2:-2, 2:-2, vs.tau] + vs.S_ss[2:-2, 2:-2, vs.tau])
            vs.S_snow = update(vs.S_snow, at[2:-2, 2:-2, vs.tau], vs.S_SNOW[2:-2, 2:-2, i])
            vs.C_in = update(vs.C_in, at[2:-2, 2:-2], extra["C_IN"][:, :, i] if virtual else vs.C_IN[2:-2, 2:-2, i])
            if bromide:   # models/svat_bromide/svat_bromide.py:343-347
                vs.M_in = update(vs.M_in, at[2:-2, 2:-2], vs.C_in[2:-2, 2:-2] * vs.prec[2:-2, 2:-2, vs.tau])
                return
            prec, snow = vs.prec[2:-2, 2:-2, vs.tau], vs.S_snow[2:-2, 2:-2, vs.tau]
            vs.C_snow = update(vs.C_snow, at[2:-2, 2:-2, vs.tau], npx.where(
                snow > 0,
                npx.where(npx.isnan(vs.C_snow[2:-2, 2:-2, vs.tau]), vs.C_in[2:-2, 2:-2],
                          (prec / (prec + snow)) * vs.C_in[2:-2, 2:-2]
                          + (snow / (prec + snow)) * vs.C_snow[2:-2, 2:-2, vs.taum1]),
                npx.nan))
            vs.C_snow = update(vs.C_snow, at[2:-2, 2:-2, vs.tau],
                               npx.where(vs.S_snow[2:-2, 2:-2, vs.tau] <= 0, npx.nan, vs.C_snow[2:-2, 2:-2, vs.tau]))
            vs.C_iso_snow = update(vs.C_iso_snow, at[2:-2, 2:-2, vs.tau], conc_to_delta(state, vs.C_snow[2:-2, 2:-2, vs.tau]))
            vs.C_in = update(vs.C_in, at[2:-2, 2:-2], npx.where(
                npx.isfinite(vs.C_snow[2:-2, 2:-2, vs.taum1]), vs.C_snow[2:-2, 2:-2, vs.taum1],
                npx.where(vs.prec[2:-2, 2:-2, vs.tau] > 0, vs.C_IN[2:-2, 2:-2, i], 0)))
            vs.C_iso_in = update(vs.C_iso_in, at[2:-2, 2:-2], conc_to_delta(state, vs.C_in[2:-2, 2:-2]))

        @roger_routine
        def set_diagnostics(self, state):
            pass

        @roger_routine
        def after_timestep(self, state):
            vs = state.variables
            vs.update(after_timestep_kernel(state))

    @roger_kernel
    def after_timestep_kernel(state):
        vs = state.variables
        vs.S_snow = update(vs.S_snow, at[2:-2, 2:-2, vs.taum1], vs.S_snow[2:-2, 2:-2, vs.tau])
        vs.prec = update(vs.prec, at[2:-2, 2:-2, vs.taum1], vs.prec[2:-2, 2:-2, vs.tau])
        if bromide:
            return KernelOutput(prec=vs.prec, S_snow=vs.S_snow)
        vs.C_snow = update(vs.C_snow, at[2:-2, 2:-2, vs.taum1], vs.C_snow[2:-2, 2:-2, vs.tau])
        return KernelOutput(prec=vs.prec, C_snow=vs.C_snow, S_snow=vs.S_snow)

    return GoldenSAS()
