"""Setup scripts used by the host-package tests.  `GoldenSVAT` has the same hook bodies as the
class tests/golden/make_golden.py drives through the *reference* (`from roger import ...` there,
`from roger_amd import ...` here) -- the point of the operator surface is that such a script
runs unchanged."""
import numpy as np

from roger_amd import KernelOutput, roger_kernel, roger_routine
from roger_amd.core.operators import at, numpy as npx, update
from roger_amd.core.surface import calc_parameters_surface_kernel
from roger_amd.models.svat import SVATSetup


@roger_kernel
def after_timestep_kernel(state):
    """A script-local tau -> taum1 rotation as every script of the reference carries one (benchmarks/SVAT_benchmark.py:183-418,
    roger/models/svat/svat.py:187-384).  On the hip backend a kernel of this name is the native rotation (roger_amd/routines.py:
    NATIVE_KERNELS), the body below is what a script author would write and never runs there."""
    vs = state.variables
    out = {}
    for name in ("S_rz", "S_ss", "S_s", "S", "z_sat", "z_wf", "theta_rz", "theta_ss", "theta", "prec", "ta", "swe", "z0"):
        a = getattr(vs, name)
        out[name] = update(a, at[2:-2, 2:-2, vs.taum1], a[2:-2, 2:-2, vs.tau])
    return KernelOutput(**out)


def make_model(params, forcing, ndays, lateral=False, global_shape=None, weights=None, stations=None, routing=None, script_hooks=None):
    """global_shape: (nx, ny) of the whole domain when `params` holds this rank's slab only (num_proc = (N, 1)).
    script_hooks: the script ALSO defines set_parameters and after_timestep itself, with the bodies every script of the reference has
    (benchmarks/SVAT_benchmark.py:105-110, 177-181) -- "plain"; or with one deviation each that the device's control part does NOT
    perform: "set_parameters_assigns" (a variable assigned in the hook), "after_timestep_assigns", "forcing_scaled" (set_forcing
    doubles the precipitation), "forcing_counts" (set_forcing keeps a counter of its own)."""
    from roger_amd.models.oned import ONEDSetup

    nx, ny = global_shape or params["lu_id"].shape
    F = forcing
    Base = ONEDSetup if lateral else SVATSetup

    class GoldenSVAT(Base):
        @roger_routine
        def set_settings(self, state):
            s = state.settings
            s.identifier = "GoldenSVAT"
            s.nx, s.ny = nx, ny
            s.runlen = 24 * 60 * 60 * ndays
            s.nitt_forc = len(F["PREC"])
            s.dx = 1
            s.dy = 1
            if routing is not None:   # examples/hillslope_scale/oneD_distributed_routing_tutorial/oneD.py:72-85
                s.enable_routing_1D = True
                s.dx, s.dy = routing["dx"], routing["dy"]
            s.x_origin = 0.0
            s.y_origin = 0.0
            s.time_origin = "2018-01-01 00:00:00"
            s.enable_groundwater_boundary = False
            s.enable_macropore_lower_boundary_condition = False
            s.enable_lateral_flow = bool(lateral)
            s.enable_adaptive_time_stepping = True
            if stations is not None:   # several meteorological stations (roger/bmimodels/svat_dist/svat_dist.py:85, 67-70)
                s.enable_distributed_input = True
                s.nstations = len(stations["station_ids"])

        @roger_routine
        def set_topography(self, state):
            if routing is None:
                return
            vs = state.variables   # oneD_distributed_routing_tutorial/oneD.py:127-203: D8 flow direction, catchment boundary
            vs.flow_dir_topo = update(vs.flow_dir_topo, at[2:-2, 2:-2], routing["flow_dir_topo"])
            vs.inner_boundary = update(vs.inner_boundary, at[2:-2, 2:-2], 1)
            vs.outer_boundary = update(vs.outer_boundary, at[2:-2, 2:-2], routing["outer_boundary"])
            vs.maskCatch = update(vs.maskCatch, at[2:-2, 2:-2], 1)

        @roger_routine
        def set_parameters_setup(self, state):
            vs = state.variables
            for k in ("lu_id", "z_soil", "dmpv", "lmpv", "theta_ac", "theta_ufc", "theta_pwp", "ks", "kf",
                      "sealing", "S_dep_tot"):
                setattr(vs, k, update(getattr(vs, k), at[2:-2, 2:-2], params[k]))
            if routing is not None:   # Strickler coefficient, oneD.py:330-335
                vs.k_st = update(vs.k_st, at[2:-2, 2:-2], routing["k_st"])
            if lateral:   # benchmarks/oneD_benchmark.py:106-117
                vs.slope = update(vs.slope, at[2:-2, 2:-2], params["slope"])
                vs.slope_per = update(vs.slope_per, at[2:-2, 2:-2], vs.slope[2:-2, 2:-2] * 100)
                vs.dmph = update(vs.dmph, at[2:-2, 2:-2], params["dmph"])
            if stations is not None:   # svat_dist.py:200-211
                vs.station_id = update(vs.station_id, at[2:-2, 2:-2], stations["station_id"])
                vs.station_ids = update(vs.station_ids, at[:], stations["station_ids"])
            if weights is not None:   # examples/catchment_scale/eberbaechle/svat_distributed/svat.py:169-186
                vs.prec_weight = update(vs.prec_weight, at[2:-2, 2:-2], weights["prec_weight"])
                vs.ta_offset = update(vs.ta_offset, at[2:-2, 2:-2], weights["ta_offset"])
                vs.pet_weight = update(vs.pet_weight, at[2:-2, 2:-2], weights["pet_weight"])

        @roger_routine
        def set_initial_conditions(self, state):
            vs = state.variables
            vs.theta_rz = update(vs.theta_rz, at[2:-2, 2:-2, : vs.taup1], params["theta_rz0"][:, :, None])
            vs.theta_ss = update(vs.theta_ss, at[2:-2, 2:-2, : vs.taup1], params["theta_ss0"][:, :, None])

        @roger_routine
        def set_forcing_setup(self, state):
            vs = state.variables
            if stations is not None:   # svat_dist.py:261-263: (n_stations, t_forc) series
                for k in ("PREC", "TA", "PET"):
                    setattr(vs, k + "_DIST", update(getattr(vs, k + "_DIST"), at[:, :], stations[k]))
            else:
                for k in ("PREC", "TA", "PET"):
                    setattr(vs, k, update(getattr(vs, k), at[:], F[k]))
            for k in ("YEAR", "MONTH", "DOY"):   # kept in vs so that run_device can hand them over
                setattr(vs, k, update(getattr(vs, k), at[:], F[k]))

        @roger_routine
        def set_forcing(self, state):
            vs = state.variables
            if vs.time % (24 * 60 * 60) == 0:
                vs.itt_day = 0
                vs.year = update(vs.year, at[1], F["YEAR"][vs.itt_forc])
                vs.month = update(vs.month, at[1], F["MONTH"][vs.itt_forc])
                vs.doy = update(vs.doy, at[1], F["DOY"][vs.itt_forc])
                sl = slice(vs.itt_forc, vs.itt_forc + 6 * 24)
                if stations is not None:   # every cell takes the series of ITS station, then the weights (svat_dist.py:280-322)
                    day = {}
                    for k in ("PREC", "TA", "PET"):
                        a = npx.zeros(vs.prec_day.shape)
                        for i, ii in enumerate(vs.station_ids):
                            mask = vs.station_id == ii
                            a = npx.where(mask[:, :, npx.newaxis], getattr(vs, k + "_DIST")[i, :][npx.newaxis, npx.newaxis, sl], a)
                        day[k] = a
                    vs.prec_day = update(vs.prec_day, at[2:-2, 2:-2, :], day["PREC"][2:-2, 2:-2, :] * vs.prec_weight[2:-2, 2:-2, npx.newaxis])
                    vs.ta_day = update(vs.ta_day, at[2:-2, 2:-2, :], day["TA"][2:-2, 2:-2, :] + vs.ta_offset[2:-2, 2:-2, npx.newaxis])
                    vs.pet_day = update(vs.pet_day, at[2:-2, 2:-2, :], day["PET"][2:-2, 2:-2, :] * vs.pet_weight[2:-2, 2:-2, npx.newaxis])
                elif weights is not None:   # eberbaechle/svat_distributed/svat.py:276-296
                    vs.prec_day = update(vs.prec_day, at[2:-2, 2:-2, :], vs.PREC[npx.newaxis, npx.newaxis, sl] * vs.prec_weight[2:-2, 2:-2, npx.newaxis])
                    vs.ta_day = update(vs.ta_day, at[2:-2, 2:-2, :], vs.TA[npx.newaxis, npx.newaxis, sl] + vs.ta_offset[2:-2, 2:-2, npx.newaxis])
                    vs.pet_day = update(vs.pet_day, at[2:-2, 2:-2, :], vs.PET[npx.newaxis, npx.newaxis, sl] * vs.pet_weight[2:-2, 2:-2, npx.newaxis])
                else:
                    vs.prec_day = update(vs.prec_day, at[:, :, :], vs.PREC[npx.newaxis, npx.newaxis, sl])
                    vs.ta_day = update(vs.ta_day, at[:, :, :], vs.TA[npx.newaxis, npx.newaxis, sl])
                    vs.pet_day = update(vs.pet_day, at[:, :, :], vs.PET[npx.newaxis, npx.newaxis, sl])
                vs.itt_forc = vs.itt_forc + 6 * 24

    if script_hooks is None:
        return GoldenSVAT(forcing=F, nx=nx, ny=ny, ndays=ndays)
    assert script_hooks in ("plain", "set_parameters_assigns", "after_timestep_assigns", "forcing_scaled", "forcing_counts")

    class ScriptSVAT(GoldenSVAT):
        @roger_routine
        def set_parameters(self, state):
            vs = state.variables

            if (vs.month[vs.tau] != vs.month[vs.taum1]) & (vs.itt > 1):
                vs.update(calc_parameters_surface_kernel(state))

        @roger_routine
        def after_timestep(self, state):
            vs = state.variables

            vs.update(after_timestep_kernel(state))

    if script_hooks == "set_parameters_assigns":
        class ScriptSVAT(ScriptSVAT):   # noqa: F811
            @roger_routine
            def set_parameters(self, state):
                vs = state.variables
                if (vs.month[vs.tau] != vs.month[vs.taum1]) & (vs.itt > 1):
                    vs.update(calc_parameters_surface_kernel(state))
                vs.c_int = update(vs.c_int, at[2:-2, 2:-2], 1.0)   # (the value it has: the trajectory stays the golden's)
    elif script_hooks == "after_timestep_assigns":
        class ScriptSVAT(ScriptSVAT):   # noqa: F811
            @roger_routine
            def after_timestep(self, state):
                vs = state.variables
                vs.update(after_timestep_kernel(state))
                vs.irr_demand = update(vs.irr_demand, at[2:-2, 2:-2], 0.0)
    elif script_hooks == "forcing_scaled":
        class ScriptSVAT(ScriptSVAT):   # noqa: F811
            @roger_routine
            def set_forcing(self, state):
                vs = state.variables
                if vs.time % (24 * 60 * 60) == 0:
                    vs.itt_day = 0
                    vs.year = update(vs.year, at[1], vs.YEAR[vs.itt_forc])
                    vs.month = update(vs.month, at[1], vs.MONTH[vs.itt_forc])
                    vs.doy = update(vs.doy, at[1], vs.DOY[vs.itt_forc])
                    sl = slice(vs.itt_forc, vs.itt_forc + 6 * 24)
                    vs.prec_day = update(vs.prec_day, at[:, :, :], 2 * vs.PREC[npx.newaxis, npx.newaxis, sl])
                    vs.ta_day = update(vs.ta_day, at[:, :, :], vs.TA[npx.newaxis, npx.newaxis, sl])
                    vs.pet_day = update(vs.pet_day, at[:, :, :], vs.PET[npx.newaxis, npx.newaxis, sl])
                    vs.itt_forc = vs.itt_forc + 6 * 24
    elif script_hooks == "forcing_counts":
        class ScriptSVAT(ScriptSVAT):   # noqa: F811
            forcing_calls = []

            @roger_routine
            def set_forcing(self, state):
                self.forcing_calls.append(int(state.variables.time))
                GoldenSVAT.set_forcing.__wrapped__(self, state)

    return ScriptSVAT(forcing=F, nx=nx, ny=ny, ndays=ndays)


def params_from_golden(g, names):
    """Recover the raw user parameters of a golden case from its recorded states."""
    nx, ny = (int(v) for v in g["nx_ny"])
    pre, ic = g["pre_params"], g["pre_ic"]
    p = {k: pre[names.index(k)].reshape(nx, ny) for k in
         ("lu_id", "z_soil", "dmpv", "lmpv", "theta_ac", "theta_ufc", "theta_pwp", "ks", "kf", "sealing", "S_dep_tot")}
    p["lu_id"] = p["lu_id"].astype(np.int32)
    if "lateral" in g.files and int(g["lateral"]):
        p["slope"] = pre[names.index("slope")].reshape(nx, ny)
        p["dmph"] = pre[names.index("dmph")].reshape(nx, ny)
    p["theta_rz0"] = ic[names.index("theta_rz")].reshape(nx, ny)
    p["theta_ss0"] = ic[names.index("theta_ss")].reshape(nx, ny)
    return p


def snapshot_from_vs(vs, names):
    cols = []
    for nm in names:
        if nm.endswith("_m1") and nm[:-3] in vs:
            a = np.asarray(getattr(vs, nm[:-3]))[2:-2, 2:-2, 0]
        else:
            a = np.asarray(getattr(vs, nm))
            a = a[2:-2, 2:-2, 1] if a.ndim == 3 else a[2:-2, 2:-2]
        cols.append(np.asarray(a, dtype=np.float64).ravel())
    return np.stack(cols)
