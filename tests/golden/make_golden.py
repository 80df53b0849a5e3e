#!/usr/bin/env python3
"""Generate golden vectors for the SVAT hot path from the reference NumPy backend.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What it does: imports the reference (`roger`, NumPy backend, float64, diskless),
drives an in-memory `SVATSetup` subclass (heterogeneous per-cell parameters,
synthetic forcing from roger_amd/forcing.py) through `setup()` and N calls of
`step()`, and records

  * the complete per-cell state after setup (the oracle's / HIP path's start
    state, *including* all derived parameters, so setup-time kernels can be
    checked separately from the step),
  * the state after `soil.calculate_parameters` inputs only (raw user
    parameters) for the setup-kernel tests,
  * per-step global scalars (dt, dt_secs, itt_day, event ids, ...),
  * full-state snapshots at selected steps, and per-routine snapshots for the
    first steps of each dt class.

Only *data* (inputs and expected outputs) is written; no reference code is
stored.  Two third-party modules the reference imports but this image lacks are
replaced by inert stand-ins that carry no arithmetic: `loguru` (logger that
swallows every call) and `h5netcdf` (never touched in diskless mode).
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def _install_standins():
    class _Null:
        def __getattr__(self, name):
            def f(*a, **k):
                return self

            return f

    lg = types.ModuleType("loguru")
    lg.logger = _Null()
    sys.modules["loguru"] = lg
    sys.modules["h5netcdf"] = types.ModuleType("h5netcdf")


def import_reference():
    _install_standins()
    sys.path.insert(0, "/root/reference")
    from roger import runtime_settings

    runtime_settings.update(backend="numpy", float_type="float64", diskless_mode=True)
    import roger  # noqa: F401

    return roger


# ---------------------------------------------------------------------------
# scenarios
# ---------------------------------------------------------------------------
def hetero_params(nx, ny, seed=42):
    """Per-cell parameters covering the land uses / soils the kernels branch on."""
    rng = np.random.default_rng(seed)
    n = nx * ny
    lu_pool = np.array([8, 5, 10, 13, 0, 98, 11, 12, 6, 7, 9, 15, 8, 10, 5, 8])
    lu = lu_pool[np.arange(n) % lu_pool.size]
    z_soil = rng.uniform(400, 2000, n).round(0)
    lmpv = np.minimum(rng.choice([0.0, 200.0, 500.0, 900.0, 1500.0], n), z_soil * 0.95).round(0)
    p = dict(
        lu_id=lu,
        z_soil=z_soil,
        dmpv=rng.choice([0.0, 25.0, 50.0, 100.0], n),
        lmpv=lmpv,
        theta_ac=rng.uniform(0.05, 0.2, n),
        theta_ufc=rng.uniform(0.08, 0.25, n),
        theta_pwp=rng.uniform(0.05, 0.3, n),
        ks=rng.uniform(0.5, 50, n),
        kf=np.full(n, 2500.0),
        sealing=np.where(lu == 0, rng.uniform(0.2, 0.8, n), 0.0),
        S_dep_tot=np.zeros(n),
    )
    th_sat = p["theta_ac"] + p["theta_ufc"] + p["theta_pwp"]
    p["theta_rz0"] = p["theta_pwp"] + rng.uniform(0.1, 0.95, n) * (th_sat - p["theta_pwp"])
    p["theta_ss0"] = p["theta_pwp"] + rng.uniform(0.1, 0.95, n) * (th_sat - p["theta_pwp"])
    # a few tie / edge cells
    p["theta_rz0"][0] = p["theta_pwp"][0] + p["theta_ufc"][0]  # exactly field capacity
    p["dmpv"][1] = 0.0
    p["lmpv"][2] = 0.0
    return {k: v.reshape(nx, ny) for k, v in p.items()}


def uniform_params(nx, ny):
    """benchmarks/SVAT_benchmark.py:92-103,117-121"""
    f = lambda v: np.full((nx, ny), v)  # noqa: E731
    return dict(
        lu_id=f(8), z_soil=f(2000.0), dmpv=f(50.0), lmpv=f(50.0), theta_ac=f(0.1), theta_ufc=f(0.1),
        theta_pwp=f(0.2), ks=f(5.0), kf=f(2500.0), sealing=f(0.0), S_dep_tot=f(0.0),
        theta_rz0=f(0.3), theta_ss0=f(0.3),
    )


TUTORIAL_INPUT = "/root/reference/examples/plot_scale/svat_tutorial/input"
EBERBAECHLE_INPUT = "/root/reference/examples/catchment_scale/eberbaechle/svat_distributed/input"


def tutorial_params():
    """BASELINE configs[0]: examples/plot_scale/svat_tutorial/config.yml (one cell)."""
    f = lambda v: np.full((1, 1), v)  # noqa: E731
    return dict(
        lu_id=f(8), z_soil=f(900.0), dmpv=f(0.0), lmpv=f(500.0), theta_ac=f(0.1), theta_ufc=f(0.1),
        theta_pwp=f(0.2), ks=f(10.0), kf=f(2500.0), sealing=f(0.0), S_dep_tot=f(0.0),
        theta_rz0=f(0.3), theta_ss0=f(0.3),
    )


def make_model(roger, params, forcing, ndays, lateral=False, weights=None, stations=None, routing=None):
    from roger import roger_routine
    from roger.models.svat import SVATSetup
    from roger.models.oneD import ONEDSetup
    from roger.core.operators import numpy as npx, update, at
    import roger.lookuptables as lut

    nx, ny = params["lu_id"].shape
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
        def set_grid(self, state):
            vs = state.variables
            vs.x = update(vs.x, at[3:-2], npx.cumsum(npx.ones(vs.x[3:-2].shape)))
            vs.y = update(vs.y, at[3:-2], npx.cumsum(npx.ones(vs.y[3:-2].shape)))

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
                return
            for k in ("PREC", "TA", "PET"):
                setattr(vs, k, update(getattr(vs, k), at[:], F[k]))

        @roger_routine
        def set_forcing(self, state):
            # same slicing as benchmarks/SVAT_benchmark.py:151-171, data held in memory
            vs = state.variables
            if vs.time % (24 * 60 * 60) == 0:
                vs.itt_day = 0
                vs.year = update(vs.year, at[1], F["YEAR"][vs.itt_forc])
                vs.month = update(vs.month, at[1], F["MONTH"][vs.itt_forc])
                vs.doy = update(vs.doy, at[1], F["DOY"][vs.itt_forc])
                sl = slice(vs.itt_forc, vs.itt_forc + 6 * 24)
                if stations is not None:
                    # every cell takes the series of ITS station (svat_dist.py:280-293: one masked assignment per station; a cell
                    # whose station_id matches none keeps zeros), then the weights (:305-322)
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
                elif weights is not None:   # station series x per-cell weight (eberbaechle/svat_distributed/svat.py:276-296)
                    vs.prec_day = update(vs.prec_day, at[2:-2, 2:-2, :],
                                         vs.PREC[npx.newaxis, npx.newaxis, sl] * vs.prec_weight[2:-2, 2:-2, npx.newaxis])
                    vs.ta_day = update(vs.ta_day, at[2:-2, 2:-2, :],
                                       vs.TA[npx.newaxis, npx.newaxis, sl] + vs.ta_offset[2:-2, 2:-2, npx.newaxis])
                    vs.pet_day = update(vs.pet_day, at[2:-2, 2:-2, :],
                                        vs.PET[npx.newaxis, npx.newaxis, sl] * vs.pet_weight[2:-2, 2:-2, npx.newaxis])
                else:
                    vs.prec_day = update(vs.prec_day, at[:, :, :], vs.PREC[npx.newaxis, npx.newaxis, sl])
                    vs.ta_day = update(vs.ta_day, at[:, :, :], vs.TA[npx.newaxis, npx.newaxis, sl])
                    vs.pet_day = update(vs.pet_day, at[:, :, :], vs.PET[npx.newaxis, npx.newaxis, sl])
                vs.itt_forc = vs.itt_forc + 6 * 24

    return GoldenSVAT()


# ---------------------------------------------------------------------------
# recording
# ---------------------------------------------------------------------------
def plane_names():
    """(plane name, reference variable, level) for every oracle plane."""
    import ctypes

    lib = ctypes.CDLL(os.path.join(REPO, "oracle", "libsvat_oracle.so"))
    lib.oc_plane_name.restype = ctypes.c_char_p
    out = []
    for i in range(lib.oc_nplanes()):
        nm = lib.oc_plane_name(i).decode()
        if nm.endswith("_m1"):
            out.append((nm, nm[:-3], 0))
        else:
            out.append((nm, nm, 1))
    return out


SCALARS = ("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")
SCALARS2 = ("event_id", "year", "month", "doy")


def snapshot(vs, planes):
    """Interior (ghost-free), flattened copy of every oracle plane -> (nplanes, n) float64."""
    cols = []
    n = None
    for _, var, lvl in planes:
        try:
            a = np.asarray(getattr(vs, var))
        except RuntimeError:   # variable not active in this configuration (e.g. lateral-flow fields in SVAT)
            cols.append(None)
            continue
        a = a[2:-2, 2:-2, lvl] if a.ndim == 3 else a[2:-2, 2:-2]
        cols.append(np.asarray(a, dtype=np.float64).ravel())
        n = cols[-1].size
    return np.stack([c if c is not None else np.zeros(n) for c in cols])


def scalars(vs):
    row = [float(np.asarray(getattr(vs, k))) for k in SCALARS]
    for k in SCALARS2:
        a = np.asarray(getattr(vs, k))
        row += [float(a[0]), float(a[1])]
    return np.array(row)


ROUTINES = (
    ("adaptive_time_stepping", "adaptive_time_stepping"),
    ("interception", "calculate_interception"),
    ("evapotranspiration", "calculate_evapotranspiration"),
    ("snow", "calculate_snow"),
    ("infiltration", "calculate_infiltration"),
    ("surface_runoff", "calculate_surface_runoff"),   # (does nothing unless settings.enable_routing_1D)
    ("subsurface_runoff", "calculate_subsurface_runoff"),
    ("capillary_rise", "calculate_capillary_rise"),
    ("soil", "calculate_soil"),  # last of the four storage routines
    ("numerics", "calc_storage"),
)


def run_case(roger, name, params, forcing, ndays, max_steps, snap_every, routine_steps, outdir, lateral=False,
             pair_every=0, weights=None, stations=None, routing=None, store_forcing=True):
    import importlib

    model = make_model(roger, params, forcing, ndays, lateral=lateral, weights=weights, stations=stations, routing=routing)
    planes = plane_names()
    rec = {}
    routine_log = {}
    state_holder = {}

    # wrap the per-routine entry points so that the state after each one can be recorded
    originals = []
    for mod_name, fn_name in ROUTINES:
        mod = importlib.import_module(f"roger.core.{mod_name}")
        orig = getattr(mod, fn_name)

        def make_wrapper(orig, tag):
            def wrapper(state, *a, **k):
                out = orig(state, *a, **k)
                step_no = state_holder.get("step")
                if step_no in routine_steps:
                    routine_log[f"r{step_no:05d}_{tag}"] = snapshot(state.variables, planes)
                    routine_log[f"r{step_no:05d}_{tag}_scal"] = scalars(state.variables)
                return out

            return wrapper

        setattr(mod, fn_name, make_wrapper(orig, fn_name))
        originals.append((mod, fn_name, orig))

    # capture the raw user parameters before the derived-parameter kernels run
    from roger.core import soil as _soil

    orig_calc_params = _soil.calculate_parameters

    def calc_params_wrapper(state):
        rec["pre_params"] = snapshot(state.variables, planes)
        return orig_calc_params(state)

    _soil.calculate_parameters = calc_params_wrapper
    from roger.core import surface as _surface

    orig_surf_params = _surface.calculate_parameters

    def surf_params_wrapper(state):
        rec["pre_surface"] = snapshot(state.variables, planes)
        return orig_surf_params(state)

    _surface.calculate_parameters = surf_params_wrapper
    orig_calc_ic = _soil.calculate_initial_conditions

    def calc_ic_wrapper(state):
        # state after set_initial_conditions + surface.calculate_initial_conditions input
        rec["pre_ic"] = snapshot(state.variables, planes)
        return orig_calc_ic(state)

    _soil.calculate_initial_conditions = calc_ic_wrapper
    try:
        model.setup()
        vs = model.state.variables
        rec["plane_names"] = np.array([p[0] for p in planes])
        rec["nx_ny"] = np.array(params["lu_id"].shape)
        rec["state0"] = snapshot(vs, planes)
        rec["scal0"] = scalars(vs)
        rec["lut_ilu"] = np.asarray(vs.lut_ilu, dtype=np.float64)
        rec["lut_gc"] = np.asarray(vs.lut_gc, dtype=np.float64)
        rec["lut_gcm"] = np.asarray(vs.lut_gcm, dtype=np.float64)
        rec["lut_rdlu"] = np.asarray(vs.lut_rdlu, dtype=np.float64)
        rec["lateral"] = np.array(int(lateral))
        if routing is not None:
            rec["routing"] = np.array(1)
            rec["routing_dx_dy"] = np.array([routing["dx"], routing["dy"]], dtype=np.float64)
        if lateral:
            rec["lut_mlms"] = np.asarray(vs.lut_mlms, dtype=np.float64)[:200]  # slopes 1..200 %
        for k, v in forcing.items():
            if store_forcing:
                rec[f"forc_{k}"] = v
        if weights is not None:
            for k, v in weights.items():
                rec[f"weight_{k}"] = np.asarray(v, dtype=np.float64).ravel()
        if stations is not None:
            for k, v in stations.items():
                rec[f"station_{k}"] = np.asarray(v)
        scal_rows = []
        snaps = {}
        step = 0
        runlen = model.state.settings.runlen
        while vs.time < runlen and step < max_steps:
            step += 1
            state_holder["step"] = step
            model.step(model.state)
            scal_rows.append(scalars(vs))
            if step % snap_every == 0 or step <= 3 or (pair_every and step % pair_every in (0, pair_every - 1)):
                snaps[f"s{step:05d}"] = snapshot(vs, planes)
        snaps[f"s{step:05d}"] = snapshot(vs, planes)
        rec["scal"] = np.stack(scal_rows)
        rec["nsteps"] = np.array(step)
        rec.update(snaps)
        rec.update(routine_log)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **rec)
        dts = rec["scal"][:, SCALARS.index("dt")]
        print(f"{name}: {step} steps (daily {np.sum(dts == 24)}, hourly {np.sum(dts == 1)}, "
              f"10min {np.sum(dts < 1)}), {len(snaps)} snapshots, {len(routine_log) // 2} routine snapshots, "
              f"max dS_num_error {float(np.max(np.asarray(vs.dS_num_error))):.3e} -> {path} "
              f"({os.path.getsize(path) / 1e6:.2f} MB)")
    finally:
        for mod, fn_name, orig in originals:
            setattr(mod, fn_name, orig)
        _soil.calculate_parameters = orig_calc_params
        _surface.calculate_parameters = orig_surf_params
        _soil.calculate_initial_conditions = orig_calc_ic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    roger = import_reference()
    from roger_amd.forcing import combo_forcing, toy_forcing

    cases = {
        # name: (params, forcing, ndays, max_steps, snap_every, routine_steps)
        "svat_hetero_combo": (hetero_params(4, 4), combo_forcing(ndays=24), 24, 100000, 40,
                              set(range(1, 4)) | {30, 31, 60, 61, 200, 201, 400, 401}),
        "svat_uniform_rain": (uniform_params(3, 2), toy_forcing("rain", ndays=6), 6, 100000, 10, {1, 2, 14, 15}),
        "svat_hetero_heavyrain": (hetero_params(3, 3, seed=7), toy_forcing("heavyrain", ndays=4), 4, 100000, 25,
                                  {1, 2, 3, 20, 21}),
        "svat_hetero_snowrain": (hetero_params(3, 3, seed=11), toy_forcing("snow+rain", ndays=8), 8, 100000, 20,
                                 {1, 2, 3, 10, 11}),
    }
    if not args.only or args.only == "svat_tutorial":
        # BASELINE configs[0]: the tutorial's single cell with its own measured forcing (PREC.txt / TA.txt / PET.txt through
        # roger_amd.forcing.forcing_from_txt = read_meteo + write_forcing's recipe, float32 as write_forcing stores it);
        # the whole year October 2010 - September 2011
        from roger_amd.forcing import forcing_from_txt

        nd = 365
        cases["svat_tutorial"] = (tutorial_params(), forcing_from_txt(TUTORIAL_INPUT, ndays=nd), nd, 100000, 250,
                                  {1, 2, 3, 500, 501})
    for name, (params, forcing, ndays, max_steps, snap_every, rsteps) in cases.items():
        if args.only and args.only != name:
            continue
        run_case(roger, name, params, forcing, ndays, max_steps, snap_every, rsteps, args.out)
    if not args.only or args.only == "svat_eberbaechle_weights":
        # BASELINE configs[4] (catchment_scale/eberbaechle/svat_distributed): the station's measured series -- the first 40 days of
        # the shipped PREC/TA/PET.txt, 2019-11-01 .. 2019-12-10, month change included -- times / plus per-cell prec_weight,
        # ta_offset, pet_weight in the setup's own set_forcing; parameters.nc is not shipped, so heterogeneous synthetic maps
        nd = 40
        nx, ny = 4, 3
        rng = np.random.default_rng(2019)
        weights = dict(prec_weight=rng.uniform(0.8, 1.3, (nx, ny)), ta_offset=rng.uniform(-5.5, 1.5, (nx, ny)),
                       pet_weight=rng.uniform(0.85, 1.15, (nx, ny)))
        from roger_amd.forcing import forcing_from_txt

        run_case(roger, "svat_eberbaechle_weights", hetero_params(nx, ny, seed=2019), forcing_from_txt(EBERBAECHLE_INPUT, ndays=nd),
                 nd, 100000, 100, {1, 2, 3, 300, 301}, args.out, weights=weights)
    if args.only == "svat_eberbaechle_full":
        # BASELINE configs[4] AS WRITTEN: the same twelve columns and weights over the station's FULL series, 2019-11-01 .. 2022-10-31
        # (157 824 ten-minute records = 1 096 days; the three input files are shipped whole as tests/golden/eberbaechle_full_input).
        # About 40 minutes of the reference: generated on request only (--only svat_eberbaechle_full).  A snapshot every 1 000 steps,
        # the scalars of every step; the forcing arrays are not stored (the test reads the text files).
        nx, ny = 4, 3
        rng = np.random.default_rng(2019)
        weights = dict(prec_weight=rng.uniform(0.8, 1.3, (nx, ny)), ta_offset=rng.uniform(-5.5, 1.5, (nx, ny)),
                       pet_weight=rng.uniform(0.85, 1.15, (nx, ny)))
        from roger_amd.forcing import forcing_from_txt

        F = forcing_from_txt(EBERBAECHLE_INPUT)
        nd = len(F["PREC"]) // 144
        run_case(roger, "svat_eberbaechle_full", hetero_params(nx, ny, seed=2019), F, nd, 10 ** 7, 1000, set(), args.out, weights=weights,
                 store_forcing=False)
    if not args.only or args.only == "svat_stations":
        # settings.enable_distributed_input: three meteorological stations, every cell mapped to one of them by vs.station_id (one
        # cell to none: it sees zeros), per-cell weights on top (roger/bmimodels/svat_dist/svat_dist.py:274-322)
        nd, nx, ny = 14, 4, 3
        rng = np.random.default_rng(77)
        series = [combo_forcing(ndays=nd, seed=sd) for sd in (5, 6, 9)]
        ids = np.array([11, 22, 33])
        cell_ids = rng.choice(ids, (nx, ny))
        cell_ids[1, 1] = 99
        stations = dict(station_ids=ids, station_id=cell_ids, **{k: np.stack([f[k] for f in series]) for k in ("PREC", "TA", "PET")})
        weights = dict(prec_weight=rng.uniform(0.8, 1.3, (nx, ny)), ta_offset=rng.uniform(-1.5, 1.5, (nx, ny)),
                       pet_weight=rng.uniform(0.85, 1.15, (nx, ny)))
        run_case(roger, "svat_stations", hetero_params(nx, ny, seed=77), series[0], nd, 100000, 60, {1, 2, 3}, args.out,
                 weights=weights, stations=stations)
    # oneD model (lateral subsurface flow, benchmarks/oneD_benchmark.py): wet start so that a perched
    # water table forms and the lateral branches are taken
    def oned_params(nx, ny, seed):
        p = hetero_params(nx, ny, seed=seed)
        rng = np.random.default_rng(seed + 1000)
        n = nx * ny
        p["slope"] = rng.choice([0.01, 0.05, 0.12, 0.3], n).reshape(nx, ny)
        p["dmph"] = rng.choice([0.0, 25.0, 50.0, 100.0], n).reshape(nx, ny)
        th_sat = p["theta_ac"] + p["theta_ufc"] + p["theta_pwp"]
        p["theta_rz0"] = p["theta_pwp"] + rng.uniform(0.6, 0.99, n).reshape(nx, ny) * (th_sat - p["theta_pwp"])
        p["theta_ss0"] = p["theta_pwp"] + rng.uniform(0.7, 1.0, n).reshape(nx, ny) * (th_sat - p["theta_pwp"])
        p["ks"] = np.minimum(p["ks"], rng.uniform(0.2, 6, n).reshape(nx, ny))
        return p

    oned = {
        "oned_hetero_combo": (oned_params(4, 4, 21), combo_forcing(ndays=16, seed=5), 16, 100000, 30,
                              {1, 2, 3, 40, 41, 120, 121}),
        "oned_hetero_heavyrain": (oned_params(3, 2, 22), toy_forcing("heavyrain", ndays=4), 4, 100000, 25, {1, 2, 20, 21}),
    }
    def oned_benchmark_params(nx, ny):
        """benchmarks/oneD_benchmark.py:99-135, 160-163 (BASELINE configs[3]): uniform columns."""
        p = uniform_params(nx, ny)
        f = lambda v: np.full((nx, ny), v)  # noqa: E731
        p.update(z_soil=f(1000.0), lmpv=f(600.0), slope=f(0.05), dmph=f(50.0))
        return p

    # configs[3]'s own parameter set: the start state of the full-size property test (3200 x 3125 columns vs one oracle column)
    oned["oned_uniform_benchmark"] = (oned_benchmark_params(3, 2), toy_forcing("heavyrain", ndays=4), 4, 100000, 25, {1, 2, 20, 21})
    for name, (params, forcing, ndays, max_steps, snap_every, rsteps) in oned.items():
        if args.only and args.only != name:
            continue
        # pairs of consecutive snapshots (k-1, k) every 10 steps: the oneD model has no snap-to-zero of
        # emptied stores (models/oneD/oneD.py vs svat.py:326-345), so long trajectories hit sign ties of
        # rounding residue; parity is therefore also pinned step by step from reference states
        run_case(roger, name, params, forcing, ndays, max_steps, snap_every, rsteps, args.out, lateral=True,
                 pair_every=10)

    # settings.enable_routing_1D (examples/hillslope_scale/oneD_distributed_routing_tutorial): surface and subsurface runoff move to the
    # D8 neighbour.  A 4 x 6 hillslope draining mostly towards +y with diagonal and sideways cells, one pit (no direction), one interior
    # outlet cell (outer_boundary = 1: takes no inflow), water leaving over the edge of the grid
    if not args.only or args.only == "oned_routing":
        nx, ny = 4, 6
        rng = np.random.default_rng(31)
        flow = np.full((nx, ny), 4)
        flow[0, 1], flow[1, 2], flow[2, 0], flow[3, 3] = 2, 8, 1, 16     # SE, SW, E, W
        flow[1, 4], flow[2, 3], flow[0, 4] = 128, 32, 64                 # NE, NW, N
        flow[3, 1] = 0                                                   # a pit
        outer = np.zeros((nx, ny), dtype=int)
        outer[2, 5] = 1
        p = oned_params(nx, ny, 23)
        routing = dict(flow_dir_topo=flow, outer_boundary=outer, k_st=rng.uniform(8, 30, (nx, ny)), dx=5, dy=5)
        run_case(roger, "oned_routing", p, toy_forcing("heavyrain", ndays=4), 4, 100000, 25, {1, 2, 30, 31, 60}, args.out,
                 lateral=True, pair_every=10, routing=routing)

    # ... the same with every step class: snowfall and melt, rain with pauses, heavy rain, dry days and a month change (combo forcing,
    # 14 days from 2018-01-20) on a 5 x 4 hillslope with random flow directions
    if not args.only or args.only == "oned_routing_combo":
        nx, ny = 5, 4
        rng = np.random.default_rng(37)
        flow = rng.choice([1, 2, 4, 8, 16, 32, 64, 128], (nx, ny))
        flow[:, :2] = 4
        flow[2, 2] = 0
        outer = np.zeros((nx, ny), dtype=int)
        outer[4, 3] = 1
        p = oned_params(nx, ny, 29)
        routing = dict(flow_dir_topo=flow, outer_boundary=outer, k_st=rng.uniform(8, 30, (nx, ny)), dx=5, dy=5)
        run_case(roger, "oned_routing_combo", p, combo_forcing(ndays=14), 14, 100000, 40, {1, 2, 100, 101}, args.out,
                 lateral=True, pair_every=10, routing=routing)

    # the reference's own routing example (examples/hillslope_scale/oneD_distributed_routing_tutorial: config.yml, oneD.py:127-335,
    # 455-483): a 1 x 20 hillslope, dx = dy = 1 m, every cell draining towards +y, uniform soil, Strickler coefficient 50, soil at field
    # capacity, the station's series (the same files as Eberbaechle's: the first 40 days, 2019-11-01 ...) with the example's weights
    # (prec x 1, ta + 1, pet x 1)
    if not args.only or args.only == "oned_routing_tutorial":
        from roger_amd.forcing import forcing_from_txt

        nx, ny, nd = 1, 20, 25
        f = lambda v: np.full((nx, ny), v)  # noqa: E731
        p = dict(lu_id=np.full((nx, ny), 8, dtype=np.int32), sealing=f(0.0), S_dep_tot=f(0.0), z_soil=f(900.0), slope=f(0.05), dmpv=f(50.0),
                 dmph=f(50.0), lmpv=f(600.0), theta_ac=f(0.1), theta_ufc=f(0.1), theta_pwp=f(0.2), ks=f(20.0), kf=f(1.0),
                 theta_rz0=f(0.3), theta_ss0=f(0.3))
        routing = dict(flow_dir_topo=np.full((nx, ny), 4), outer_boundary=np.zeros((nx, ny), dtype=int), k_st=f(50.0), dx=1, dy=1)
        weights = dict(prec_weight=f(1.0), ta_offset=f(1.0), pet_weight=f(1.0))
        run_case(roger, "oned_routing_tutorial", p, forcing_from_txt(EBERBAECHLE_INPUT, ndays=nd), nd, 100000, 40, {1, 2, 200, 201}, args.out,
                 lateral=True, pair_every=25, routing=routing, weights=weights)


if __name__ == "__main__":
    main()
