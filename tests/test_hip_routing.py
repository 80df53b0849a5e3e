"""GPU: settings.enable_routing_1D -- the D8 routing of surface and subsurface runoff (rh_surface_routing, rh_subsurface_routing,
rh_step_routed) against the reference's own run of a routed hillslope (tests/golden/oned_routing.npz: 4 x 6 cells with all eight
flow directions, a pit, an interior outlet; 240 steps) and, for the halo path of a decomposed run, a domain cut in two."""
import numpy as np
import pytest

from golden_util import ROUTING_CASES, ROUTING_STEP_CASES, compare, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from roger_amd import _native as N

    N.load()
    return N


def routed_ctx(native, g, names, key="state0", scal_key="scal0", columns=None):
    """A routing context holding the golden snapshot `key`; columns = (x0, x1): only that slab of the grid."""
    import hip_util as H

    nx, ny = (int(v) for v in g["nx_ny"])
    dx, dy = (float(v) for v in g["routing_dx_dy"])
    snap = np.asarray(g[key])
    if columns is not None:
        x0, x1 = columns
        snap = snap.reshape(snap.shape[0], nx, ny)[:, x0:x1].reshape(snap.shape[0], -1)
        nx = x1 - x0
    ctx = native.Context(nx, ny, enable_lateral_flow=1, enable_routing_1D=1, dx=dx, dy=dy)
    H.upload_snapshot(ctx, snap, names)
    ctx.set_scalars(H.scalars_from_row(g[scal_key] if isinstance(scal_key, str) else scal_key))
    ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    ctx.set_lut_mlms(g["lut_mlms"])
    return ctx


@pytest.mark.parametrize("case", ROUTING_STEP_CASES)
def test_single_steps_from_reference_states(native, case):
    """One routed step from the reference's state k-1 gives the reference's state k (every pair of consecutive snapshots)."""
    import hip_util as H

    g, names, forcing = load_case(case)
    have = sorted(int(k[1:]) for k in g.files if k.startswith("s") and k[1:].isdigit())
    pairs = [k for k in have if k - 1 in have and k >= 2]
    assert len(pairs) >= 12
    for k in pairs:
        ctx = routed_ctx(native, g, names, key=f"s{k - 1:05d}", scal_key=g["scal"][k - 2])
        s = ctx.get_scalars()
        if s.time % 86400:   # mid-day: hand over the forcing the reference took at midnight
            i0 = s.itt_forc - 144
            ctx.set_forcing_day(*[forcing[v][i0:i0 + 144] for v in ("PREC", "TA", "PET")])
        monthly = H.HipForcingDriver(ctx, forcing).before_step()
        ctx.step_routed(monthly)
        np.testing.assert_array_equal(H.scalars_to_row(ctx.get_scalars()), g["scal"][k - 1], err_msg=f"step {k}")
        compare(H.download_snapshot(ctx, names), g[f"s{k:05d}"], names, what=f"{case} single step {k}")
        ctx.close()


@pytest.mark.parametrize("case", ROUTING_CASES)
def test_trajectory_golden(native, case):
    """Free-running from the initial state: all planes at the stored steps, the integer scalars exactly at every step."""
    import hip_util as H

    g, names, forcing = load_case(case)
    ctx = routed_ctx(native, g, names)
    drv = H.HipForcingDriver(ctx, forcing)
    checked = 0
    for step in range(1, int(g["nsteps"]) + 1):
        ctx.step_routed(drv.before_step())
        np.testing.assert_array_equal(H.scalars_to_row(ctx.get_scalars()), g["scal"][step - 1], err_msg=f"{case} scalars step {step}")
        key = f"s{step:05d}"
        if key in g.files:
            compare(H.download_snapshot(ctx, names), g[key], names, what=f"{case} step {step}")
            checked += 1
    assert checked >= 15
    ctx.close()


@pytest.mark.parametrize("case", ROUTING_STEP_CASES)
def test_routing_entry_points_golden(native, case):
    """rh_surface_routing and rh_subsurface_runoff + rh_subsurface_routing: state before the reference routine in, state after it out."""
    import hip_util as H

    g, names, _ = load_case(case)
    steps = sorted({int(k[1:6]) for k in g.files if k.startswith("r") and k.endswith("_calculate_surface_runoff")})
    assert len(steps) >= 4
    for step in steps:
        for prev, cur, entries in (("calculate_infiltration", "calculate_surface_runoff", ("rh_surface_routing",)),
                                   ("calculate_surface_runoff", "calculate_subsurface_runoff", ("rh_subsurface_runoff", "rh_subsurface_routing"))):
            kp = f"r{step:05d}_{prev}"
            ctx = routed_ctx(native, g, names, key=kp, scal_key=kp + "_scal")
            for e in entries:
                ctx.call(e)
            compare(H.download_snapshot(ctx, names), g[f"r{step:05d}_{cur}"], names, what=f"{case} step {step} {cur}")
            ctx.close()


@pytest.mark.parametrize("case", ROUTING_STEP_CASES)
def test_halo_columns_of_a_cut_domain(native, case):
    """The grid cut in two along x (as a (2, 1) decomposition cuts it), one context per half, the edge columns handed over by hand
    (rh_route_get_edges / rh_route_set_halo: what the RCCL exchange of rh_surface_routing does between ranks): both routings give the
    single-domain result bit for bit -- water crosses the cut straight and diagonally."""
    import hip_util as H

    g, names, _ = load_case(case)
    nx, ny = (int(v) for v in g["nx_ny"])
    if nx < 2:
        pytest.skip("a single column of cells along x cannot be cut")
    cut = nx // 2
    steps = sorted({int(k[1:6]) for k in g.files if k.startswith("r") and k.endswith("_calculate_surface_runoff")})
    crossed = 0
    for step in steps:
        for which, kp in ((0, f"r{step:05d}_calculate_infiltration"), (1, None)):
            if which == 0:
                whole = routed_ctx(native, g, names, key=kp, scal_key=kp + "_scal")
                halves = [routed_ctx(native, g, names, key=kp, scal_key=kp + "_scal", columns=c) for c in ((0, cut), (cut, nx))]
            else:   # continue: the lateral flow, then its routing
                for c in [whole] + halves:
                    c.call("rh_subsurface_runoff")
            whole.call("rh_surface_routing" if which == 0 else "rh_subsurface_routing")
            for h in halves:
                h.route_out(which)
            if which == 0:
                st = [h.route_static_edges() for h in halves]
                halves[0].route_set_halo(1, flow_dir=st[1][0], mask=st[1][2])   # the right half's x = 0 column is the left half's x = nx halo
                halves[1].route_set_halo(0, flow_dir=st[0][1], mask=st[0][3])
            e = [h.route_edges(which) for h in halves]
            halves[0].route_set_halo(1, q=e[1][0])
            halves[1].route_set_halo(0, q=e[0][1])
            crossed += int((e[0][1] != 0).any() or (e[1][0] != 0).any())
            for h in halves:
                h.route_in(which)
            ref = H.download_snapshot(whole, names).reshape(len(names), nx, ny)
            got = np.concatenate([H.download_snapshot(h, names).reshape(len(names), -1, ny) for h in halves], axis=1)
            assert np.array_equal(got, ref, equal_nan=True), f"step {step} routing {which}: planes {[names[p] for p in np.unique(np.argwhere(got != ref)[:, 0])][:6]}"
        for c in [whole] + halves:
            c.close()
    assert crossed >= 2, "no water crossed the cut: the test would not see the halo path"


def test_routed_steps_with_device_hooks_and_one_rank_communicator(native):
    """rh_run_steps on a routing context (device-side set_forcing / set_parameters hooks, then rh_step_routed) reproduces the golden
    trajectory; with a one-rank RCCL communicator attached (the predicate words and the edge columns then take the multi-rank code path
    up to the exchange itself) the result is bit-identical."""
    import hip_util as H

    g, names, forcing = load_case(ROUTING_CASES[0])
    nsteps = int(g["nsteps"])
    out = []
    for with_comm in (False, True):
        ctx = routed_ctx(native, g, names)
        ctx.set_forcing_series(forcing)
        if with_comm:
            ctx.comm_init(native.comm_unique_id(), 1, 0)
        ctx.run_steps(nsteps)
        np.testing.assert_array_equal(H.scalars_to_row(ctx.get_scalars()), g["scal"][nsteps - 1])
        out.append(H.download_snapshot(ctx, names))
        ctx.close()
    compare(out[0], g[f"s{nsteps:05d}"], names, what="routed run_steps")
    assert np.array_equal(out[0], out[1], equal_nan=True)


def test_device_driven_routed_steps_equal_the_routine_by_routine_step(native, monkeypatch):
    """rh_run_steps on a routing context forms the control part in one kernel from the summary bits the first pass posted and selects
    the step's forcing inside that pass (6 launches per step); RH_ROUTED_BY_ROUTINE=1 keeps rh_step_routed per step (predicate passes
    over the arena, 17 launches): the same bits, also when the run is split and the host touches a plane in between."""
    import hip_util as H

    g, names, forcing = load_case(ROUTING_CASES[0])
    nsteps = int(g["nsteps"])
    out = {}
    for mode in ("device", "routine", "split"):
        if mode == "routine":
            monkeypatch.setenv("RH_ROUTED_BY_ROUTINE", "1")
        else:
            monkeypatch.delenv("RH_ROUTED_BY_ROUTINE", raising=False)
        ctx = routed_ctx(native, g, names)
        ctx.set_forcing_series(forcing)
        if mode == "split":
            ctx.run_steps(37)
            ctx.upload("z0", ctx.download("z0"))   # the host touches a plane: the summary bits are rebuilt from the arena
            ctx.run_steps(1)
            ctx.run_steps(nsteps - 38)
        else:
            ctx.run_steps(nsteps)
        out[mode] = (H.scalars_to_row(ctx.get_scalars()), H.download_snapshot(ctx, names))
        ctx.close()
    for mode in ("routine", "split"):
        np.testing.assert_array_equal(out["device"][0], out[mode][0])
        assert np.array_equal(out["device"][1], out[mode][1], equal_nan=True), mode
    # with the output accumulators configured (the accumulator kernel then runs between the numerics and the rotation): the same state,
    # and daily sums equal to those of the routine-by-routine step
    sums = {}
    for mode in ("device", "routine"):
        if mode == "routine":
            monkeypatch.setenv("RH_ROUTED_BY_ROUTINE", "1")
        else:
            monkeypatch.delenv("RH_ROUTED_BY_ROUTINE", raising=False)
        ctx = routed_ctx(native, g, names)
        ctx.set_forcing_series(forcing)
        ctx.diag_configure(rate=("q_sur_out", "q_sub_in", "aet", "prec"), collect=("S", "z0"), n_slots=8)
        ctx.run_steps(nsteps)
        assert np.array_equal(H.download_snapshot(ctx, names), out["device"][1], equal_nan=True), mode
        sums[mode] = np.stack([ctx.diag_download(k, slot) for slot in range(4) for k in ("q_sur_out", "q_sub_in", "aet", "prec", "S", "z0")])
        ctx.close()
    assert np.array_equal(sums["device"], sums["routine"]) and sums["device"][0].sum() + sums["device"][1].sum() > 0


def test_fused_step_is_refused_and_lateral_flow_is_required(native):
    g, names, forcing = load_case(ROUTING_CASES[0])
    ctx = routed_ctx(native, g, names)
    ctx.set_forcing_day(*[forcing[v][:144] for v in ("PREC", "TA", "PET")])
    with pytest.raises(native.NativeError, match="routine by routine"):
        ctx.step(False)
    ctx.close()
    with pytest.raises(native.NativeError, match="enable_lateral_flow"):
        native.Context(3, 3, enable_routing_1D=1)


def _tiled_case(native, oracle, g, names, nx, ny, rng, flow=None):
    """A grid of nx x ny cells drawn from the golden hillslope's cells (every per-cell plane of a cell travels together), with new
    routing parameters: an oracle state and a device context holding the same start state."""
    import hip_util as H
    from golden_util import configure_settings

    src = np.asarray(g["state0"])
    pick = rng.integers(0, src.shape[1], nx * ny)
    snap = src[:, pick].copy()
    if flow is None:
        flow = rng.choice([0, 1, 2, 4, 8, 16, 32, 64, 128], nx * ny, p=[0.04] + [0.12] * 8)
    snap[names.index("flow_dir_topo")] = flow
    snap[names.index("outer_boundary")] = rng.random(nx * ny) < 0.03
    snap[names.index("k_st")] = rng.uniform(8, 30, nx * ny)
    dx, dy = (float(v) for v in g["routing_dx_dy"])
    st = oracle.OracleState(nx * ny)
    st.load_snapshot(snap, names)
    st.load_scalars(g["scal0"])
    st.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    configure_settings(st.settings, g)
    st.settings.nx, st.settings.ny = nx, ny
    ctx = native.Context(nx, ny, enable_lateral_flow=1, enable_routing_1D=1, dx=dx, dy=dy)
    H.upload_snapshot(ctx, snap, names)
    ctx.set_scalars(H.scalars_from_row(g["scal0"]))
    ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    ctx.set_lut_mlms(g["lut_mlms"])
    return st, ctx


@pytest.fixture(scope="module")
def oracle():
    import oracle_binding as ob

    ob.lib()
    return ob


def test_random_flow_directions_over_many_tiles_against_oracle(native, oracle):
    """70 x 45 cells (50 arena tiles, so the D8 gather reads across tiles), every flow direction and pits at random, outlets sprinkled
    in: the routed step tracks the oracle on every plane -- free-running over the first 100 steps, then step by step from the oracle's
    state (the oneD columns have no snap-to-zero of emptied stores: from step 123 on a residue of 3e-18 mm in a saturated root zone
    decides a branch, as in the oneD goldens, and routing would carry such a tie downstream) -- and water does move between cells."""
    import hip_util as H
    from golden_util import compare_bulk

    g, names, forcing = load_case(ROUTING_CASES[0])
    nx, ny = 70, 45
    st, ctx = _tiled_case(native, oracle, g, names, nx, ny, np.random.default_rng(8))
    odrv, hdrv = oracle.ForcingDriver(forcing), H.HipForcingDriver(ctx, forcing)
    moved = 0.0
    for step in range(1, 161):
        if step > 100:
            H.upload_snapshot(ctx, st.snapshot(names), names)
        pd, td, ed, monthly = odrv.before_step(st)
        st.step(pd, td, ed, monthly)
        assert hdrv.before_step() == monthly
        ctx.step_routed(monthly)
        if step <= 3 or step % 20 == 0 or step > 100:
            np.testing.assert_array_equal(H.scalars_to_row(ctx.get_scalars()), st.scalars_row(), err_msg=f"step {step}")
            compare_bulk(H.download_snapshot(ctx, names), st.snapshot(names), names, what=f"step {step}", frac_bulk=0.9999 if step > 100 else 0.999)
            moved += float(ctx.download("q_sur_in").sum() + ctx.download("q_sub_in").sum())
    assert moved > 1.0, "no water was routed: the test would not see the gather"
    ctx.close()


def test_full_size_routed_columns_equal_a_strip(native, oracle):
    """10^6 cells all draining towards +y: every x-row of the grid is the same hillslope, so the state does not depend on x, and it
    equals the oracle's run of one 1 x 1000 strip."""
    import hip_util as H
    from golden_util import ATOL, RTOL

    g, names, forcing = load_case(ROUTING_CASES[0])
    nx, ny = 1000, 1000
    rng = np.random.default_rng(3)
    strip_pick = rng.integers(0, np.asarray(g["state0"]).shape[1], ny)
    src = np.asarray(g["state0"])[:, strip_pick].copy()
    src[names.index("flow_dir_topo")] = 4
    src[names.index("outer_boundary")] = 0
    src[names.index("k_st")] = rng.uniform(8, 30, ny)
    dx, dy = (float(v) for v in g["routing_dx_dy"])
    from golden_util import configure_settings

    st = oracle.OracleState(ny)
    st.load_snapshot(src, names)
    st.load_scalars(g["scal0"])
    st.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    configure_settings(st.settings, g)
    st.settings.nx, st.settings.ny = 1, ny
    ctx = native.Context(nx, ny, enable_lateral_flow=1, enable_routing_1D=1, dx=dx, dy=dy)
    for row, nm in zip(src, names):
        if nm in ctx.index:
            ctx.upload(nm, np.tile(row, nx))
    ctx.set_scalars(H.scalars_from_row(g["scal0"]))
    ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    ctx.set_lut_mlms(g["lut_mlms"])
    odrv, hdrv = oracle.ForcingDriver(forcing), H.HipForcingDriver(ctx, forcing)
    for step in range(60):
        pd, td, ed, monthly = odrv.before_step(st)
        st.step(pd, td, ed, monthly)
        assert hdrv.before_step() == monthly
        ctx.step_routed(monthly)
    np.testing.assert_array_equal(H.scalars_to_row(ctx.get_scalars()), st.scalars_row())
    ref = st.snapshot(names)
    routed = 0.0
    for nm in ("S", "S_rz", "S_ss", "theta_rz", "theta_ss", "z0", "z_sat", "q_sur_out", "q_sur_in", "q_sub_out", "q_sub_in", "q_sub", "q_ss", "aet"):
        got = ctx.download(nm).reshape(nx, ny)
        assert np.array_equal(got.min(axis=0), got.max(axis=0), equal_nan=True), nm   # the same hillslope in every row
        r = ref[names.index(nm)]
        with np.errstate(invalid="ignore"):
            bad = ~((np.abs(got[0] - r) <= ATOL + 1e-8 * np.abs(r)) | (np.isnan(got[0]) & np.isnan(r)))
        assert bad.mean() <= 0.002, (nm, int(bad.sum()), got[0][bad][:3], r[bad][:3])
        if nm in ("q_sur_in", "q_sub_in"):
            routed += float(np.nansum(np.abs(r)))
    assert routed > 0.1
    ctx.close()


def test_month_change_inside_device_driven_routed_steps(native, oracle, monkeypatch):
    """The month changes on the third day of the run: set_parameters' surface parameters are then recomputed inside the first pass
    (device-driven stepping, D->monthly) or by a kernel of their own (routine by routine) -- both track the oracle, whose driver decides
    on the host, and agree bit for bit."""
    import hip_util as H
    from golden_util import compare_bulk
    from roger_amd.forcing import combo_forcing

    g, names, _ = load_case(ROUTING_CASES[0])
    F = {k: np.array(v) for k, v in combo_forcing(ndays=8).items()}
    F["MONTH"] = np.where(np.arange(F["MONTH"].size) < 2 * 144, 4, 5).astype(F["MONTH"].dtype)   # April -> May: the vegetation tables differ
    F["DOY"] = (119 + np.arange(F["DOY"].size) // 144).astype(F["DOY"].dtype)
    nx, ny = 20, 16
    out = {}
    for mode in ("device", "routine"):
        if mode == "routine":
            monkeypatch.setenv("RH_ROUTED_BY_ROUTINE", "1")
        else:
            monkeypatch.delenv("RH_ROUTED_BY_ROUTINE", raising=False)
        st, ctx = _tiled_case(native, oracle, g, names, nx, ny, np.random.default_rng(21))
        for s_ in (st.scal,):   # the golden's start state sits in January: move the calendar of both sides to the series' April
            s_.month[0] = s_.month[1] = 4
            s_.doy[0] = s_.doy[1] = 119
        ctx.set_scalars(H.scalars_from_row(st.scalars_row()))
        ctx.set_forcing_series(F)
        odrv = oracle.ForcingDriver(F)
        months, changed = set(), 0
        gc0 = st.planes["ground_cover"].copy()
        for step in range(1, 91):
            if step > 12:   # from here on step by step from the oracle's state (residue ties of the oneD columns, see above)
                H.upload_snapshot(ctx, st.snapshot(names), names)
            pd, td, ed, monthly = odrv.before_step(st)
            changed += int(monthly)
            st.step(pd, td, ed, monthly)
            months.add(int(st.scal.month[1]))
            ctx.run_steps(1)
            if step % 10 == 0 or monthly or step > 12:
                np.testing.assert_array_equal(H.scalars_to_row(ctx.get_scalars()), st.scalars_row(), err_msg=f"{mode} step {step}")
                compare_bulk(H.download_snapshot(ctx, names), st.snapshot(names), names, what=f"{mode} step {step}")
        assert months == {4, 5} and changed == 1 and not np.array_equal(gc0, st.planes["ground_cover"])
        out[mode] = H.download_snapshot(ctx, names)
        ctx.close()
    assert np.array_equal(out["device"], out["routine"], equal_nan=True)
