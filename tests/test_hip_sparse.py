"""Sparse stores inside rh_run_steps (k_step<..., LAZY, SPARSE>): every step of a call that another step of the same call follows
leaves out the stores of the planes the step only PRODUCES (tools/liveness.py; include/roger_hip.h).  What a caller can observe must
not change:

  * the planes in question are never read: poisoned with NaN before EVERY step, the golden trajectories of the reference still come
    out (all planes at the stored steps, integer scalars at every step);
  * rh_run_steps in pieces ending on every stored step of the golden trajectories gives the reference's state on every plane, and
    the same bits as the library with the option switched off;
  * an accumulator that was given such a plane switches the option off."""
import os

import numpy as np
import pytest

from golden_util import CASES, ONED_TIE_COLUMNS, TieTracker, compare, deviating_columns, load_case
from test_hip_parity import _ctx, native  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES)
def test_pure_output_planes_are_never_read(native, case):  # noqa: F811
    import hip_util as H

    g, names, forcing = load_case(case)
    ctx = _ctx(native, g, names)
    poison = ctx.pure_output_planes()
    assert len(poison) >= 60
    nan = np.full(ctx.n, np.nan)
    drv = H.HipForcingDriver(ctx, forcing)
    ties, checked = (TieTracker.for_case(case, names, ctx.n)), 0
    for step in range(1, int(g["nsteps"]) + 1):
        for nm in poison:
            ctx.upload(nm, nan)
        ctx.step(drv.before_step())
        s = ctx.get_scalars()
        np.testing.assert_array_equal(H.scalars_to_row(s), g["scal"][step - 1], err_msg=f"{case} scalars step {step}")
        assert s.sanity_ok == 1
        key = f"s{step:05d}"
        if key in g.files:
            if ties:
                ties.check(H.download_snapshot(ctx, names), g[key], step, what=case)
            else:
                compare(H.download_snapshot(ctx, names), g[key], names, what=f"{case} step {step}, pure outputs poisoned before the step")
            checked += 1
    assert checked >= 3
    ctx.close()


def _pieces(native, g, names, forcing, stops, sparse):  # noqa: F811
    import hip_util as H

    if sparse:
        os.environ.pop("RH_NO_SPARSE_STORES", None)
    else:
        os.environ["RH_NO_SPARSE_STORES"] = "1"
    try:
        ctx = _ctx(native, g, names)
    finally:
        os.environ.pop("RH_NO_SPARSE_STORES", None)
    ctx.set_forcing_series(forcing)
    out, done, used = [], 0, 0
    for stop in stops:
        ctx.run_steps(stop - done)
        n, done = stop - done, stop
        used += ctx.sparse_steps()
        if sparse and n >= 3:
            assert ctx.sparse_steps() >= n - 2, (n, ctx.sparse_steps())   # all but the last (and the first, eager after an upload / download)
        out.append((H.scalars_to_row(ctx.get_scalars()), H.download_snapshot(ctx, names)))
    ctx.close()
    return out, used


@pytest.mark.parametrize("case", CASES)
def test_run_steps_in_pieces_leaves_every_plane_as_full_steps_do(native, case):  # noqa: F811
    g, names, forcing = load_case(case)
    stops = sorted(int(k[1:]) for k in g.files if k.startswith("s") and k[1:].isdigit())
    a, used = _pieces(native, g, names, forcing, stops, sparse=True)
    b, none = _pieces(native, g, names, forcing, stops, sparse=False)
    assert used > 0 and none == 0
    ties = TieTracker.for_case(case, names, int(np.prod(g["nx_ny"])))
    for stop, (sa, pa), (sb, pb) in zip(stops, a, b):
        np.testing.assert_array_equal(sa, sb, err_msg=f"{case} scalars after step {stop}")
        np.testing.assert_array_equal(pa, pb, err_msg=f"{case} planes after step {stop}: sparse stores changed what the caller sees")
        np.testing.assert_array_equal(sa, g["scal"][stop - 1], err_msg=f"{case} scalars step {stop}")
        if ties:
            ties.check(pa, g[f"s{stop:05d}"], stop, what=case)
        else:
            compare(pa, g[f"s{stop:05d}"], names, what=f"{case} step {stop} (rh_run_steps in pieces)")


def test_accumulated_pure_output_planes_are_kept_and_the_option_stays_on(native, monkeypatch):  # noqa: F811
    """An accumulator that was given pure-output planes (q_ss, aet ...) used to switch the sparse stores off; since the end of round 3 the
    sparse kernel stores THOSE planes after all (DevState::keep) and leaves the other ~70 unwritten.  The daily sums equal, bit for bit,
    the sums of a context that stores everything (RH_NO_SPARSE_STORES), over every day of the run, and so does the state at the end."""
    import hip_util as H

    g, names, forcing = load_case("svat_hetero_combo")
    assert {"q_ss", "aet"} <= set(_ctx(native, g, names).pure_output_planes())
    rate, collect = ["q_ss", "aet", "inf_mat_rz", "prec"], ["S_fp_rz", "theta"]
    out = []
    for no_sparse in (False, True):
        if no_sparse:
            monkeypatch.setenv("RH_NO_SPARSE_STORES", "1")
        ctx = _ctx(native, g, names)
        ctx.set_forcing_series(forcing)
        ctx.diag_configure(rate=["S_fp_rz"], collect=[], n_slots=4)   # state only
        ctx.run_steps(30)
        assert ctx.sparse_steps() == 0 if no_sparse else ctx.sparse_steps() >= 28   # (the first step after an upload is eager)
        ctx.diag_configure(rate=rate, collect=collect, n_slots=16)
        ctx.run_steps(200)
        assert ctx.sparse_steps() == 0 if no_sparse else ctx.sparse_steps() >= 198
        sums = {nm: [ctx.diag_download(nm, slot) for slot in range(4)] for nm in rate + collect}
        out.append((sums, H.download_snapshot(ctx, names), [ctx.diag_steps(slot) for slot in range(4)]))
        ctx.close()
    assert out[0][2] == out[1][2] and sum(out[0][2]) > 0
    for nm in rate + collect:
        for slot in range(4):
            np.testing.assert_array_equal(out[0][0][nm][slot], out[1][0][nm][slot], err_msg=f"{nm} slot {slot}")
    assert any(np.any(out[0][0]["q_ss"][slot] != 0) for slot in range(4)) and any(np.any(out[0][0]["aet"][slot] != 0) for slot in range(4))
    assert np.array_equal(out[0][1], out[1][1], equal_nan=True)


# ---- the routed step (settings.enable_routing_1D): three passes, a pass also keeps what a later pass of the same step loads --------
@pytest.mark.parametrize("case", ["oned_routing", "oned_routing_tutorial"])
def test_routed_pure_output_planes_are_never_read(native, case):  # noqa: F811
    """The planes the routed step only produces (RH_SPARSE_FIELDS_ROUTED: the flow analysis over the three passes with the gathers
    as assignments of q_*_in) poisoned with NaN before every step: the reference's routed trajectory still comes out."""
    import hip_util as H
    from test_hip_routing import routed_ctx

    g, names, forcing = load_case(case)
    ctx = routed_ctx(native, g, names)
    poison = [nm for nm in ctx.pure_output_planes() if ctx.index[nm] < ctx.planes_held]
    assert len(poison) >= 100 and "q_sur_out" not in poison and "q_sub_out" not in poison and "q_ss" in poison and "S_rz" in poison
    nan = np.full(ctx.n, np.nan)
    drv = H.HipForcingDriver(ctx, forcing)
    checked = 0
    for step in range(1, int(g["nsteps"]) + 1):
        for nm in poison:
            ctx.upload(nm, nan)
        ctx.step_routed(drv.before_step())
        np.testing.assert_array_equal(H.scalars_to_row(ctx.get_scalars()), g["scal"][step - 1], err_msg=f"{case} scalars step {step}")
        key = f"s{step:05d}"
        if key in g.files:
            compare(H.download_snapshot(ctx, names), g[key], names, what=f"{case} step {step}, pure outputs poisoned before the step")
            checked += 1
    assert checked >= 3
    ctx.close()


@pytest.mark.parametrize("case", ["oned_routing", "oned_routing_tutorial"])
def test_routed_run_steps_in_pieces_leaves_every_plane_as_full_steps_do(native, case):  # noqa: F811
    import hip_util as H
    from test_hip_routing import routed_ctx

    g, names, forcing = load_case(case)
    stops = sorted(int(k[1:]) for k in g.files if k.startswith("s") and k[1:].isdigit())
    runs = {}
    for sparse in (True, False):
        if not sparse:
            os.environ["RH_NO_SPARSE_STORES"] = "1"
        try:
            ctx = routed_ctx(native, g, names)
        finally:
            os.environ.pop("RH_NO_SPARSE_STORES", None)
        ctx.set_forcing_series(forcing)
        out, done, used = [], 0, 0
        for stop in stops:
            ctx.run_steps(stop - done)
            used += ctx.sparse_steps()
            if sparse and stop - done >= 2:
                assert ctx.sparse_steps() == stop - done - 1
            done = stop
            out.append((H.scalars_to_row(ctx.get_scalars()), H.download_snapshot(ctx, names)))
        runs[sparse] = (out, used)
        ctx.close()
    assert runs[True][1] > 0 and runs[False][1] == 0
    for stop, (sa, pa), (sb, pb) in zip(stops, runs[True][0], runs[False][0]):
        np.testing.assert_array_equal(sa, sb)
        assert np.array_equal(pa, pb, equal_nan=True), f"{case} step {stop}: sparse stores changed what the caller sees"
        np.testing.assert_array_equal(sa, g["scal"][stop - 1])
        compare(pa, g[f"s{stop:05d}"], names, what=f"{case} step {stop} (routed rh_run_steps in pieces)")
