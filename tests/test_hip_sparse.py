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
    ties, checked = (TieTracker(names, ctx.n) if case in ONED_TIE_COLUMNS else None), 0
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
    ties = TieTracker(names, int(np.prod(g["nx_ny"]))) if case in ONED_TIE_COLUMNS else None
    for stop, (sa, pa), (sb, pb) in zip(stops, a, b):
        np.testing.assert_array_equal(sa, sb, err_msg=f"{case} scalars after step {stop}")
        np.testing.assert_array_equal(pa, pb, err_msg=f"{case} planes after step {stop}: sparse stores changed what the caller sees")
        np.testing.assert_array_equal(sa, g["scal"][stop - 1], err_msg=f"{case} scalars step {stop}")
        if ties:
            ties.check(pa, g[f"s{stop:05d}"], stop, what=case)
        else:
            compare(pa, g[f"s{stop:05d}"], names, what=f"{case} step {stop} (rh_run_steps in pieces)")


def test_an_accumulated_pure_output_plane_switches_the_option_off(native):  # noqa: F811
    g, names, forcing = load_case("svat_hetero_combo")
    ctx = _ctx(native, g, names)
    assert "q_ss" in ctx.pure_output_planes() and "S_fp_rz" not in ctx.pure_output_planes()
    ctx.set_forcing_series(forcing)
    ctx.diag_configure(rate=["S_fp_rz"], collect=[], n_slots=4)   # state only: sparse stores stay on
    ctx.run_steps(30)
    assert ctx.sparse_steps() >= 28
    ctx.diag_configure(rate=["q_ss", "aet"], collect=["S_fp_rz"], n_slots=4)
    ctx.run_steps(30)
    assert ctx.sparse_steps() == 0
    ctx.close()


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
