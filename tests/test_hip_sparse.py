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

from golden_util import CASES, ONED_TIE_COLUMNS, compare, deviating_columns, load_case
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
    off, checked = set(), 0
    for step in range(1, int(g["nsteps"]) + 1):
        for nm in poison:
            ctx.upload(nm, nan)
        ctx.step(drv.before_step())
        s = ctx.get_scalars()
        np.testing.assert_array_equal(H.scalars_to_row(s), g["scal"][step - 1], err_msg=f"{case} scalars step {step}")
        assert s.sanity_ok == 1
        key = f"s{step:05d}"
        if key in g.files:
            if case in ONED_TIE_COLUMNS:
                off |= deviating_columns(H.download_snapshot(ctx, names), g[key])
                assert len(off) <= ONED_TIE_COLUMNS[case] + 1, f"{case} step {step}: columns {sorted(off)} deviate"
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
    off = set()
    for stop, (sa, pa), (sb, pb) in zip(stops, a, b):
        np.testing.assert_array_equal(sa, sb, err_msg=f"{case} scalars after step {stop}")
        np.testing.assert_array_equal(pa, pb, err_msg=f"{case} planes after step {stop}: sparse stores changed what the caller sees")
        np.testing.assert_array_equal(sa, g["scal"][stop - 1], err_msg=f"{case} scalars step {stop}")
        if case in ONED_TIE_COLUMNS:
            off |= deviating_columns(pa, g[f"s{stop:05d}"])
            assert len(off) <= ONED_TIE_COLUMNS[case] + 1
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
