"""The parameter planes of the fused step as the lazy kernels read them (DevState::pmask, include/roger_hip.h: rh_param_stats):

  * a parameter plane with ONE value over a wavefront's 64 columns is read as one element per wave;
  * the 15 parameters calc_parameters_soil derives from the primaries (roger/core/soil.py:143-557) are evaluated by the stages instead
    of being loaded, where the planes hold exactly what the setup kernel computed.

Both are decided per wavefront from the planes themselves, on the device, whenever somebody other than the fused kernel may have changed
them; so is a third bit: every column of the wave lies in the catchment (maskCatch == 1), and the wave runs the copy of the pipeline in
which the mask is the constant 1 (the reference multiplies every assignment by the mask: ~ 350 multiplications by 1.0 per column and step
fold away).  What a caller can observe must not change: every plane after rh_run_steps equals, BIT FOR BIT, what a context with both switched
off (RH_NO_PARAM_UNIFORM / RH_NO_PARAM_DERIVE at rh_create) leaves -- on the golden cases as they are, with a derived plane overwritten
by other values on some columns, and with a parameter changed between two calls."""
import os

import numpy as np
import pytest

from golden_util import CASES, load_case
from test_hip_parity import _ctx, native  # noqa: F401

pytestmark = pytest.mark.gpu
SWITCHES = ("RH_NO_PARAM_UNIFORM", "RH_NO_PARAM_DERIVE", "RH_NO_MASK_CONSTANT")


def make_ctx(native, g, names, forcing, plain):  # noqa: F811
    for k in SWITCHES:
        os.environ.pop(k, None)
        if plain:
            os.environ[k] = "1"
    try:
        ctx = _ctx(native, g, names)
    finally:
        for k in SWITCHES:
            os.environ.pop(k, None)
    ctx.set_forcing_series(forcing)
    return ctx


def all_planes(ctx, names):
    import hip_util as H

    return H.scalars_to_row(ctx.get_scalars()), H.download_snapshot(ctx, names)


def run_both(native, case, edit=None, stops=(7, 40, 41, 90)):  # noqa: F811
    g, names, forcing = load_case(case)
    nsteps = int(g["nsteps"])
    stops = sorted({min(s, nsteps) for s in stops})
    out = []
    for plain in (False, True):
        ctx = make_ctx(native, g, names, forcing, plain)
        if edit is not None:
            edit(ctx, 0)
        rows, done, stats = [], 0, []
        for k, stop in enumerate(stops):
            ctx.run_steps(stop - done)
            done = stop
            stats.append(ctx.param_stats())
            rows.append(all_planes(ctx, names))
            if edit is not None:
                edit(ctx, k + 1)
        out.append((rows, stats))
        ctx.close()
    (fast, fstats), (plain_rows, pstats) = out
    assert all(s == (0.0, 0.0) for s in pstats), pstats
    for (sa, pa), (sb, pb) in zip(fast, plain_rows):
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(pa, pb)      # every plane, bit for bit (NaN == NaN position-wise)
    return fstats


@pytest.mark.parametrize("case", CASES)
def test_parameter_fast_paths_change_no_bit(native, case):  # noqa: F811
    stats = run_both(native, case)
    assert all(d == 1.0 for d, _ in stats), stats   # the goldens' planes are what the setup kernels computed: derived everywhere
    if "uniform" in case:
        assert all(u > 150 for _, u in stats), stats   # one value per plane over the grid: nearly every parameter load is one element


def test_overwritten_derived_plane_is_loaded_again(native):  # noqa: F811
    """A derived plane that holds OTHER values on some columns (a script that sets theta_sat or wfs itself after setup): the waves of those
    columns load the planes, the others keep deriving; the step uses the uploaded values (both contexts do, bit for bit)."""
    def edit(ctx, k):
        if k == 0:
            for nm, f in (("wfs", 1.25), ("S_pwp_ss", 0.5)):
                a = ctx.download(nm)
                a[: max(1, ctx.n // 3)] *= f
                ctx.upload(nm, a)

    stats = run_both(native, "svat_hetero_combo", edit=edit)
    assert all(0.0 <= d < 1.0 for d, _ in stats), stats


def test_parameter_changed_between_calls_is_seen(native):  # noqa: F811
    """An upload between two rh_run_steps calls (time-variant parameters set by a hook): the wave words are formed again from the planes
    -- a uniform plane that stops being uniform is read column by column, a primary that changes without its derived planes turns the
    derivation off for its waves (the reference keeps using the planes it has, so must the kernel)."""
    def edit(ctx, k):
        if k == 1:
            a = ctx.download("ks")
            a[::2] *= 1.5
            ctx.upload("ks", a)
        if k == 2:
            a = ctx.download("theta_pwp")
            a[: max(1, ctx.n // 2)] += 0.01
            ctx.upload("theta_pwp", a)

    stats = run_both(native, "svat_uniform_rain", edit=edit)
    assert stats[0][0] == 1.0 and stats[-1][0] < 1.0, stats
    assert stats[1][1] < stats[0][1], stats


@pytest.mark.parametrize("lateral", [False, True])
@pytest.mark.parametrize("layout", ["random", "blocks"])
def test_many_wavefronts_with_a_ragged_last_one(lateral, layout):
    """1 000 columns = 15 full wavefronts and one of 40 lanes, heterogeneous parameters drawn per column ("random": no wave is uniform)
    or per block of 96 columns ("blocks": soil / land-use units -- some waves uniform, some straddle a boundary), 150 device-driven steps
    over the combo forcing incl. a month change: fast and plain contexts agree on every plane, bit for bit."""
    from roger_amd.forcing import combo_forcing
    from roger_amd.svat import create_svat, hetero_params

    nx, ny = 40, 25
    n = nx * ny
    p = hetero_params(n, seed=5)
    if layout == "blocks":
        block = np.arange(n) // 96
        p = {k: np.asarray(v)[block * 96] for k, v in p.items()}
    if lateral:
        p.update(slope=0.05, slope_per=5, dmph=50.0)
        p["z_soil"] = np.maximum(p["z_soil"], 800.0)
    forcing = combo_forcing(ndays=45)
    res, stats = [], []
    for plain in (False, True):
        for k in SWITCHES:
            os.environ.pop(k, None)
            if plain:
                os.environ[k] = "1"
        try:
            ctx = create_svat(nx, ny, params=p, lateral=lateral)
        finally:
            for k in SWITCHES:
                os.environ.pop(k, None)
        ctx.set_forcing_series(forcing)
        ctx.run_steps(100)
        ctx.run_steps(50)
        stats.append(ctx.param_stats())
        s = ctx.get_scalars()
        assert s.sanity_ok == 1 and s.itt == 150
        res.append({nm: ctx.download(nm) for nm, _ in ctx.planes[: ctx.planes_held]})
        ctx.close()
    assert stats[1] == (0.0, 0.0) and stats[0][0] == 1.0
    if layout == "blocks":
        assert stats[0][1] > 40.0, stats     # the waves inside a block read their parameters as one element
    for nm in res[0]:
        np.testing.assert_array_equal(res[0][nm], res[1][nm], err_msg=nm)
