"""Multi-GPU stepping from C (rh_comm_init + rh_run_steps_dist: ncclAllReduce -> control kernel -> fused kernel per step, no
host code between the steps) on ONE GPU with a one-rank RCCL communicator: the all-reduce is then a copy and the trajectory must
equal rh_run_steps' bit for bit.  The two-rank protocol itself is covered on CPU by tests/test_distributed_gloo.py (gloo ranks
over the oracle double); RCCL does not place two ranks on one device.  Replaces what the reference does per step through rank 0
(roger/core/adaptive_time_stepping_dist_safe.py:6-26)."""
import numpy as np
import pytest

from golden_util import compare, load_case

pytestmark = pytest.mark.gpu


def _ctx(native, g, names):
    import hip_util as H

    nx, ny = (int(v) for v in g["nx_ny"])
    ctx = native.Context(nx, ny)
    H.upload_snapshot(ctx, g["state0"], names)
    ctx.set_scalars(H.scalars_from_row(g["scal0"]))
    ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    return ctx


def test_one_rank_rccl_equals_run_steps():
    import hip_util as H
    from roger_amd import _native as native

    g, names, forcing = load_case("svat_hetero_combo")
    a, b = _ctx(native, g, names), _ctx(native, g, names)
    for c in (a, b):
        c.set_forcing_series(forcing)
    with pytest.raises(native.NativeError, match="communicator"):
        b.run_steps_dist(1)
    b.comm_init(native.comm_unique_id(), 1, 0)
    done = 0
    for step in (1, 2, 40, 41, 120, 240):
        a.run_steps(step - done)
        b.run_steps_dist(step - done)
        done = step
        np.testing.assert_array_equal(H.scalars_to_row(a.get_scalars()), H.scalars_to_row(b.get_scalars()), err_msg=f"step {step}")
        np.testing.assert_array_equal(H.scalars_to_row(b.get_scalars()), g["scal"][step - 1], err_msg=f"step {step} vs golden")
        if step == 41:   # the host touches the planes in between: the exchange restarts from the arena
            for c in (a, b):
                c.upload("S_dep", c.download("S_dep"))
    sa, sb = H.download_snapshot(a, names), H.download_snapshot(b, names)
    np.testing.assert_array_equal(sa, sb)
    compare(sb, g["s00240"], names, what="rh_run_steps_dist, one rank, step 240")
    assert b.get_scalars().sanity_ok == 1
    # mixing the two drivers on one context
    b.run_steps(7)
    b.run_steps_dist(5)
    a.run_steps(12)
    np.testing.assert_array_equal(H.download_snapshot(a, names), H.download_snapshot(b, names))
    a.close()
    b.close()


def test_comm_through_torch_process_group():
    """comm_init_torch: the id travels through torch.distributed (here a one-rank gloo group stands in for the launcher's)."""
    import os

    import torch.distributed as dist

    import hip_util as H
    from roger_amd import _native as native

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g, names, forcing = load_case("svat_uniform_rain")
        ctx = _ctx(native, g, names)
        ctx.set_forcing_series(forcing)
        ctx.comm_init_torch()
        ctx.run_steps_dist(int(g["nsteps"]))
        np.testing.assert_array_equal(H.scalars_to_row(ctx.get_scalars()), g["scal"][-1])
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_time_limit_with_the_multi_gpu_stepping():
    """rh_set_time_limit under rh_run_steps_dist (the all-reduce and the control kernel run in front of every step; the control
    kernel finds the run over): the same stop, the same state as rh_run_steps under the limit, and the communicator is reported."""
    import hip_util as H
    from roger_amd import _native as native

    g, names, forcing = load_case("svat_hetero_combo")
    a, b = _ctx(native, g, names), _ctx(native, g, names)
    for c in (a, b):
        c.set_forcing_series(forcing)
    b.comm_init(native.comm_unique_id(), 1, 0)
    assert b.comm_info() == (1, 0) and a.comm_info() == (1, 0)
    stop = 80
    t_stop = int(g["scal"][stop - 1][1])
    for c in (a, b):
        c.set_time_limit(t_stop)
    a.run_steps(stop + 25)
    b.run_steps_dist(stop + 25)
    for c in (a, b):
        np.testing.assert_array_equal(H.scalars_to_row(c.get_scalars()), g["scal"][stop - 1])
    sa, sb = H.download_snapshot(a, names), H.download_snapshot(b, names)
    np.testing.assert_array_equal(sa, sb)
    compare(sb, g[f"s{stop:05d}"], names, what="rh_run_steps_dist stopped by the time limit")
    b.set_time_limit(None)
    a.set_time_limit(None)
    a.run_steps(9)
    b.run_steps_dist(9)
    np.testing.assert_array_equal(H.download_snapshot(a, names), H.download_snapshot(b, names))
    np.testing.assert_array_equal(H.scalars_to_row(b.get_scalars()), g["scal"][stop + 8])
    a.close()
    b.close()


def test_time_limit_is_refused_where_the_control_part_does_not_observe_it():
    """Per-cell forcing takes the predicate-kernel control path, which does not look at the limit: rh_run_steps says so instead of
    running past it."""
    from golden_util import load_weights
    from roger_amd import _native as native

    g, names, forcing = load_case("svat_eberbaechle_weights")
    w = load_weights(g)
    ctx = _ctx(native, g, names)
    ctx.set_forcing_series(forcing)
    ctx.set_forcing_weights(w["prec_weight"], w["ta_offset"], w["pet_weight"])
    ctx.run_steps(3)
    ctx.set_time_limit(10 * 86400)
    with pytest.raises(native.NativeError, match="rh_set_time_limit"):
        ctx.run_steps(5)
    ctx.set_time_limit(None)
    ctx.run_steps(5)
    ctx.close()
