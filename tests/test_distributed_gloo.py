"""N > 1 path on CPU: two `gloo` ranks, each owning half of the grid (split along x as
`num_proc=(2, 1)`), step in lock-step through roger_amd.distributed.PhasedStepper -- the same
orchestration the HIP backend uses with RCCL -- and must reproduce the single-domain reference
trajectory.  The oracle stands in for the device (OracleContext test double)."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def test_decomposition_api():
    from roger_amd import distributed as D

    D.validate_decomposition(8, 6, (2, 1), 2)
    with pytest.raises(RuntimeError):
        D.validate_decomposition(8, 6, (2, 1), 3)      # roger/distributed.py:128-131
    with pytest.raises(ValueError):
        D.validate_decomposition(9, 6, (2, 1), 2)      # :133-134
    assert D.get_chunk_size(8, 6, (2, 1)) == (4, 6)
    assert D.proc_rank_to_index(3, (2, 2)) == (1, 1) and D.proc_index_to_rank(1, 1, (2, 2)) == 3
    g, l = D.get_chunk_slices(8, 6, (2, 1), rank=1)
    assert g == (slice(4, 8), slice(0, 6)) and l == (slice(0, 4), slice(0, 6))
    g, l = D.get_chunk_slices(8, 6, (2, 1), rank=0, include_overlap=True)  # :166-170
    assert g == (slice(0, 6), slice(0, 10)) and l == (slice(0, 6), slice(0, 10))


def _worker(rank, world, port, case, nsteps, out_dir, one_exchange=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_context as OC
    from golden_util import load_case
    from roger_amd.distributed import PhasedStepper, get_chunk_slices

    g, names, forcing = load_case(case)
    nx, ny = (int(v) for v in g["nx_ny"])
    (gx, gy), _ = get_chunk_slices(nx, ny, (world, 1), rank)
    sel = np.arange(nx * ny).reshape(nx, ny)[gx, gy].ravel()
    ctx = OC.OracleContext(nx // world, ny)
    for row, nm in zip(g["state0"], names):
        ctx.upload(nm, row[sel])
    from hip_util import scalars_from_row

    ctx.set_scalars(scalars_from_row(g["scal0"]))
    ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    ctx.set_forcing_series(forcing)
    PhasedStepper(OC.OraclePhases(ctx, one_exchange=one_exchange)).run(nsteps)
    snap = np.stack([ctx.download(nm).astype(np.float64) for nm in names])
    s = ctx.get_scalars()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), snap=snap, sel=sel,
             scal=np.array([s.itt, s.time, s.dt_secs, s.itt_day, s.event_id[1], s.event_id_counter, s.dt]))
    dist.destroy_process_group()


def test_summary_derivation_matches_direct_evaluation(oracle):
    """Single domain: the predicate words derived from the summary word (the one-exchange protocol) equal the ones
    evaluated column by column, on every step of a trajectory with rain, snow, pauses and all three step classes;
    the trajectory itself matches the reference."""
    import oracle_context as OC
    from golden_util import compare, load_case
    from hip_util import scalars_from_row

    g, names, forcing = load_case("svat_hetero_combo")
    nx, ny = (int(v) for v in g["nx_ny"])
    ctx = OC.OracleContext(nx, ny)
    for row, nm in zip(g["state0"], names):
        ctx.upload(nm, row)
    ctx.set_scalars(scalars_from_row(g["scal0"]))
    ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    ctx.set_forcing_series(forcing)
    classes = set()
    for step in range(1, 241):
        ctx._hooks()
        w0, w1, w1_direct = ctx.finish_from_summary(ctx.summary_word(), -1)
        assert w1 == w1_direct, f"step {step}: derived {w1:#x}, direct {w1_direct:#x}"
        classes.add(int(ctx.get_scalars().dt_secs))
    assert classes == {600, 3600, 86400}
    snap = np.stack([ctx.download(nm).astype(np.float64) for nm in names])
    compare(snap, g["s00240"], names, what="summary protocol, single domain, step 240")


@pytest.mark.parametrize("case,nsteps,one_exchange", [("svat_hetero_combo", 120, False), ("svat_hetero_combo", 240, False),
                                                      ("svat_hetero_combo", 240, True)])
def test_two_ranks_reproduce_single_domain(tmp_path, oracle, case, nsteps, one_exchange):
    from golden_util import compare, load_case

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, case, nsteps, str(tmp_path), one_exchange), nprocs=2, join=True)
    g, names, _ = load_case(case)
    ref = g[f"s{nsteps:05d}"]
    got = np.full_like(ref, np.nan)
    for r in range(2):
        d = np.load(tmp_path / f"rank{r}.npz")
        got[:, d["sel"]] = d["snap"]
        gs = g["scal"][nsteps - 1]
        np.testing.assert_array_equal(d["scal"], [gs[0], gs[1], gs[2], gs[3], gs[9], gs[6], gs[7]])
    compare(got, ref, names, what=f"{case}: 2 ranks vs single-domain reference, step {nsteps}")
