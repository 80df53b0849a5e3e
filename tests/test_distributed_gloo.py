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


def _halo_worker(rank, world, port, num_proc, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch

    from roger_amd import distributed as D

    nxl, nyl = 3, 4
    px, py = D.proc_rank_to_index(rank, num_proc)
    # interior cell (i, j) of the global grid holds 100 * i + j (+ 0.5 in the second layer); ghosts start as -1
    gi = np.arange(px * nxl, (px + 1) * nxl)[:, None] * 100.0 + np.arange(py * nyl, (py + 1) * nyl)[None, :]
    a = -np.ones((nxl + 4, nyl + 4, 2))
    a[2:-2, 2:-2, 0] = gi
    a[2:-2, 2:-2, 1] = gi + 0.5
    out_np = D.exchange_overlap(a, num_proc)
    out_t = D.exchange_overlap(torch.from_numpy(a.copy()), num_proc).numpy()
    assert np.array_equal(out_np, out_t)
    ok = D.global_and(rank >= 0) and not D.global_and(rank == 0) and D.global_or(rank == 0)
    tot = D.global_sum(float(rank)), D.global_max(rank), D.global_min(rank)
    np.savez(os.path.join(out_dir, f"halo{rank}.npz"), a=out_np, ok=ok, tot=np.array(tot, dtype=float))
    dist.destroy_process_group()


@pytest.mark.parametrize("num_proc", [(2, 1), (2, 2)])
def test_exchange_overlap_and_reductions(tmp_path, num_proc):
    """The ghost frame comes from the neighbours' interiors (edges and corners), stays untouched at the domain edge;
    the global reductions agree on every rank (roger/distributed.py:223-416)."""
    from roger_amd import distributed as D

    world = num_proc[0] * num_proc[1]
    port = 29500 + (os.getpid() % 2000) + 7
    mp.spawn(_halo_worker, args=(world, port, num_proc, str(tmp_path)), nprocs=world, join=True)
    nxl, nyl = 3, 4
    for rank in range(world):
        d = np.load(tmp_path / f"halo{rank}.npz")
        a = d["a"]
        px, py = D.proc_rank_to_index(rank, num_proc)
        for li in range(nxl + 4):
            for lj in range(nyl + 4):
                i, j = px * nxl + li - 2, py * nyl + lj - 2          # global cell this local (ghost) cell stands for
                inside = 0 <= i < nxl * num_proc[0] and 0 <= j < nyl * num_proc[1]
                want = 100.0 * i + j if inside else -1.0
                assert a[li, lj, 0] == want and a[li, lj, 1] == (want + 0.5 if inside else -1.0), (rank, li, lj)
        assert bool(d["ok"]) and list(d["tot"]) == [sum(range(world)), world - 1, 0]


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


def _setup_worker(rank, world, port, case, ndays, out_dir, device_hooks):
    """RogerSetup on two ranks (num_proc = (2, 1)): the hook-preserving run() and run_device(), the device replaced by the double."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from roger_amd import runtime_settings

    runtime_settings.update(num_proc=(world, 1))     # before the core is imported (locked afterwards, as in the reference)
    import oracle_binding as ob
    import oracle_context
    import svat_scripts as S
    from golden_util import load_case
    from roger_amd import _native
    from roger_amd.distributed import get_chunk_slices

    _native.Context = oracle_context.OracleContext
    _native.plane_table = lambda: list(zip(ob.plane_names(), ob.plane_is_int()))
    g, names, forcing = load_case(case)
    nx, ny = (int(v) for v in g["nx_ny"])
    (gx, gy), _ = get_chunk_slices(nx, ny, (world, 1), rank)
    p = {k: v[gx, gy] for k, v in S.params_from_golden(g, names).items()}
    model = S.make_model(p, forcing, ndays, global_shape=(nx, ny))
    model.setup()
    vs = model.state.variables
    assert np.asarray(vs.lu_id).shape == (nx // world + 4, ny + 4)
    if device_hooks:
        steps = int(np.sum(g["scal"][:, 1] <= ndays * 86400))
        model.run_device(steps)
    else:
        model.run()
    snap = S.snapshot_from_vs(vs, names)
    sel = np.arange(nx * ny).reshape(nx, ny)[gx, gy].ravel()
    np.savez(os.path.join(out_dir, f"setup{rank}.npz"), snap=snap, sel=sel,
             scal=np.array([int(vs.itt), int(vs.time), int(vs.dt_secs), int(vs.itt_day), int(vs.event_id_counter)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("device_hooks", [False, True])
def test_roger_setup_on_two_ranks(tmp_path, oracle, device_hooks):
    """`RogerSetup.run()` (user hooks on the host, rh_adaptive_dt in three parts with the predicate words all-reduced in between) and
    `run_device()` (one exchange of the summary word per step) with num_proc = (2, 1) reproduce the single-domain reference run:
    every rank derives the same dt, event ids and forcing window.  (The reference decides dt through rank 0,
    adaptive_time_stepping_dist_safe.py:6-26.)  A world size that does not match num_proc is refused (roger/distributed.py:128-131)."""
    from golden_util import compare, load_case

    case, ndays = "svat_hetero_combo", 24    # the whole golden run: its last step carries a snapshot
    port = 29500 + (os.getpid() % 2000) + 11 + int(device_hooks)
    mp.spawn(_setup_worker, args=(2, port, case, ndays, str(tmp_path), device_hooks), nprocs=2, join=True)
    g, names, _ = load_case(case)
    nsteps = int(np.sum(g["scal"][:, 1] <= ndays * 86400))
    key = f"s{nsteps:05d}"
    gs = g["scal"][nsteps - 1]
    got = None
    for r in range(2):
        d = np.load(tmp_path / f"setup{r}.npz")
        np.testing.assert_array_equal(d["scal"], [gs[0], gs[1], gs[2], gs[3], gs[6]], err_msg=f"rank {r}")
        if key in g.files:
            got = np.full_like(g[key], np.nan) if got is None else got
            got[:, d["sel"]] = d["snap"]
    if got is not None:
        compare(got, g[key], names, what=f"RogerSetup on 2 ranks, step {nsteps}")


def _mismatch_worker(rank, world, port):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    import oracle_context
    import svat_scripts as S
    from golden_util import load_case
    from roger_amd import _native

    _native.Context = oracle_context.OracleContext
    _native.plane_table = lambda: list(zip(ob.plane_names(), ob.plane_is_int()))
    g, names, forcing = load_case("svat_uniform_rain")
    model = S.make_model(S.params_from_golden(g, names), forcing, 1)     # num_proc left at (1, 1) on two ranks
    try:
        model.setup()
    except RuntimeError as e:
        assert "does not match size of communicator" in str(e)
    else:
        raise AssertionError("setup() accepted num_proc = (1, 1) on two ranks")
    dist.destroy_process_group()


def test_world_size_must_match_num_proc():
    mp.spawn(_mismatch_worker, args=(2, 29500 + (os.getpid() % 2000) + 17), nprocs=2, join=True)


def _routed_worker(rank, world, port, out_dir, device_hooks):
    """The routed hillslope (settings.enable_routing_1D) on two ranks: every rank holds half of the x range; routed water crosses the cut
    straight and diagonally, the edge columns travel between the out and the in part of both routings."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from roger_amd import runtime_settings

    runtime_settings.update(num_proc=(world, 1))
    import oracle_binding as ob
    import oracle_context
    import svat_scripts as S
    from golden_util import load_case, routing_of
    from roger_amd import _native
    from roger_amd.distributed import get_chunk_slices

    _native.Context = oracle_context.OracleContext
    _native.plane_table = lambda: list(zip(ob.plane_names(), ob.plane_is_int()))
    g, names, forcing = load_case("oned_routing")
    nx, ny = (int(v) for v in g["nx_ny"])
    (gx, gy), _ = get_chunk_slices(nx, ny, (world, 1), rank)
    p = {k: v[gx, gy] for k, v in S.params_from_golden(g, names).items()}
    ndays = len(forcing["PREC"]) // 144
    model = S.make_model(p, forcing, ndays, lateral=True, global_shape=(nx, ny), routing=routing_of(g, names, columns=(gx.start, gx.stop)))
    model.setup()
    vs = model.state.variables
    nsteps = int(g["nsteps"])
    if device_hooks:
        model.run_device(nsteps)
    else:
        for _ in range(nsteps):
            model.step(model.state)
    snap = S.snapshot_from_vs(vs, names)
    sel = np.arange(nx * ny).reshape(nx, ny)[gx, gy].ravel()
    np.savez(os.path.join(out_dir, f"routed{rank}.npz"), snap=snap, sel=sel,
             scal=np.array([int(vs.itt), int(vs.time), int(vs.dt_secs), int(vs.itt_day), int(vs.event_id_counter)]))
    model.state.backend_context.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("device_hooks", [False, True])
def test_routing_on_two_ranks(tmp_path, oracle, device_hooks):
    """settings.enable_routing_1D with num_proc = (2, 1): the two halves of the routed hillslope, exchanging their edge columns in both
    routings of every step, reproduce the reference's single-domain run (which the reference itself could not: it never exchanges the
    routed water between processes)."""
    from golden_util import compare, load_case

    port = 29500 + (os.getpid() % 2000) + 23 + int(device_hooks)
    mp.spawn(_routed_worker, args=(2, port, str(tmp_path), device_hooks), nprocs=2, join=True)
    g, names, _ = load_case("oned_routing")
    nsteps = int(g["nsteps"])
    ref, gs = g[f"s{nsteps:05d}"], g["scal"][nsteps - 1]
    got = np.full_like(ref, np.nan)
    for r in range(2):
        d = np.load(tmp_path / f"routed{r}.npz")
        np.testing.assert_array_equal(d["scal"], [gs[0], gs[1], gs[2], gs[3], gs[6]], err_msg=f"rank {r}")
        got[:, d["sel"]] = d["snap"]
    compare(got, ref, names, what=f"routing on 2 ranks, step {nsteps}")


def _per_cell_worker(rank, world, port, case, nsteps, out_dir):
    """run_device() on two ranks with per-cell forcing (station weights / several stations): the three-phase protocol, both
    predicate words evaluated over every rank's columns and all-reduced (rh_run_steps_dist and the summary path refuse
    per-cell forcing)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from roger_amd import runtime_settings

    runtime_settings.update(num_proc=(world, 1))
    import oracle_binding as ob
    import oracle_context
    import svat_scripts as S
    from golden_util import load_case, load_stations, load_weights
    from roger_amd import _native
    from roger_amd.distributed import get_chunk_slices

    _native.Context = oracle_context.OracleContext
    _native.plane_table = lambda: list(zip(ob.plane_names(), ob.plane_is_int()))
    g, names, forcing = load_case(case)
    nx, ny = (int(v) for v in g["nx_ny"])
    (gx, gy), _ = get_chunk_slices(nx, ny, (world, 1), rank)
    p = {k: v[gx, gy] for k, v in S.params_from_golden(g, names).items()}
    w = load_weights(g)
    weights = {k: v.reshape(nx, ny)[gx, gy] for k, v in w.items()} if w else None
    stations = None
    if load_stations(g) is not None:
        stations = dict(station_ids=g["station_station_ids"], station_id=np.asarray(g["station_station_id"]).reshape(nx, ny)[gx, gy],
                        PREC=g["station_PREC"], TA=g["station_TA"], PET=g["station_PET"])
    model = S.make_model(p, forcing, len(forcing["PREC"]) // 144, global_shape=(nx, ny), weights=weights, stations=stations)
    model.setup()
    model.run_device(nsteps)
    assert model._per_cell_forcing
    vs = model.state.variables
    sel = np.arange(nx * ny).reshape(nx, ny)[gx, gy].ravel()
    np.savez(os.path.join(out_dir, f"percell{rank}.npz"), snap=S.snapshot_from_vs(vs, names), sel=sel,
             scal=np.array([int(vs.itt), int(vs.time), int(vs.dt_secs), int(vs.itt_day)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("case,nsteps", [("svat_eberbaechle_weights", 100), ("svat_stations", 120)])
def test_per_cell_forcing_on_two_ranks(tmp_path, oracle, case, nsteps):
    """The distributed catchment setups (eberbaechle/svat_distributed/svat.py:276-296; svat_dist.py:274-322) with num_proc = (2, 1):
    snow falls on part of the grid only, so the step length is decided by columns of BOTH ranks; the two-rank run_device() reproduces
    the reference's single-domain run."""
    from golden_util import compare, load_case

    port = 29500 + (os.getpid() % 2000) + 31 + (case == "svat_stations")
    mp.spawn(_per_cell_worker, args=(2, port, case, nsteps, str(tmp_path)), nprocs=2, join=True)
    g, names, _ = load_case(case)
    gs = g["scal"][nsteps - 1]
    got = np.full_like(g[f"s{nsteps:05d}"], np.nan)
    for r in range(2):
        d = np.load(tmp_path / f"percell{r}.npz")
        np.testing.assert_array_equal(d["scal"], [gs[0], gs[1], gs[2], gs[3]], err_msg=f"rank {r}")
        got[:, d["sel"]] = d["snap"]
    compare(got, g[f"s{nsteps:05d}"], names, what=f"{case} on 2 ranks, step {nsteps}")
