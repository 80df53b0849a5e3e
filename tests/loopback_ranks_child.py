"""Child process of tests/test_hip_loopback_ranks.py: TWO ranks as two threads of this process on one GPU, each with its own context,
joined by the loopback communicator (tests/loopback_nccl.cpp through RH_RCCL_LIB) -- the library's own multi-rank code with nranks = 2.

    python tests/loopback_ranks_child.py allreduce | routing | routing_by_routine
"""
import ctypes as C
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import hip_util as H  # noqa: E402
from golden_util import ROUTING_CASES, ROUTING_STEP_CASES, compare, load_case  # noqa: E402
from roger_amd import _native as native  # noqa: E402


def run_ranks(fn, n=2):
    """fn(rank) on n threads; the first exception of any rank is raised here (a rank that dies leaves its peer waiting: time limit)."""
    errs, out = [None] * n, [None] * n

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            errs[r] = e

    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=240)
    alive = [t.is_alive() for t in th]
    for e in errs:
        if e is not None:
            raise e
    assert not any(alive), f"ranks still waiting in a collective: {alive}"
    return out


def loopback_counts():
    lib = C.CDLL(os.environ["RH_RCCL_LIB"])
    s, r, a = C.c_ulong(), C.c_ulong(), C.c_ulong()
    lib.loopback_counts(C.byref(s), C.byref(r), C.byref(a))
    return s.value, r.value, a.value


def slab(g, key, nx, ny, x0, x1):
    snap = np.asarray(g[key])
    return snap.reshape(snap.shape[0], nx, ny)[:, x0:x1].reshape(snap.shape[0], -1)


def svat_ctx(g, names, forcing, columns=None):
    nx, ny = (int(v) for v in g["nx_ny"])
    x0, x1 = columns or (0, nx)
    ctx = native.Context(x1 - x0, ny)
    H.upload_snapshot(ctx, slab(g, "state0", nx, ny, x0, x1), names)
    ctx.set_scalars(H.scalars_from_row(g["scal0"]))
    ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    ctx.set_forcing_series(forcing)
    return ctx


def scenario_allreduce():
    """rh_run_steps_dist on slabs of the heterogeneous 4 x 4 SVAT case (two halves; four slabs one column wide): the global predicates
    (time step class, events, snow, sanity) are all-reduced between the ranks every step; every rank must hold the golden run's scalars at
    every checked step and together they its state, and equal a single context's rh_run_steps bit for bit."""
    g, names, forcing = load_case("svat_hetero_combo")
    nx, ny = (int(v) for v in g["nx_ny"])
    checks = (1, 2, 40, 41, 120, 240)
    whole = svat_ctx(g, names, forcing)
    done, want = 0, []
    for step in checks:
        whole.run_steps(step - done)
        done = step
        want.append(H.scalars_to_row(whole.get_scalars()))
        np.testing.assert_array_equal(want[-1], g["scal"][step - 1], err_msg=f"single context, step {step}")
        if step == 41:   # the host touches planes in between: the exchange restarts from the arena
            whole.upload("S_dep", whole.download("S_dep"))
    ref = H.download_snapshot(whole, names).reshape(len(names), nx, ny)
    whole.close()
    for edges in ([0, nx // 2, nx], list(range(nx + 1))):
        nr = len(edges) - 1
        uid = native.comm_unique_id()
        _, _, a0 = loopback_counts()

        def rank(r):
            ctx = svat_ctx(g, names, forcing, columns=(edges[r], edges[r + 1]))
            ctx.comm_init(uid, nr, r)
            assert ctx.comm_info() == (nr, r)
            done, rows = 0, []
            for step in checks:
                ctx.run_steps_dist(step - done)
                done = step
                rows.append(H.scalars_to_row(ctx.get_scalars()))
                if step == 41:
                    ctx.upload("S_dep", ctx.download("S_dep"))
            snap = H.download_snapshot(ctx, names)
            ctx.close()
            return rows, snap

        res = run_ranks(rank, nr)
        for k, step in enumerate(checks):
            for r in range(nr):
                np.testing.assert_array_equal(res[r][0][k], want[k], err_msg=f"rank {r} of {nr}, step {step}: scalars differ from the single domain")
        got = np.concatenate([res[r][1].reshape(len(names), -1, ny) for r in range(nr)], axis=1)
        assert np.array_equal(got, ref, equal_nan=True), f"{nr} ranks: planes differ: {[names[p] for p in np.unique(np.argwhere(got != ref)[:, 0])][:8]}"
        compare(got.reshape(len(names), -1), g["s00240"], names, what=f"{nr} ranks, rh_run_steps_dist, step 240")
        s, _, a1 = loopback_counts()
        assert a1 - a0 >= 240 and s == 0, (s, a1 - a0)
        print(f"allreduce: {nr} ranks == single domain == golden over 240 steps; {a1 - a0} all-reduces through the loopback communicator")


def routed_ctx(g, names, key="state0", scal_key="scal0", columns=None):
    nx, ny = (int(v) for v in g["nx_ny"])
    dx, dy = (float(v) for v in g["routing_dx_dy"])
    x0, x1 = columns or (0, nx)
    ctx = native.Context(x1 - x0, ny, enable_lateral_flow=1, enable_routing_1D=1, dx=dx, dy=dy)
    H.upload_snapshot(ctx, slab(g, key, nx, ny, x0, x1), names)
    ctx.set_scalars(H.scalars_from_row(g[scal_key] if isinstance(scal_key, str) else scal_key))
    ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
    ctx.set_lut_mlms(g["lut_mlms"])
    return ctx


def scenario_routing():
    """Device-driven routed steps (rh_run_steps on routing contexts with a communicator of 2 and of 3 - 4 ranks): per step the predicate
    words are all-reduced and the edge columns of both routings go through route_exchange's grouped send / receive -- with more than two
    ranks the inner ranks talk to both neighbours in one group, and a slab one column wide sends the same column both ways.  The slabs
    together equal the single domain bit for bit and the reference's golden trajectory; water crosses the cuts."""
    for case in ROUTING_STEP_CASES:
        g, names, forcing = load_case(case)
        nx, ny = (int(v) for v in g["nx_ny"])
        if nx < 2:
            continue
        nsteps = int(g["nsteps"])
        whole = routed_ctx(g, names)
        whole.set_forcing_series(forcing)
        whole.run_steps(nsteps)
        want = H.scalars_to_row(whole.get_scalars())
        np.testing.assert_array_equal(want, g["scal"][nsteps - 1])
        ref = H.download_snapshot(whole, names).reshape(len(names), nx, ny)
        whole.close()
        splits = [[0, nx // 2, nx], list(range(nx + 1)) if nx <= 4 else [0, 2, 3, nx]]   # two halves; one-column slabs / an inner one-column slab
        for edges in splits:
            nr = len(edges) - 1
            uid = native.comm_unique_id()
            s0, _, a0 = loopback_counts()

            def rank(r):
                ctx = routed_ctx(g, names, columns=(edges[r], edges[r + 1]))
                ctx.set_forcing_series(forcing)
                ctx.comm_init(uid, nr, r)
                assert ctx.comm_info() == (nr, r)
                ctx.run_steps(nsteps)
                row = H.scalars_to_row(ctx.get_scalars())
                snap = H.download_snapshot(ctx, names)
                ctx.close()
                return row, snap

            res = run_ranks(rank, nr)
            for r in range(nr):
                np.testing.assert_array_equal(res[r][0], want, err_msg=f"{case} rank {r} of {nr}: scalars")
            got = np.concatenate([res[r][1].reshape(len(names), -1, ny) for r in range(nr)], axis=1)
            assert np.array_equal(got, ref, equal_nan=True), f"{case}, {nr} ranks: planes differ: {[names[p] for p in np.unique(np.argwhere(got != ref)[:, 0])][:8]}"
            if case in ROUTING_CASES:   # (the combo case is pinned step by step only: free-running it meets oneD residue ties)
                compare(got.reshape(len(names), -1), g[f"s{nsteps:05d}"], names, what=f"{case}: {nr} ranks, routed run_steps")
            s1, _, a1 = loopback_counts()
            # per step: a surface and a subsurface exchange, one send per neighbour and rank (+ the static edges once)
            assert s1 - s0 >= 4 * (nr - 1) * nsteps and a1 - a0 >= nsteps, (case, nr, s1 - s0, a1 - a0)
            print(f"routing {case}: {nr} ranks == single domain{' == golden' if case in ROUTING_CASES else ''} over {nsteps} steps; {s1 - s0} sends, {a1 - a0} all-reduces")


def scenario_routing_by_routine():
    """The routine-by-routine entry points a hook-preserving RogerSetup.step() uses: rh_surface_routing / rh_subsurface_routing on two
    halves with the communicator equal the single domain at every stored step of the golden run (the exchange inside route_all)."""
    case = ROUTING_STEP_CASES[0]
    g, names, _ = load_case(case)
    nx, ny = (int(v) for v in g["nx_ny"])
    cut = nx // 2
    steps = sorted({int(k[1:6]) for k in g.files if k.startswith("r") and k.endswith("_calculate_surface_runoff")})[:6]
    uids = {step: native.comm_unique_id() for step in steps}   # a communicator per stored step (the contexts are rebuilt from the golden state)

    def rank(r):
        out = []
        for step in steps:
            kp = f"r{step:05d}_calculate_infiltration"
            ctx = routed_ctx(g, names, key=kp, scal_key=kp + "_scal", columns=((0, cut), (cut, nx))[r])
            ctx.comm_init(uids[step], 2, r)
            ctx.call("rh_surface_routing")
            ctx.call("rh_subsurface_runoff")
            ctx.call("rh_subsurface_routing")
            out.append(H.download_snapshot(ctx, names))
            ctx.close()
        return out

    res = run_ranks(rank)
    for k, step in enumerate(steps):
        kp = f"r{step:05d}_calculate_infiltration"
        whole = routed_ctx(g, names, key=kp, scal_key=kp + "_scal")
        whole.call("rh_surface_routing")
        whole.call("rh_subsurface_runoff")
        whole.call("rh_subsurface_routing")
        ref = H.download_snapshot(whole, names).reshape(len(names), nx, ny)
        got = np.concatenate([res[r][k].reshape(len(names), -1, ny) for r in range(2)], axis=1)
        assert np.array_equal(got, ref, equal_nan=True), f"step {step}: planes differ: {[names[p] for p in np.unique(np.argwhere(got != ref)[:, 0])][:8]}"
        whole.close()
    print(f"routing by routine {case}: 2 ranks == single domain at {len(steps)} golden steps")


if __name__ == "__main__":
    {"allreduce": scenario_allreduce, "routing": scenario_routing, "routing_by_routine": scenario_routing_by_routine}[sys.argv[1]]()
