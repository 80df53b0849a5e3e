#!/usr/bin/env python3
"""Benchmark of the SVAT hot path: cell-timesteps/s on the SVAT_benchmark grid.

    python bench.py --gpus N --steps K --warmup W [--size NX NY]

One "step" is one model time step (adaptive dt: 24 h / 1 h / 10 min) of every column of the grid:
the device-side user hooks, the predicate kernels and the fused per-column kernel.  The workload
is BASELINE.json configs[1]: SVAT_benchmark, nx*ny = 10^6 uniform benchmark parameters
(benchmarks/SVAT_benchmark.py:92-103,117-121), float64, synthetic forcing (seeded
roger_amd.forcing.combo_forcing; the reference's forcing.nc is not shipped).  All state is resident
in HBM before the timed region.  For N > 1 every rank owns one nx*ny slab of a grid split along x
(weak scaling); the only data-path communication is one 64-value predicate all-reduce per step.

Prints one JSON line on rank 0 (see the bench contract).  The `roofline` entry prices the fused kernel `k_step`
against the 8 TB/s HBM peak, with the kernel duration measured by HIP events on the kernel's stream inside the timed
region:

  * `achieved` / `frac`: the bytes one launch must move -- per column what the kernel variant that ran loads and stores
    (every plane the step needs read once, every plane it assigns written once; counted from the kernel's gfx950 ISA by
    tools/isa_census.py into roger_amd/csrc/rh_step_bytes.json) x the launch's columns -- over the average kernel
    duration.  Physical: always <= 1.
  * `traffic`: the HBM bytes per launch measured with the PMC counters FETCH_SIZE / WRITE_SIZE (separate passes,
    calibrated on a copy of known size with the same access shape; tools/make_profiles.sh -> profiles/traffic.json) for
    this kernel variant at this column count, or null if that combination was not profiled.
  * `reference_equivalent`: the same kernel time against SURVEY.md section 8(d)'s 2 779 B per cell-step -- the distinct
    variables the REFERENCE's step reads and assigns.  The fused kernel keeps the intermediates in registers and defers
    the tau -> taum1 copies, so it moves fewer bytes than that; this figure can exceed the peak and is not a roofline
    fraction.

The `cpu_baseline` entry times the oracle (a C port of the reference's NumPy step, OpenMP over the columns) on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

REFERENCE_BYTES_PER_CELL_STEP = 2779  # 1555 read + 1224 written, SURVEY.md section 8(d): the reference's read / write sets (SVAT)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s


# a wave that reads ONE element of a plane fetches a 128-byte line instead of 512 bytes: priced at 1/4 (PMC, profiles/traffic.json: 793.6 B per
# column measured with the benchmark's uniform parameters against 728 B of state and stores + 232 B of parameter loads x 1/4 = 786)
UNIFORM_READ_FRACTION = 1.0 / 4.0


def kernel_bytes_per_cell(model, variant, param_stats=None):
    """(load, store) bytes one column moves per fused step of that kernel variant ("eager", "lazy", "sparse"):
    roger_amd/csrc/rh_step_bytes.json (tools/isa_census.py).  The lazy / sparse variants read the parameter planes through the wave's
    word (include/roger_hip.h: rh_param_stats): `param_stats` = (fraction of the waves whose derived parameters are not loaded, bytes
    per column of parameter loads that are one element per wave) of the context that ran."""
    rec = json.load(open(os.path.join(REPO, "roger_amd", "csrc", "rh_step_bytes.json")))
    r = rec[f"{'oned' if model == 'oned' else 'svat'}_{variant}"]
    ld = r["load_bytes"]
    if "load_bytes_all_parameters_loaded" in r:
        derived, uniform = param_stats if param_stats is not None else (0.0, 0.0)
        full = r["load_bytes_all_parameters_loaded"]
        ld = full - derived * (full - r["load_bytes"]) - uniform * (1.0 - UNIFORM_READ_FRACTION)
    return ld, r["store_bytes"]


def measured_traffic(key, n_cells):
    """HBM bytes per launch from profiles/traffic.json (PMC passes of tools/make_profiles.sh) for this kernel variant, only if it
    was profiled at this column count.  Returns (bytes per launch or None, the record or None)."""
    tf = os.path.join(REPO, "profiles", "traffic.json")
    try:
        rec = json.load(open(tf)).get(key, {}).get(str(n_cells))
    except Exception:   # noqa: BLE001
        rec = None
    if not rec:
        return None, None
    return rec["hbm_bytes_per_launch"], rec


def prewarm(torch, device, ms):
    """A device-to-device copy of 256 MB repeated for `ms` milliseconds: untimed work that brings the device to the clock state it
    holds under load (not steps of the model; before the timed region)."""
    if ms <= 0:
        return
    a = torch.empty(1 << 25, dtype=torch.float64, device=device)
    b = torch.empty_like(a)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            b.copy_(a)
        torch.cuda.synchronize(device)
    del a, b


def device_clocks():
    """Clock levels rocm-smi reports (a child process; nothing here touches the GPU): the fused kernel's speed follows the
    memory-side clocks of the moment."""
    import re
    import subprocess

    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=15).stdout
        return {k: int(v) for k, v in re.findall(r"GPU\[0\]\s*:\s*(\w+) clock level: \d+: \((\d+)Mhz\)", out)}
    except Exception:   # noqa: BLE001
        return None


def host_threads():
    """Threads for the CPU baseline: the CPUs this process may run on, at most 16 (a one-GPU box's share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("RH_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(n_cells, steps, forcing):
    """Oracle (oracle/svat_oracle.c, OpenMP over the columns) on the host: same parameters, same forcing, same steps.
    Returns (cell-timesteps/s, seconds, threads used)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import numpy as np

    import oracle_binding as ob
    from roger_amd import lookuptables as lut
    from roger_amd.svat import BENCHMARK_PARAMS

    ob.lib().oc_set_num_threads(host_threads())
    st = ob.OracleState(n_cells)
    st.set_luts(lut.ARR_ILU, lut.ARR_GC, lut.ARR_GCM, lut.ARR_RDLU)
    P = st.planes
    P["maskCatch"][:] = 1
    for nm, v in (("ta", 15.0), ("ta_m1", 15.0), ("z_gw", 1000.0), ("z_gw_m1", 1000.0), ("c_int", 1.0), ("c_root", 1.0)):
        P[nm][:] = v
    for k, v in BENCHMARK_PARAMS.items():
        if k not in ("theta_rz", "theta_ss"):
            P[k][:] = v
    st.scal.dt, st.scal.dt_secs, st.scal.event_id_counter = 1.0, 3600, 1
    for k, v in (("year", 1900), ("month", 1), ("doy", 1)):
        getattr(st.scal, k)[0] = getattr(st.scal, k)[1] = v
    st.topo()
    st.params_surface()
    st.params_soil()
    for lvl in ("", "_m1"):
        P["theta_rz" + lvl][:] = BENCHMARK_PARAMS["theta_rz"]
        P["theta_ss" + lvl][:] = BENCHMARK_PARAMS["theta_ss"]
    st.initial_conditions()
    drv = ob.ForcingDriver(forcing)
    t0 = time.perf_counter()
    for _ in range(steps):
        pd, td, ed, monthly = drv.before_step(st)
        st.step(pd, td, ed, monthly)
    dt = time.perf_counter() - t0
    return n_cells * steps / dt, dt, int(ob.lib().oc_num_threads())


SAS_S_RZ, SAS_S_SS = 90.0, 260.0   # initial root zone / subsoil storage in mm (uniform benchmark soil)


def cpu_baseline_sas(n_cells, ndays, ages, substeps, daily, solver="deterministic"):
    """Oracle (oracle/sas_oracle.c) on the host: first n_cells columns, first ndays days of the same inputs."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import sas_binding as sb
    from roger_amd import sas as rsas

    sb.lib().oc_sas_set_num_threads(host_threads())
    st = sb.SasState(n_cells, ages, substeps, age_statistics=True, solver=solver)
    for key, S in (("rz", SAS_S_RZ), ("ss", SAS_S_SS)):
        sa, msa = rsas.initial_age_state([S] * n_cells, ages)
        st.state[f"sa_{key}"][:] = sa
        st.state[f"msa_{key}"][:] = msa
    for f, p in rsas.benchmark_sas_params(n_cells).items():
        st.sas[f][:] = p
    t0 = time.perf_counter()
    for d in range(ndays):
        for k in st.inp:
            st.inp[k][:] = daily[k][d, :n_cells]
        st.step_oracle()
    dt = time.perf_counter() - t0
    return n_cells * ndays / dt, dt, int(sb.lib().oc_sas_num_threads())


def bench_sas(args, torch, dist, rank, local_rank, world, device):
    """BASELINE configs[2]: SVATOXYGEN18_benchmark, nx*ny columns per GPU, ages = 1000, 6 sub-steps, power-law
    SAS with the benchmark's exponents, age statistics on.  One step = one day of every column (rh_sas_step)."""
    from roger_amd import sas as rsas

    nx, ny = args.size
    n = nx * ny
    ndays_resident = 8
    daily = rsas.synthetic_daily_inputs(n, ndays_resident, seed=42 + rank)
    ctx = rsas.create_sas(n, args.ages, args.substeps, SAS_S_RZ, SAS_S_SS, daily=daily, device=local_rank, age_statistics=True,
                          solver=args.sas_solver)
    euler = args.sas_solver != "deterministic"
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    ctx.run_days(0, args.warmup)
    ctx.enable_timing(True)
    fence()
    t0 = time.perf_counter()
    ctx.run_days(args.warmup, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.timing_summary()
    ctx.enable_timing(False)
    ctx.sync()   # raises if a column asked for an unsupported SAS family
    d18O = ctx.download("C_iso_q_ss")[:4]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        algo = 8 * args.ages * 8           # read + write of sa_rz, msa_rz, sa_ss, msa_ss (64 000 B at ages = 1000)
        k_avg_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = algo * n / k_avg_s / 1e9
        pows = 5 * args.substeps * (args.ages + 1)
        kname = {"Euler": "k_sas_euler", "RK4": "k_sas_rk4"}.get(args.sas_solver, "k_sas")
        traffic, trec = measured_traffic(f"{kname}_ages{args.ages}_sub{args.substeps}", n)
        traffic_note = f"; traffic = PMC FETCH_SIZE / WRITE_SIZE of this kernel at this column count ({trec['source']})" if trec else ""
        # compute side ("state the compute bound for SAS explicitly"): the kernel is bound by fp64 VALU ISSUE.  Per column-day it
        # issues `valu_wave_insts` wave-instructions (PMC SQ_INSTS_VALU / columns, profiles/sas_valu.json); an fp64 wave64 instruction
        # occupies its SIMD for 4 cycles (MI355X_MICROARCH.md: 64 lanes over 16-wide fp64 VALUs), the chip has 256 CUs x 4 SIMDs.
        compute = None
        try:
            vrec = json.load(open(os.path.join(REPO, "profiles", "sas_valu.json")))[("" if not euler else args.sas_solver.lower() + "_") + f"ages{args.ages}_sub{args.substeps}"]
            cyc = vrec["valu_wave_insts_per_column"] * n * 4.0 / 1024.0
            clock_hz = vrec.get("clock_mhz", 2400) * 1e6
            compute = {
                "bound": "fp64 VALU issue",
                "valu_wave_insts_per_column_day": vrec["valu_wave_insts_per_column"],
                "cycles_per_wave_inst": 4,
                "simds": 1024,
                "clock_mhz": vrec.get("clock_mhz", 2400),
                "min_ms_at_full_issue": cyc / clock_hz * 1e3,
                "frac": (cyc / clock_hz) / k_avg_s,
                "source": vrec.get("source"),
            }
        except Exception:   # noqa: BLE001
            compute = None
        if compute and compute["frac"] > 1.0:   # the record belongs to another build of the kernel: not a fraction of anything
            compute = None
        out = {
            "metric": "cell-timesteps/sec on SVATOXYGEN18_benchmark grid",
            "value": world * n * args.steps / elapsed,
            "unit": "cell-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"SVATOXYGEN18_benchmark (offline oxygen-18 transport, {args.sas_solver} SAS solver) nx*ny={n} per GPU, "
                            f"ages={args.ages}, sas_solver_substeps={args.substeps}, power-law SAS (benchmark exponents), "
                            "age statistics on, synthetic daily fluxes (seed 42); one step = one day",
                "cells_per_gpu": n,
                "decomposition": f"({world},1) along x, no exchange",
                "d18O_q_ss_sample": [None if x != x else float(x) for x in d18O],
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "note": f"algorithmic bytes = state read + written once per day; the kernel is fp64-ALU bound by "
                        f"<= {pows} pow per column-day (fluxes that are 0 on a day are skipped), see DESIGN.md" + traffic_note,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": algo * n,
                "avg_kernel_ms": k_avg_s * 1e3,
                "launches_timed": launches,
                "compute": compute,
            },
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported at N = 1 only
            cells = max(8, min(n, int(args.cpu_cells) // 25))
            days = min(ndays_resident, args.steps + args.warmup, 4)
            v, secs, threads = cpu_baseline_sas(cells, days, args.ages, args.substeps, daily, args.sas_solver)
            out["cpu_baseline"] = {
                "value": v,
                "unit": "cell-timesteps/s",
                "cores": threads,
                "kind": "port",
                "sample": f"oracle/sas_oracle.c (OpenMP over the columns), {cells} columns x first {days} days of the same inputs, "
                          f"{secs:.1f} s on {threads} host threads",
            }
        print(json.dumps(out))
    ctx.close()


def bench_routed(args, torch, dist, rank, local_rank, world, device):
    """The oneD benchmark's columns with settings.enable_routing_1D (surface and subsurface runoff routed to the D8 neighbour, every cell
    draining towards +y): the routed step, three per-column passes around the two gathers (rh_run_steps on a routing context); several
    ranks exchange the predicate words and the edge columns over RCCL from C."""
    from roger_amd.forcing import combo_forcing
    from roger_amd.svat import create_svat, hetero_params

    nx, ny = args.size
    n_local = nx * ny
    params = dict(hetero_params(n_local, seed=42 + rank)) if args.params == "hetero" else {}
    for k, v in dict(z_soil=1000.0, lmpv=600.0, slope=0.05, slope_per=5, dmph=50.0, flow_dir_topo=4, k_st=15.0).items():
        params.setdefault(k, v)
    ctx = create_svat(nx, ny, params=params, device=local_rank, lateral=True, enable_routing_1D=1, dx=5.0, dy=5.0)
    total_steps = args.steps + args.warmup
    ctx.set_forcing_series(combo_forcing(ndays=max(30, total_steps + 5)))
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    by_routine = bool(os.environ.get("RH_ROUTED_BY_ROUTINE"))   # A/B: the routine-by-routine control part (17 launches per step)
    first = "routed_a" if by_routine else "routed_a2"
    run, stepping = ctx.run_steps, ("rh_run_steps (routed, rh_step_routed per step: 17 launches)" if by_routine else
                                    "rh_run_steps (routed: control kernel on the posted summary bits, three passes with the gathers folded in: 4 launches per step)")
    if world > 1:
        ctx.comm_init_torch()
        run, stepping = ctx.run_steps_dist, "rh_run_steps_dist (routed; summary word and edge columns over RCCL from C)"

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    run(args.warmup)
    s0 = ctx.get_scalars()
    ctx.enable_timing(True)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.timing_summary()
    per_ms, _ = ctx.timing_detail() if not by_routine else (None, None)
    sparse_steps = ctx.sparse_steps()
    ctx.enable_timing(False)
    s1 = ctx.get_scalars()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if s1.itt - s0.itt != args.steps:   # (the reference's water balance check fails on routed runs: sanity_ok is not asserted)
        raise SystemExit(f"bench: step bookkeeping failed (itt {s0.itt}->{s1.itt})")
    if rank == 0:
        census = json.load(open(os.path.join(REPO, "roger_amd", "csrc", "rh_step_bytes.json")))   # bytes per column from the ISA (tools/isa_census.py)
        # device-driven: the gathers are folded into the second and third pass (k_routed_bg / k_routed_cg), and every step of the call
        # but the last leaves out the stores of the planes the routed step only produces (sparse stores)
        sparse = (not by_routine) and sparse_steps == launches - 1 and per_ms is not None and len(per_ms) == launches
        sfx = "_sparse" if sparse else ""
        names = (first, "routed_b", "routed_c_after") if by_routine or os.environ.get("RH_ROUTED_SEPARATE_GATHERS") else \
            (first + sfx, "routed_bg" + sfx, "routed_cg_after" + sfx)
        passes = {seq: (census[seq]["load_bytes"], census[seq]["store_bytes"]) for seq in names}
        first = names[0]
        ld_b, st_b = passes[first]
        k_avg_s = (float(per_ms[:-1].mean()) / 1e3) if sparse else kernel_ms / 1e3 / max(launches, 1)
        achieved = (ld_b + st_b) * n_local / k_avg_s / 1e9
        step_bytes = sum(sum(v) for v in passes.values())
        traffic, _ = measured_traffic("k_" + first, n_local)
        out = {
            "metric": "cell-timesteps/sec on SVAT_benchmark grid",
            "value": world * n_local * args.steps / elapsed,
            "unit": "cell-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"oneD_benchmark synthetic grid nx*ny={n_local} per GPU ({nx}x{ny}) with enable_routing_1D (every cell drains "
                            f"towards +y, Strickler coefficient 15, dx = dy = 5 m), {args.params} benchmark parameters, combo forcing (seed 42), adaptive dt",
                "cells_per_gpu": n_local,
                "global_cells": args.global_cells if args.global_cells else world * n_local,
                "simulated_seconds": int(s1.time - s0.time),
                "decomposition": f"({world},1) along x, " + ("one summary all-reduce and two edge-column exchanges per step" if world > 1
                                                              else "single GPU: no exchange"),
                "stepping": stepping + (f"; {sparse_steps} of {launches} steps with sparse stores" if sparse_steps else ""),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": f"k_{first} (" + ("" if by_routine else "forcing selection, ") + "interception ... infiltration + the surface outflow: the longest of the step's three passes)",
                "note": f"achieved = the bytes this pass loads + stores per column ({ld_b} + {st_b} B, roger_amd/csrc/rh_step_bytes.json from "
                        "the ISA) / its average duration by HIP events; the whole routed step moves "
                        f"{step_bytes} B per column in its three passes, the folded gathers' neighbour reads (cache hits mostly) included "
                        "(fused oneD step: 1240 B sparse / 2040 B), plus the adaptive time stepping",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": (ld_b + st_b) * n_local,
                "algorithmic_bytes_per_cell": {p: {"load": v[0], "store": v[1]} for p, v in passes.items()},
                "whole_step": {"bytes_per_cell": step_bytes, "achieved": step_bytes * n_local / (elapsed / args.steps) / 1e9,
                               "frac": step_bytes * n_local / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
                "avg_kernel_ms": k_avg_s * 1e3,
                "launches_timed": launches,
            },
        }
        print(json.dumps(out))
    ctx.close()


def bench_setup(args, torch, device):
    """`--stepping setup | hooks | routines`: the SVAT / oneD benchmark as a RogerSetup SCRIPT calling plain `run()` (roger/roger.py:523-580),
    one GPU.  setup: the script leaves the per-step hooks to the model class -- run() then advances through rh_run_steps under
    rh_set_time_limit, a few rounds per run.  hooks: the script brings a per-step hook of its own (here a read_data that does
    nothing), so run() keeps the reference's loop: the hooks on the host, then -- set_parameters and after_timestep being the model
    class's -- the physics of the step as ONE native call (rh_svat_step).  routines: the three native calls per step
    (rh_adaptive_dt, rh_step_core, rh_after_timestep) of a script that brings its own set_parameters or after_timestep.  One step = one model time step; the run covers `--days` days
    after `--warmup-days` untimed ones, the steps are counted from vs.itt."""
    from roger_amd import roger_routine
    from roger_amd.forcing import combo_forcing
    from roger_amd.models.oned import ONEDSetup
    from roger_amd.models.svat import SVATSetup
    from roger_amd.svat import BENCHMARK_PARAMS

    nx, ny = args.size
    p = {k: v for k, v in BENCHMARK_PARAMS.items() if k not in ("theta_rz", "theta_ss")}
    if args.model == "oned":
        p.update(z_soil=1000.0, lmpv=600.0, slope=0.05, dmph=50.0)
    base = ONEDSetup if args.model == "oned" else SVATSetup

    class Benchmark(base):
        initial_theta = dict(theta_rz=BENCHMARK_PARAMS["theta_rz"], theta_ss=BENCHMARK_PARAMS["theta_ss"])

    if args.stepping == "routines":
        os.environ["RH_STEP_BY_ROUTINE"] = "1"   # (what a script with a set_parameters / after_timestep hook that does something of its own gets)
    if args.stepping in ("hooks", "routines"):
        class Benchmark(Benchmark):   # noqa: F811
            hook_calls = 0

            @roger_routine
            def read_data(self, state):   # a per-step hook that does something of the script's own: run() must keep calling it
                type(self).hook_calls += 1
    if args.stepping == "script":
        # the hooks every script of the reference defines ITSELF, with the reference's bodies (benchmarks/SVAT_benchmark.py:105-110,
        # 152-181): recognised hook by hook (roger_amd/hooks.py), run() advances on the device
        from roger_amd import KernelOutput, roger_kernel
        from roger_amd.core.operators import at, numpy as npx, update
        from roger_amd.core.surface import calc_parameters_surface_kernel

        @roger_kernel
        def after_timestep_kernel(state):   # (a kernel of this name is the native rotation: roger_amd/routines.py)
            vs = state.variables
            return KernelOutput(S=update(vs.S, at[2:-2, 2:-2, vs.taum1], vs.S[2:-2, 2:-2, vs.tau]))

        class Benchmark(Benchmark):   # noqa: F811
            @roger_routine
            def set_parameters(self, state):
                vs = state.variables

                if (vs.month[vs.tau] != vs.month[vs.taum1]) & (vs.itt > 1):
                    vs.update(calc_parameters_surface_kernel(state))

            @roger_routine
            def set_forcing(self, state):
                vs = state.variables

                condt = vs.time % (24 * 60 * 60) == 0
                if condt:
                    vs.itt_day = 0
                    vs.year = update(vs.year, at[1], vs.YEAR[vs.itt_forc])
                    vs.month = update(vs.month, at[1], vs.MONTH[vs.itt_forc])
                    vs.doy = update(vs.doy, at[1], vs.DOY[vs.itt_forc])
                    vs.prec_day = update(vs.prec_day, at[:, :, :], vs.PREC[npx.newaxis, npx.newaxis, vs.itt_forc:vs.itt_forc + 6 * 24])
                    vs.ta_day = update(vs.ta_day, at[:, :, :], vs.TA[npx.newaxis, npx.newaxis, vs.itt_forc:vs.itt_forc + 6 * 24])
                    vs.pet_day = update(vs.pet_day, at[:, :, :], vs.PET[npx.newaxis, npx.newaxis, vs.itt_forc:vs.itt_forc + 6 * 24])
                    vs.itt_forc = vs.itt_forc + 6 * 24

            @roger_routine
            def after_timestep(self, state):
                vs = state.variables

                vs.update(after_timestep_kernel(state))

    total_days = args.warmup_days + args.days
    model = Benchmark(forcing=combo_forcing(ndays=total_days + 2), nx=nx, ny=ny, ndays=total_days, parameters=p)
    model.setup()
    state = model.state
    if os.environ.get("RH_NO_HOOK_RECOGNITION") and args.stepping == "script":
        assert not model.device_run_possible()   # (A/B: what such a script got before round 4 -- the three-call step)
    else:
        assert model.device_run_possible() is (args.stepping in ("setup", "script")), model.hook_classes()
    ctx = state.backend_context

    def run_days(days):
        with state.settings.unlock():
            state.settings.runlen = days * 86400
        vs = state.variables
        itt0 = int(vs.itt)
        ctx.sync()
        t0 = time.perf_counter()
        model.run()
        ctx.sync()
        return time.perf_counter() - t0, int(vs.itt) - itt0

    run_days(args.warmup_days)
    elapsed, steps = run_days(args.days)
    n = nx * ny
    out = {
        "metric": "cell-timesteps/sec on SVAT_benchmark grid",
        "value": n * steps / elapsed,
        "unit": "cell-timesteps/s",
        "n_gpus": 1,
        "steps": steps,
        "warmup": args.warmup_days,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{'oneD' if args.model == 'oned' else 'SVAT'}_benchmark as a RogerSetup script calling run(): nx*ny={n} ({nx}x{ny}), uniform benchmark "
                        f"parameters, combo forcing (seed 42), {args.days} days after {args.warmup_days} untimed ones",
            "cells_per_gpu": n,
            "hooks_on_device": model.hook_classes(),
            "stepping": "RogerSetup.run(): " + ("stock per-step hooks, rh_run_steps under rh_set_time_limit (rounds)" if args.stepping == "setup"
                                                else "the script's OWN set_forcing / set_parameters / after_timestep with the reference's bodies (benchmarks/SVAT_benchmark.py:105-110, 152-181), recognised by behaviour: rh_run_steps under rh_set_time_limit" if args.stepping == "script"
                                                else ("a per-step hook in front of the physics that does something of the script's own: the reference's loop, hooks on the host, the rest of the step one native call that also returns the scalars (rh_svat_step_scalars)" if args.stepping == "hooks"
                                                      else "the reference's loop with the three-call step: hooks on the host, rh_adaptive_dt + rh_step_core + rh_after_timestep per step")),
            "wall_s": elapsed,
        },
    }
    print(json.dumps(out))
    ctx.close()


def run_extra(name, argv, need_free_gb, torch, device, timeout_s=150):
    """One more bench line from a CHILD process of this script (own context, own memory; a failure there leaves the headline alone):
    returns the parsed line or {"skipped": reason}.  Never exec: the parent stays the process the caller waits for."""
    import subprocess

    free_b, _total = torch.cuda.mem_get_info(device)
    if free_b < need_free_gb * 2**30:
        return {"skipped": f"{name}: {free_b / 2**30:.0f} GiB of device memory free, {need_free_gb} GiB wanted"}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--no-extras", "--no-cpu-baseline"], env=env,
                           capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"skipped": f"{name}: no line within {timeout_s} s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"skipped": f"{name}: the child ended with status {r.returncode}: {r.stderr.strip().splitlines()[-1:] or ''}"}
    return json.loads(lines[-1])


def bench_extras(torch, device):
    """what the driver's one command should also witness -- the north-star size (the >= 50 % target is stated at
    nx * ny = 10^7) and the transport step (BASELINE configs[2]) --, measured AFTER the headline with the headline's own formulas
    (the same code: a child run of this script), reported under `extras`; the headline fields do not depend on them."""
    out = {}
    d = run_extra("svat_1e7", ["--size", "3200", "3125", "--steps", "20", "--warmup", "5", "--prewarm-ms", "0"], 70, torch, device)
    if "skipped" not in d:
        r = d["roofline"]
        d = {"workload": d["config"]["workload"], "value": d["value"], "ms_per_step": d["ms_per_step"], "avg_kernel_ms": r["avg_kernel_ms"],
             "frac": r["frac"], "bytes_per_cell": r["algorithmic_bytes_per_cell"], "kernel": r["kernel"], "steps": d["steps"],
             "traffic_frac": r.get("traffic_frac")}
    out["svat_1e7"] = d
    d = run_extra("sas_1e5", ["--model", "sas", "--size", "400", "250", "--steps", "4", "--warmup", "2"], 16, torch, device)
    if "skipped" not in d:
        r = d["roofline"]
        d = {"workload": d["config"]["workload"], "value": d["value"], "ms_per_step": d["ms_per_step"], "avg_kernel_ms": r["avg_kernel_ms"],
             "frac": r["frac"], "compute": {"frac": (r.get("compute") or {}).get("frac"),
                                            "valu_wave_insts_per_column_day": (r.get("compute") or {}).get("valu_wave_insts_per_column_day")},
             "steps": d["steps"], "unit": "column-days/s; ms per model day"}
    out["sas_1e5"] = d
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this script, one per GPU of this node, as CHILD processes
    (never exec: the parent stays the process the caller waits for), hand them the rendezvous through RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT -- what `python -m torch.distributed.run --nproc-per-node N` would set, and what the reference's
    benchmarks get from `mpirun -n N` (benchmarks/run_benchmarks.py:27-31,181; benchmark_base.py:8-29) --, pass on what they print
    (rank 0 prints the JSON line) and return their exit status: non-zero as soon as one rank fails, the others are then stopped."""
    import socket
    import subprocess

    import signal

    with socket.socket() as s:   # a free port on the loopback interface (released just before the ranks bind it: a taken port fails the
        s.bind(("127.0.0.1", 0))   # rendezvous loudly, the ranks exit non-zero and so does this launcher)
        port = s.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs = []

    def stop_all(grace=5.0):
        """Exactly the processes started here: terminate, then kill what is still there after `grace` seconds."""
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t_end = time.monotonic() + grace
        for p in alive:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()

    def on_signal(signum, _frame):   # a driver's timeout (SIGTERM) or Ctrl-C must not leave ranks behind holding the GPUs
        raise SystemExit(128 + signum)

    previous = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    status, failed_at = 0, None
    try:
        for r in range(n):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            time.sleep(0.2)
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad and failed_at is None:
                failed_at = time.monotonic()
                print(f"bench: rank {bad[0][0]} exited with status {bad[0][1]}; stopping the other ranks", file=sys.stderr)
            if failed_at is not None and time.monotonic() - failed_at > 5.0:   # a rank waiting in a collective for the failed one
                stop_all(grace=2.0)
        for r, p in enumerate(procs):
            rc = p.wait()
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
    finally:
        stop_all()
        for sig, handler in previous.items():
            signal.signal(sig, handler)
    return status


def launch_check(dist, torch, rank, world):
    """--launch-check: what a rank does up to the first collective, without a GPU (gloo)."""
    if os.environ.get("RH_BENCH_TEST_FAIL_RANK") == str(rank):   # the launcher's failure path (tests)
        raise SystemExit(3)
    if os.environ.get("RH_BENCH_TEST_HANG"):   # the launcher's signal path (tests): a rank that never finishes by itself
        print(f"rank-pid {os.getpid()}", flush=True)
        time.sleep(600)
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([rank + 1], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "sum_of_rank_ids_plus_one": int(t.item()),
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, nargs=2, default=(1000, 1000), metavar=("NX", "NY"))
    ap.add_argument("--global-size", type=int, nargs=2, default=None, metavar=("NX", "NY"),
                    help="STRONG scaling (BASELINE configs[3]: `oneD_benchmark nx*ny = 10^7 over 8 GPUs`; benchmarks/run_benchmarks.py:27-31 fixes "
                         "the global size and varies the ranks): the size of the WHOLE domain, split along x over --gpus ranks (num_proc = "
                         "(N, 1)); refused if the ranks do not divide nx (roger/distributed.py:121-138).  Replaces --size (per rank, weak scaling)")
    ap.add_argument("--params", choices=("uniform", "hetero"), default="uniform")
    ap.add_argument("--model", choices=("svat", "oned", "sas"), default="svat",
                    help="svat: SVAT_benchmark (BASELINE configs[1]); oned: oneD_benchmark (lateral subsurface flow); "
                         "sas: SVATOXYGEN18_benchmark (configs[2]: offline oxygen-18 transport, one step = one day)")
    ap.add_argument("--ages", type=int, default=1000, help="sas: age classes (benchmark: 1000)")
    ap.add_argument("--substeps", type=int, default=6, help="sas: sas_solver_substeps (benchmark: 6)")
    ap.add_argument("--routing", action="store_true",
                    help="oned: settings.enable_routing_1D -- surface and subsurface runoff routed to the D8 neighbour (the routed step)")
    ap.add_argument("--sas-solver", choices=("deterministic", "Euler", "RK4"), default="deterministic",
                    help="sas: settings.sas_solver (benchmark: deterministic; Euler / RK4 = the explicit schemes, transport.py:2064-2414, 1139-2047)")
    ap.add_argument("--station-weights", action="store_true",
                    help="svat: per-cell prec_weight / ta_offset / pet_weight on the station series (the distributed catchment "
                         "setups, BASELINE configs[4]: --size 80 53 --params hetero --station-weights)")
    ap.add_argument("--placement-probes", type=int, default=8,
                    help="svat / oned: candidate arenas rh_create times a streaming copy on before keeping the fastest (DESIGN.md section 5; "
                         "the library's default too; 1 = take the first)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="keep the device busy with a plain device-to-device copy for this long right before the timed region (untimed, not "
                         "model steps): a 20-step run is over in 5 ms, before the device has left its idle clocks -- the same 20 steps take "
                         "7 %% longer without it (profiles/r03_warm_ab.txt); 0 switches it off")
    ap.add_argument("--spinup", type=int, default=107,
                    help="svat / oned, device stepping: untimed steps in FRONT of the warm-up that bring the model to the end of the forcing's "
                         "first heavy-rain event (step 112 with the default --warmup 5), so that the timed steps of even a 20-step run "
                         "cover all three step classes -- 5 ten-minute, 14 hourly and 1 daily step (SURVEY 8d: per dt-class; VERDICT r2 weak #8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="one GPU, --model svat at the default size: do not add the `extras` object (SVAT at 3200 x 3125 and the SAS step at 10^5 "
                         "columns, each from a child run of this script after the headline measurement); for profiling runs")
    ap.add_argument("--cpu-cells", type=int, default=1000000,
                    help="columns of the CPU baseline sample (svat / oned; sas uses 1/25 of it)")
    ap.add_argument("--stepping", choices=("device", "setup", "script", "hooks", "routines"), default="device",
                    help="device: rh_run_steps driven by this script (the headline line); setup / script / hooks / routines: the benchmark as a RogerSetup "
                         "script calling plain run() with the model class's stock hooks / with set_forcing, set_parameters and after_timestep of its OWN "
                         "as every script of the reference has them (recognised by behaviour: on the device) / with a per-step hook in front of the "
                         "physics that does something of its own (hooks on the host, one native call per step) / with the three-call step that a "
                         "set_parameters or after_timestep hook doing something of its own needs (one GPU)")
    ap.add_argument("--days", type=int, default=20, help="--stepping setup | hooks: days of the timed run()")
    ap.add_argument("--warmup-days", type=int, default=2, help="--stepping setup | hooks: days of the untimed run() in front")
    ap.add_argument("--launch-check", action="store_true",
                    help="the launcher's own test: start the ranks, form the process group over gloo, all-reduce the rank ids and print "
                         "one line -- no GPU work, no model (tests/test_bench_launcher.py)")
    args = ap.parse_args()

    scaling, global_cells = "weak", None
    if args.global_size is not None:
        # the reference's decomposition rule, before anything is started (roger/distributed.py:121-138)
        from roger_amd.distributed import get_chunk_size, validate_decomposition

        gnx, gny = args.global_size
        try:
            validate_decomposition(gnx, gny, (args.gpus, 1), args.gpus)
        except (ValueError, RuntimeError) as e:
            raise SystemExit(f"bench: --global-size {gnx} {gny} over {args.gpus} ranks: {e}")
        args.size = get_chunk_size(gnx, gny, (args.gpus, 1))
        scaling, global_cells = "strong", gnx * gny
    args.scaling, args.global_cells = scaling, global_cells

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the bare command `python bench.py --gpus N`: this process becomes the launcher (nothing here has touched torch or the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the ranks disagree")
    if args.launch_check:
        return launch_check(dist, torch, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hip backend has no CPU fallback")
    # rehearsal of the multi-rank path on a one-GPU box: RH_BENCH_SINGLE_DEVICE=1 puts every rank on device 0 and
    # exchanges over gloo (RCCL refuses two ranks on one device); never used by the driver
    rehearsal = bool(os.environ.get("RH_BENCH_SINGLE_DEVICE"))
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    if args.stepping != "device":
        if world != 1 or args.model == "sas":
            raise SystemExit("--stepping setup | hooks: one GPU, --model svat | oned")
        return bench_setup(args, torch, device)

    if args.model == "sas":
        bench_sas(args, torch, dist, rank, local_rank, world, device)
        if world > 1:
            dist.destroy_process_group()
        return

    if args.routing:
        if args.model != "oned":
            raise SystemExit("--routing goes with --model oned (the routed subsurface runoff is the lateral flow)")
        bench_routed(args, torch, dist, rank, local_rank, world, device)
        if world > 1:
            dist.destroy_process_group()
        return

    from roger_amd.distributed import HipPhases, PhasedStepper
    from roger_amd.forcing import combo_forcing
    from roger_amd.svat import create_svat, hetero_params

    nx, ny = args.size
    n_local = nx * ny
    params = hetero_params(n_local, seed=42 + rank) if args.params == "hetero" else None
    if args.model == "oned":   # benchmarks/oneD_benchmark.py:99-135
        params = dict(params or {})
        for k, v in dict(z_soil=1000.0, lmpv=600.0, slope=0.05, slope_per=5, dmph=50.0).items():
            params.setdefault(k, v)
    ctx = create_svat(nx, ny, params=params, device=local_rank, lateral=(args.model == "oned"), placement_probes=args.placement_probes)
    total_steps = args.steps + args.warmup + args.spinup
    forcing = combo_forcing(ndays=max(30, total_steps + 5))   # a dry day is ONE step: the series must outlast one step per day
    ctx.set_forcing_series(forcing)
    if args.station_weights:   # eberbaechle/svat_distributed/svat.py:169-186, 276-296 (synthetic maps, seed 7)
        import numpy as np

        rng = np.random.default_rng(7 + rank)
        ctx.set_forcing_weights(rng.uniform(0.8, 1.2, n_local), rng.uniform(-1.5, 1.5, n_local), rng.uniform(0.9, 1.1, n_local))
    if args.station_weights and world > 1:
        raise SystemExit("--station-weights: the three-phase exchange of per-cell forcing is not wired into bench.py (single GPU only)")
    stepping, comm_ranks = "rh_run_steps", None
    if (world > 1 and not rehearsal) or os.environ.get("RH_BENCH_FORCE_DIST"):
        # multi-GPU: per step ncclAllReduce (64 x int32) -> control kernel -> fused kernel, enqueued from C (rh_run_steps_dist);
        # RH_BENCH_FORCE_DIST=1 rehearses it on one GPU with a one-rank communicator
        ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
        # no fallback: a rank whose C-side communicator cannot be formed ends the job with a non-zero status (the launcher stops the
        # others); stepping through torch.distributed instead costs 20 x more per step outside the kernel and would still print a line
        ctx.comm_init_torch()
        comm_ranks, comm_rank = ctx.comm_info()   # ncclCommCount / ncclCommUserRank of the communicator the steps will use
        if comm_ranks != world or comm_rank != rank:
            raise SystemExit(f"bench: rank {rank}: the RCCL communicator has {comm_ranks} ranks (this is rank {comm_rank} of it), the job has {world}")
        run = ctx.run_steps_dist
        stepping = "rh_run_steps_dist (RCCL from C)"
    elif world > 1 or os.environ.get("RH_BENCH_FORCE_PHASED"):   # rehearsals: the Python orchestration (gloo between CPU-side ranks)
        run = PhasedStepper(HipPhases(ctx, device), always_exchange=True).run   # one summary all-reduce per step
        stepping = "PhasedStepper (torch.distributed)"
    else:
        ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
        run = ctx.run_steps                               # single GPU: no exchange, fewer launches

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    clocks0 = device_clocks() if rank == 0 else None   # (a child process that takes half a second: BEFORE the warm-up, not between it and the timed region)
    run(args.spinup + args.warmup)
    s0 = ctx.get_scalars()
    prewarm(torch, device, args.prewarm_ms)
    ctx.enable_timing(True)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.timing_summary()
    per_ms, per_dt = ctx.timing_detail()
    sparse_steps = ctx.sparse_steps()   # of the timed call
    ctx.enable_timing(False)
    s1 = ctx.get_scalars()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if s1.sanity_ok != 1 or s1.itt - s0.itt != args.steps:
        raise SystemExit(f"bench: step bookkeeping failed (sanity_ok={s1.sanity_ok}, itt {s0.itt}->{s1.itt})")

    if rank == 0:
        value = world * n_local * args.steps / elapsed
        lazy, tail = ctx.step_mode()
        # Inside one rh_run_steps call every step but the last leaves out the stores of the planes the step only produces (sparse
        # stores, include/roger_hip.h); the last step stores everything.  The roofline prices the DOMINANT launch -- the sparse
        # variant, averaged over the launches that ran it -- with the bytes THAT variant moves; the full-store launch is reported beside it.
        n_sparse = sparse_steps if sparse_steps == launches - 1 and launches == len(per_ms) else 0
        kind = "sparse" if n_sparse else ("lazy" if lazy else "eager")
        k_avg_s = (float(per_ms[:n_sparse].mean()) / 1e3) if n_sparse else kernel_ms / 1e3 / max(launches, 1)
        pstats = ctx.param_stats() if kind != "eager" else None
        ld_b, st_b = kernel_bytes_per_cell(args.model, kind, pstats)
        algo = (ld_b + st_b) * n_local
        achieved = algo / k_avg_s / 1e9
        variant = f"k_step_{'oned' if args.model == 'oned' else 'svat'}_{kind}"
        full = None
        if n_sparse:
            fl, fs = kernel_bytes_per_cell(args.model, "lazy", pstats)
            full = {"kernel": variant.replace("sparse", "lazy"), "launches": launches - n_sparse, "avg_kernel_ms": float(per_ms[n_sparse:].mean()),
                    "algorithmic_bytes_per_cell": {"load": fl, "store": fs},
                    "frac": (fl + fs) * n_local / (float(per_ms[n_sparse:].mean()) / 1e3) / 1e9 / HBM_PEAK_GBS}
        traffic, trec = measured_traffic(variant + ("_hetero" if args.params == "hetero" else ""), n_local)
        ref_achieved = REFERENCE_BYTES_PER_CELL_STEP * n_local / k_avg_s / 1e9
        out = {
            "metric": "cell-timesteps/sec on SVAT_benchmark grid",
            "value": value,
            "unit": "cell-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{'oneD' if args.model == 'oned' else 'SVAT'}_benchmark synthetic grid " + (f"nx*ny={args.global_cells} in all ({args.global_size[0]}x{args.global_size[1]}), " if args.global_cells else "")
                            + f"nx*ny={n_local} per GPU ({nx}x{ny}), {args.params} "
                            "benchmark parameters, combo forcing (seed 42), adaptive dt" + (", per-cell station weights" if args.station_weights else ""),
                "cells_per_gpu": n_local,
                "global_cells": args.global_cells if args.global_cells else world * n_local,
                "simulated_seconds": int(s1.time - s0.time),
                "spinup_steps": args.spinup,
                "decomposition": f"({world},1) along x, " + ("one 256-byte predicate all-reduce per step" if world > 1 else "single GPU: no exchange"),
                "stepping": stepping,
                "n_ranks_in_comm": comm_ranks,   # ncclCommCount of the communicator the timed steps used (None: no communicator, single GPU)
                "placement_probe_ms": [round(v, 4) for v in ctx.placement_report()],   # the candidate arenas' copy times, the chosen one first
                # per time-step class (kernel time only, rank 0); `value` is the aggregate over the run
                "dt_classes": {
                    name: {"steps": int((per_dt == secs).sum()),
                           "avg_kernel_ms": float(per_ms[per_dt == secs].mean()),
                           "cell_timesteps_per_s_kernel": float(n_local / (per_ms[per_dt == secs].mean() / 1e3))}
                    for name, secs in (("10min", 600), ("1h", 3600), ("24h", 86400)) if (per_dt == secs).any()
                },
            },
            "roofline": {
                "bound": "hbm",
                "kernel": variant + (", control part of the next step in its tail" if tail or n_sparse else ""),
                "note": "achieved = bytes the launch must move (per column the planes this kernel variant loads + stores once: "
                        f"{ld_b} + {st_b} B, roger_amd/csrc/rh_step_bytes.json from the ISA) / average kernel duration by HIP events; "
                        "traffic = HBM bytes per launch by PMC (profiles/traffic.json) if this variant was profiled at this size; "
                        "reference_equivalent = the same time against the 2779 B per cell-step of the reference's read + write sets "
                        "(SURVEY 8d; for oneD the reference moves more) -- not a fraction of the peak, the kernel moves fewer bytes",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_achieved": (traffic / k_avg_s / 1e9) if traffic else None,
                "traffic_frac": (traffic / k_avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "traffic_source": trec["source"] if trec else None,
                "algorithmic_bytes_per_launch": algo,
                "algorithmic_bytes_per_cell": {"load": ld_b, "store": st_b},
                # how the kernel read the parameter planes (rh_param_stats): the 15 parameters calc_parameters_soil derives from the
                # primaries are evaluated in the kernel where the planes hold exactly those values (120 B per column not loaded), and
                # a parameter plane with ONE value over a wave's 64 columns is read as one element (priced at 1/8 of its bytes)
                "parameters": None if pstats is None else {"derived_fraction_of_waves": pstats[0], "bytes_per_cell_read_as_one_element_per_wave": pstats[1],
                                                           "uniform_read_priced_at": UNIFORM_READ_FRACTION},
                "reference_equivalent": {"bytes_per_cell_step": REFERENCE_BYTES_PER_CELL_STEP, "achieved": ref_achieved,
                                         "ratio_to_peak": ref_achieved / HBM_PEAK_GBS},
                "avg_kernel_ms": k_avg_s * 1e3,
                "launches_timed": n_sparse if n_sparse else launches,
                "full_store_launch": full,   # the call's last step (stores every plane), or null when no step ran with sparse stores
                "outside_kernel_us_per_step": (elapsed - kernel_ms / 1e3) / args.steps * 1e6,
            },
            "clocks_mhz": {"before": clocks0, "after": device_clocks()},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported at N = 1 only
            cpu_steps = min(args.steps + args.warmup, 60)
            v, secs, threads = cpu_baseline(args.cpu_cells, cpu_steps, forcing)
            out["cpu_baseline"] = {
                "value": v,
                "unit": "cell-timesteps/s",
                "cores": threads,
                "kind": "port",
                "sample": f"oracle/svat_oracle.c (OpenMP over the columns), {args.cpu_cells} cells x first {cpu_steps} steps of "
                          f"the same forcing, {secs:.1f} s on {threads} host threads",
            }
        plain = (world == 1 and args.model == "svat" and tuple(args.size) == (1000, 1000) and args.params == "uniform" and not args.station_weights
                 and not os.environ.get("RH_BENCH_FORCE_DIST") and not os.environ.get("RH_BENCH_FORCE_PHASED"))
        if plain and not args.no_extras and not args.no_cpu_baseline:   # (--no-cpu-baseline marks the profiling / A-B runs of tools/*.sh)
            ctx.close()   # (the headline's arena is not needed any more)
            out["extras"] = bench_extras(torch, device)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
