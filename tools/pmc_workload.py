#!/usr/bin/env python3
"""Workload for the PMC passes: a calibration copy of known size, then fused SVAT steps.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_workload.py

tools/pmc_summarise.py turns the two counter CSVs into profiles/traffic.json.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from roger_amd import _native as N          # noqa: E402
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402

NX, NY = (int(v) for v in os.environ.get("RH_PMC_SIZE", "1000x1000").split("x"))
MODEL = os.environ.get("RH_PMC_MODEL", "svat")   # svat | oned (lateral subsurface flow, benchmarks/oneD_benchmark.py:99-135) | routed (oned + enable_routing_1D)
PARAMS = os.environ.get("RH_PMC_PARAMS", "uniform")   # uniform: the benchmark's (every wave reads its parameter planes as one element); hetero: drawn per column
CALIB_PLANES = 96

if __name__ == "__main__":
    scratch = N.Context(NX, NY)
    first = next(i for i, (_, is_int) in enumerate(scratch.planes) if not is_int)
    for _ in range(5):
        scratch.calibrate_copy(first, first + 100, CALIB_PLANES)
    scratch.sync()
    scratch.close()
    params = dict(z_soil=1000.0, lmpv=600.0, slope=0.05, slope_per=5, dmph=50.0) if MODEL in ("oned", "routed") else None
    if PARAMS == "hetero":
        from roger_amd.svat import hetero_params

        params = dict(hetero_params(NX * NY, seed=42), **(params or {}))
    if MODEL == "routed":   # the workload of bench.py --model oned --routing
        params.update(flow_dir_topo=4, k_st=15.0)
        ctx = create_svat(NX, NY, params=params, lateral=True, enable_routing_1D=1, dx=5.0, dy=5.0)
    else:
        ctx = create_svat(NX, NY, params=params, lateral=(MODEL == "oned"))
    ctx.set_forcing_series(combo_forcing(ndays=30))
    ctx.run_steps(25)
    ctx.sync()
    ctx.close()
