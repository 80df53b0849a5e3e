#!/usr/bin/env python3
"""Small fixed SAS workload for rocprofv3 runs: n columns x 1000 ages, a few days.
usage: python3 tools/sas_workload.py [n_cells] [days] [substeps] [age_statistics 0|1] [solver deterministic|Euler]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from roger_amd import _native as N  # noqa: E402
from roger_amd import sas as rsas  # noqa: E402

if os.environ.get("RH_PMC_CALIB"):   # calibration copy of known size first (tools/pmc_summarise.py)
    scratch = N.Context(1000, 1000)
    first = next(i for i, (_, is_int) in enumerate(scratch.planes) if not is_int)
    for _ in range(5):
        scratch.calibrate_copy(first, first + 100, 96)
    scratch.sync()
    scratch.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
days = int(sys.argv[2]) if len(sys.argv) > 2 else 4
substeps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
stats = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
solver = sys.argv[5] if len(sys.argv) > 5 else "deterministic"
daily = rsas.synthetic_daily_inputs(n, 8, seed=42)
ctx = rsas.create_sas(n, 1000, substeps, 90.0, 260.0, daily=daily, age_statistics=stats, solver=solver)
ctx.run_days(0, days)
ctx.sync()
print("d18O q_ss", ctx.download("C_iso_q_ss")[:3])
ctx.close()
