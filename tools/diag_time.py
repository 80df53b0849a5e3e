#!/usr/bin/env python3
"""What device-side output accumulators cost per step at 10^6 columns: the benchmark's fused steps (rh_run_steps) with the daily sums of
typical output variables -- fluxes the sparse kernel would not store (q_ss, aet, transp, evap_soil, inf_mat_rz, inf_mp_rz) and state
(theta, S_fp_rz) -- against the same run with no accumulators; RH_NO_SPARSE_STORES=1 shows what the accumulators cost when they
switched the sparse stores off (until the end of round 3).   gpurun -- python3 tools/diag_time.py [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rate, collect = ["q_ss", "aet", "transp", "evap_soil", "inf_mat_rz", "inf_mp_rz"], ["theta", "S_fp_rz"]
    for label, with_diag in (("no accumulators", False), (f"{len(rate)} rate + {len(collect)} collect accumulators", True)):
        ctx = create_svat(1000, 1000)
        ctx.set_forcing_series(combo_forcing(ndays=60))
        if with_diag:
            ctx.diag_configure(rate=rate, collect=collect, n_slots=4)
        ctx.run_steps(120)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.run_steps(steps)
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps * 1e3
        print(f"{label}: {dt:.4f} ms per step ({1e6 / dt / 1e6:.2f} x 10^9 cell-timesteps/s), sparse steps {ctx.sparse_steps()} of {steps}"
              + (" [RH_NO_SPARSE_STORES]" if os.environ.get("RH_NO_SPARSE_STORES") else ""))
        ctx.close()
