#!/usr/bin/env python3
"""BASELINE configs[4] as written, as a timing: the Eberbaechle setup's shape (80 x 53 columns), heterogeneous synthetic parameter maps
(`parameters.nc` is not shipped; seed 42), per-cell prec_weight / ta_offset / pet_weight (seed 7) on the station's FULL 2019-2022 series
(tests/golden/eberbaechle_full_input), stepped on the device with rh_run_steps.  Prints one JSON line (profiles/).
    python tools/eberbaechle_full.py [nx=80 ny=53]"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from roger_amd.forcing import forcing_from_txt   # noqa: E402
from roger_amd.svat import create_svat, hetero_params  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (80, 53)
n = nx * ny
F = forcing_from_txt(os.path.join(REPO, "tests", "golden", "eberbaechle_full_input"))
ndays = len(F["PREC"]) // 144
ctx = create_svat(nx, ny, params=hetero_params(n, seed=42))
ctx.set_forcing_series(F)
rng = np.random.default_rng(7)
ctx.set_forcing_weights(rng.uniform(0.8, 1.3, n), rng.uniform(-5.5, 1.5, n), rng.uniform(0.85, 1.15, n))
# (per-cell forcing: rh_set_time_limit is the summary path's; the rounds are bounded from the host: a step covers at most a day)
ctx.enable_timing(True)
t0 = time.perf_counter()
steps = 0
while True:
    s = ctx.get_scalars()
    remaining = ndays * 86400 - s.time
    if remaining <= 0:
        break
    k = -(-remaining // 86400)
    ctx.run_steps(int(k))
    steps += int(k)
ctx.sync()
wall = time.perf_counter() - t0
s = ctx.get_scalars()
ms, dts = ctx.timing_detail()
print(json.dumps({
    "workload": f"catchment_scale Eberbaechle shape {nx}x{ny}, heterogeneous synthetic parameters, per-cell station weights, the station's full "
                f"series 2019-11-01 .. 2022-10-31 ({len(F['PREC'])} ten-minute records), rh_run_steps in rounds of ceil(remaining days) steps",
    "days": ndays, "steps": int(s.itt), "sanity_ok": int(s.sanity_ok), "wall_s": wall, "steps_per_s": s.itt / wall,
    "cell_timesteps_per_s": s.itt * n / wall, "ms_per_step": wall / s.itt * 1e3,
    "fused_kernel_us_avg": float(ms.mean() * 1e3),
    "step_classes": {str(int(k)): int((dts == k).sum()) for k in (600, 3600, 86400)},
}))
ctx.close()
