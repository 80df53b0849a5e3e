"""How much of k_sas is the power function?  10^5 columns x 1000 ages, every flux with the same exponent: a general one
(exp2(k log2 .)), 0.5 (square-root path) and 1 (no function at all).  Diagnostics only."""
import sys
import time

import torch

sys.path.insert(0, ".")
from roger_amd import sas as rsas  # noqa: E402

n = 100000
daily = rsas.synthetic_daily_inputs(n, 8, seed=42)
for k in (0.3, 0.5, 1.0):
    params = rsas.benchmark_sas_params(n)
    for f in params:
        params[f][:, 1] = k
    ctx = rsas.create_sas(n, 1000, 6, 90.0, 260.0, sas_params=params, daily=daily, device=0, age_statistics=True)
    ctx.run_days(0, 3)
    ctx.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.run_days(3, 8)
    ctx.sync()
    print(f"k = {k}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms per day", flush=True)
    ctx.close()
