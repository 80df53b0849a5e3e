"""k_sas with the anion kernels (bromide) against the isotope kernels at the benchmark's shape: 10^5 columns x 1000 ages,
6 sub-steps, age statistics.  Diagnostics only."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from roger_amd import _native, sas as rsas  # noqa: E402

n, ages, sub = 100000, 1000, 6
daily = rsas.synthetic_daily_inputs(n, 8, seed=42)
for tracer in ("oxygen18", "bromide"):
    ctx = _native.SasContext(n, ages, sub, device=0, forcing_days=8, age_statistics=True, tracer=tracer)
    for f, p in rsas.benchmark_sas_params(n).items():
        ctx.upload(f"sas_params_{f}", p)
    for key, S in (("rz", 90.0), ("ss", 260.0)):
        sa, msa = rsas.initial_age_state([S] * n, ages)
        ctx.upload(f"sa_{key}", sa)
        ctx.upload(f"msa_{key}", msa if tracer == "oxygen18" else sa * 3.0)     # 3 mg/l in every age class
    for k, a in daily.items():
        ctx.upload(k, a if tracer == "oxygen18" or k != "C_in" else np.full_like(a, 2.0))
    if tracer == "bromide":
        ctx.upload("S_sat_rz", np.full(n, 400.0))
    ctx.run_days(0, 3)
    ctx.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.run_days(3, 8)
    ctx.sync()
    print(f"{tracer}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms per day", flush=True)
    ctx.close()
