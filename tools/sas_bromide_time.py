"""The SAS kernels with the anion kernels (bromide) against the isotope kernels at the benchmark's shape: 10^5 columns x 1000 ages,
6 sub-steps, age statistics; for the deterministic solver and the explicit ones.  Diagnostics only.
    python tools/sas_bromide_time.py [deterministic Euler RK4]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from roger_amd import _native, sas as rsas  # noqa: E402

n, ages, sub = 100000, 1000, 6
daily = rsas.synthetic_daily_inputs(n, 8, seed=42)
solvers = sys.argv[1:] or ["deterministic"]
for solver, tracer in [(s, t) for s in solvers for t in ("oxygen18", "bromide")]:
    ctx = _native.SasContext(n, ages, sub, device=0, forcing_days=8, age_statistics=True, tracer=tracer, solver=solver)
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
    ndays = 8 if solver == "deterministic" else (4 if solver == "Euler" else 2)
    ctx.run_days(0, 3)
    ctx.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.run_days(3, ndays)
    ctx.sync()
    print(f"{solver} {tracer}: {(time.perf_counter() - t0) / ndays * 1e3:.2f} ms per day", flush=True)
    ctx.close()
