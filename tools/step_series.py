#!/usr/bin/env python3
"""The fused kernel's duration step by step over the first steps of the bench workload (10^6 columns, combo forcing): is the first minute
of a run slower because of the device (ramp) or because of the state (what the columns do in those steps)?
    python tools/step_series.py [steps=120] [prewarm_ms=0]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("RH_PLACEMENT_PROBES", "1")   # the raw level of each allocation: no placement probing (the library's default is 8 candidates)
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
prewarm = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ctx = create_svat(1000, 1000)
ctx.set_forcing_series(combo_forcing(ndays=steps + 40))
if prewarm > 0:
    a = torch.empty(1 << 25, dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < prewarm:
        for _ in range(8):
            b.copy_(a)
        torch.cuda.synchronize()
ctx.enable_timing(True)
ctx.run_steps(steps)
ctx.sync()
per_ms, per_dt = ctx.timing_detail()
per_ms, per_dt = np.asarray(per_ms), np.asarray(per_dt)
for k in range(0, steps, 10):
    print(f"steps {k:3d}..{k + 9:3d}: kernel {per_ms[k:k + 10].mean():.4f} ms  (dt classes {sorted(set(int(v) for v in per_dt[k:k + 10]))})", flush=True)
# the same steps again from the same start state, after the run above has kept the device busy
ctx2 = create_svat(1000, 1000)
ctx2.set_forcing_series(combo_forcing(ndays=steps + 40))
ctx2.enable_timing(True)
ctx2.run_steps(steps)
ctx2.sync()
p2 = np.asarray(ctx2.timing_detail()[0])
print("second context, same steps, device already busy:")
for k in range(0, steps, 10):
    print(f"steps {k:3d}..{k + 9:3d}: kernel {p2[k:k + 10].mean():.4f} ms", flush=True)
# ... and alternately: does the level belong to the context (its arena) or to the moment?
for rnd in range(3):
    for name, c in (("first", ctx), ("second", ctx2)):
        c.enable_timing(True)
        c.run_steps(60)
        c.sync()
        p = np.asarray(c.timing_detail()[0])
        print(f"round {rnd} {name:6s} context: kernel {p.mean():.4f} ms over {p.size} steps", flush=True)
ctx3 = create_svat(1000, 1000)
ctx3.set_forcing_series(combo_forcing(ndays=steps + 40))
ctx3.enable_timing(True)
ctx3.run_steps(120)
ctx3.sync()
p3 = np.asarray(ctx3.timing_detail()[0])
print(f"third context: first 10 steps {p3[:10].mean():.4f}, steps 60..119 {p3[60:].mean():.4f} ms")
for name, c in (("first", ctx), ("second", ctx2), ("third", ctx3)):
    print(name, "arena placement report:", c.placement_report())
