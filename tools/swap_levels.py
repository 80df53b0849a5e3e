#!/usr/bin/env python3
"""Does the fused kernel's speed level belong to the arena or to the rest of the context?  K contexts alive in one process (the
staircase of tools/arena_levels.py); per context the level of the fused step; then the first
and the last context exchange their arenas (rh_debug_swap_arenas) and everything is measured again.
    python tools/swap_levels.py [contexts=6]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("RH_PLACEMENT_PROBES", "1")
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
F = combo_forcing(ndays=400)


def level(c, steps=60):
    c.enable_timing(True)
    c.run_steps(steps)
    c.sync()
    v = float(np.mean(c.timing_detail()[0]))
    c.enable_timing(False)
    return v


ctxs = []
for _ in range(k):
    c = create_svat(1000, 1000)
    c.set_forcing_series(F)
    c.run_steps(60)
    c.sync()
    ctxs.append(c)
addr = lambda c: c._lib.rh_plane_device_ptr(c._h, 0)   # noqa: E731
print("arena addresses      : " + " ".join(f"{addr(c):x}" for c in ctxs))
print("fused step, ms       : " + " ".join(f"{level(c):.4f}" for c in ctxs), flush=True)
print("fused step, ms       : " + " ".join(f"{level(c):.4f}" for c in ctxs), flush=True)
a, b = ctxs[0], ctxs[-1]
assert a._lib.rh_debug_swap_arenas(a._h, b._h) == 0
print("first <-> last arena swapped")
print("arena addresses      : " + " ".join(f"{addr(c):x}" for c in ctxs))
print("fused step, ms       : " + " ".join(f"{level(c):.4f}" for c in ctxs), flush=True)
print("fused step, ms       : " + " ".join(f"{level(c):.4f}" for c in ctxs), flush=True)
