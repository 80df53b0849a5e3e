#!/usr/bin/env python3
"""Does the placement level change when the SAME allocation is used at a shifted base?  Three contexts (three allocations, each padded by
8 MiB), the 96-plane calibration copy with the fused kernel's access shape timed at shifts of 0 ... 8 MiB inside each allocation.
    python tools/arena_phase.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["RH_PLACEMENT_PROBES"] = "1"
os.environ["RH_ARENA_PAD_KB"] = str(8 * 1024)
from roger_amd import _native as N  # noqa: E402

shifts = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
ctxs = [N.Context(1000, 1000) for _ in range(3)]
first = next(i for i, (_, is_int) in enumerate(ctxs[0].planes) if not is_int)
print("shift KiB: " + " ".join(f"{s:6d}" for s in shifts))
for j, c in enumerate(ctxs):
    row = []
    for sh in shifts:
        os.environ["RH_CALIB_SHIFT_KB"] = str(sh)
        for _ in range(3):
            c.calibrate_copy(first, first + 100, 96)
        c.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            c.calibrate_copy(first, first + 100, 96)
        c.sync()
        row.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"context {j}: " + " ".join(f"{v:6.4f}" for v in row), flush=True)
