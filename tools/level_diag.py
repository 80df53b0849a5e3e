#!/usr/bin/env python3
"""Does the fused kernel's speed level belong to the arena (placement) or to the moment (clocks / power state)?
One process, K contexts created up front (each its own arena), then rounds over all of them for a while: per round and context the
average k_step duration, with the clocks rocm-smi reports next to it.

    python tools/level_diag.py [contexts=3] [seconds=40] [nx=1000] [ny=1000]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("RH_PLACEMENT_PROBES", "1")   # the raw level of each allocation: no placement probing (the library's default is 8 candidates)
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
        keep = [ln.split(":", 1)[1].strip() if ":" in ln else ln for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "fclk", "Power", "junction", "memory)"))]
        return " | ".join(keep)[:300]
    except Exception as e:   # noqa: BLE001
        return f"(rocm-smi: {e})"


if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 40
    nx, ny = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1000, 1000)
    F = combo_forcing(ndays=400)
    ctxs = []
    for _ in range(k):
        c = create_svat(nx, ny)
        c.set_forcing_series(F)
        c.run_steps(5)
        c.sync()
        ctxs.append(c)
    t0 = time.time()
    rnd = 0
    while time.time() - t0 < secs:
        rnd += 1
        row = []
        for c in ctxs:
            c.enable_timing(True)
            c.run_steps(40)
            ms, n = c.timing_summary()
            c.enable_timing(False)
            row.append(ms / n)
        print(f"t={time.time() - t0:6.1f}s round {rnd:3d}: " + "  ".join(f"{x:.4f}" for x in row) + (f"   {clocks()}" if rnd % 5 == 1 else ""), flush=True)
    for c in ctxs:
        c.close()
