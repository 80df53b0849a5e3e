#!/usr/bin/env python3
"""What moves with the fused kernel's speed level: rounds of fused steps in one process, `rocm-smi --showmetrics` (the driver's gpu_metrics
table: clocks per XCD, temperatures, power, the accumulated residency counters of the power / thermal limiters) read before and after
every round; prints per round the kernel's average duration and every numeric metric, as value or as increase over the round.

    python tools/level_metrics.py [rounds=10] [steps=400] [pause_s=0]"""
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("RH_PLACEMENT_PROBES", "1")   # the raw level of each allocation: no placement probing (the library's default is 8 candidates)
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402


def metrics():
    try:
        out = subprocess.run(["rocm-smi", "--showmetrics"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)}
    rec = {}
    for ln in out.splitlines():
        m = re.match(r"GPU\[0\]\s*:\s*([^:]+?)\s*:\s*(.*)$", ln)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        nums = re.findall(r"-?\d+(?:\.\d+)?", val)
        if nums and "N/A" not in val:
            rec[key] = [float(v) for v in nums]
    return rec


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    pause = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    ctx = create_svat(1000, 1000)
    ctx.set_forcing_series(combo_forcing(ndays=rounds * steps + 40))
    ctx.run_steps(5)
    ctx.sync()
    first = None
    for r in range(rounds):
        m0 = metrics()
        ctx.enable_timing(True)
        t0 = time.perf_counter()
        ctx.run_steps(steps)
        ctx.sync()
        wall = time.perf_counter() - t0
        ms, n = ctx.timing_summary()
        ctx.enable_timing(False)
        m1 = metrics()
        if first is None:
            first = m0
            print("metrics available:", len(m0), "keys", flush=True)
        moved = []
        for k in sorted(m1):
            if k in m0 and len(m0[k]) == len(m1[k]):
                d = [b - a for a, b in zip(m0[k], m1[k])]
                if any(abs(x) > 0 for x in d):
                    moved.append(f"{k}: {m1[k][:8]} (+{[round(x, 3) for x in d[:8]]})")
        print(f"round {r}: kernel {ms / max(n, 1):.4f} ms over {n} launches, wall {wall / steps * 1e3:.4f} ms per step", flush=True)
        for ln in moved:
            print("    " + ln, flush=True)
        if pause:
            time.sleep(pause)
    ctx.close()
