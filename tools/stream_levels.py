#!/usr/bin/env python3
"""Does the fused kernel's speed level (tools/arena_levels.py) follow the ARENA or the STREAM (HIP maps streams onto a few hardware queues)?
K contexts alive in one process; (A) each on the stream rh_create made for it, (B) all on ONE stream, (C) the first and the last context
moved over K streams created here.     python tools/stream_levels.py [contexts=6]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("RH_PLACEMENT_PROBES", "1")
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402

hip = C.CDLL("libamdhip64.so")
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
print("A  own streams        : " + " ".join(f"{level(c):.4f}" for c in ctxs), flush=True)
print("A  own streams (again): " + " ".join(f"{level(c):.4f}" for c in ctxs), flush=True)
streams = []
for _ in range(k):
    s = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(s)) == 0
    streams.append(s)
for j, s in enumerate(streams):
    for c in ctxs:
        c.set_stream(s.value)
    print(f"B  all on new stream {j}: " + " ".join(f"{level(c):.4f}" for c in ctxs), flush=True)
for c in ctxs:
    c.set_stream(None)
print("B  all on the null stream: " + " ".join(f"{level(c):.4f}" for c in ctxs), flush=True)
for which in (0, k - 1):
    row = []
    for s in streams:
        ctxs[which].set_stream(s.value)
        row.append(level(ctxs[which]))
    print(f"C  context {which} over the {k} streams: " + " ".join(f"{v:.4f}" for v in row), flush=True)
