#!/usr/bin/env python3
"""The fused kernel's speed level per arena: K contexts alive in one process, each stepped in turn (three rounds), with the device address of
its arena -- does the level follow the address?
    python tools/arena_levels.py [contexts=8]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("RH_PLACEMENT_PROBES", "1")   # the raw level of each allocation: no placement probing (the library's default is 8 candidates)
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = combo_forcing(ndays=400)
ctxs = []
for _ in range(k):
    c = create_svat(1000, 1000)
    c.set_forcing_series(F)
    c.run_steps(60)
    c.sync()
    ctxs.append(c)
for rnd in range(3):
    row = []
    for c in ctxs:
        c.enable_timing(True)
        c.run_steps(60)
        c.sync()
        row.append(float(np.mean(c.timing_detail()[0])))
    print(f"round {rnd}: " + " ".join(f"{v:.4f}" for v in row), flush=True)
for j, c in enumerate(ctxs):
    p = c._lib.rh_plane_device_ptr(c._h, 0)
    print(f"context {j}: arena at 0x{p:014x}  (offset from the first {(p - ctxs[0]._lib.rh_plane_device_ptr(ctxs[0]._h, 0)) / 2**20:10.1f} MiB, mod 1 GiB {(p % 2**30) / 2**20:7.1f} MiB, mod 2 MiB {p % 2**21})")
