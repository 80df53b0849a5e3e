#!/usr/bin/env python3
"""Does the fused kernel's level follow the arena's address, or the size of its allocation?  Contexts of one process whose arenas are shifted
inside their allocation (RH_ARENA_OFFSET_KB) or whose allocation is padded behind the arena (RH_ARENA_PAD_KB; both read by rh_create),
stepped in turn; placement probing off.
    python tools/arena_offsets.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402

# (offset, pad) in MiB: the arena starts `offset` into its allocation, which is `pad` longer than the arena needs
offs = [(0, 0), (896, 0), (0, 896), (0, 0), (0, 896), (896, 0), (0, 1792), (0, 0), (0, 448), (448, 448), (0, 896), (0, 0)]
os.environ["RH_PLACEMENT_PROBES"] = "1"
F = combo_forcing(ndays=400)
ctxs = []
for o, pad in offs:
    os.environ["RH_ARENA_OFFSET_KB"] = str(o * 1024)
    os.environ["RH_ARENA_PAD_KB"] = str(pad * 1024)
    c = create_svat(1000, 1000)
    c.set_forcing_series(F)
    c.run_steps(60)
    c.sync()
    ctxs.append(c)
for rnd in range(2):
    row = []
    for c in ctxs:
        c.enable_timing(True)
        c.run_steps(60)
        c.sync()
        row.append(float(np.mean(c.timing_detail()[0])))
print("offset MiB | pad MiB | arena address mod 1 GiB (MiB) | kernel ms (second round)")
for (o, pad), c, v in zip(offs, ctxs, row):
    p = c._lib.rh_plane_device_ptr(c._h, 0)
    print(f"{o:6d} | {pad:6d} | {(p % 2**30) / 2**20:8.1f} | {v:.4f}   0x{p:x}")
