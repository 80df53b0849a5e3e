#!/usr/bin/env python3
"""Workload for tools/arena_level_counters.sh: RH_LEVEL_CONTEXTS (4) contexts alive in one process; 30 fused steps on the first, then 30 on the last (the
arenas sit at different speed levels, tools/arena_levels.py); the dispatches of the two are told apart by their order."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("RH_PLACEMENT_PROBES", "1")   # the raw level of each allocation: no placement probing (the library's default is 8 candidates)
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat        # noqa: E402

F = combo_forcing(ndays=200)
ctxs = []
for _ in range(int(os.environ.get("RH_LEVEL_CONTEXTS", "4"))):
    c = create_svat(1000, 1000)
    c.set_forcing_series(F)
    c.run_steps(3)      # (eager first step, then lazy: 3 launches of warm-up per context, 12 in all before the measured ones)
    c.sync()
    ctxs.append(c)
for c in (ctxs[0], ctxs[-1]):
    c.run_steps(30)
    c.sync()
