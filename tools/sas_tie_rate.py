"""Fraction of (day, column) pairs of every SAS golden case that the device reproduces at rtol 1e-10 when each day is
restarted from the reference's state (the rest are residue ties).  Run on the GPU box."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from sas_binding import SAS_CASES, SasGolden, column_deviation
from test_hip_sas import make_ctx, push, pull
for case in SAS_CASES:
    g = SasGolden(case); st = g.new_state(); ctx = make_ctx(st); tight = []
    for d in range(1, g.ndays + 1):
        g.load_state(st, d - 1); g.load_inputs(st, d); push(ctx, st); ctx.step(0); pull(ctx, st)
        tight.append(column_deviation(st, lambda k: g.day(d, k), g.n, bool(g.stats)))
    tight = np.array(tight); print(case, round(float(tight.mean()), 4), f"{np.count_nonzero(~tight)}/{tight.size}")
    ctx.close()
