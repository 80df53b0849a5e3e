"""Diagnostic: random columns on the device vs the oracle, with the SAS code of the deviating flux."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from test_hip_sas import random_problem, clone, make_ctx, push, pull
from sas_binding import FLUXES

for (n, ages, sub) in [(300, 17, 4), (150, 100, 5)]:
    st = random_problem(n, ages, sub, seed=ages + n)
    ref = clone(st)
    ctx = make_ctx(st)
    push(ctx, st)
    ctx.step(0)
    pull(ctx, st)
    ref.step_oracle()
    for f in FLUXES:
        a, b = st.out["tt_" + f], ref.out["tt_" + f]
        dev = np.abs(a - b).max(axis=1)
        bad = np.where(dev > 1e-6)[0]
        for c in bad[:6]:
            k = np.abs(a[c] - b[c]).argmax()
            print(f"n={n} ages={ages} {f}: col {c} dev {dev[c]:.3e} at age {k} code {int(st.sas[f][c, 0])} p1 {st.sas[f][c, 1]:.3f} flux {st.inp[f][c]:.3f} "
                  f"gpu {a[c][max(0,k-2):k+3]} ref {b[c][max(0,k-2):k+3]}")
    ctx.close()
