"""Diagnostic: which random columns deviate between the HIP SAS step and the oracle, and by how much."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from test_hip_sas import random_problem, clone, make_ctx, push, pull
from sas_binding import FLUXES

for (n, ages, sub) in [(150, 100, 5), (300, 17, 4), (96, 1000, 6)]:
    st = random_problem(n, ages, sub, seed=ages + n)
    ref = clone(st)
    ctx = make_ctx(st)
    for day in range(3):
        if day == 0:
            push(ctx, st)
        ctx.step(0)
        pull(ctx, st)
        ref.step_oracle()
        print(f"n={n} ages={ages} sub={sub} day={day}")
        for k in ["tt_" + f for f in FLUXES] + ["sa_rz", "sa_ss", "C_rz", "C_ss"]:
            a = st.out[k] if k in st.out else st.state[k]
            b = ref.out[k] if k in ref.out else ref.state[k]
            ok = np.isclose(a, b, rtol=1e-10, atol=1e-12, equal_nan=True).reshape(n, -1).all(axis=1)
            dev = np.nanmax(np.abs(a - b).reshape(n, -1), axis=1)
            bad = np.where(~ok)[0]
            if len(bad):
                f = k[3:] if k.startswith("tt_") else None
                codes = st.sas[f][bad, 0] if f else None
                print(f"  {k}: {len(bad)} bad cols {bad[:10]} dev {dev[bad][:6]} codes {codes[:10] if codes is not None else ''} mask {st.maskCatch[bad][:10]}")
    ctx.close()
