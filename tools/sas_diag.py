"""Diagnostic: which (day, column) pairs of a SAS golden case deviate on the device, by how much, and with which
SAS codes.  usage: python tools/sas_diag.py <case>"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import sas_binding as sb
from test_hip_sas import make_ctx, pull, push

case = sys.argv[1] if len(sys.argv) > 1 else "sas_families_a50"
g = sb.SasGolden(case)
st = g.new_state()
ctx = make_ctx(st)
codes = np.stack([g.z[f"sas_{f}"][:, 0].astype(int) for f in sb.FLUXES], axis=1)
for d in range(1, g.ndays + 1):
    g.load_state(st, d - 1)
    g.load_inputs(st, d)
    push(ctx, st)
    ctx.step(0)
    pull(ctx, st)
    for f in sb.FLUXES:
        a, b = st.out[f"tt_{f}"], g.day(d, f"tt_{f}")
        dev = np.abs(a - b).max(axis=1)
        for c in np.where(dev > 1e-9)[0]:
            print(f"day {d} col {c} tt_{f}: dev {dev[c]:.3e} codes(evap,transp,q_rz,q_ss,cpr)={codes[c]} flux={st.inp[f][c]:.4f} "
                  f"p={g.z['sas_' + f][c][:5].round(3)}")
ctx.close()

if len(sys.argv) > 3:   # detail: python tools/sas_diag.py <case> <day> <col> <flux>
    d, c, f = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    st = g.new_state()
    ctx = make_ctx(st)
    g.load_state(st, d - 1)
    g.load_inputs(st, d)
    push(ctx, st)
    ctx.step(0)
    pull(ctx, st)
    np.set_printoptions(precision=4, linewidth=220)
    a, b = st.out[f"tt_{f}"][c], g.day(d, f"tt_{f}")[c]
    print("tt dev by age:", (a - b))
    print("tt ref:", b)
    src = "rz" if f in ("evap_soil", "transp", "q_rz") else "ss"
    print("sa before:", g.day(d - 1, f"sa_{src}")[c])
    print("params:", g.z["sas_" + f][c], "flux", st.inp[f][c])
    for ff in sb.FLUXES:
        a, b = st.out[f"tt_{ff}"][c], g.day(d, f"tt_{ff}")[c]
        print(f"{ff}: code {int(g.z['sas_' + ff][c][0])} flux {st.inp[ff][c]:.4f} max|dev| {np.abs(a - b).max():.3e} sum gpu {a.sum():.15f} ref {b.sum():.15f}")
    for k in ("sa_rz", "sa_ss"):
        a, b = st.state[k][c], g.day(d, k)[c]
        print(k, "max dev", np.abs(a - b).max(), "sum gpu", a.sum(), "ref", b.sum())
