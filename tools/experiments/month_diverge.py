#!/usr/bin/env python3
"""Which planes leave the oracle at the month change of tests/test_hip_parity.py::test_month_change_inside_device_driven_steps (diagnosis only)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import hip_util as H                # noqa: E402
import oracle_binding as ob         # noqa: E402
from golden_util import configure_settings, is_lateral, load_case   # noqa: E402
from roger_amd import _native as N  # noqa: E402
from roger_amd.forcing import combo_forcing   # noqa: E402

N.load()
case = sys.argv[1] if len(sys.argv) > 1 else "oned_hetero_combo"
g, names, _ = load_case(case)
F = {k: np.array(v) for k, v in combo_forcing(ndays=8).items()}
F["MONTH"] = np.where(np.arange(F["MONTH"].size) < 2 * 144, 4, 5).astype(F["MONTH"].dtype)
F["DOY"] = (119 + np.arange(F["DOY"].size) // 144).astype(F["DOY"].dtype)
rng = np.random.default_rng(4)
src = np.asarray(g["state0"])
nx, ny = 24, 20
snap = src[:, rng.integers(0, src.shape[1], nx * ny)].copy()
st = ob.OracleState(nx * ny)
st.load_snapshot(snap, names)
st.load_scalars(g["scal0"])
st.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
configure_settings(st.settings, g)
st.settings.nx, st.settings.ny = nx, ny
st.scal.month[0] = st.scal.month[1] = 4
st.scal.doy[0] = st.scal.doy[1] = 119
ctx = N.Context(nx, ny, enable_lateral_flow=int(is_lateral(g)))
H.upload_snapshot(ctx, snap, names)
ctx.set_scalars(H.scalars_from_row(st.scalars_row()))
ctx.set_luts(g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
if is_lateral(g):
    ctx.set_lut_mlms(g["lut_mlms"])
ctx.set_forcing_series(F)
odrv = ob.ForcingDriver(F)
for step in range(1, 8):
    prev = st.snapshot(names)
    pd, td, ed, monthly = odrv.before_step(st)
    st.step(pd, td, ed, monthly)
    ctx.run_steps(1)
    got, ref = H.download_snapshot(ctx, names), st.snapshot(names)
    with np.errstate(all="ignore"):
        ok = (got == ref) | (np.isnan(got) & np.isnan(ref)) | (np.abs(got - ref) <= 1e-10 + 1e-10 * np.abs(ref))
    bad = np.argwhere(~ok)
    print(f"step {step} monthly {monthly} dt {st.scal.dt_secs}: {len(bad)} values off in {len(np.unique(bad[:, 1])) if len(bad) else 0} cells")
    if len(bad):
        planes = sorted({names[p] for p in bad[:, 0]})
        print("   planes:", planes[:40])
        i = bad[0][1]
        print(f"   cell {i}: lu_id {ref[names.index('lu_id'), i]}")
        for p in sorted(set(bad[bad[:, 1] == i][:, 0]))[:30]:
            print(f"      {names[p]:20s} got {got[p, i]!r:26} ref {ref[p, i]!r:26} before {prev[p, i]!r}")
        lus = sorted({int(ref[names.index('lu_id'), c]) for c in np.unique(bad[:, 1])})
        print("   land uses of the deviating cells:", lus)
        break
