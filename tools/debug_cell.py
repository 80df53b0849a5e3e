"""Debug helper (GPU box): find where one column of the HIP run leaves the oracle."""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob
import hip_util as H
from golden_util import load_case
from roger_amd import _native as N
from roger_amd.forcing import combo_forcing
import test_hip_parity as T

g, names, _ = load_case("svat_hetero_combo")
luts = (g["lut_ilu"], g["lut_gc"], g["lut_gcm"], g["lut_rdlu"])
st = T._oracle_setup(ob, 64, 64, 123, luts)
onames = st.names
ctx = N.Context(64, 64)
H.upload_snapshot(ctx, st.snapshot(), onames)
ctx.set_scalars(H.scalars_from_row(st.scalars_row())); ctx.set_luts(*luts)
F = combo_forcing(ndays=12)
odrv = ob.ForcingDriver(F); hdrv = H.HipForcingDriver(ctx, F)
cell = int(sys.argv[1]) if len(sys.argv) > 1 else 1884
for step in range(1, 130):
    pd, td, ed, m = odrv.before_step(st); st.step(pd, td, ed, m)
    ctx.step(hdrv.before_step())
    got = H.download_snapshot(ctx, onames); ref = st.snapshot()
    with np.errstate(all="ignore"):
        rel = np.abs(got - ref) / (np.abs(ref) + 1e-30)
        rel = np.where((got == ref) | (np.abs(got - ref) < 1e-13), 0, rel)
    worst = rel.max(axis=0)
    c = rel[:, cell]
    if c.max() > 1e-12:
        idx = np.argsort(-c)[:10]
        print(f"step {step} dt {st.scal.dt}: cell {cell} max rel {c.max():.2e}; worst cells overall {np.argsort(-worst)[:3]} {np.sort(worst)[-3:]}")
        for i in idx:
            if c[i] > 1e-13:
                print(f"   {onames[i]:24s} got {got[i, cell]!r} ref {ref[i, cell]!r}")
        if c.max() > 1e-9:
            break
print("params of cell:", {k: st.planes[k][cell] for k in ("lu_id","z_soil","dmpv","lmpv","theta_ac","theta_ufc","theta_pwp","ks","z_root","theta_sat","lambda_bc","ha","n_salv","m_bc")})
