#!/usr/bin/env python3
"""Where the device's routed step first leaves the oracle on the random 70 x 45 grid of tests/test_hip_routing.py (diagnosis only)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import hip_util as H                # noqa: E402
import oracle_binding as ob         # noqa: E402
from golden_util import ROUTING_CASES, load_case   # noqa: E402
from roger_amd import _native as N  # noqa: E402
import test_hip_routing as T        # noqa: E402

N.load()
g, names, forcing = load_case(ROUTING_CASES[0])
nx, ny = 70, 45
st, ctx = T._tiled_case(N, ob, g, names, nx, ny, np.random.default_rng(8))
odrv, hdrv = ob.ForcingDriver(forcing), H.HipForcingDriver(ctx, forcing)
resync = "--resync" in sys.argv
seen = 0
for step in range(1, 161):
    prev = st.snapshot(names)
    pd, td, ed, monthly = odrv.before_step(st)
    st.step(pd, td, ed, monthly)
    hdrv.before_step()
    ctx.step_routed(monthly)
    got, ref = H.download_snapshot(ctx, names), st.snapshot(names)
    with np.errstate(all="ignore"):
        ok = (got == ref) | (np.isnan(got) & np.isnan(ref)) | (np.abs(got - ref) <= 1e-10 + 1e-10 * np.abs(ref))
    if not ok.all():
        bad = np.argwhere(~ok)
        cells = np.unique(bad[:, 1])
        print(f"step {step} dt_secs {st.scal.dt_secs}: {len(bad)} values in {len(cells)} cells; planes {sorted({names[p] for p in bad[:, 0]})[:12]}")
        for p, i in bad[:6]:
            print(f"   {names[p]}[{i} = ({i // ny},{i % ny})] got {got[p, i]!r} ref {ref[p, i]!r}")
        if seen == 0:
            i = bad[0][1]
            print(f"   cell {i}: every deviating plane (got / ref), and its inputs")
            for p in sorted(set(bad[bad[:, 1] == i][:, 0])):
                if not names[p].endswith("_m1"):
                    print(f"      {names[p]:18s} {got[p, i]!r:26} {ref[p, i]!r}")
            for nm in ("lu_id", "z_soil", "theta_rz", "theta_fc", "theta_sat", "z0", "prec", "q_sur_in", "q_sub_in", "q_sub_in_rz", "q_sub_in_ss", "z_sat", "inf_mat_rz", "inf_mp_rz", "q_rz", "q_ss", "cpr_rz", "S_fp_rz", "S_lp_rz", "S_fp_ss", "S_lp_ss", "flow_dir_topo", "outer_boundary", "z_wf", "no_wf", "event_id"):
                if nm in names:
                    print(f"      in {nm:16s} {got[names.index(nm), i]!r:26} {ref[names.index(nm), i]!r}   before: {prev[names.index(nm), i]!r}")
        seen += 1
        if resync:   # continue from the oracle's state: every report is then a fresh one-step deviation
            H.upload_snapshot(ctx, ref, names)
        if seen >= 2:
            break
