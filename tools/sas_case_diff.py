#!/usr/bin/env python3
"""Per-variable deviation of the device from a SAS golden case, each day restarted from the reference's state.

    python tools/sas_case_diff.py sas_euler_a40 [day]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from sas_binding import SasGolden  # noqa: E402
from test_hip_sas import make_ctx, pull, push  # noqa: E402


def main():
    case = sys.argv[1]
    g = SasGolden(case)
    days = [int(sys.argv[2])] if len(sys.argv) > 2 else range(1, g.ndays + 1)
    st = g.new_state()
    ctx = make_ctx(st)
    for d in days:
        g.load_state(st, d - 1)
        g.load_inputs(st, d)
        push(ctx, st)
        ctx.step(0)
        pull(ctx, st)
        for k in list(st.out) + list(st.state):
            a = st.out[k] if k in st.out else st.state[k]
            try:
                b = g.day(d, k)
            except KeyError:
                continue
            ok = np.isclose(a, b, rtol=1e-10, atol=1e-12, equal_nan=True)
            if not ok.all():
                idx = np.argwhere(~ok)
                print(f"day {d} {k}: {len(idx)} off, first {idx[:4].tolist()} got {a[tuple(idx[0])]!r} want {b[tuple(idx[0])]!r}")
    ctx.close()


if __name__ == "__main__":
    main()
