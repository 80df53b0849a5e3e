#!/usr/bin/env python3
"""Which deviation of the device kernel from the reference's arithmetic produces its residue ties?  (VERDICT r2 next #9.)

The device differs from the reference in the last bits of three things: the power function, the ORDER of the cumulative sums over the
age axis (a wave scan instead of numpy's running sum) and the sub-step distributions (accumulated directly instead of as differences of
accumulated cumulative sums).  The CPU oracle can take either summation order (oracle/sas_oracle.c, OC_SAS_DEVICE_ORDER): this tool
counts, per golden case, the (day, column) pairs that miss 1e-10 when every day restarts from the reference's state -- the criterion of
tests/sas_binding.DEVICE_TIES -- for each combination.  Runs on the CPU:

    python tools/sas_tie_causes.py [case ...]
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["sas_power_a40", "sas_families_a50", "sas_mixed_a70", "sas_stats_a30", "sas_deuterium_a40"]
LABEL = {0: "reference order (the oracle as tested)", 1: "cumulative sums in the wave scan's order", 2: "sub-step distributions accumulated directly",
         3: "both"}

CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
from sas_binding import SasGolden, column_deviation
out = []
for case in sys.argv[2:]:
    g = SasGolden(case)
    st = g.new_state()
    tight = []
    for d in range(1, g.ndays + 1):
        g.load_state(st, d - 1)
        g.load_inputs(st, d)
        st.step_oracle()
        try:
            tight.append(column_deviation(st, lambda k: g.day(d, k), g.n, bool(g.stats)))
        except AssertionError as e:
            tight.append(np.zeros(g.n, bool))
    tight = np.array(tight)
    out.append(f"{case}={int(np.count_nonzero(~tight))}/{tight.size}")
print(" ".join(out))
"""

if __name__ == "__main__":
    cases = sys.argv[1:] or CASES
    for order in (0, 1, 2, 3):
        env = dict(os.environ, OC_SAS_DEVICE_ORDER=str(order))
        r = subprocess.run([sys.executable, "-c", CHILD, REPO] + cases, env=env, capture_output=True, text=True)
        print(f"order {order} ({LABEL[order]}): {r.stdout.strip() or r.stderr.strip()[-300:]}")
