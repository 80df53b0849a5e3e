#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of a translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py roger_hip [pattern]      # or rh_sas_det_iso, rh_sas_rk4_anion, ... (roger_amd/build.py UNITS)
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from roger_amd.build import CSRC, FLAGS, HIPCC  # noqa: E402


def main():
    unit = sys.argv[1] if len(sys.argv) > 1 else "roger_hip"
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    extra = [a for a in sys.argv[3:]]
    cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, unit + ".hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    dem = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch':>7s} {'occ':>3s} {'LDS':>6s}")
    for (name, r), d in zip(rows.items(), dem):
        d = re.sub(r"\(Arena.*", "", d)
        if pat and not pat.search(d):
            continue
        print(f"{d[:70]:70s} {r.get('VGPRs', ''):>5s} {r.get('AGPRs', ''):>5s} {r.get('SGPRs', ''):>5s} {r.get('VGPRs Spill', ''):>6s} "
              f"{r.get('SGPRs Spill', ''):>6s} {r.get('ScratchSize [bytes/lane]', ''):>7s} {r.get('Occupancy [waves/SIMD]', ''):>3s} "
              f"{r.get('LDS Size [bytes/block]', ''):>6s}")


if __name__ == "__main__":
    main()
