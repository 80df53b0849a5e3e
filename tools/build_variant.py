#!/usr/bin/env python3
"""Builds a kernel variant next to the product library, for A/B measurements on the GPU box:

    python tools/build_variant.py nt3 -DRH_NT=3        -> roger_amd/variants/libroger_hip_nt3.so
    ROGER_HIP_LIB=roger_amd/variants/libroger_hip_nt3.so python bench.py ...

(roger_amd/variants/ is git-ignored; it travels with gpurun like the product .so.)"""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from roger_amd.build import CSRC, FLAGS, HIPCC, PKG  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(PKG, "variants", f"libroger_hip_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
with tempfile.TemporaryDirectory() as tmp:
    jobs = []
    for unit in ("roger_hip", "rh_sas"):
        cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, unit + ".hip"), "-o", os.path.join(tmp, unit + ".o")]
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise SystemExit(" ".join(cmd))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(tmp, u + ".o") for u in ("roger_hip", "rh_sas")] + ["-o", out], check=True)
print(out)
