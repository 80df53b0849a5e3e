#!/usr/bin/env python3
"""Builds a kernel variant next to the product library, for A/B measurements on the GPU box:

    python tools/build_variant.py nt3 -DRH_NT=3                          -> roger_amd/variants/libroger_hip_nt3.so
    python tools/build_variant.py ew4 --unit rh_sas_euler_iso -DRH_EULER_WAVES=4    (only that unit is recompiled with the flags)
    ROGER_HIP_LIB=roger_amd/variants/libroger_hip_nt3.so python bench.py ...

(roger_amd/variants/ is git-ignored; it travels with gpurun like the product .so.)"""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from roger_amd.build import CSRC, FLAGS, HIPCC, OBJ, PKG, UNITS, build_native  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
only = None
if "--unit" in extra:
    i = extra.index("--unit")
    only = extra[i + 1]
    del extra[i:i + 2]
build_native()   # the product's objects, reused for the units the flags do not concern
out = os.path.join(PKG, "variants", f"libroger_hip_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
with tempfile.TemporaryDirectory() as tmp:
    jobs, objs = [], []
    for unit in UNITS:
        if only and unit != only:
            objs.append(os.path.join(OBJ, unit + ".o"))
            continue
        obj = os.path.join(tmp, unit + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, unit + ".hip"), "-o", obj]
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise SystemExit(" ".join(cmd))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
print(out)
