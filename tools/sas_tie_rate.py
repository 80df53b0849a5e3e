#!/usr/bin/env python3
"""Residue ties of the device against the reference / the oracle, the numbers behind
tests/sas_binding.py DEVICE_TIES and DEVICE_NEW_TIES.  Run on the GPU box:

    python -m pytest tests/test_hip_sas.py -m gpu -q -s -k "single_days or random_columns" | grep TIES
"""
import subprocess
import sys

if __name__ == "__main__":
    out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hip_sas.py", "-m", "gpu", "-q", "-s", "-k", "single_days or random_columns"],
                         capture_output=True, text=True).stdout
    print("\n".join(ln for ln in out.splitlines() if "TIES" in ln or "passed" in ln or "failed" in ln))
