"""INTEGRATION.md executed (build container only; skipped where the reference is absent): the reference's own tutorial script
runs unmodified through the reference's driver under backend="hip" -- the edits of INTEGRATION.md sections 1-4 applied in memory, the
process routines bound to the C ABI's entry points (served by the device double here) -- and reproduces the golden trajectory of the
NumPy backend.  tests/integration/reference_hip_backend.py is the script; a subprocess, because it rewires imports."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference lives in the build container only")
def test_reference_tutorial_script_runs_unmodified_on_the_hip_backend():
    r = subprocess.run([sys.executable, os.path.join(HERE, "integration", "reference_hip_backend.py"), "--days", "90"],
                       capture_output=True, text=True, cwd="/tmp", timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ran unmodified under backend='hip'" in r.stdout and "scalars identical at every step" in r.stdout
