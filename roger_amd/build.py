"""Builds the native library (HIP kernels + C ABI) in-tree for gfx950.

`python -m roger_amd.build` or `roger_amd.build.build_native()`.  hipcc cross-compiles without a
GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libroger_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: the reference evaluates every product and sum separately (NumPy ufuncs);
# contracting a*b+c into an FMA changes roundings and can flip the model's `>`/`>=` masks.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-I" + os.path.join(REPO, "include"), "-I" + CSRC]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_native(force=False, verbose=False):
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_sets.py")], check=True)
    sources = [os.path.join(CSRC, f) for f in ("roger_hip.hip", "rh_sas.hip", "rh_physics.h", "rh_col.h", "rh_sets.inc", "rh_sas_tables.inc")]
    sources += [os.path.join(REPO, "include", f) for f in ("roger_hip.h", "rh_fields.def", "roger_hip_sas.h",
                                                           "rh_sas_arrays.def")]
    if force or _newer(LIB, sources):
        # the two translation units compile side by side (each takes about a minute), then one link
        import tempfile

        with tempfile.TemporaryDirectory() as tmp:
            jobs = []
            for unit in ("roger_hip", "rh_sas"):
                cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, unit + ".hip"), "-o", os.path.join(tmp, unit + ".o")]
                if verbose:
                    print(" ".join(cmd))
                jobs.append((cmd, subprocess.Popen(cmd)))
            for cmd, job in jobs:
                if job.wait() != 0:
                    raise subprocess.CalledProcessError(job.returncode, cmd)
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(tmp, u + ".o") for u in ("roger_hip", "rh_sas")] + ["-o", LIB]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
