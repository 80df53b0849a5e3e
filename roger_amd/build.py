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


# translation units and what each depends on (besides its own .hip): compiled side by side, objects cached in roger_amd/_obj
_SAS_DEPS = ("rh_sas_dev.h", "rh_sas_tables.inc", "include/roger_hip_sas.h", "include/rh_sas_arrays.def", "include/roger_hip.h")
UNITS = {
    "roger_hip": ("rh_physics.h", "rh_pow.h", "rh_col.h", "rh_sets.inc", "include/roger_hip.h", "include/rh_fields.def"),
    "rh_sas": _SAS_DEPS,                                            # the SAS C ABI
    "rh_sas_det_iso": _SAS_DEPS + ("rh_sas_kernels.h",),            # the deterministic SAS kernels: isotopes ...
    "rh_sas_det_anion": _SAS_DEPS + ("rh_sas_kernels.h",),          # ... and anions
    "rh_sas_euler_iso": _SAS_DEPS + ("rh_sas_solvers_impl.h",),     # the explicit solvers
    "rh_sas_euler_anion": _SAS_DEPS + ("rh_sas_solvers_impl.h",),
    "rh_sas_rk4_iso": _SAS_DEPS + ("rh_sas_solvers_impl.h",),
    "rh_sas_rk4_anion": _SAS_DEPS + ("rh_sas_solvers_impl.h",),
}
OBJ = os.path.join(PKG, "_obj")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _dep_path(d):
    return os.path.join(REPO, d) if d.startswith("include/") else os.path.join(CSRC, d)


def build_native(force=False, verbose=False):
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_sets.py")], check=True)
    os.makedirs(OBJ, exist_ok=True)
    this = os.path.abspath(__file__)   # the flags live here
    jobs, objs = [], []
    for unit, deps in UNITS.items():
        src, obj = os.path.join(CSRC, unit + ".hip"), os.path.join(OBJ, unit + ".o")
        objs.append(obj)
        if force or _newer(obj, [src, this] + [_dep_path(d) for d in deps]):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise subprocess.CalledProcessError(job.returncode, cmd)
    if force or jobs or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
