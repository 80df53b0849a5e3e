"""Builds a ready-to-step SVAT state on the device from user parameters, using only the native
setup kernels.  This is the sequence of `RogerSetup.setup()` for the SVAT model
(roger/roger.py:258-336): parameters -> surface/soil derived parameters -> initial conditions.
"""
import numpy as np

from . import _native as N
from . import lookuptables as lut

# benchmarks/SVAT_benchmark.py:92-103,117-121
BENCHMARK_PARAMS = dict(lu_id=8, z_soil=2000.0, dmpv=50.0, lmpv=50.0, theta_ac=0.1, theta_ufc=0.1, theta_pwp=0.2,
                        ks=5.0, kf=2500.0, theta_rz=0.3, theta_ss=0.3)


def hetero_params(n, seed=42):
    """Seeded heterogeneous parameter maps (SURVEY.md section 8d, config 2 variant)."""
    rng = np.random.default_rng(seed)
    p = dict(
        lu_id=rng.choice([5, 8, 10, 13], n),
        z_soil=rng.uniform(600, 2000, n).round(0),
        dmpv=rng.choice([0.0, 25.0, 50.0, 100.0], n),
        theta_ac=rng.uniform(0.05, 0.2, n),
        theta_ufc=rng.uniform(0.08, 0.25, n),
        theta_pwp=rng.uniform(0.05, 0.3, n),
        ks=rng.uniform(1, 50, n),
        kf=np.full(n, 2500.0),
    )
    p["lmpv"] = np.minimum(rng.choice([50.0, 300.0, 600.0, 1000.0], n), p["z_soil"] * 0.9).round(0)
    sat = p["theta_ac"] + p["theta_ufc"] + p["theta_pwp"]
    p["theta_rz"] = p["theta_pwp"] + rng.uniform(0.2, 0.9, n) * (sat - p["theta_pwp"])
    p["theta_ss"] = p["theta_pwp"] + rng.uniform(0.2, 0.9, n) * (sat - p["theta_pwp"])
    return p


def create_svat(nx, ny, params=None, device=0, lateral=False, **settings):
    """Returns a `_native.Context` holding a fully initialised SVAT state.

    params: dict of scalars or arrays (n_cells) for lu_id, z_soil, dmpv, lmpv, theta_ac, theta_ufc,
    theta_pwp, ks, kf (+ optional sealing, S_dep_tot) and the initial theta_rz, theta_ss.
    """
    if lateral:
        settings = dict(settings, enable_lateral_flow=1)
    ctx = N.Context(nx, ny, device=device, **settings)
    n = ctx.n
    p = dict(BENCHMARK_PARAMS)
    p.update(params or {})
    ctx.set_luts(lut.ARR_ILU, lut.ARR_GC, lut.ARR_GCM, lut.ARR_RDLU)
    if lateral:
        ctx.set_lut_mlms(lut.ARR_MLMS)

    def full(v, dtype):
        a = np.asarray(v)
        return np.full(n, a, dtype=dtype) if a.ndim == 0 else np.ascontiguousarray(a, dtype=dtype).reshape(n)

    for k, v in p.items():
        if k in ("theta_rz", "theta_ss"):
            continue
        ctx.upload(k, full(v, ctx.dtype_of(k)))
    ctx.call("rh_topo")
    ctx.call("rh_params_surface")
    ctx.call("rh_params_soil")
    if lateral:
        ctx.call("rh_params_lateral")
    for k in ("theta_rz", "theta_ss"):
        a = full(p[k], np.float64)
        ctx.upload(k, a)
        ctx.upload(k + "_m1", a)
    ctx.call("rh_initial_conditions")
    ctx.sync()
    return ctx
