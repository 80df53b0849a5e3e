"""Input validators with the reference's error behaviour (roger/core/numerics.py:7-121): they raise
ValueError with the same messages.  Run on the host mirrors at setup time."""
import numpy as npx


def _interior(a):
    return a[2:-2, 2:-2]


def validate_parameters_surface(state):
    vs = state.variables
    m = _interior(vs.maskCatch)
    seal = _interior(vs.sealing)
    if npx.any(((seal > 1) | (seal < 0)) & m):
        raise ValueError("sealing-parameter is out of range.")
    lu = _interior(vs.lu_id)
    if npx.any(((lu > 1000) | (lu < 0)) & m):
        raise ValueError("lu_id-parameter is out of range.")
    if npx.any(npx.isnan(seal) & m):
        raise ValueError("sealing-parameter contains non-numeric values.")


def validate_parameters_soil(state):
    vs = state.variables
    m = _interior(vs.maskCatch)
    g = lambda name: _interior(getattr(vs, name))  # noqa: E731
    z_soil, lmpv, ks = g("z_soil"), g("lmpv"), g("ks")
    tsum = g("theta_pwp") + g("theta_ufc") + g("theta_ac")
    if npx.any((z_soil > 0) & ((tsum > 0.99) | (tsum < 0.01)) & m):
        raise ValueError("theta-parameters are out of range.")
    if npx.any((z_soil > 0) & ((ks > 10000) | (ks < 0)) & m):
        raise ValueError("ks-parameter is out of range.")
    if npx.any((z_soil > 0) & ((lmpv > z_soil) | (lmpv < 0)) & m):
        raise ValueError("lmpv-parameter is out of range.")
    if npx.any((z_soil <= 0) & m):
        raise ValueError("z_soil-parameter contains zero or negative values.")
    if npx.any((_interior(vs.z_gw)[..., 1] * 1000 < z_soil) & m):
        raise ValueError("One or more z_gw-parameter value are less than z_soil-parameter value.")
    for name in ("theta_ac", "theta_ufc", "theta_pwp"):
        if npx.any((g(name) <= 0) & m):
            raise ValueError(f"{name}-parameter contains zero or negative values.")
    for name in ("z_soil", "dmpv", "lmpv", "theta_pwp", "theta_ufc", "theta_ac", "ks", "kf"):
        if npx.any(npx.isnan(g(name)) & m):
            raise ValueError(f"{name}-parameter contains non-numeric values.")


def validate_initial_conditions_soil(state):
    vs = state.variables
    m = _interior(vs.maskCatch)
    th_sat = _interior(vs.theta_sat)
    if npx.any((_interior(vs.theta_rz)[..., 0] > th_sat) & m):
        raise ValueError("theta_rz is too high.")
    if npx.any((_interior(vs.theta_ss)[..., 0] > th_sat) & m):
        raise ValueError("theta_ss is too high.")
