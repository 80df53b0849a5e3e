"""Host-side setup of the offline oxygen-18 transport problem (SAS, deterministic solver) on the hip
backend: what `SVATOXYGEN18_benchmark.py` does in `set_parameters_setup`, `set_initial_conditions`
and `set_forcing`, written against the C ABI of include/roger_hip_sas.h.

The product path: there is no CPU implementation here; every step runs in the kernels of csrc/rh_sas_kernels.h / rh_sas_solvers_impl.h."""
import numpy as np

from . import _native

FLUXES = ("evap_soil", "transp", "q_rz", "q_ss", "cpr_rz")
# power-law exponents of the benchmark (benchmarks/SVATOXYGEN18_benchmark.py:129-138)
BENCHMARK_SAS_K = dict(evap_soil=0.2, cpr_rz=0.2, transp=0.5, q_rz=1.5, q_ss=1.5)
VSMOW_CONC18O = 2005.2e-6   # roger/settings.py:76


def delta_to_conc(delta_iso, vsmow=VSMOW_CONC18O):
    """roger/core/transport.py:315-325 (oxygen-18 branch)."""
    delta_iso = np.asarray(delta_iso, dtype=np.float64)
    return vsmow * (delta_iso / 1000. + 1.) / (1. + (delta_iso / 1000. + 1.) * vsmow)


def conc_to_delta(conc, vsmow=VSMOW_CONC18O, d18O_min=-20.0, d18O_max=0.0):
    """roger/core/transport.py:328-340 (oxygen-18 branch)."""
    conc = np.asarray(conc, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = 1000. * (conc / (vsmow * (1. - conc)) - 1.)
        return np.where((d < d18O_min) | (d > d18O_max), np.nan, d)


def benchmark_sas_params(n):
    """vs.sas_params_<flux>[..., 0] = 6 (power), [..., 1] = k; SVATOXYGEN18_benchmark.py:129-138."""
    out = {}
    for f in FLUXES:
        p = np.zeros((n, 8))
        p[:, 0] = 6
        p[:, 1] = BENCHMARK_SAS_K[f]
        out[f] = p
    return out


def initial_age_state(S, ages, delta0=-10.0):
    """sa, msa of one store as `set_initial_conditions` builds them (SVATOXYGEN18_benchmark.py:172-340):
    the storage S spread evenly over the age classes 1 .. ages-1 (`diff(linspace(0, S, ages))`),
    isotope signal delta0 in every class but the first."""
    S = np.asarray(S, dtype=np.float64)
    sa = np.zeros(S.shape + (ages,))
    sa[..., 1:] = np.diff(np.linspace(np.zeros_like(S), S, ages, axis=-1), axis=-1)
    msa = np.full(S.shape + (ages,), float(delta_to_conc(delta0)))
    msa[..., 0] = 0.0
    return sa, msa


def synthetic_daily_inputs(n, ndays, seed=42, delta_prec=-7.0):
    """Daily fluxes in mm/day for `ndays` days x `n` columns, as the reference's `set_forcing` would read them
    from a SVAT run (SVATOXYGEN18_benchmark.py:342-437).  The reference's SVAT.nc is not shipped: the values
    are synthetic with the magnitudes of a Central-European summer (wet days: infiltration and percolation;
    dry days: evapotranspiration and capillary rise), a domain-wide wet/dry sequence and +-20 % variation
    between columns.  delta-18O of precipitation is -7 permil (:380-382)."""
    rng = np.random.default_rng(seed)
    wet = rng.uniform(size=ndays) < 0.45
    out = {k: np.zeros((ndays, n)) for k in _native.DAILY_INPUTS}
    for d in range(ndays):
        var = lambda: rng.uniform(0.8, 1.2, n)  # noqa: E731
        if wet[d]:
            p = rng.uniform(4, 25)
            out["inf_mat_rz"][d] = 0.7 * p * var()
            out["inf_pf_rz"][d] = 0.15 * p * var()
            out["inf_pf_ss"][d] = 0.05 * p * var()
            out["q_rz"][d] = 0.35 * p * var()
            out["q_ss"][d] = 0.25 * p * var()
            out["evap_soil"][d] = rng.uniform(0.2, 0.8) * var()
            out["transp"][d] = rng.uniform(0.8, 2.0) * var()
        else:
            out["evap_soil"][d] = rng.uniform(0.3, 1.2) * var()
            out["transp"][d] = rng.uniform(1.5, 4.0) * var()
            out["q_rz"][d] = rng.uniform(0.0, 0.6) * var()
            out["q_ss"][d] = rng.uniform(0.2, 1.5) * var()
            out["cpr_rz"][d] = rng.uniform(0.0, 0.5) * var()
        out["C_in"][d] = float(delta_to_conc(delta_prec)) if wet[d] else 0.0
    return out


def create_sas(n_cells, ages, substeps, S_rz, S_ss, sas_params=None, daily=None, device=0, age_statistics=False,
               keep_distributions=False, chunk_cells=32768, delta0=-10.0, solver="deterministic"):
    """A SasContext with the benchmark's initial conditions, parameters and daily inputs on the device.
    S_rz, S_ss: scalars or (n_cells,) arrays of the initial storages in mm."""
    daily = daily if daily is not None else synthetic_daily_inputs(n_cells, 1)
    ndays = next(iter(daily.values())).shape[0]
    ctx = _native.SasContext(n_cells, ages, substeps, device=device, forcing_days=ndays,
                             age_statistics=age_statistics, keep_distributions=keep_distributions, solver=solver)
    sas_params = sas_params if sas_params is not None else benchmark_sas_params(n_cells)
    for f in FLUXES:
        ctx.upload(f"sas_params_{f}", sas_params[f])
    for k in _native.DAILY_INPUTS:
        ctx.upload(k, daily[k])
    S_rz = np.broadcast_to(np.asarray(S_rz, dtype=np.float64), (n_cells,))
    S_ss = np.broadcast_to(np.asarray(S_ss, dtype=np.float64), (n_cells,))
    for c0 in range(0, n_cells, chunk_cells):
        c1 = min(n_cells, c0 + chunk_cells)
        for key, S in (("rz", S_rz), ("ss", S_ss)):
            sa, msa = initial_age_state(S[c0:c1], ages, delta0)
            ctx.upload_cells(f"sa_{key}", c0, sa)
            ctx.upload_cells(f"msa_{key}", c0, msa)
    return ctx
