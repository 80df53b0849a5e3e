"""GPU: device-side daily accumulators (rh_diag_*) and the SVAT -> oxygen-18 transport chain on the device.

The SAS golden files hold, as `in_*`, the daily flux sums and end-of-day storages of the REFERENCE's SVAT run that
fed the reference's transport model (tests/golden/make_golden_sas.py: run_svat_daily).  Here the same SVAT setup
runs on the hip backend with device-side hooks and the device-side accumulators; its daily sums must equal the
reference's (rtol 1e-9: sums of up to 144 step values that each agree to 1e-10)."""
import os
import sys

import numpy as np
import pytest

import sas_binding as sb

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
pytestmark = pytest.mark.gpu

RATE = ("prec", "inf_mat_rz", "inf_mp_rz", "inf_sc_rz", "inf_ss", "transp", "evap_soil", "cpr_rz", "q_rz", "q_ss")
COLLECT = ("S_rz", "S_ss", "S_snow")
SEEDS = {"sas_power_a40": 3, "sas_mixed_a70": 5, "sas_stats_a30": 11}   # tests/golden/make_golden_sas.py: cases


def run_svat_with_accumulators(g, case):
    from make_golden import hetero_params            # parameter generator of the golden runs (no reference import)
    from roger_amd.forcing import combo_forcing
    from svat_scripts import make_model

    seed = SEEDS[case]
    model = make_model(hetero_params(g.nx, g.ny, seed=seed), combo_forcing(ndays=g.ndays, seed=seed), g.ndays)
    model.setup()
    ctx = model.state.backend_context
    ctx.diag_configure(rate=RATE, collect=COLLECT, n_slots=g.ndays + 1)
    model.enable_device_hooks()
    vs = model.state.variables
    while vs.time < g.ndays * 86400:
        model.run_device(1)
    assert vs.time == g.ndays * 86400
    return model, ctx


@pytest.mark.parametrize("case", ["sas_stats_a30", "sas_power_a40"])
def test_daily_sums_match_reference(case):
    g = sb.SasGolden(case)
    model, ctx = run_svat_with_accumulators(g, case)
    steps = [ctx.diag_steps(d - 1) for d in range(1, g.ndays + 1)]
    assert sum(steps) == model.state.variables.itt and set(steps) <= set(range(1, 145)) and min(steps) == 1   # dry days: one step
    for d in range(1, g.ndays + 1):
        for k in RATE + COLLECT:
            want = g.z[f"in_{k}"][:, :, d].reshape(-1)
            got = ctx.diag_download(k, d - 1)
            assert np.allclose(got, want, rtol=1e-9, atol=1e-10), f"{case} day {d} {k}: {np.abs(got - want).max()}"
    ctx.close()


def test_svat_to_transport_on_device():
    """The chain without the host in the data path: daily sums accumulate on the device, are copied device-to-device
    into the transport context's daily inputs, and the oxygen-18 step runs on them; compared with the reference's
    transport run (loose bound: free-running trajectory)."""
    from roger_amd import _native

    case = "sas_power_a40"
    g = sb.SasGolden(case)
    model, ctx = run_svat_with_accumulators(g, case)
    n = g.n
    sas = _native.SasContext(n, g.ages, g.substeps, forcing_days=g.ndays, keep_distributions=True)
    for f in sb.FLUXES:
        sas.upload(f"sas_params_{f}", g.z[f"sas_{f}"])
    for k in ("sa_rz", "msa_rz", "sa_ss", "msa_ss"):
        sas.upload(k, g.day(0, k))
    sas.upload("C_in", np.stack([g.day(d, "C_in") for d in range(1, g.ndays + 1)]))   # isotope input: host-side hook
    # device-to-device: row d-1 of each daily input <- the accumulator of day d
    ctx.sync()
    for d in range(1, g.ndays + 1):
        for dst, srcs in (("inf_mat_rz", ("inf_mat_rz",)), ("inf_pf_ss", ("inf_ss",)), ("evap_soil", ("evap_soil",)),
                          ("transp", ("transp",)), ("q_rz", ("q_rz",)), ("q_ss", ("q_ss",)), ("cpr_rz", ("cpr_rz",))):
            sas.set_daily_from_device(dst, d - 1, ctx.diag_device_ptr(srcs[0], d - 1))
    # inf_pf_rz = inf_mp_rz + inf_sc_rz (SVATOXYGEN18_benchmark.py:352): two accumulators, summed on the host here
    sas.upload("inf_pf_rz", np.stack([ctx.diag_download("inf_mp_rz", d - 1) + ctx.diag_download("inf_sc_rz", d - 1)
                                      for d in range(1, g.ndays + 1)]))
    ctx.sync()
    sas.run_days(0, g.ndays)
    for k in ("sa_rz", "sa_ss"):
        assert np.isclose(sas.download(k), g.day(g.ndays, k), rtol=5e-3, atol=5e-3).all(), k
    for k in ("C_iso_q_ss", "C_iso_rz", "C_iso_ss"):
        assert np.isclose(sas.download(k), g.day(g.ndays, k), rtol=5e-3, atol=5e-3, equal_nan=True).all(), k
    sas.close()
    ctx.close()


def test_accumulators_keep_summing_on_the_routine_path_after_a_limited_device_run(monkeypatch):
    """a device run under rh_set_time_limit ends with halted launches, whose prologue leaves DevState::skipped = 1.
    The accumulator kernel behind the routine-by-routine step (rh_step_core: no fused launch in front) must not look at that flag --
    it returned at once and the daily sums silently stopped.  Day 1 on the device under a limit, day 2 through the three-call step
    with the hooks on the host: both days' sums equal the reference's."""
    from make_golden import hetero_params
    from roger_amd.forcing import combo_forcing
    from svat_scripts import make_model

    case = "sas_stats_a30"
    g = sb.SasGolden(case)
    seed = SEEDS[case]
    model = make_model(hetero_params(g.nx, g.ny, seed=seed), combo_forcing(ndays=g.ndays, seed=seed), g.ndays)
    model.setup()
    ctx = model.state.backend_context
    ctx.diag_configure(rate=RATE, collect=COLLECT, n_slots=g.ndays + 1)
    vs = model.state.variables
    ctx.set_time_limit(86400)
    model.run_device(200, final=False)      # far more launches than the day has steps: the trailing ones are halted
    ctx.set_time_limit(None)
    assert vs.time == 86400
    monkeypatch.setenv("RH_STEP_BY_ROUTINE", "1")
    assert not model._fused_host_step_possible()
    while vs.time < 2 * 86400:
        model.step(model.state)
    ctx.sync()
    for d in (1, 2):
        assert ctx.diag_steps(d - 1) >= 1
        for k in RATE + COLLECT:
            want = g.z[f"in_{k}"][:, :, d].reshape(-1)
            got = ctx.diag_download(k, d - 1)
            assert np.allclose(got, want, rtol=1e-9, atol=1e-10), f"day {d} {k}: {np.abs(got - want).max()}"
    ctx.close()
