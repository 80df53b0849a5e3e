"""The SAS / oxygen-18 oracle (oracle/sas_oracle.c) against the reference's own outputs
(tests/golden/sas_*.npz, produced by tests/golden/make_golden_sas.py from the reference NumPy backend).

Bar: floating point, rtol 1e-10 / atol 1e-12 (the restatement keeps numpy's operation order, in
practice it is bit-exact up to libm `pow`); NaN patterns identical."""
import os

import numpy as np
import pytest

from sas_binding import ANION_SOLVER_CASES, FIRST_TIE, ORACLE_TIES, check_day_loose, FLUXES, INFS, SAS_CASES, SOLVER_CASES, STAT_Q, STAT_TARGETS, GOLDEN, SasGolden, compare_sas

CASES = [c for c in SAS_CASES + SOLVER_CASES if os.path.exists(os.path.join(GOLDEN, f"{c}.npz"))]


def compare_msa(got, want, sa, what):
    """Isotope signal per age class, compared where the age class holds water.  An emptied class
    (`sa - flux * (sa / flux)`) keeps a rounding residue of either sign before it is snapped to 0,
    and `msa = where(sa <= 0, 0, msa)` turns that sign into 0-or-kept; numpy's AVX-512 `pow` and
    libm's differ in the last bit for ~5 % of arguments, so the signal of EMPTY classes (weight 0 in
    every later formula) is not reproducible between the two and is excluded."""
    holds = np.asarray(sa) > 0
    compare_sas(np.where(holds, got, 0.0), np.where(holds, want, 0.0), what)


def check_day(g, st, d, tag):
    for f in FLUXES:
        for pre in ("tt", "mtt", "TT", "C", "C_iso"):
            compare_sas(st.out[f"{pre}_{f}"], g.day(d, f"{pre}_{f}"), f"{tag} day {d} {pre}_{f}")
    for f in INFS:
        for pre in ("C", "C_iso"):
            compare_sas(st.out[f"{pre}_{f}"], g.day(d, f"{pre}_{f}"), f"{tag} day {d} {pre}_{f}")
    for k in ("C_rz", "C_ss", "C_s", "C_iso_rz", "C_iso_ss", "C_iso_s"):
        compare_sas(st.out[k], g.day(d, k), f"{tag} day {d} {k}")
    # the reference ages sa_rz/sa_ss (and only those) at the end of the step
    for k in ("sa_rz", "sa_ss"):
        compare_sas(st.state[k], g.day(d, k), f"{tag} day {d} {k}")
        compare_msa(st.state["m" + k], g.day(d, "m" + k), g.day(d, k), f"{tag} day {d} m{k}")
    compare_sas(st.out["sa_s"], g.day(d, "sa_s"), f"{tag} day {d} sa_s")
    compare_msa(st.out["msa_s"], g.day(d, "msa_s"), g.day(d, "sa_s"), f"{tag} day {d} msa_s")
    if g.stats:
        for w, p in STAT_TARGETS:
            for q in STAT_Q:
                compare_sas(st.out[f"{p}{q}_{w}"], g.day(d, f"{p}{q}_{w}"), f"{tag} day {d} {p}{q}_{w}")


@pytest.mark.parametrize("case", CASES)
def test_single_days_from_reference_states(case):
    """Every golden day restarted from the reference's own state: all outputs at rtol 1e-10.  Residue ties
    (sas_binding.FIRST_TIE: numpy's pow vs libm's) may take single (day, column) pairs off: at most ONE more than measured
    for the case (sas_binding.ORACLE_TIES), and those within the loose bound."""
    from sas_binding import column_deviation

    g = SasGolden(case)
    st = g.new_state()
    tight = []
    for d in range(1, g.ndays + 1):
        g.load_state(st, d - 1)
        g.load_inputs(st, d)
        st.step_oracle()
        tight.append(column_deviation(st, lambda k: g.day(d, k), g.n, bool(g.stats)))
    tight = np.array(tight)
    ties = int(np.count_nonzero(~tight))
    assert ties <= ORACLE_TIES[case] + 1, f"{case}: {ties} of {tight.size} (day, column) pairs deviate (measured {ORACLE_TIES[case]}): {np.argwhere(~tight)[:8]}"


@pytest.mark.parametrize("case", CASES)
def test_trajectory(case):
    """Free-running from the initial state.  Tight until the first residue tie (module docstring of
    sas_binding.FIRST_TIE), bounded afterwards."""
    g = SasGolden(case)
    st = g.new_state()
    g.load_state(st, 0)
    for d in range(1, g.ndays + 1):
        g.load_inputs(st, d)
        st.step_oracle()
        if d < FIRST_TIE.get(case, g.ndays + 1):
            check_day(g, st, d, case)
        else:
            check_day_loose(g, st, d, case)


def test_cases_present():
    assert CASES, "no SAS golden file found"
    g = SasGolden(CASES[0])
    # the goldens exercise what they are meant to: water leaves by every flux and isotope signals are finite
    for f in FLUXES:
        tot = sum(np.nansum(g.day(d, f"tt_{f}")) for d in range(1, g.ndays + 1))
        assert tot > 0, f
    assert np.isfinite(g.day(g.ndays, "C_iso_q_ss")).any()


def test_rescale_after_warmup():
    """soil.rescale_SA (rescale_sa_msa_iso_soil_kernel, core/soil.py:1250-1395): the reference's state after its
    warm-up run (w000_*) -> rescaled state (d000_*)."""
    g = SasGolden("sas_warmup_a30")
    st = g.new_state()
    for k in st.state:
        st.state[k][:] = g.z[f"w000_{k}"]
    for k in st.S_init:
        st.S_init[k][:] = g.z[k]
    st.rescale_oracle()
    for k in ("sa_rz", "sa_ss"):
        compare_sas(st.state[k], g.day(0, k), f"rescale {k}")
        compare_msa(st.state["m" + k], g.day(0, "m" + k), g.day(0, k), f"rescale m{k}")
    compare_sas(st.out["sa_s"], g.day(0, "sa_s"), "rescale sa_s")
    compare_msa(st.out["msa_s"], g.day(0, "msa_s"), g.day(0, "sa_s"), "rescale msa_s")
    for k in ("C_rz", "C_ss", "C_s", "C_iso_rz", "C_iso_ss", "C_iso_s"):
        compare_sas(st.out[k], g.day(0, k), f"rescale {k}")
    assert abs(st.state["sa_rz"].sum(axis=1) - g.z["S_rz_init"]).max() < 1e-9


# --- bromide: the reference's anion kernels (msa = solute mass by age) -------------------------------------------

def bromide_names(stats):
    names = [f"{pre}_{f}" for f in FLUXES for pre in ("tt", "TT")]
    names += [f"{pre}_{f}" for f in FLUXES[1:] for pre in ("mtt", "C", "M")]
    names += [f"{pre}_{f}" for f in INFS for pre in ("C", "M")]
    names += ["C_rz", "C_ss", "C_s", "M_rz", "M_ss", "M_s", "sa_rz", "sa_ss", "msa_rz", "msa_ss", "sa_s", "msa_s"]
    if stats:
        # the reference never assigns rt10 / rt90 of root zone and subsoil
        names += [f"{p}{q}_{w}" for w, p in STAT_TARGETS for q in STAT_Q if not (w in ("rz", "ss") and q in ("10", "90"))]
    return names


def check_day_bromide(g, st, d, tag, rtol=1e-10):
    for k in bromide_names(bool(g.stats)):
        got = st.state[k] if k in st.state else st.out[k]
        compare_sas(got, g.day(d, k), f"{tag} day {d} {k}", rtol=rtol, atol=1e-11)


def test_bromide_single_days_from_reference_states():
    """calc_*_transport_anion_kernel day by day from the reference's own states (sas_bromide_a40: two crop columns whose
    uptake stops above 80 % saturation, pulses of bromide on three wet days)."""
    g = SasGolden("sas_bromide_a40")
    assert g.tracer == "bromide"
    st = g.new_state()
    for d in range(1, g.ndays + 1):
        g.load_state(st, d - 1)
        g.load_inputs(st, d)
        st.step_oracle()
        check_day_bromide(g, st, d, "bromide")
    assert max(g.day(d, "M_q_ss").max() for d in range(1, g.ndays + 1)) > 1.0   # the pulse reaches the percolation
    assert any((g.day(d, "M_transp") == 0).any() and (g.day(d, "M_transp") > 0).any() for d in range(1, g.ndays + 1))


def test_bromide_trajectory():
    g = SasGolden("sas_bromide_a40")
    st = g.new_state()
    g.load_state(st, 0)
    for d in range(1, g.ndays + 1):
        g.load_inputs(st, d)
        st.step_oracle()
        check_day_bromide(g, st, d, "bromide trajectory", rtol=1e-9)
    # mass balance of the whole run: what came in is in the soil or left with transpiration / percolation
    m_in = sum(g.day(d, "M_inf_mat_rz") + g.day(d, "M_inf_pf_rz") + g.day(d, "M_inf_pf_ss") for d in range(1, g.ndays + 1))
    m_out = sum(g.day(d, "M_transp") + g.day(d, "M_q_ss") for d in range(1, g.ndays + 1))
    # (not exact: an age class emptied of water drops what solute it still held, `msa = where(sa <= 0, 0, msa)`)
    np.testing.assert_allclose(st.out["M_s"], m_in - m_out, rtol=1e-4, atol=1e-9)
    assert (st.out["M_s"] <= (m_in - m_out) * (1 + 1e-12) + 1e-9).all()


def explicit_anion_names(stats):
    """What svat_transport_model_euler / _rk4 and the storages assign for the anions: no solute masses of the fluxes (M_*), nothing for
    the soil evaporation's solute."""
    return [k for k in bromide_names(stats) if not (k.startswith("M_") and k not in ("M_rz", "M_ss", "M_s"))]


@pytest.mark.parametrize("case", ANION_SOLVER_CASES)
def test_explicit_solvers_with_the_anion_kernels(case):
    """settings.sas_solver = "Euler" / "RK4" with bromide (the reference's bromide tutorial runs RK4): day by day from the reference's
    states, then free-running."""
    g = SasGolden(case)
    assert g.tracer == "bromide" and g.solver in ("Euler", "RK4")
    names = explicit_anion_names(bool(g.stats))
    st = g.new_state()
    for d in range(1, g.ndays + 1):
        g.load_state(st, d - 1)
        g.load_inputs(st, d)
        st.step_oracle()
        for k in names:
            compare_sas(st.state[k] if k in st.state else st.out[k], g.day(d, k), f"{case} day {d} {k}", rtol=1e-10, atol=1e-11)
    g.load_state(st, 0)
    for d in range(1, g.ndays + 1):
        g.load_inputs(st, d)
        st.step_oracle()
        for k in names:
            compare_sas(st.state[k] if k in st.state else st.out[k], g.day(d, k), f"{case} trajectory day {d} {k}", rtol=1e-9, atol=1e-11)
    assert max(g.day(d, "C_q_ss").max() for d in range(1, g.ndays + 1)) > 0   # solute reaches the percolation


def test_bromide_rescale_after_warmup():
    """rescale_sa_msa_anion_soil_kernel, bromide branch (core/soil.py:1399-1506): the reference's state after its warm-up
    run (w000_*) -> rescaled age vectors, the soil emptied of bromide (d000_*); then the run proper, free-running."""
    g = SasGolden("sas_bromide_warmup_a30")
    st = g.new_state()
    for k in st.state:
        st.state[k][:] = g.z[f"w000_{k}"]
    assert st.state["msa_rz"].any()          # the warm-up run left bromide in the soil
    for k in st.S_init:
        st.S_init[k][:] = g.z[k]
    st.rescale_oracle()
    for k in ("sa_rz", "sa_ss", "msa_rz", "msa_ss"):
        compare_sas(st.state[k], g.day(0, k), f"bromide rescale {k}")
    for k in ("sa_s", "msa_s", "C_rz", "C_ss", "C_s", "M_rz", "M_ss", "M_s"):
        compare_sas(st.out[k], g.day(0, k), f"bromide rescale {k}")
    for d in range(1, g.ndays + 1):
        g.load_inputs(st, d)
        st.step_oracle()
        check_day_bromide(g, st, d, "bromide after warm-up", rtol=1e-9)


@pytest.mark.parametrize("case", ["sas_chloride_warmup_a30", "sas_chloride_rk4_warmup_a30", "sas_virtualtracer_euler_a30"])
def test_chloride_rescale_and_run(case):
    """settings.enable_chloride: the anion kernels as for bromide; soil.rescale_SA scales the solute with the water
    (rescale_sa_msa_anion_soil_kernel, core/soil.py:1507-1640).  Reference state after its warm-up run -> rescaled state
    -> ten days free-running, age statistics on.  Also with the RK4 solver (warm-up and run)."""
    g = SasGolden(case)
    assert g.tracer in ("chloride", "virtualtracer")   # (under an explicit solver the virtual tracer IS chloride: its soil evaporation carries nothing)
    st = g.new_state()
    for k in st.state:
        st.state[k][:] = g.z[f"w000_{k}"]
    for k in st.S_init:
        st.S_init[k][:] = g.z[k]
    st.rescale_oracle()
    for k in ("sa_rz", "sa_ss", "msa_rz", "msa_ss"):
        compare_sas(st.state[k], g.day(0, k), f"chloride rescale {k}")
    for k in ("sa_s", "msa_s", "C_rz", "C_ss", "C_s"):
        compare_sas(st.out[k], g.day(0, k), f"chloride rescale {k}")
    assert st.state["msa_rz"].any()
    for d in range(1, g.ndays + 1):
        g.load_inputs(st, d)
        st.step_oracle()
        if g.solver == "deterministic":
            check_day_bromide(g, st, d, "chloride after warm-up", rtol=1e-9)
        else:
            for k in explicit_anion_names(bool(g.stats)):
                compare_sas(st.state[k] if k in st.state else st.out[k], g.day(d, k), f"chloride {g.solver} after warm-up day {d} {k}", rtol=1e-9, atol=1e-11)


def test_threads_do_not_change_results():
    """oc_sas_step forks over the columns from 256 columns on (OpenMP): one thread against four, bit for bit."""
    import sas_binding as sb

    g = SasGolden("sas_stats_a30")
    reps = 80
    outs = []
    for threads in (1, 4):
        sb.lib().oc_sas_set_num_threads(threads)
        st = sb.SasState(g.n * reps, g.ages, g.substeps, True)
        st.maskCatch[:] = np.tile(g.z["maskCatch"], reps)
        for f in FLUXES:
            st.sas[f][:] = np.tile(g.z[f"sas_{f}"], (reps, 1))
        one = g.new_state()
        g.load_state(one, 0)
        for k in st.state:
            st.state[k][:] = np.tile(one.state[k], (reps, 1))
        for d in (1, 2, 3):
            g.load_inputs(one, d)
            for k in st.inp:
                st.inp[k][:] = np.tile(one.inp[k], reps)
            st.step_oracle()
        outs.append({**{k: v.copy() for k, v in st.state.items()}, **{k: v.copy() for k, v in st.out.items()}})
    sb.lib().oc_sas_set_num_threads(1)
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)


def test_virtualtracer_rescale_and_run():
    """settings.enable_virtualtracer: the anion kernels, the soil evaporation takes the tracer along at alpha = 1
    (calc_evaporation_transport_virtualtracer_kernel), rescaling as for chloride."""
    g = SasGolden("sas_virtualtracer_a30")
    assert g.tracer == "virtualtracer"
    st = g.new_state()
    for k in st.state:
        st.state[k][:] = g.z[f"w000_{k}"]
    for k in st.S_init:
        st.S_init[k][:] = g.z[k]
    st.rescale_oracle()
    for k in ("sa_rz", "sa_ss", "msa_rz", "msa_ss"):
        compare_sas(st.state[k], g.day(0, k), f"virtual tracer rescale {k}")
    names = bromide_names(bool(g.stats)) + ["mtt_evap_soil", "C_evap_soil", "M_evap_soil"]
    took = 0.0
    for d in range(1, g.ndays + 1):
        g.load_inputs(st, d)
        st.step_oracle()
        for k in names:
            got = st.state[k] if k in st.state else st.out[k]
            compare_sas(got, g.day(d, k), f"virtual tracer day {d} {k}", rtol=1e-9, atol=1e-11)
        took += st.out["M_evap_soil"].sum()
    assert took > 0      # the evaporation did take tracer
