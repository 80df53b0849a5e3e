"""The host package's offline-transport flow (RogerSetup.setup / step, RogerVariables synchronisation,
core.transport) with the oracle-backed SasContext stand-in: the very setup class that produced the SAS golden
vectors through the reference (tests/sas_scripts.py) runs through roger_amd and reproduces them."""
import numpy as np
import pytest

import sas_binding as sb
from sas_scripts import make_transport_model
from test_oracle_sas import compare_msa


def golden_inputs(g):
    svat = {k[3:]: g.z[k] for k in g.z.files if k.startswith("in_")}
    sas = {f: g.z[f"sas_{f}"].reshape(g.nx, g.ny, 8) for f in sb.FLUXES}
    return svat, sas


def interior(a, lvl=None):
    a = np.asarray(a)[2:-2, 2:-2]
    if lvl is not None:
        a = a[:, :, lvl]
    return a.reshape((-1,) + a.shape[2:])


def run_and_compare(g, model, first_tie=None, warmup=0):
    model.setup()
    model.warmup(repeat=warmup)
    vs = model.state.variables
    assert vs.itt == 0 and vs.time == 0 and model.state.settings.warmup_done
    if warmup:
        for k in ("sa_rz", "sa_ss"):
            sb.compare_sas(interior(getattr(vs, k), 1), g.day(0, k), f"after warm-up {k}")
        for k in ("C_rz", "C_iso_ss", "C_s"):
            sb.compare_sas(interior(getattr(vs, k), 1), g.day(0, k), f"after warm-up {k}")
    for d in range(1, g.ndays + 1):
        model.step(model.state)
        assert vs.itt == d and vs.time == d * 86400
        if first_tie is not None and d >= first_tie:
            continue
        for k in ("sa_rz", "sa_ss"):
            sb.compare_sas(interior(getattr(vs, k), 1), g.day(d, k), f"day {d} {k}")
            compare_msa(interior(getattr(vs, "m" + k), 1), g.day(d, "m" + k), g.day(d, k), f"day {d} m{k}")
            sb.compare_sas(interior(getattr(vs, k), 0), g.day(d, k), f"day {d} {k}[taum1]")
        for f in sb.FLUXES:
            for pre in ("tt", "TT", "C", "C_iso"):
                sb.compare_sas(interior(getattr(vs, f"{pre}_{f}")), g.day(d, f"{pre}_{f}"), f"day {d} {pre}_{f}")
        for k in ("C_rz", "C_ss", "C_s", "C_iso_rz", "C_iso_ss", "C_iso_s"):
            sb.compare_sas(interior(getattr(vs, k), 1), g.day(d, k), f"day {d} {k}")
        for k in ("C_in", "C_iso_in", "C_inf_mat_rz", "C_iso_inf_pf_ss"):
            sb.compare_sas(interior(getattr(vs, k)), g.day(d, k), f"day {d} {k}")
        for k in ("C_snow", "C_iso_snow"):      # host-side hook arithmetic (set_forcing / after_timestep)
            sb.compare_sas(interior(getattr(vs, k), 1), g.day(d, k), f"day {d} {k}")
        if g.stats:
            for w, p in sb.STAT_TARGETS:
                for q in sb.STAT_Q:
                    sb.compare_sas(interior(getattr(vs, f"{p}{q}_{w}")), g.day(d, f"{p}{q}_{w}"), f"day {d} {p}{q}_{w}", rtol=1e-9)


@pytest.fixture
def oracle_sas(monkeypatch):
    from oracle_sas_context import OracleSasContext
    from roger_amd import _native

    monkeypatch.setattr(_native, "SasContext", OracleSasContext)


@pytest.mark.parametrize("case", ["sas_stats_a30", "sas_mixed_a70", "sas_deuterium_a40", "sas_euler_a40", "sas_rk4_a40"])
def test_transport_setup_through_host_package(oracle_sas, case):
    g = sb.SasGolden(case)
    svat, sas = golden_inputs(g)
    model = make_transport_model("roger_amd", svat, sas, g.ages, g.substeps, g.ndays, bool(g.stats), tracer=g.tracer, solver=g.solver)
    run_and_compare(g, model, first_tie=sb.FIRST_TIE.get(case))


def test_transport_warmup_and_run(oracle_sas):
    """setup(); warmup(); run() -- the call sequence of benchmarks/SVATOXYGEN18_benchmark.py:483-485: one whole run,
    soil.rescale_SA, clock reset, then the run proper."""
    g = sb.SasGolden("sas_warmup_a30")
    svat, sas = golden_inputs(g)
    model = make_transport_model("roger_amd", svat, sas, g.ages, g.substeps, g.ndays, bool(g.stats), warmup_days=g.ndays)
    run_and_compare(g, model, warmup=1)


def test_transport_settings_are_checked(oracle_sas):
    g = sb.SasGolden("sas_stats_a30")
    svat, sas = golden_inputs(g)
    with pytest.raises(NotImplementedError, match="deterministic"):
        make_transport_model("roger_amd", svat, sas, g.ages, g.substeps, g.ndays, False).__class__(
            override=dict(sas_solver="Heun")).setup()
    with pytest.raises(ValueError, match="settings.h"):   # benchmarks/SVATOXYGEN18_benchmark.py:30-31
        make_transport_model("roger_amd", svat, sas, g.ages, g.substeps, g.ndays, False).__class__(
            override=dict(sas_solver="Euler")).setup()
    with pytest.raises(NotImplementedError, match="oxygen-18"):
        make_transport_model("roger_amd", svat, sas, g.ages, g.substeps, g.ndays, False).__class__(
            override=dict(enable_oxygen18=False)).setup()
    m = make_transport_model("roger_amd", svat, sas, g.ages, g.substeps, g.ndays, False)
    m.setup()
    with pytest.raises(ValueError, match="unexpected shape"):
        with m.state.variables.unlock():
            m.state.variables.sa_rz = np.zeros((3, 3))


# --- bromide ---------------------------------------------------------------------------------------------------------

def bromide_model(g, pkg="roger_amd", warmup_days=0):
    svat, sas = golden_inputs(g)
    shape = (g.nx, g.ny)
    C_IN = np.zeros(shape + (g.ndays + 1,))
    for d in range(1, g.ndays + 1):
        C_IN[:, :, d] = g.day(d, "C_in").reshape(shape)
    extra = {k: g.z[k].reshape(shape) for k in ("alpha_transp", "alpha_q", "lu_id")}
    extra["C_IN"] = C_IN
    if g.tracer in ("chloride", "virtualtracer"):   # the background concentrations tests/golden/make_golden_sas.py starts from
        extra.update(C_init_rz=4.0, C_init_ss=9.0)
    return make_transport_model(pkg, svat, sas, g.ages, g.substeps, g.ndays, bool(g.stats), warmup_days=warmup_days,
                                tracer=g.tracer, extra=extra, solver=g.solver)


TWO_LEVELS = ("sa_rz", "sa_ss", "msa_rz", "msa_ss", "sa_s", "msa_s", "C_rz", "C_ss", "C_s", "M_rz", "M_ss", "M_s")   # (x, y, timesteps, ...)


def run_and_compare_bromide(g, model, rtol=1e-9, warmup=0):
    """The setup class that produced sas_bromide_a40 through the reference (settings.enable_bromide, the anion kernels),
    run free through roger_amd: every day of the trajectory."""
    from test_oracle_sas import bromide_names, explicit_anion_names

    if g.solver != "deterministic":   # the explicit models never assign the solute masses of the fluxes
        bromide_names = explicit_anion_names

    model.setup()
    model.warmup(repeat=warmup)
    vs = model.state.variables
    if warmup:   # soil.rescale_SA: rescaled age vectors, the soil emptied of bromide / chloride scaled with the water
        for k in ("sa_rz", "sa_ss", "msa_rz", "msa_ss", "C_rz", "C_s"):
            sb.compare_sas(interior(getattr(vs, k), 1), g.day(0, k), f"{g.tracer} after warm-up {k}", rtol=rtol, atol=1e-11)
        assert np.asarray(vs.msa_rz).any() == (g.tracer != "bromide") and vs.itt == 0
    for d in range(1, g.ndays + 1):
        model.step(model.state)
        assert vs.itt == d
        for k in bromide_names(bool(g.stats)) + ["C_in", "M_in"]:
            v = getattr(vs, k)
            lvl = 1 if k in TWO_LEVELS else None
            sb.compare_sas(interior(v, lvl), g.day(d, k), f"bromide day {d} {k}", rtol=rtol, atol=1e-11)
        for k in ("M_rz", "C_ss", "sa_rz"):   # after_substep_anion: taum1 follows tau
            sb.compare_sas(interior(getattr(vs, k), 0), g.day(d, k), f"bromide day {d} {k}[taum1]", rtol=rtol, atol=1e-11)


def test_bromide_setup_through_host_package(oracle_sas):
    g = sb.SasGolden("sas_bromide_a40")
    run_and_compare_bromide(g, bromide_model(g))


@pytest.mark.parametrize("case", sb.ANION_SOLVER_CASES)
def test_bromide_with_explicit_solvers_through_host_package(oracle_sas, case):
    """settings.sas_solver = "Euler" / "RK4" with settings.enable_bromide (examples/plot_scale/svat_bromide_tutorial runs RK4)."""
    g = sb.SasGolden(case)
    run_and_compare_bromide(g, bromide_model(g))


def test_bromide_warmup_and_run(oracle_sas):
    g = sb.SasGolden("sas_bromide_warmup_a30")
    run_and_compare_bromide(g, bromide_model(g, warmup_days=g.ndays), warmup=1)


@pytest.mark.parametrize("case", ["sas_chloride_warmup_a30", "sas_chloride_rk4_warmup_a30", "sas_virtualtracer_euler_a30"])
def test_chloride_warmup_and_run(oracle_sas, case):
    """settings.enable_chloride: setup(); warmup(); run() against the reference's chloride run (deterministic and RK4)."""
    g = sb.SasGolden(case)
    run_and_compare_bromide(g, bromide_model(g, warmup_days=g.ndays), warmup=1)


def test_virtualtracer_warmup_and_run(oracle_sas):
    """settings.enable_virtualtracer through the host package (no C_IN series in the reference: set_forcing assigns C_in)."""
    g = sb.SasGolden("sas_virtualtracer_a30")
    run_and_compare_bromide(g, bromide_model(g, warmup_days=g.ndays), warmup=1)


def test_transport_needs_exactly_one_tracer():
    g = sb.SasGolden("sas_bromide_a40")
    model = bromide_model(g)

    class Both(type(model)):
        def set_settings(self, state):
            super().set_settings(state)
            with state.settings.unlock():
                state.settings.enable_oxygen18 = True

    with pytest.raises(NotImplementedError, match="exactly one"):
        Both().setup()


def test_transport_output_diagnostics(oracle_sas, tmp_path):
    """state.diagnostics of the offline transport model: nothing during the warm-up, record 0 = the rescaled initial state,
    then a record per day with the step's values -- concentrations, age statistics and an age-resolved distribution."""
    from roger_amd import roger_routine, runtime_settings as rs
    from nc_util import netcdf_file

    g = sb.SasGolden("sas_warmup_a30")
    svat, sas = golden_inputs(g)
    base = make_transport_model("roger_amd", svat, sas, g.ages, g.substeps, g.ndays, bool(g.stats), warmup_days=g.ndays)

    class WithOutput(type(base)):
        @roger_routine
        def set_diagnostics(self, state):
            d = state.diagnostics
            d["collect"].output_variables = ["C_iso_q_ss", "C_iso_rz", "tt_q_ss", "sa_s"]
            d["collect"].output_frequency = 24 * 60 * 60
            d["collect"].sampling_frequency = 1
            d["collect"].base_output_path = str(tmp_path)

    prev = rs.diskless_mode
    object.__setattr__(rs, "diskless_mode", False)
    try:
        model = WithOutput()
        model.setup()
        model.warmup()
        model.run()
    finally:
        object.__setattr__(rs, "diskless_mode", prev)
    ident = model.state.settings.identifier
    with netcdf_file(str(tmp_path / f"{ident}.collect.nc"), "r", mmap=False) as f:
        assert list(f.variables["Time"][:]) == [float(k) for k in range(g.ndays + 1)]
        assert f.variables["tt_q_ss"].dimensions == ("Time", "ages", "y", "x") and f.dimensions["ages"] == g.ages
        C = f.variables["C_iso_q_ss"][:].copy()
        tt = f.variables["tt_q_ss"][:].copy()
        C_rz = f.variables["C_iso_rz"][:].copy()
    for d in range(1, g.ndays + 1):
        want = g.day(d, "C_iso_q_ss").reshape(g.nx, g.ny).T
        assert np.allclose(C[d], want, rtol=5e-3, atol=5e-3, equal_nan=True), d
        assert np.allclose(tt[d], g.day(d, "tt_q_ss").reshape(g.nx, g.ny, g.ages).T, rtol=5e-3, atol=5e-3), d
    assert np.allclose(C_rz[0], g.day(0, "C_iso_rz").reshape(g.nx, g.ny).T, rtol=1e-9, equal_nan=True)   # after the rescaling
    with pytest.raises(NotImplementedError, match="before"):
        class Aged(type(base)):
            @roger_routine
            def set_diagnostics(self, state):
                state.diagnostics["collect"].output_variables = ["sa_rz"]
                state.diagnostics["collect"].output_frequency = 24 * 60 * 60
                state.diagnostics["collect"].sampling_frequency = 1

        Aged().setup()
