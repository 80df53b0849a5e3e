"""Host package on the real device: the setup script of the golden generator run through
roger_amd (RogerSetup.setup/step/run, run_device) matches the reference trajectory."""
import numpy as np
import pytest

from golden_util import compare, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,by_routine", [("svat_uniform_rain", False), ("svat_hetero_combo", False), ("oned_hetero_heavyrain", False),
                                             ("svat_hetero_combo", True), ("oned_hetero_heavyrain", True)])
def test_setup_step_reproduce_reference(case, by_routine, monkeypatch):
    """step() of a script with its own set_forcing: the physics as one native call (rh_svat_step: the model class's set_parameters and
    after_timestep are the fused kernel's); by_routine: the three-call step of a script that also brings one of those two hooks."""
    import svat_scripts as S
    from golden_util import is_lateral

    if by_routine:
        monkeypatch.setenv("RH_STEP_BY_ROUTINE", "1")

    g, names, forcing = load_case(case)
    ndays = len(forcing["PREC"]) // 144
    model = S.make_model(S.params_from_golden(g, names), forcing, ndays, lateral=is_lateral(g))
    model.setup()
    vs = model.state.variables
    compare(S.snapshot_from_vs(vs, names), g["state0"], names, what=f"{case} after setup()")
    nsteps = min(int(g["nsteps"]), 130)
    for step in range(1, nsteps + 1):
        model.step(model.state)
        key = f"s{step:05d}"
        if key in g.files:
            for i, k in enumerate(("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")):
                assert getattr(vs, k) == g["scal"][step - 1][i], (step, k)
            compare(S.snapshot_from_vs(vs, names), g[key], names, what=f"{case} step {step}")


@pytest.mark.parametrize("case", ["svat_stations", "svat_eberbaechle_weights"])
@pytest.mark.parametrize("device_hooks", [False, True])
def test_distributed_input_setups_on_device(case, device_hooks):
    """settings.enable_distributed_input (several stations, vs.station_id) and the per-cell weights of the distributed catchment
    setups through RogerSetup on the device: hooks on the host (per-cell prec_day / ta_day / pet_day handed over every midnight,
    transposed on upload) and hooks on the device (rh_set_forcing_stations / rh_set_forcing_weights) against the reference's runs of
    the same setup scripts.  (tests/test_host_package.py::test_distributed_input_setups is the same on the CPU double.)"""
    import svat_scripts as S
    from golden_util import load_stations, load_weights

    g, names, forcing = load_case(case)
    nx, ny = (int(v) for v in g["nx_ny"])
    w = load_weights(g)
    weights = {k: v.reshape(nx, ny) for k, v in w.items()} if w else None
    stations = None
    if load_stations(g) is not None:
        stations = dict(station_ids=g["station_station_ids"], station_id=g["station_station_id"], PREC=g["station_PREC"],
                        TA=g["station_TA"], PET=g["station_PET"])
    nsteps = int(g["nsteps"])
    model = S.make_model(S.params_from_golden(g, names), forcing, len(forcing["PREC"]) // 144, weights=weights, stations=stations)
    model.setup()
    if device_hooks:
        model.run_device(nsteps)
    else:
        for _ in range(nsteps):
            model.step(model.state)
    vs = model.state.variables
    gs = g["scal"][nsteps - 1]
    assert (int(vs.itt), int(vs.time), int(vs.dt_secs), int(vs.itt_day)) == (gs[0], gs[1], gs[2], gs[3])
    compare(S.snapshot_from_vs(vs, names), g[f"s{nsteps:05d}"], names, what=f"{case} step {nsteps}")


def test_tutorial_year_on_device():
    """BASELINE configs[0]: examples/plot_scale/svat_tutorial (one cell, config.yml parameters, a year of measured forcing
    read by forcing_from_txt in the golden generator): setup() and the whole year through RogerSetup.step with the hooks on
    the host, and once more with the hooks on the device (run_device)."""
    import svat_scripts as S

    g, names, forcing = load_case("svat_tutorial")
    nsteps = int(g["nsteps"])
    model = S.make_model(S.params_from_golden(g, names), forcing, 365)
    model.setup()
    vs = model.state.variables
    compare(S.snapshot_from_vs(vs, names), g["state0"], names, what="tutorial after setup()")
    for step in range(1, nsteps + 1):
        model.step(model.state)
        key = f"s{step:05d}"
        if key in g.files:
            for i, k in enumerate(("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")):
                assert getattr(vs, k) == g["scal"][step - 1][i], (step, k)
            compare(S.snapshot_from_vs(vs, names), g[key], names, what=f"tutorial step {step}")
    assert vs.time == 365 * 86400
    dev = S.make_model(S.params_from_golden(g, names), forcing, 365)
    dev.setup()
    dev.run_device(nsteps)
    assert dev.state.variables.time == 365 * 86400
    compare(S.snapshot_from_vs(dev.state.variables, names), g[f"s{nsteps:05d}"], names, what="tutorial, hooks on the device")


def test_run_device_matches_reference():
    import svat_scripts as S

    g, names, forcing = load_case("svat_hetero_combo")
    model = S.make_model(S.params_from_golden(g, names), forcing, 24)
    model.setup()
    nsteps = int(g["nsteps"])
    model.run_device(nsteps)
    vs = model.state.variables
    assert vs.itt == nsteps and vs.time == g["scal"][nsteps - 1][1]
    compare(S.snapshot_from_vs(vs, names), g[f"s{nsteps:05d}"], names, what="run_device final state")


def test_output_diagnostics_on_device(tmp_path):
    """state.diagnostics (rate / collect / average, daily) on the real context: once with the hooks on the host (a record
    per completed day while stepping), once with run_device (all days resident on the device, flushed at the end) -- the
    same files; the daily sums equal the per-step values added up on the host."""
    from roger_amd import diagnostics, runtime_settings as rs
    from test_host_package import _diagnostics_model, check_diagnostics_files

    ndays = 3
    prev = rs.diskless_mode
    object.__setattr__(rs, "diskless_mode", False)
    try:
        (tmp_path / "a").mkdir()
        (tmp_path / "b").mkdir()
        g, names, model = _diagnostics_model("svat_hetero_combo", ndays, tmp_path / "a")
        model.setup()
        vs = model.state.variables
        acc, sums = 0.0, []
        while vs.time < ndays * 86400:
            model.step(model.state)
            acc = acc + np.asarray(vs.aet)[2:-2, 2:-2].T
            if vs.time % 86400 == 0:
                sums.append(acc)
                acc = 0.0
        nsteps = int(vs.itt)
        diagnostics.close(model.state)
        g, names, dev = _diagnostics_model("svat_hetero_combo", ndays, tmp_path / "b")
        dev.setup()
        dev.run_device(nsteps)
    finally:
        object.__setattr__(rs, "diskless_mode", prev)
    nx, ny = (int(v) for v in g["nx_ny"])
    ra, ca, ta = check_diagnostics_files(model, tmp_path / "a", ndays, nx, ny)
    rb, cb, tb = check_diagnostics_files(dev, tmp_path / "b", ndays, nx, ny)
    np.testing.assert_allclose(ra["aet"][1:], np.stack(sums), rtol=1e-13, atol=1e-13)
    for k in ra:
        np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)
    for k in ca:
        np.testing.assert_array_equal(ca[k], cb[k], err_msg=k)
    np.testing.assert_array_equal(ta, tb)


def test_tutorial_example_from_text_inputs(tmp_path):
    """examples/svat_tutorial.py end to end on the first 30 days of the tutorial's measured inputs (tests/golden/
    tutorial_input): text files -> device -> <identifier>.rate.nc / .collect.nc.  The files close the water balance of the
    soil column day by day, the precipitation record equals the daily sums of PREC.txt, and the final storages equal the
    reference's trajectory (golden svat_tutorial) at the same time."""
    import importlib.util
    import os

    from golden_util import GOLDEN_DIR
    from roger_amd import runtime_settings as rs
    from nc_util import netcdf_file

    spec = importlib.util.spec_from_file_location("svat_tutorial_example", os.path.join(os.path.dirname(GOLDEN_DIR), "..", "examples", "svat_tutorial.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    prev = rs.diskless_mode
    object.__setattr__(rs, "diskless_mode", False)
    try:
        model = ex.main([os.path.join(GOLDEN_DIR, "tutorial_input"), "--days", "30", "--out", str(tmp_path)])
    finally:
        object.__setattr__(rs, "diskless_mode", prev)
    with netcdf_file(str(tmp_path / "SVAT.rate.nc"), "r", mmap=False) as f:
        rate = {k: f.variables[k][:, 0, 0].copy() for k in ex.CONFIG["OUTPUT_RATE"]}
        assert f.variables["Time"].shape == (31,)
    with netcdf_file(str(tmp_path / "SVAT.collect.nc"), "r", mmap=False) as f:
        S = f.variables["S"][:, 0, 0].copy()
    g, names, forcing = load_case("svat_tutorial")
    np.testing.assert_allclose(rate["prec"][1:], forcing["PREC"][: 30 * 144].reshape(30, 144).sum(axis=1), rtol=1e-12, atol=1e-12)
    # water balance of the whole column: dS = prec - aet - q_ss - surface runoff (none leaves a flat, unsealed cell here)
    dS = np.diff(S)
    np.testing.assert_allclose(dS, (rate["prec"] - rate["aet"] - rate["q_ss"] - rate["q_hof"] - rate["q_sof"])[1:], atol=1e-9)
    step = int(np.searchsorted(g["scal"][:, 1], 30 * 86400)) + 1        # the reference's step that ends day 30
    assert g["scal"][step - 1, 1] == 30 * 86400 and model.state.variables.itt == step


def test_output_diagnostics_hourly_on_device(tmp_path):
    """rh_diag_set_interval(3600) on the real context, hooks on the device: the same records as stepping with the hooks on
    the host (one per hour while stepping hourly or finer, one per daily step)."""
    import svat_scripts as S
    from roger_amd import diagnostics, roger_routine, runtime_settings as rs
    from nc_util import netcdf_file

    g, names, forcing = load_case("svat_hetero_combo")

    def build(out):
        base = S.make_model(S.params_from_golden(g, names), forcing, 3)

        class Hourly(type(base)):
            @roger_routine
            def set_diagnostics(self, state):
                d = state.diagnostics["rate"]
                d.output_variables, d.output_frequency, d.sampling_frequency, d.base_output_path = ["prec", "aet", "q_ss"], 3600, 1, str(out)

        return Hourly()

    prev = rs.diskless_mode
    object.__setattr__(rs, "diskless_mode", False)
    try:
        (tmp_path / "a").mkdir()
        (tmp_path / "b").mkdir()
        host = build(tmp_path / "a")
        host.setup()
        vs = host.state.variables
        while vs.time < 3 * 86400:
            host.step(host.state)
        diagnostics.close(host.state)
        dev = build(tmp_path / "b")
        dev.setup()
        dev.run_device(int(vs.itt))
    finally:
        object.__setattr__(rs, "diskless_mode", prev)
    files = [netcdf_file(str(tmp_path / d / f"{host.state.settings.identifier}.rate.nc"), "r", mmap=False) for d in ("a", "b")]
    try:
        ta, tb = (f.variables["Time"][:].copy() for f in files)
        np.testing.assert_array_equal(ta, tb)
        assert len(ta) > 10 and np.isclose(np.diff(ta) * 24, 1).any()
        for k in ("prec", "aet", "q_ss"):
            np.testing.assert_array_equal(files[0].variables[k][:], files[1].variables[k][:], err_msg=k)
    finally:
        for f in files:
            f.close()


def test_routed_setup_on_device():
    """settings.enable_routing_1D through the host package on the device: `step()` (hooks on the host, rh_adaptive_dt + rh_step_core,
    which runs routine by routine with the two gathers for a routing context) and `run_device()` against the reference's routed run."""
    import svat_scripts as S
    from golden_util import routing_of

    g, names, forcing = load_case("oned_routing")
    ndays = len(forcing["PREC"]) // 144
    nsteps = int(g["nsteps"])
    model = S.make_model(S.params_from_golden(g, names), forcing, ndays, lateral=True, routing=routing_of(g, names))
    model.setup()
    vs = model.state.variables
    compare(S.snapshot_from_vs(vs, names), g["state0"], names, what="after setup()")
    for step in range(1, nsteps + 1):
        model.step(model.state)
        key = f"s{step:05d}"
        if key in g.files:
            for i, k in enumerate(("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")):
                assert getattr(vs, k) == g["scal"][step - 1][i], (step, k)
            compare(S.snapshot_from_vs(vs, names), g[key], names, what=f"step {step}")
    fast = S.make_model(S.params_from_golden(g, names), forcing, ndays, lateral=True, routing=routing_of(g, names))
    fast.setup()
    fast.run_device(nsteps)
    compare(S.snapshot_from_vs(fast.state.variables, names), g[f"s{nsteps:05d}"], names, what="run_device")


def test_routing_example_from_text_inputs(tmp_path):
    """examples/oned_routing_tutorial.py end to end on the first 25 days of the example's measured inputs (the Eberbaechle fixture: the
    reference ships the same three files with both setups): text files -> device hooks with the example's station weights -> routed steps
    -> ONED.rate.nc / ONED.collect.nc; the final state equals the reference's run of the same hillslope (golden oned_routing_tutorial),
    and the routed fluxes of the files chain down the slope (what leaves a cell arrives in the next)."""
    import importlib.util
    import os

    import svat_scripts as S
    from golden_util import GOLDEN_DIR
    from roger_amd import runtime_settings as rs
    from nc_util import netcdf_file

    spec = importlib.util.spec_from_file_location("oned_routing_example", os.path.join(os.path.dirname(GOLDEN_DIR), "..", "examples", "oned_routing_tutorial.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    prev = rs.diskless_mode
    object.__setattr__(rs, "diskless_mode", False)
    try:
        model = ex.main([os.path.join(GOLDEN_DIR, "eberbaechle_input"), "--days", "25", "--out", str(tmp_path)])
    finally:
        object.__setattr__(rs, "diskless_mode", prev)
    g, names, _ = load_case("oned_routing_tutorial")
    nsteps = int(g["nsteps"])
    vs = model.state.variables
    assert vs.itt == nsteps and vs.time == 25 * 86400
    compare(S.snapshot_from_vs(vs, names), g[f"s{nsteps:05d}"], names, what="routing example, day 25")
    with netcdf_file(str(tmp_path / "ONED.rate.nc"), "r", mmap=False) as f:
        out, inn = f.variables["q_sub_out"][:, :, 0].copy(), f.variables["q_sub_in"][:, :, 0].copy()   # (Time, y, x)
        assert f.variables["Time"].shape == (26,)
    assert out[1:].sum() > 0
    np.testing.assert_allclose(inn[1:, 1:], out[1:, :-1], rtol=1e-9, atol=1e-12)   # every cell drains into the next one along y


@pytest.mark.parametrize("case", ["svat_hetero_combo", "oned_hetero_heavyrain"])
def test_run_with_stock_hooks_on_the_device(case, tmp_path):
    """`RogerSetup.run()` of a setup that leaves the per-step hooks to the model class (SVATSetup / ONEDSetup) advances through
    rh_run_steps under rh_set_time_limit -- a few rounds, no host code between steps -- and ends in the reference's state at the
    end of the run with the reference's step count (tests/test_host_package.py runs the same against the hook-preserving loop)."""
    import svat_scripts as S
    from test_host_package import _stock_model

    ndays = min(12, len(load_case(case)[2]["PREC"]) // 144)
    g, names, m = _stock_model(case, ndays)
    m.setup()
    assert m.device_run_possible()
    rounds = []
    inner = m.run_device
    m.run_device = lambda n, final=True: (rounds.append(n), inner(n, final=final))[1]
    m.run()
    vs = m.state.variables
    nsteps = int(np.sum(g["scal"][:, 1] <= ndays * 86400))
    assert (int(vs.itt), int(vs.time)) == (nsteps, ndays * 86400) and len(rounds) <= 6
    key = f"s{nsteps:05d}"
    if key in g.files:
        compare(S.snapshot_from_vs(vs, names), g[key], names, what=f"{case}: run() on the device, step {nsteps}")
    gs = g["scal"][nsteps - 1]
    assert (int(vs.dt_secs), int(vs.itt_day), int(vs.event_id_counter)) == (gs[2], gs[3], gs[6])


@pytest.mark.parametrize("case", ["svat_hetero_combo", "oned_hetero_heavyrain"])
def test_reference_shaped_script_runs_on_the_device(case):
    """Every script the reference ships defines set_forcing, set_parameters and after_timestep itself (benchmarks/SVAT_benchmark.py:
    105-110, 152-181); a script of that shape is recognised hook by hook (roger_amd/hooks.py) and its run() advances on the device."""
    import svat_scripts as S
    from test_host_package import run_script_case

    model, g, names = run_script_case(case, "plain")
    rounds = []
    inner = model.run_device
    model.run_device = lambda n, final=True: (rounds.append(n), inner(n, final=final))[1]
    model.run()
    assert rounds, "run() did not take the device path"
    vs = model.state.variables
    nsteps = int(g["nsteps"])
    assert (int(vs.itt), int(vs.time)) == (nsteps, int(g["scal"][nsteps - 1][1]))
    compare(S.snapshot_from_vs(vs, names), g[f"s{nsteps:05d}"], names, what=f"{case}: reference-shaped script, run() on the device")


@pytest.mark.parametrize("script_hooks", ["set_parameters_assigns", "after_timestep_assigns", "forcing_scaled", "forcing_counts"])
def test_a_hook_that_does_something_of_its_own_stays_on_the_host(script_hooks):
    """A hook that assigns a variable, changes the forcing or keeps a counter of its own is the host's: the reference's loop -- through
    the lean loop (one native call per step that also returns the scalars, rh_svat_step_scalars) when set_parameters and
    after_timestep are the device's, the three-call step otherwise."""
    import svat_scripts as S
    from test_host_package import SCRIPT_CASES, run_script_case

    model, g, names = run_script_case("svat_hetero_combo", script_hooks)
    assert model._lean_host_loop_possible() is SCRIPT_CASES[script_hooks][2]
    model.run_device = None
    model.run()
    vs = model.state.variables
    nsteps = int(g["nsteps"])
    if script_hooks == "forcing_scaled":
        assert int(vs.time) == int(g["scal"][nsteps - 1][1])
        return
    assert (int(vs.itt), int(vs.time)) == (nsteps, int(g["scal"][nsteps - 1][1]))
    compare(S.snapshot_from_vs(vs, names), g[f"s{nsteps:05d}"], names, what=f"{script_hooks}: hooks on the host")
    if script_hooks == "forcing_counts":
        assert type(model).forcing_calls == [int(t) for t in [0] + list(g["scal"][: nsteps - 1, 1])]


def test_lean_loop_equals_step_by_step(monkeypatch):
    """run() through the lean loop (hooks as plain functions, rh_svat_step_scalars) and run() through step() (RH_NO_LEAN_LOOP=1) end in
    the same state bit for bit, with the daily output of both written alike."""
    import svat_scripts as S
    from test_host_package import run_script_case

    snaps = []
    for lean in (True, False):
        if not lean:
            monkeypatch.setenv("RH_NO_LEAN_LOOP", "1")
        model, g, names = run_script_case("svat_hetero_combo", "forcing_counts")
        type(model).forcing_calls.clear()
        assert model._lean_host_loop_possible() is lean
        model.run()
        vs = model.state.variables
        snaps.append((int(vs.itt), int(vs.time), S.snapshot_from_vs(vs, names)))
    assert snaps[0][:2] == snaps[1][:2]
    np.testing.assert_array_equal(snaps[0][2], snaps[1][2])
