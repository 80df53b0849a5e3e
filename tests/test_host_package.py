"""Host package (operator surface mirroring the reference) on CPU, with the oracle standing in
for the device through the OracleContext test double."""
import numpy as np
import pytest

from golden_util import compare, load_case


@pytest.fixture
def oracle_backend(monkeypatch, oracle):
    import oracle_context
    from roger_amd import _native

    monkeypatch.setattr(_native, "Context", oracle_context.OracleContext)
    monkeypatch.setattr(_native, "plane_table", lambda: list(zip(oracle.plane_names(), oracle.plane_is_int())))
    return oracle_context


def test_runtime_settings():
    from roger_amd.runtime import RuntimeSettings

    rs = RuntimeSettings()
    assert rs.backend == "hip" and rs.float_type == "float64"
    rs.update(num_proc=(2, 1), profile_mode="true")
    assert rs.num_proc == (2, 1) and rs.profile_mode is True
    with pytest.raises(ValueError):
        rs.update(backend="numpy")          # roger/runtime.py:30-40 parse_choice
    with pytest.raises(AttributeError):
        rs.update(no_such_setting=1)
    rs.__locked__ = True
    with pytest.raises(RuntimeError):
        rs.update(device="gpu")             # locked after core import, roger/core/__init__.py:42-44


def test_operators_are_value_semantics():
    from roger_amd.core.operators import at, update, update_add, update_multiply
    from roger_amd.variables import allocate

    a = allocate({"x": 2, "y": 3}, ("x", "y"))
    assert a.shape == (6, 7) and not a.flags.writeable   # +4 ghosts, immutable (variables.py:6431-6433)
    b = update(a, at[2:-2, 2:-2], 5.0)
    assert a.sum() == 0 and b[2:-2, 2:-2].sum() == 30 and not b.flags.writeable
    c = update_add(b, at[2, 2], 1.0)
    d = update_multiply(c, at[2:-2, 2:-2], 2.0)
    assert c[2, 2] == 6 and d[2, 2] == 12 and b[2, 2] == 5


def test_variables_contract(oracle_backend):
    """Shape/dtype validation and locking as roger/state.py:240-251,112-140."""
    from roger_amd.state import RogerState

    st = RogerState()
    with st.settings.unlock():
        st.settings.nx, st.settings.ny = 3, 2
    with pytest.raises(RuntimeError):
        st.settings.nx = 5                  # locked
    with pytest.raises(RuntimeError):
        st.variables                        # not initialised yet
    st.initialize_variables()
    vs = st.variables
    assert vs.z_soil.shape == (7, 6) and vs.theta_rz.shape == (7, 6, 2) and vs.prec_day.shape == (7, 6, 144)
    assert vs.ta[2, 2, 1] == 15 and vs.z_gw[3, 3, 0] == 1000 and bool(vs.maskCatch[2, 2])
    assert vs.tau == 1 and vs.taum1 == 0 and vs.dt == 1 and vs.dt_secs == 3600 and list(vs.month) == [1, 1]
    with pytest.raises(RuntimeError):
        vs.z_soil = np.zeros((7, 6))        # locked outside routines
    with vs.unlock():
        with pytest.raises(ValueError):
            vs.z_soil = np.zeros((3, 2))    # wrong shape
        with pytest.raises(AttributeError):
            vs.not_a_variable = 1
        vs.z_soil = np.full((7, 6), 900.0)
        vs.time = 86400
    assert vs.time == 86400
    vs.flush_to_device()
    assert np.all(st.backend_context.download("z_soil") == 900.0)
    assert st.backend_context.get_scalars().time == 86400


@pytest.mark.parametrize("case,by_routine", [("svat_uniform_rain", False), ("svat_hetero_snowrain", False), ("oned_hetero_heavyrain", False),
                                             ("svat_tutorial", False), ("svat_hetero_snowrain", True), ("oned_hetero_heavyrain", True)])
def test_setup_and_run_reproduce_reference(oracle_backend, case, by_routine, monkeypatch):
    """The same setup script the golden generator ran through the reference, run through this
    package: setup() state and the trajectory of step() match the reference.  The script brings its own set_forcing and leaves
    set_parameters / after_timestep to the model class: step() then runs the physics as ONE native call (rh_svat_step);
    by_routine: the three-call step (rh_adaptive_dt, rh_step_core, rh_after_timestep) a script with a set_parameters or
    after_timestep hook of its own gets."""
    import svat_scripts as S

    if by_routine:
        monkeypatch.setenv("RH_STEP_BY_ROUTINE", "1")

    g, names, forcing = load_case(case)
    ndays = len(forcing["PREC"]) // 144
    from golden_util import is_lateral

    model = S.make_model(S.params_from_golden(g, names), forcing, ndays, lateral=is_lateral(g))
    model.setup()
    vs = model.state.variables
    compare(S.snapshot_from_vs(vs, names), g["state0"], names, what=f"{case} after setup()")
    nsteps = int(g["nsteps"])
    if case == "svat_tutorial":   # a year of single-cell steps: the first 600 here (25 days, snapshots at 250 and 500);
        nsteps = 600              # the whole year runs in test_oracle_golden (oracle) and test_hip_host_package (device)
    for step in range(1, nsteps + 1):
        model.step(model.state)
        for i, k in enumerate(("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")):
            assert getattr(vs, k) == g["scal"][step - 1][i], (step, k)
        key = f"s{step:05d}"
        if key in g.files:
            compare(S.snapshot_from_vs(vs, names), g[key], names, what=f"{case} step {step}")
    assert vs.time >= model.state.settings.runlen or case == "svat_tutorial"


def test_run_and_run_device_agree(oracle_backend):
    """`run()` (user hooks on the host every step) and `run_device()` (hooks on the device) give
    the same trajectory."""
    import svat_scripts as S

    g, names, forcing = load_case("svat_hetero_combo")
    ndays = 6
    a = S.make_model(S.params_from_golden(g, names), forcing, ndays)
    b = S.make_model(S.params_from_golden(g, names), forcing, ndays)
    a.setup()
    b.setup()
    a.run()
    nsteps = a.state.variables.itt
    b.run_device(nsteps)
    assert b.state.variables.time == a.state.variables.time
    np.testing.assert_array_equal(S.snapshot_from_vs(a.state.variables, names), S.snapshot_from_vs(b.state.variables, names))


def test_scope_errors(oracle_backend):
    """Switches outside the hot path fail loudly instead of silently doing something else."""
    import svat_scripts as S

    g, names, forcing = load_case("svat_uniform_rain")
    model = S.make_model(S.params_from_golden(g, names), forcing, 1)
    model.override_settings = {"enable_routing_1D": True}
    with pytest.raises(NotImplementedError):
        model.setup()
    bad = S.params_from_golden(g, names)
    bad["ks"] = bad["ks"] * -1
    with pytest.raises(ValueError, match="ks-parameter is out of range"):
        S.make_model(bad, forcing, 1).setup()   # numerics.validate_parameters_soil


def test_restart_round_trip(tmp_path, monkeypatch):
    """roger/restart.py on the hip backend: the file the model writes through its settings (restart_output_filename with format
    syntax, restart_frequency; write_restart at the end of run()) is an HDF5 file with the group "core" in the reference's layout;
    a fresh model that names it in restart_input_filename reads it at the end of setup() and continues identically to the
    uninterrupted run."""
    import oracle_context as OC
    from golden_util import load_case
    from roger_amd import _native, h5lite, restart
    from svat_scripts import make_model, params_from_golden

    monkeypatch.setattr(_native, "Context", OC.OracleContext)
    monkeypatch.chdir(tmp_path)
    g, names, forcing = load_case("svat_hetero_combo")
    p = params_from_golden(g, names)
    a = make_model(p, forcing, 2)
    a.override_settings = dict(write_restart=True, restart_output_filename="{identifier}_{itt:0>4d}.restart.h5")
    a.setup()
    a.run()                                   # two days; the restart file is written in run()'s finally clause
    n = int(a.state.variables.itt)
    fname = tmp_path / f"GoldenSVAT_{n:0>4d}.restart.h5"
    assert fname.is_file()
    groups = h5lite.read(fname)
    core = groups["core"]
    nx, ny = (int(v) for v in g["nx_ny"])
    assert core["S_rz"].shape == (nx + 4, ny + 4, 2) and core["lu_id"].shape == (nx + 4, ny + 4) and core["itt"].shape == ()
    assert int(core["itt"]) == n and int(core["time"]) == 2 * 86400
    have = [k for k in restart.REFERENCE_RESTART_VARIABLES if k in a.state.var_meta and k not in restart._LATERAL_ONLY]
    assert len(have) > 120 and all(k in core for k in have)
    # "core" is EXACTLY the reference's list for this model (its reader fails on any other dataset, roger/restart.py:15-16); the
    # rest of the arena sits in a group of its own, and the reference's diagnostic groups exist
    assert set(core) == set(have) | set(restart.REFERENCE_ONLY_RESTART_VARIABLES) and "dmph" not in core
    assert "prec" in groups[restart.MORE_GROUP] and "z_wf_m1" not in core and {"average", "snapshot", "tracer_monitor", "water_monitor"} <= set(groups)
    np.testing.assert_array_equal(core["S_rz"], np.asarray(a.state.variables.S_rz))
    # a fresh model restarts from the file at the end of its setup() ...
    b = make_model(p, forcing, 4)
    b.override_settings = dict(restart_input_filename=str(fname))
    b.setup()
    assert b.state.variables.itt == n and b.state.variables.time == 2 * 86400
    # ... and continues like the uninterrupted run
    c = make_model(p, forcing, 4)
    c.setup()
    c.run()
    with b.state.settings.unlock():
        b.state.settings.runlen = 2 * 86400   # two more days
    b.run()
    assert b.state.variables.itt == c.state.variables.itt
    for nm in ("S", "theta_rz", "z_wf", "q_ss", "swe", "event_id", "time_event0", "itt_forc", "prec", "ta", "z0", "S_dep"):
        np.testing.assert_array_equal(np.asarray(getattr(b.state.variables, nm)), np.asarray(getattr(c.state.variables, nm)), err_msg=nm)
    # a missing file is an error, as in the reference (restart.py:84-85), not a silent cold start
    d = make_model(p, forcing, 1)
    d.override_settings = dict(restart_input_filename=str(tmp_path / "nothing.h5"))
    with pytest.raises(IOError, match="not found"):
        d.setup()


@pytest.mark.parametrize("case", ["svat_stations", "svat_eberbaechle_weights"])
@pytest.mark.parametrize("device_hooks", [False, True])
def test_distributed_input_setups(oracle_backend, case, device_hooks):
    """The distributed catchment setups on the host package: per-cell weights on one station's series
    (eberbaechle/svat_distributed/svat.py:169-186, 276-296) and settings.enable_distributed_input with several stations
    (roger/bmimodels/svat_dist/svat_dist.py:200-211, 261-322) -- the same hooks the golden generator drove through the reference, with
    the hooks on the host (run) and handed to the device (run_device: rh_set_forcing_stations / rh_set_forcing_weights)."""
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
    nsteps = 120 if case == "svat_stations" else 100    # steps that carry a snapshot
    model = S.make_model(S.params_from_golden(g, names), forcing, len(forcing["PREC"]) // 144, weights=weights, stations=stations)
    model.setup()
    compare(S.snapshot_from_vs(model.state.variables, names), g["state0"], names, what=f"{case} after setup")
    if device_hooks:
        model.run_device(nsteps)
    else:
        for _ in range(nsteps):
            model.step(model.state)
    vs = model.state.variables
    gs = g["scal"][nsteps - 1]
    assert (int(vs.itt), int(vs.time), int(vs.dt_secs), int(vs.itt_day)) == (gs[0], gs[1], gs[2], gs[3])
    compare(S.snapshot_from_vs(vs, names), g[f"s{nsteps:05d}"], names, what=f"{case} step {nsteps}")


def test_forcing_from_text_inputs():
    """roger_amd.forcing.forcing_from_txt (read_meteo + write_forcing's recipe, roger/io_tools/csv.py:10-104,
    roger/tools/setup.py:469-620) on the first 30 days of the tutorial's measured inputs (tests/golden/tutorial_input:
    data rows of examples/plot_scale/svat_tutorial/input) against the arrays the reference's own pipeline produced for the
    golden run svat_tutorial."""
    import os

    from golden_util import GOLDEN_DIR
    from roger_amd.forcing import SLOTS_PER_DAY, forcing_from_txt

    g = np.load(os.path.join(GOLDEN_DIR, "svat_tutorial.npz"))
    F = forcing_from_txt(os.path.join(GOLDEN_DIR, "tutorial_input"))
    n = 30 * SLOTS_PER_DAY
    for k in ("PREC", "TA", "PET", "YEAR", "MONTH", "DOY"):
        assert F[k].shape == (n,) and F[k].dtype == g[f"forc_{k}"].dtype, k
        np.testing.assert_array_equal(F[k], g[f"forc_{k}"][:n], err_msg=k)
    assert F["PREC"].sum() > 10 and F["DOY"][0] == 274    # 1 October 2010
    # float64 on request (write_forcing(float_type="float64")): not rounded through float32
    F64 = forcing_from_txt(os.path.join(GOLDEN_DIR, "tutorial_input"), float_type="float64", ndays=2)
    assert F64["PREC"].shape == (2 * SLOTS_PER_DAY,) and F64["PREC"][0] == 0.42 and F["PREC"][0] != 0.42


def test_forcing_from_text_inputs_eberbaechle():
    """The same reader on BASELINE configs[4]'s measured station series (tests/golden/eberbaechle_input: the first 40 days of
    examples/catchment_scale/eberbaechle/svat_distributed/input, 2019-11-01 .. 2019-12-10) against the arrays the golden run
    svat_eberbaechle_weights was driven with."""
    import os

    from golden_util import GOLDEN_DIR
    from roger_amd.forcing import SLOTS_PER_DAY, forcing_from_txt

    g = np.load(os.path.join(GOLDEN_DIR, "svat_eberbaechle_weights.npz"))
    F = forcing_from_txt(os.path.join(GOLDEN_DIR, "eberbaechle_input"))
    for k in ("PREC", "TA", "PET", "YEAR", "MONTH", "DOY"):
        assert F[k].shape == (40 * SLOTS_PER_DAY,), k
        np.testing.assert_array_equal(F[k], g[f"forc_{k}"], err_msg=k)
    assert F["DOY"][0] == 305 and F["MONTH"][-1] == 12


def _diagnostics_model(case, ndays, tmp_path, device_hooks=False):
    import svat_scripts as S
    from roger_amd import roger_routine

    g, names, forcing = load_case(case)
    base = S.make_model(S.params_from_golden(g, names), forcing, ndays)

    class WithOutput(type(base)):
        @roger_routine
        def set_diagnostics(self, state):   # as the reference's setup scripts do (examples/plot_scale/svat_tutorial/svat.py)
            d = state.diagnostics
            d["rate"].output_variables = ["prec", "aet", "q_ss", "inf_mat_rz"]
            d["rate"].output_frequency = 24 * 60 * 60
            d["rate"].sampling_frequency = 1
            d["rate"].base_output_path = str(tmp_path)
            d["collect"].output_variables = ["S_rz", "S_ss", "theta"]
            d["collect"].output_frequency = 24 * 60 * 60
            d["collect"].sampling_frequency = 1
            d["collect"].base_output_path = str(tmp_path)
            d["average"].output_variables = ["ta"]
            d["average"].output_frequency = 24 * 60 * 60
            d["average"].sampling_frequency = 1
            d["average"].base_output_path = str(tmp_path)

    model = WithOutput()
    return g, names, model


def check_diagnostics_files(model, tmp_path, ndays, nx, ny, reference_sums=None):
    from nc_util import netcdf_file

    ident = model.state.settings.identifier
    with netcdf_file(str(tmp_path / f"{ident}.rate.nc"), "r", mmap=False) as f:
        assert f.dimensions["x"] == nx and f.dimensions["y"] == ny and f.dimensions["Time"] is None
        t = f.variables["Time"]
        assert t.units == b"days" and list(t[:]) == [float(k) for k in range(ndays + 1)]
        prec = f.variables["prec"]
        assert prec.dimensions == ("Time", "y", "x") and prec.shape == (ndays + 1, ny, nx) and prec.units == b"mm/dt"
        rate = {k: f.variables[k][:].copy() for k in ("prec", "aet", "q_ss", "inf_mat_rz")}
    with netcdf_file(str(tmp_path / f"{ident}.collect.nc"), "r", mmap=False) as f:
        collect = {k: f.variables[k][:].copy() for k in ("S_rz", "S_ss", "theta")}
    with netcdf_file(str(tmp_path / f"{ident}.average.nc"), "r", mmap=False) as f:
        ta = f.variables["ta"][:].copy()
    assert (rate["prec"][0] == 0).all() and rate["prec"][1:].sum() > 0          # record 0: initial values
    assert np.isfinite(ta).all() and collect["S_rz"].shape == (ndays + 1, ny, nx)
    return rate, collect, ta


def test_output_diagnostics_daily_files(oracle_backend, tmp_path):
    """`state.diagnostics` as in the reference's setup scripts: rate / collect / average with daily output become
    <identifier>.rate.nc ... in the reference's layout (x, y, unlimited Time; (Time, y, x) variables; record 0 = initial
    values).  The rate file's daily sums equal the sums of the per-step values of a run without diagnostics."""
    from roger_amd import runtime_settings as rs

    ndays = 4
    g, names, model = _diagnostics_model("svat_hetero_combo", ndays, tmp_path)
    prev = rs.diskless_mode
    object.__setattr__(rs, "diskless_mode", False)   # (runtime settings are locked once the core modules are imported)
    try:
        model.setup()
        vs = model.state.variables
        sums = {k: [] for k in ("prec", "aet", "q_ss")}
        day_acc = {k: 0.0 for k in sums}
        last = {}
        while vs.time < ndays * 86400:
            model.step(model.state)
            for k in sums:
                day_acc[k] = day_acc[k] + np.asarray(getattr(vs, k))[2:-2, 2:-2][..., 1].T if np.asarray(getattr(vs, k)).ndim == 3 \
                    else day_acc[k] + np.asarray(getattr(vs, k))[2:-2, 2:-2].T
            if vs.time % 86400 == 0:
                for k in sums:
                    sums[k].append(day_acc[k])
                    day_acc[k] = 0.0
                last = {k: np.asarray(getattr(vs, k))[2:-2, 2:-2, 1].T.copy() for k in ("S_rz", "S_ss", "theta")}
        from roger_amd import diagnostics

        diagnostics.close(model.state)
    finally:
        object.__setattr__(rs, "diskless_mode", prev)
    nx, ny = (int(v) for v in g["nx_ny"])
    rate, collect, ta = check_diagnostics_files(model, tmp_path, ndays, nx, ny)
    for k in sums:
        np.testing.assert_allclose(rate[k][1:], np.stack(sums[k]), rtol=1e-13, atol=1e-13, err_msg=k)
    for k in last:
        np.testing.assert_array_equal(collect[k][-1], last[k], err_msg=k)
    with pytest.raises(IOError):     # the files exist now: a second setup refuses to overwrite them
        object.__setattr__(rs, "diskless_mode", False)
        try:
            _diagnostics_model("svat_hetero_combo", ndays, tmp_path)[2].setup()
        finally:
            object.__setattr__(rs, "diskless_mode", prev)


def test_output_diagnostics_scope(oracle_backend, tmp_path):
    import svat_scripts as S
    from roger_amd import roger_routine

    g, names, forcing = load_case("svat_uniform_rain")
    base = S.make_model(S.params_from_golden(g, names), forcing, 1)

    class TwoFrequencies(type(base)):
        @roger_routine
        def set_diagnostics(self, state):
            state.diagnostics["rate"].output_variables = ["prec"]
            state.diagnostics["rate"].output_frequency = 60 * 60
            state.diagnostics["rate"].sampling_frequency = 1
            state.diagnostics["collect"].output_variables = ["S_rz"]
            state.diagnostics["collect"].output_frequency = 24 * 60 * 60
            state.diagnostics["collect"].sampling_frequency = 1

    with pytest.raises(NotImplementedError, match="ONE output interval"):
        TwoFrequencies().setup()

    class Maximum(type(base)):
        @roger_routine
        def set_diagnostics(self, state):
            state.diagnostics["maximum"].output_variables = ["prec"]
            state.diagnostics["maximum"].output_frequency = 24 * 60 * 60

    with pytest.raises(NotImplementedError, match="native: rate, collect, average"):
        Maximum().setup()


def test_output_diagnostics_hourly(oracle_backend, tmp_path):
    """output_frequency = 3600: a record whenever a step ends on the hour (roger/diagnostics/api.py:56-58) -- one per hour
    while the model steps hourly or in ten-minute steps, ONE at the end of a daily step.  The records add up to the daily
    file's sums and carry the reference's time stamps."""
    import svat_scripts as S
    from roger_amd import diagnostics, roger_routine, runtime_settings as rs
    from nc_util import netcdf_file

    g, names, forcing = load_case("svat_hetero_combo")
    base = S.make_model(S.params_from_golden(g, names), forcing, 4)

    class Hourly(type(base)):
        @roger_routine
        def set_diagnostics(self, state):
            d = state.diagnostics["rate"]
            d.output_variables, d.output_frequency, d.sampling_frequency, d.base_output_path = ["prec", "aet"], 60 * 60, 1, str(tmp_path)

    prev = rs.diskless_mode
    object.__setattr__(rs, "diskless_mode", False)
    try:
        model = Hourly()
        model.setup()
        vs = model.state.variables
        want_t, want_prec, acc = [0.0], [], 0.0
        while vs.time < 4 * 86400:
            model.step(model.state)
            acc = acc + np.asarray(vs.prec)[2:-2, 2:-2, 1].T
            if vs.time % 3600 == 0:
                want_t.append(vs.time / 86400)
                want_prec.append(acc)
                acc = 0.0
        diagnostics.close(model.state)
    finally:
        object.__setattr__(rs, "diskless_mode", prev)
    with netcdf_file(str(tmp_path / f"{model.state.settings.identifier}.rate.nc"), "r", mmap=False) as f:
        t, prec = f.variables["Time"][:].copy(), f.variables["prec"][:].copy()
    np.testing.assert_allclose(t, want_t, rtol=0, atol=1e-12)
    np.testing.assert_allclose(prec[1:], np.stack(want_prec), rtol=1e-13, atol=1e-13)
    steps = np.diff(t) * 24
    assert np.isclose(steps, 1).any() and np.isclose(steps, 24).any()     # hourly records and whole-day records


def test_routed_setup_reproduces_reference(oracle_backend):
    """settings.enable_routing_1D through the host package: the routed hillslope the golden generator ran through the reference
    (tests/golden/oned_routing.npz), same setup script, `step()` with the user hooks on the host and `run_device()`."""
    import svat_scripts as S
    from golden_util import routing_of

    g, names, forcing = load_case("oned_routing")
    ndays = len(forcing["PREC"]) // 144
    model = S.make_model(S.params_from_golden(g, names), forcing, ndays, lateral=True, routing=routing_of(g, names))
    model.setup()
    vs = model.state.variables
    compare(S.snapshot_from_vs(vs, names), g["state0"], names, what="after setup()")
    nsteps = int(g["nsteps"])
    for step in range(1, nsteps + 1):
        model.step(model.state)
        for i, k in enumerate(("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")):
            assert getattr(vs, k) == g["scal"][step - 1][i], (step, k)
        key = f"s{step:05d}"
        if key in g.files:
            compare(S.snapshot_from_vs(vs, names), g[key], names, what=f"step {step}")
    fast = S.make_model(S.params_from_golden(g, names), forcing, ndays, lateral=True, routing=routing_of(g, names))
    fast.setup()
    fast.run_device(nsteps)
    compare(S.snapshot_from_vs(fast.state.variables, names), g[f"s{nsteps:05d}"], names, what="run_device")


def _stock_model(case, ndays, tmp_path=None, override_hook=False):
    """The ready-made model class with its stock per-step hooks (SVATSetup / ONEDSetup), parameters and forcing of a golden case."""
    import svat_scripts as S
    from golden_util import is_lateral
    from roger_amd.models.oned import ONEDSetup
    from roger_amd.models.svat import SVATSetup
    from roger_amd.routines import roger_routine

    g, names, forcing = load_case(case)
    nx, ny = (int(v) for v in g["nx_ny"])
    p = {k: np.asarray(v).reshape(nx, ny) for k, v in S.params_from_golden(g, names).items()}
    theta = {"theta_rz": p.pop("theta_rz0")[:, :, None], "theta_ss": p.pop("theta_ss0")[:, :, None]}   # both time levels
    base = ONEDSetup if is_lateral(g) else SVATSetup

    class Model(base):
        initial_theta = theta

        @roger_routine
        def set_diagnostics(self, state):
            if tmp_path is not None:
                d = state.diagnostics
                d["rate"].output_variables = ["prec", "aet", "q_ss"]
                d["collect"].output_variables = ["S_rz", "theta"]
                for k in ("rate", "collect"):
                    d[k].output_frequency, d[k].sampling_frequency, d[k].base_output_path = 86400, 1, str(tmp_path)

    if override_hook:   # a setup script with a per-step hook of its own: run() must keep calling it
        calls = []

        class Model(Model):   # noqa: F811
            @roger_routine
            def read_data(self, state):
                calls.append(int(state.variables.itt))

        Model.calls = calls
    m = Model(forcing=forcing, nx=nx, ny=ny, ndays=ndays, parameters=p)
    return g, names, m


@pytest.mark.parametrize("case", ["svat_hetero_combo", "oned_hetero_heavyrain"])
def test_run_with_stock_hooks_stays_on_the_device(oracle_backend, case, tmp_path):
    """`RogerSetup.run()` (roger/roger.py:523-580) of a setup script that leaves the per-step hooks to the model class advances
    through rh_run_steps in rounds of ceil(remaining / day) steps -- no host code between two steps -- and ends in the same state,
    step count and output files as the hook-preserving loop, which a script with a hook of its own still gets."""
    import svat_scripts as S
    from roger_amd import runtime_settings as rs

    ndays = min(6, len(load_case(case)[2]["PREC"]) // 144)
    prev = rs.diskless_mode
    object.__setattr__(rs, "diskless_mode", False)
    try:
        out = {}
        for tag, override in (("device", False), ("hooks", True)):
            d = tmp_path / tag
            d.mkdir()
            g, names, m = _stock_model(case, ndays, d, override_hook=override)
            m.setup()
            assert m.device_run_possible() is (not override)
            rounds = []
            if not override:
                inner = m.run_device
                m.run_device = lambda n, final=True: (rounds.append(n), inner(n, final=final))[1]
            m.run()
            vs = m.state.variables
            out[tag] = (int(vs.itt), int(vs.time), S.snapshot_from_vs(vs, names), d, rounds, getattr(type(m), "calls", None))
    finally:
        object.__setattr__(rs, "diskless_mode", prev)
    a, b = out["device"], out["hooks"]
    assert a[0] == b[0] and a[1] == b[1] == ndays * 86400
    assert a[4][0] == ndays and len(a[4]) <= 8 and sum(a[4]) >= a[0]          # a few generous rounds (the device stops at the end of the run)
    assert b[5] == list(range(a[0]))                                          # the script's own hook ran before every step
    np.testing.assert_array_equal(a[2], b[2])
    from nc_util import netcdf_file

    for kind, keys in (("rate", ("prec", "aet", "q_ss")), ("collect", ("S_rz", "theta"))):
        ident = m.state.settings.identifier
        with netcdf_file(str(a[3] / f"{ident}.{kind}.nc"), "r", mmap=False) as fa, netcdf_file(str(b[3] / f"{ident}.{kind}.nc"), "r", mmap=False) as fb:
            assert fa.variables["Time"].shape == (ndays + 1,)
            np.testing.assert_array_equal(fa.variables["Time"][:], fb.variables["Time"][:])
            for k in keys:
                np.testing.assert_allclose(fa.variables[k][:], fb.variables[k][:], rtol=1e-13, atol=1e-13, err_msg=f"{kind} {k}")


# ---------------------------------------------------------------------------------------------------------------------------------
# reference-shaped scripts: every script the reference ships defines set_forcing, set_parameters and after_timestep ITSELF
# (benchmarks/SVAT_benchmark.py:105-110, 152-181).  Which of them the device performs is decided by behaviour (roger_amd/hooks.py).
# ---------------------------------------------------------------------------------------------------------------------------------
SCRIPT_CASES = {
    # script_hooks: (hooks the device performs, run() stays on the device, the step is one native call)
    "plain": (dict(read_data=True, set_boundary_conditions=True, set_forcing=True, set_parameters=True, after_timestep=True), True, True),
    "set_parameters_assigns": (dict(set_forcing=True, set_parameters=False, after_timestep=True), False, False),
    "after_timestep_assigns": (dict(set_forcing=True, set_parameters=True, after_timestep=False), False, False),
    "forcing_scaled": (dict(set_forcing=False, set_parameters=True, after_timestep=True), False, True),
    "forcing_counts": (dict(set_forcing=False, set_parameters=True, after_timestep=True), False, True),
}


def run_script_case(case, script_hooks, ndays=None, check_golden=True):
    """run() of a reference-shaped script; returns (model, golden, names).  Shared with tests/test_hip_host_package.py."""
    import svat_scripts as S
    from golden_util import is_lateral

    g, names, forcing = load_case(case)
    ndays = ndays or len(forcing["PREC"]) // 144
    model = S.make_model(S.params_from_golden(g, names), forcing, ndays, lateral=is_lateral(g), script_hooks=script_hooks)
    model.setup()
    want, on_device, one_call = SCRIPT_CASES[script_hooks]
    classes = model.hook_classes()
    for hook, expected in want.items():
        assert classes[hook] is expected, (script_hooks, hook, classes)
    assert model.device_run_possible() is on_device
    assert model._fused_host_step_possible() is one_call
    return model, g, names


@pytest.mark.parametrize("case", ["svat_hetero_combo", "oned_hetero_heavyrain"])
def test_reference_shaped_script_runs_on_the_device(oracle_backend, case):
    """Own set_forcing slicing the series at midnight, own set_parameters with the month test, own after_timestep calling a
    script-local after_timestep_kernel: recognised hook by hook, run() advances on the device and ends where the reference ends."""
    model, g, names = run_script_case(case, "plain")
    rounds = []
    inner = model.run_device
    model.run_device = lambda n, final=True: (rounds.append(n), inner(n, final=final))[1]
    model.run()
    assert rounds, "run() did not take the device path"
    vs = model.state.variables
    nsteps = int(g["nsteps"])
    assert (int(vs.itt), int(vs.time)) == (nsteps, int(g["scal"][nsteps - 1][1]))
    compare(S_snapshot(vs, names), g[f"s{nsteps:05d}"], names, what=f"{case}: reference-shaped script, run() on the device")


def S_snapshot(vs, names):
    import svat_scripts as S

    return S.snapshot_from_vs(vs, names)


@pytest.mark.parametrize("script_hooks", ["set_parameters_assigns", "after_timestep_assigns", "forcing_scaled", "forcing_counts"])
def test_a_hook_that_does_something_of_its_own_stays_on_the_host(oracle_backend, script_hooks):
    """A hook that assigns a variable itself, changes the forcing or keeps a counter is NOT the device's: run() keeps the reference's
    loop (the three-call step when set_parameters / after_timestep are the script's, one native call otherwise) and -- where the
    deviation leaves the values alone -- still ends in the golden state; a counting hook is called exactly once per step."""
    model, g, names = run_script_case("svat_hetero_combo", script_hooks)
    model.run_device = None   # (run() must not take it)
    model.run()
    vs = model.state.variables
    nsteps = int(g["nsteps"])
    if script_hooks == "forcing_scaled":
        assert int(vs.time) == int(g["scal"][nsteps - 1][1]) and int(vs.itt) != 0
        return
    assert (int(vs.itt), int(vs.time)) == (nsteps, int(g["scal"][nsteps - 1][1]))
    compare(S_snapshot(vs, names), g[f"s{nsteps:05d}"], names, what=f"{script_hooks}: hooks on the host")
    if script_hooks == "forcing_counts":
        assert type(model).forcing_calls == [int(t) for t in [0] + list(g["scal"][: nsteps - 1, 1])]


def test_hook_recognition_can_be_switched_off(oracle_backend, monkeypatch):
    import svat_scripts as S

    monkeypatch.setenv("RH_NO_HOOK_RECOGNITION", "1")
    g, names, forcing = load_case("svat_hetero_combo")
    model = S.make_model(S.params_from_golden(g, names), forcing, 24, script_hooks="plain")
    model.setup()
    assert not any(model.hook_classes()[h] for h in ("set_forcing", "set_parameters", "after_timestep"))
    assert not model.device_run_possible() and not model._fused_host_step_possible()


def test_probes_leave_no_trace(oracle_backend):
    """Probing runs the hooks against the RECORDING state: nothing is assigned, no kernel runs, the scalars are untouched."""
    import svat_scripts as S

    g, names, forcing = load_case("svat_hetero_combo")
    model = S.make_model(S.params_from_golden(g, names), forcing, 24, script_hooks="plain")
    model.setup()
    vs = model.state.variables
    before = S.snapshot_from_vs(vs, names)
    scal = [getattr(vs, k) for k in ("itt", "time", "itt_forc", "itt_day")] + [list(vs.month), list(vs.year)]
    day = np.array(vs.prec_day)
    model.hook_classes()
    assert vs._probe is None
    np.testing.assert_array_equal(before, S.snapshot_from_vs(vs, names))
    assert scal == [getattr(vs, k) for k in ("itt", "time", "itt_forc", "itt_day")] + [list(vs.month), list(vs.year)]
    np.testing.assert_array_equal(day, np.array(vs.prec_day))
