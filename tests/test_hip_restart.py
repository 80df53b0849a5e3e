"""Restart on the GPU (roger/restart.py:32-67, 129-174): write at step k, fresh context, read, continue --
bit for bit the uninterrupted run, device-side output accumulators of the interrupted day included."""
import numpy as np
import pytest

from golden_util import load_case

pytestmark = pytest.mark.gpu


def _model(tmp_path, ndays, tag, **override):
    import svat_scripts as S
    from roger_amd import roger_routine

    g, names, forcing = load_case("svat_hetero_combo")
    base = S.make_model(S.params_from_golden(g, names), forcing, ndays)

    class WithOutput(type(base)):
        @roger_routine
        def set_diagnostics(self, state):
            d = state.diagnostics
            for kind, names_ in (("rate", ["prec", "aet", "q_ss", "inf_mat_rz"]), ("collect", ["S_rz", "S_ss", "swe"])):
                d[kind].output_variables = names_
                d[kind].output_frequency = 24 * 60 * 60
                d[kind].sampling_frequency = 1
                d[kind].base_output_path = str(tmp_path / tag)

    (tmp_path / tag).mkdir(exist_ok=True)
    m = WithOutput()
    m.override_settings = override
    return m, names


def test_restart_round_trip_on_the_device(tmp_path, monkeypatch):
    from roger_amd import h5lite, restart, runtime_settings as rs

    monkeypatch.chdir(tmp_path)
    object.__setattr__(rs, "diskless_mode", False)   # (runtime settings are locked once the core modules are imported)
    ndays = 8
    a, names = _model(tmp_path, ndays, "a")
    a.setup()
    a.run()                                           # the uninterrupted run
    # the interrupted one: stop in the MIDDLE of a day (the accumulators hold a partial sum), write, restart, finish
    b, _ = _model(tmp_path, ndays, "b")
    b.setup()
    k = 0
    while not (b.state.variables.time > 3 * 86400 and b.state.variables.time % 86400 != 0):
        b.step(b.state)
        k += 1
    fname = restart.write_restart(b.state, filename=str(tmp_path / "mid_{itt:0>4d}.restart.h5"))
    core = h5lite.read(fname)["core"]
    assert int(core["itt"]) == k and core["S_rz"].shape[2] == 2 and "hip_diag" in h5lite.read(fname)
    c, _ = _model(tmp_path, ndays, "c", restart_input_filename=fname)
    c.setup()
    assert int(c.state.variables.itt) == k and int(c.state.variables.time) == int(b.state.variables.time)
    with c.state.settings.unlock():
        c.state.settings.runlen = ndays * 86400 - int(c.state.variables.time)
    c.run()
    va, vc = a.state.variables, c.state.variables
    assert int(va.itt) == int(vc.itt) and int(va.time) == int(vc.time)
    for nm in names:
        base = nm[:-3] if nm.endswith("_m1") else nm
        np.testing.assert_array_equal(np.asarray(getattr(va, base)), np.asarray(getattr(vc, base)), err_msg=base)
    # the daily sums of the day the restart fell into come out the same (partial sums were carried over)
    from nc_util import netcdf_file

    for kind, var in (("rate", "aet"), ("rate", "prec"), ("collect", "S_rz")):
        with netcdf_file(str(tmp_path / "a" / f"GoldenSVAT.{kind}.nc"), "r", mmap=False) as fa, \
                netcdf_file(str(tmp_path / "c" / f"GoldenSVAT.{kind}.nc"), "r", mmap=False) as fc:
            ra, rc = np.asarray(fa.variables[var][:]), np.asarray(fc.variables[var][:])
            day = int(b.state.variables.time // 86400)              # 0-based index of the interrupted day; record day + 1
            np.testing.assert_array_equal(ra[day + 1:], rc[-(ra.shape[0] - day - 1):], err_msg=f"{kind} {var}")
