"""Restart files go back and forth between the reference and roger_amd ("lets reference and build exchange
states").

* tests/golden/restart/reference_restart.h5 was written by the REFERENCE's own `write_restart` (roger/restart.py:130-174, h5py,
  chunked + gzip) three days into the heterogeneous 4 x 4 SVAT case; reference_restart.npz holds what a fresh REFERENCE model made
  of it: its state right after `read_restart` at the end of setup(), and after two more days
  (tests/golden/make_restart_fixture.py, run under /opt/conda/bin/python3.9 in the build container).  roger_amd reads the same
  file (h5lite) and must be in the same state, and end in the same state.
* The other direction needs the reference itself and runs in the build container only: a file `roger_amd.restart.write_restart`
  wrote is read by the reference's `read_restart` (roger/restart.py:70-127), which continues two days; roger_amd, restarted from the
  same file, ends in the same state."""
import os
import subprocess

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, compare, load_case

FIX = os.path.join(GOLDEN_DIR, "restart")
CONDA_PY = "/opt/conda/bin/python3.9"
SCAL = ("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")


def _model(ndays, **override):
    import svat_scripts as S

    g, names, forcing = load_case("svat_hetero_combo")
    m = S.make_model(S.params_from_golden(g, names), forcing, ndays)
    m.override_settings = dict(override)
    return m, names


def _rows(ref, names):
    """Rows of a fixture snapshot (every oracle plane, the routing's included) for the planes of this model."""
    have = list(ref["plane_names"])
    return [have.index(nm) for nm in names]


def _scalars(vs):
    row = [float(getattr(vs, k)) for k in SCAL]
    for k in ("event_id", "year", "month", "doy"):
        a = np.asarray(getattr(vs, k))
        row += [float(a[0]), float(a[1])]
    return np.array(row)


def _continue_two_days(m):
    vs = m.state.variables
    t_end = int(vs.time) + 2 * 86400
    while vs.time < t_end:
        m.step(m.state)


def _check_reads_reference():
    import svat_scripts as S

    ref = np.load(os.path.join(FIX, "reference_restart.npz"))
    m, names = _model(5, restart_input_filename=os.path.join(FIX, "reference_restart.h5"))
    sel = _rows(ref, names)
    m.setup()
    vs = m.state.variables
    np.testing.assert_array_equal(_scalars(vs), ref["after_read_scal"])
    assert int(vs.time) == 3 * 86400
    compare(S.snapshot_from_vs(vs, names), ref["after_read"][sel], names, what="state right after reading the reference's restart file")
    _continue_two_days(m)
    np.testing.assert_array_equal(_scalars(vs), ref["restarted_scal"])
    compare(S.snapshot_from_vs(vs, names), ref["restarted"][sel], names, what="two days after restarting from the reference's file")
    # (the reference's own restarted run IS its uninterrupted run in this case: the fixture generator printed the difference)
    compare(ref["restarted"][sel], ref["uninterrupted"][sel], names, what="fixture: restarted vs uninterrupted reference")


def test_reads_a_restart_file_the_reference_wrote(monkeypatch, oracle):
    import oracle_context
    from roger_amd import _native

    monkeypatch.setattr(_native, "Context", oracle_context.OracleContext)
    monkeypatch.setattr(_native, "plane_table", lambda: list(zip(oracle.plane_names(), oracle.plane_is_int())))
    _check_reads_reference()


@pytest.mark.gpu
def test_reads_a_restart_file_the_reference_wrote_on_the_device():
    _check_reads_reference()


def test_the_reference_file_is_what_h5py_reads():
    """h5lite on the reference's file: every dataset of the group "core", shapes with the ghost frame and the trailing time axis."""
    from roger_amd import h5lite, restart

    core = h5lite.read(os.path.join(FIX, "reference_restart.h5"))["core"]
    assert core["S_rz"].shape == (8, 8, 2) and core["lu_id"].shape == (8, 8) and core["itt"].shape == () and int(core["time"]) == 3 * 86400
    missing = [k for k in restart.REFERENCE_RESTART_VARIABLES if k not in core and k not in ("slope", "slope_per", "dmph") and
               not k.startswith("z_sat_layer_")]
    assert not missing, missing


@pytest.mark.skipif(not (os.path.isdir("/root/reference") and os.path.exists(CONDA_PY)),
                    reason="needs the reference and an interpreter with h5py (build container only)")
def test_the_reference_reads_a_restart_file_roger_amd_wrote(monkeypatch, oracle, tmp_path):
    import oracle_context
    import svat_scripts as S
    from roger_amd import _native, restart

    monkeypatch.setattr(_native, "Context", oracle_context.OracleContext)
    monkeypatch.setattr(_native, "plane_table", lambda: list(zip(oracle.plane_names(), oracle.plane_is_int())))
    a, names = _model(3)
    a.setup()
    a.run()
    fname = str(tmp_path / "roger_amd_restart.h5")
    restart.write_restart(a.state, filename=fname)
    out = str(tmp_path / "reference_continued.npz")
    r = subprocess.run([CONDA_PY, os.path.join(GOLDEN_DIR, "make_restart_fixture.py"), "read", fname, "2", out], capture_output=True, text=True,
                       cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = np.load(out)
    b, _ = _model(5, restart_input_filename=fname)
    b.setup()
    vs = b.state.variables
    # what the reference made of the file right after reading it: its restart list, nothing else (roger/restart.py:96-105)
    sel = _rows(ref, names)
    np.testing.assert_array_equal(ref["after_read_scal"][:8], _scalars(vs)[:8])
    for nm in ("S_fp_rz", "S_lp_ss", "swe", "z_wf", "theta_rz", "S_int_top", "z0", "t_event_csum", "prec_event_csum"):
        i = list(names).index(nm)
        np.testing.assert_array_equal(ref["after_read"][sel][i], S.snapshot_from_vs(vs, names)[i], err_msg=nm)
    _continue_two_days(b)
    np.testing.assert_array_equal(_scalars(vs), ref["restarted_scal"])
    compare(S.snapshot_from_vs(vs, names), ref["restarted"][sel], names, what="the reference and roger_amd, both restarted from roger_amd's file")
