#!/opt/conda/bin/python3.9
"""Restart files exchanged with the REFERENCE.  Runs in the build container only, under the
interpreter that has h5py and the reference's numpy pin:

    /opt/conda/bin/python3.9 tests/golden/make_restart_fixture.py write
        the reference (NumPy backend, float64) runs the heterogeneous 4 x 4 SVAT case of make_golden.py for three days, its own
        `write_restart` (roger/restart.py:130-174) writes tests/golden/restart/reference_restart.h5; a FRESH reference model names
        that file in settings.restart_input_filename, reads it at the end of setup() (roger/restart.py:70-127) and continues for
        two days -> tests/golden/restart/reference_restart.npz: the restarted reference's state and scalars, and the uninterrupted
        run's for comparison (the reference's restart list lacks part of a step's state: its own restarted run is not its
        uninterrupted run).
    /opt/conda/bin/python3.9 tests/golden/make_restart_fixture.py read FILE NDAYS OUT.npz
        the reference reads a restart file that roger_amd wrote and continues NDAYS days (tests/test_restart_interchange.py).

Only data is written.  `loguru` and `h5netcdf` are replaced by the inert stand-ins of make_golden.py; h5py is the real one."""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG   # noqa: E402

OUT = os.path.join(HERE, "restart")
NDAYS_BEFORE, NDAYS_AFTER = 3, 2


def import_reference():
    MG._install_standins()
    sys.path.insert(0, "/root/reference")
    from roger import runtime_settings

    runtime_settings.update(backend="numpy", float_type="float64", diskless_mode=False, hdf5_gzip_compression=True)
    import roger  # noqa: F401

    return roger


def model(roger, ndays, **override):
    from roger_amd.forcing import combo_forcing

    m = MG.make_model(roger, MG.hetero_params(4, 4), combo_forcing(ndays=24), ndays)
    m.override_settings = dict(override)
    return m


def run_until(m, t_end):
    vs = m.state.variables
    while vs.time < t_end:
        m.step(m.state)


def main():
    roger = import_reference()
    planes = MG.plane_names()
    if sys.argv[1] == "write":
        os.makedirs(OUT, exist_ok=True)
        fname = os.path.join(OUT, "reference_restart.h5")
        if os.path.exists(fname):
            os.remove(fname)
        a = model(roger, NDAYS_BEFORE + NDAYS_AFTER, restart_output_filename=fname)
        a.setup()
        run_until(a, NDAYS_BEFORE * 86400)
        from roger import restart

        restart.write_restart(a.state, force=True)
        rec = dict(plane_names=np.array([p[0] for p in planes]), at_restart=MG.snapshot(a.state.variables, planes),
                   at_restart_scal=MG.scalars(a.state.variables))
        run_until(a, (NDAYS_BEFORE + NDAYS_AFTER) * 86400)
        rec["uninterrupted"], rec["uninterrupted_scal"] = MG.snapshot(a.state.variables, planes), MG.scalars(a.state.variables)
        b = model(roger, NDAYS_BEFORE + NDAYS_AFTER, restart_input_filename=fname)
        b.setup()
        rec["after_read"], rec["after_read_scal"] = MG.snapshot(b.state.variables, planes), MG.scalars(b.state.variables)
        run_until(b, (NDAYS_BEFORE + NDAYS_AFTER) * 86400)
        rec["restarted"], rec["restarted_scal"] = MG.snapshot(b.state.variables, planes), MG.scalars(b.state.variables)
        np.savez_compressed(os.path.join(OUT, "reference_restart.npz"), **rec)
        import h5py

        with h5py.File(fname, "r") as f:
            n = len(f["core"])
        print(f"{fname}: {os.path.getsize(fname) / 1e3:.0f} kB, {n} datasets in 'core'; restarted run: itt {int(b.state.variables.itt)}, "
              f"uninterrupted: itt {int(a.state.variables.itt)}; max |restarted - uninterrupted| S = "
              f"{float(np.max(np.abs(np.asarray(a.state.variables.S) - np.asarray(b.state.variables.S)))):.3e}")
    elif sys.argv[1] == "read":
        fname, ndays, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
        b = model(roger, NDAYS_BEFORE + NDAYS_AFTER + ndays, restart_input_filename=fname)
        b.setup()
        rec = dict(plane_names=np.array([p[0] for p in planes]), after_read=MG.snapshot(b.state.variables, planes),
                   after_read_scal=MG.scalars(b.state.variables))
        run_until(b, int(b.state.variables.time) + ndays * 86400)
        rec["restarted"], rec["restarted_scal"] = MG.snapshot(b.state.variables, planes), MG.scalars(b.state.variables)
        np.savez_compressed(out, **rec)
        print(f"the reference read {fname} and continued {ndays} days: itt {int(b.state.variables.itt)}, time {int(b.state.variables.time)}")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
