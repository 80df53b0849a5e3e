"""Host package on the real device: the setup script of the golden generator run through
roger_amd (RogerSetup.setup/step/run, run_device) matches the reference trajectory."""
import numpy as np
import pytest

from golden_util import compare, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["svat_uniform_rain", "svat_hetero_combo", "oned_hetero_heavyrain"])
def test_setup_step_reproduce_reference(case):
    import svat_scripts as S
    from golden_util import is_lateral

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
