"""tools/liveness.py: the flow analysis that decides which planes the fused step only PRODUCES (k_step<..., SPARSE> leaves their
stores out inside an rh_run_steps call).  Hand-written cases for every construct the analysis distinguishes, then facts about the
real rh_physics.h that a wrong analysis would get wrong.  (The GPU suite checks the same property dynamically: the planes are
poisoned before every step of the golden trajectories, tests/test_hip_sparse.py.)"""
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_sets as G   # noqa: E402
import liveness as L   # noqa: E402


def pure(src, stages=("rt_a",)):
    funcs = G.parse_functions(G.strip_comments(src))
    p, s = L.pure_outputs(L.Analyser(funcs), list(stages))
    return p, s


def test_assigned_then_read_is_a_pure_output():
    p, s = pure("RH_DEV void rt_a(Col &c) { c.x = c.p * 2; c.y = c.x + 1; }")
    assert p == {"x", "y"} and s.ue == {"p"}


def test_read_before_the_assignment_is_not():
    p, _ = pure("RH_DEV void rt_a(Col &c) { c.y = c.x + 1; c.x = 0; }")
    assert p == {"y"}


def test_update_operators_read():
    p, _ = pure("RH_DEV void rt_a(Col &c) { c.x += c.p; c.y = 1; c.y -= 2; }")
    assert p == {"y"}


def test_a_conditional_assignment_keeps_the_old_value_on_one_path():
    p, s = pure("RH_DEV void rt_a(Col &c, const StepCtx &X) { if (X.sel >= 0) { c.x = 1; } c.y = 2; }")
    assert p == {"y"} and "x" in s.may - s.deff


def test_both_branches_assign():
    p, _ = pure("RH_DEV void rt_a(Col &c) { if (c.m > 4) { c.x = c.p; } else { c.x = 0.0; } if (c.m) c.y = 1; else c.y = 2; }")
    assert p == {"x", "y"}


def test_else_if_chain_without_a_final_else():
    p, _ = pure("RH_DEV void rt_a(Col &c) { if (c.m > 4) c.x = 1; else if (c.m > 2) c.x = 2; }")
    assert p == set()


def test_a_select_on_the_old_value_reads_it():
    p, _ = pure("RH_DEV void rt_a(Col &c) { c.x = (c.m > 0 ? 1.0 : c.x) * c.mk; }")
    assert p == set()


def test_loop_bodies_may_not_run():
    p, _ = pure("RH_DEV void rt_a(Col &c) { for (int it = 0; it < c.n; ++it) { c.x = it; } c.y = 0; for (int k = 0; k < 3; ++k) c.y += c.x; }")
    assert p == set() or p == {"y"}
    assert "x" not in p


def test_helper_summaries_compose_in_order():
    src = """
    RH_DEV void h_first(Col &c, double mk) { c.x = c.p * mk; }
    RH_DEV void h_second(Col &c) { c.y = c.x; c.z = c.z + 1; }
    RH_DEV void rt_a(Col &c) { h_first(c, 1.0); h_second(c); }
    """
    p, s = pure(src)
    assert p == {"x", "y"} and s.ue == {"p", "z"}


def test_a_helper_called_under_a_condition_assigns_conditionally():
    src = """
    RH_DEV void h_set(Col &c) { c.x = 1; }
    RH_DEV void rt_a(Col &c, const StepCtx &X) { if (X.cond1) h_set(c); c.y = (c.m ? 1.0 : 0.0); }
    """
    assert pure(src)[0] == {"y"}


def test_a_helper_inside_a_ternary_is_conditional():
    src = """
    RH_DEV double h_val(Col &c) { c.x = 1; return 2.0; }
    RH_DEV void rt_a(Col &c) { c.y = c.m > 0 ? h_val(c) : 0.0; }
    """
    assert pure(src)[0] == {"y"}


def test_reference_parameters_read_and_may_write():
    src = """
    RH_DEV double h_take(double &store, double cap) { store -= cap; return cap; }
    RH_DEV void rt_a(Col &c) { c.e = h_take(c.s, c.cap); }
    """
    p, s = pure(src)
    assert p == {"e"} and "s" in s.ue


def test_nothing_is_definite_behind_an_early_return():
    src = "RH_DEV void rt_a(Col &c) { c.x = 1; if (c.m > 0) return; c.y = 2; }"
    p, _ = pure(src)
    assert p == {"x"}


def test_stages_of_a_sequence():
    src = """
    RH_DEV void rt_a(Col &c) { c.flux = c.store * 0.1; c.store -= c.flux; }
    RH_DEV void rt_b(Col &c) { c.bal = c.store + c.flux - c.last; c.last = c.store; }
    """
    p, s = pure(src, ("rt_a", "rt_b"))
    assert p == {"flux", "bal"} and s.ue == {"store", "last"}


def test_deferred_rotation_copies_are_not_reads():
    src = """
    RH_DEV void h_rotate(Col &c) { c.q_m1 = c.q; c.s_m1 = c.s; }
    RH_DEV void rt_a(Col &c) { c.q = c.s * 0.5; c.s -= c.q; c.d = c.s - c.s_m1; h_rotate(c); }
    """
    funcs = G.parse_functions(G.strip_comments(src))
    p, s = L.pure_outputs(L.Analyser(funcs), ["rt_a"], [("q_m1", "q"), ("s_m1", "s")])
    assert {"q", "d", "q_m1"} <= p and "s" not in p and "s_m1" in s.ue


# ---- the real header ------------------------------------------------------------------------------------------------------------
def _real():
    funcs = G.parse_functions(G.strip_comments(open(G.SRC).read()))
    return funcs, G.rotation_pairs(funcs)


def _sparse_fields(model):
    txt = open(G.OUT).read()
    line = next(ln for ln in txt.splitlines() if ln.startswith(f"#define RH_SPARSE_FIELDS_{model}(X)"))
    return set(re.findall(r"X\((\w+)\)", line))


def test_facts_about_the_real_step():
    sv, on = _sparse_fields("SVAT"), _sparse_fields("ONED")
    # fluxes and diagnostics of a step
    for f in ("q_ss", "q_rz", "aet", "transp", "evap_soil", "inf_mat_pot", "q_hof", "q_sof", "dS_num_error", "int_prec", "snow_melt"):
        assert f in sv and f in on, f
    # state and everything a step takes over from the last one
    for f in ("S_fp_rz", "S_lp_rz", "S_fp_ss", "S_lp_ss", "swe", "S_snow", "z0", "S_dep", "theta_rz", "z_wf", "z_sat",
              "inf_mat_rz",      # calc_inf_mp uses LAST step's inf_mat_rz (infiltration.py:888-904)
              "prec", "ta",      # assigned under a condition by the selection, sampled for the summary bits
              "t_event_csum", "prec_event_csum", "maskCatch", "lu_id", "ks", "z_soil"):
        assert f not in sv and f not in on, f
    # X whose X_m1 the step reads: the lazy kernel fills that register from the X plane
    for f in ("S_rz", "S_ss", "S", "S_s"):
        assert f not in sv, f
    assert not any(f.endswith("_m1") for f in sv | on)
    assert {"q_sub", "q_sub_rz", "q_sub_ss", "z_sat_layer_1"} <= on - sv
    assert len(sv) >= 60 and len(on) >= 80


def test_sparse_planes_are_stored_by_the_full_kernel_and_never_loaded_for_their_value():
    funcs, pairs = _real()
    sets = G.analyse(funcs)
    for model, seqs in (("SVAT", ("step", "step_monthly")), ("ONED", ("step_lateral", "step_lateral_monthly"))):
        sp = _sparse_fields(model)
        for seq in seqs:
            written = set().union(*[sets[rt][1] for rt in G.SEQUENCES[seq]])
            assert sp <= written, (model, seq, sorted(sp - written))


def test_generated_sets_are_in_sync():
    before = open(G.OUT).read()
    G.main()
    assert open(G.OUT).read() == before
