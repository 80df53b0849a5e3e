#!/usr/bin/env python3
"""Derive the load/store plane sets of every routine in roger_amd/csrc/rh_physics.h.

For each `rt_*` device function (and, transitively, the `h_*` helpers it calls) this scans the
body for `c.<field>` accesses:

  load set  = every field mentioned (the compiler drops loads whose value is never used, so an
              over-approximation costs nothing),
  store set = every field that is assigned (`=`, `+=`, `-=`, `*=`, `/=`) or passed to a helper
              parameter declared as a non-const `double &` / `int &`.

Output: roger_amd/csrc/rh_sets.inc with `RH_SET_LOAD_<routine>(X)` / `RH_SET_STORE_<routine>(X)`
X-macros.  Run by __graft_entry__.build() before hipcc; the result is also committed so that the
sets are reviewable.
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import liveness  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(REPO, "roger_amd", "csrc", "rh_physics.h")
OUT = os.path.join(REPO, "roger_amd", "csrc", "rh_sets.inc")
FIELDS_DEF = os.path.join(REPO, "include", "rh_fields.def")


def strip_comments(s):
    s = re.sub(r"//[^\n]*", "", s)
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def field_names():
    txt = strip_comments(open(FIELDS_DEF).read())
    names = []
    for name, _typ, levels in re.findall(r"RH_FIELD\((\w+),\s*(\w+),\s*(\d)\)", txt):
        names.append(name)
        if levels == "2":
            names.append(name + "_m1")
    return names


def split_args(s):
    args, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def parse_functions(src):
    funcs = {}
    for m in re.finditer(r"RH_DEV\s+[\w\s]+?\b((?:rt|rd|rl|h)_\w+)\s*\(([^)]*)\)\s*\{", src):
        name, params = m.group(1), m.group(2)
        i = m.end()
        depth = 1
        while depth:
            ch = src[i]
            depth += ch == "{"
            depth -= ch == "}"
            i += 1
        body = src[m.end(): i - 1]
        plist = split_args(params)
        refpos = [k for k, p in enumerate(plist) if re.match(r"^(double|int)\s*&", p)]
        colpos = [k for k, p in enumerate(plist) if re.match(r"^(const\s+)?Col\s*&", p)]
        funcs[name] = dict(body=body, refpos=refpos, colpos=colpos)
    return funcs


def matching_paren(s, start):
    depth = 0
    for i in range(start, len(s)):
        if s[i] == "(":
            depth += 1
        elif s[i] == ")":
            depth -= 1
            if depth == 0:
                return i
    raise ValueError("unbalanced parentheses")


def analyse(funcs):
    direct = {}
    for name, f in funcs.items():
        body = f["body"]
        mention = set(re.findall(r"\bc\.(\w+)", body))
        write = set(re.findall(r"\bc\.(\w+)\s*(?:=(?!=)|\+=|-=|\*=|/=)", body))
        calls = []
        for m in re.finditer(r"\b((?:rt|rd|rl|h)_\w+)\s*\(", body):
            callee = m.group(1)
            if callee not in funcs:
                continue
            close = matching_paren(body, m.end() - 1)
            args = split_args(body[m.end(): close])
            for k in funcs[callee]["refpos"]:
                if k < len(args):
                    fm = re.fullmatch(r"c\.(\w+)", args[k])
                    if fm:
                        write.add(fm.group(1))
            passes_col = any(k < len(args) and args[k] == "c" for k in funcs[callee]["colpos"])
            calls.append((callee, passes_col))
        direct[name] = dict(mention=mention, write=write, calls=calls)

    def closure(name, seen=()):
        d = direct[name]
        mention, write = set(d["mention"]), set(d["write"])
        for callee, passes_col in d["calls"]:
            if callee in seen or not passes_col:
                continue
            m2, w2 = closure(callee, seen + (name,))
            mention |= m2
            write |= w2
        return mention, write

    return {name: closure(name) for name in funcs if name.startswith(("rt_", "rd_", "rl_"))}


# The fused step as a sequence of stages.  For a sequence the generator emits per-stage sets so that
# the kernel can load a plane right before the first routine that mentions it and store a plane
# right after the last routine that assigns it (short live ranges -> fewer registers -> two waves
# per SIMD, one computing while the other waits on memory):
#   LOAD_<rt>  = mention(rt) minus everything mentioned by earlier stages (already in registers)
#   STORE_<rt> = write(rt)   minus everything written again by a later stage
#   ROT_<rt>   = the tau -> taum1 copies of after_timestep (`h_rotate`: c.X_m1 = c.X) that can be done right
#                after this stage: X has its final value (no later stage assigns it) and nobody reads the old
#                X_m1 any more.  The copy is a store of the register holding X into the X_m1 plane, so X's
#                live range ends here instead of at the end of the kernel (the 30 rotated values were the
#                kernel's spills).  X_m1 then leaves the after_timestep stage's store set.
_COMMON_HEAD = ["rt_interception", "rt_evapotranspiration", "rt_snow", "rt_inf_events", "rt_inf_matrix", "rt_inf_macropores",
                "rt_inf_cracks", "rt_inf_finish"]
SEQUENCES = {
    "step": ["rt_select_prec", "rt_select_pet"] + _COMMON_HEAD + ["rt_subsurface_runoff", "rt_capillary_rise", "rt_storage",
                                                "rt_num_error", "rt_after_timestep"],
    "step_monthly": ["rt_select_prec", "rt_select_pet", "rt_params_surface"] + _COMMON_HEAD + [
        "rt_subsurface_runoff", "rt_capillary_rise", "rt_storage", "rt_num_error", "rt_after_timestep"],
    "step_lateral": ["rt_select_prec", "rt_select_pet"] + _COMMON_HEAD + ["rt_subsurface_runoff_lateral", "rt_capillary_rise",
                                                        "rt_storage", "rt_num_error_lateral", "rt_after_timestep_oned"],
    "step_lateral_monthly": ["rt_select_prec", "rt_select_pet", "rt_params_surface"] + _COMMON_HEAD + [
        "rt_subsurface_runoff_lateral", "rt_capillary_rise", "rt_storage", "rt_num_error_lateral",
        "rt_after_timestep_oned"],
}
# The three passes of the routed step (settings.enable_routing_1D, roger_hip.hip k_routed_a / _b / _c): staged like the fused step -- a
# plane is loaded right before the first stage that mentions it and stored right after the last stage that assigns it -- without the
# rotation and summary machinery of the fused kernel.
PLAIN_SEQUENCES = {
    "routed_a": ["rt_interception", "rt_evapotranspiration", "rt_snow", "rt_inf_events", "rt_inf_matrix", "rt_inf_macropores",
                 "rt_inf_cracks", "rt_inf_finish_routed", "rt_route_surface_out"],
    # device-driven stepping (rh_run_steps): the selection of the step's forcing [and the monthly surface parameters] in front
    "routed_a2": ["rt_select_prec", "rt_select_pet", "rt_interception", "rt_evapotranspiration", "rt_snow", "rt_inf_events", "rt_inf_matrix",
                  "rt_inf_macropores", "rt_inf_cracks", "rt_inf_finish_routed", "rt_route_surface_out"],
    "routed_a2_monthly": ["rt_select_prec", "rt_select_pet", "rt_params_surface", "rt_interception", "rt_evapotranspiration", "rt_snow",
                          "rt_inf_events", "rt_inf_matrix", "rt_inf_macropores", "rt_inf_cracks", "rt_inf_finish_routed",
                          "rt_route_surface_out"],
    # the step core of the hook-preserving flow (rh_step_core: everything between the set_parameters and after_timestep hooks), staged
    "core": _COMMON_HEAD + ["rt_subsurface_runoff", "rt_capillary_rise", "rt_storage", "rt_num_error"],
    "core_lateral": _COMMON_HEAD + ["rt_subsurface_runoff_lateral", "rt_capillary_rise", "rt_storage", "rt_num_error_lateral"],
    "routed_b": ["rt_route_surface_in", "rt_subsurface_runoff_lateral", "rt_route_subsurface_out"],
    "routed_c": ["rt_route_subsurface_in", "rt_capillary_rise", "rt_storage", "rt_num_error_routed"],
    "routed_c_after": ["rt_route_subsurface_in", "rt_capillary_rise", "rt_storage", "rt_num_error_routed", "rt_after_timestep_oned"],
}
PLAIN_CHECK = {"routed_a": "rt_routed_a", "routed_b": "rt_routed_b", "routed_c": "rt_routed_c", "routed_c_after": "rt_routed_c_after",
               "core": "rt_step_core", "core_lateral": "rt_step_core_lateral"}
# stages before which long-lived, momentarily unused planes are evicted from registers (see main())
EVICT_BEFORE = set(filter(None, os.environ.get("RH_EVICT", "").split(",")))
# each sequence must cover exactly what the corresponding single-function routine does
SEQUENCE_CHECK = {"step": "rt_step", "step_monthly": "rt_step_monthly", "step_lateral": "rt_step_lateral",
                  "step_lateral_monthly": "rt_step_lateral_monthly"}


def rotation_pairs(funcs):
    """[(X_m1, X)] of h_rotate."""
    return re.findall(r"\bc\.(\w+_m1)\s*=\s*c\.(\w+)\s*;", funcs["h_rotate"]["body"])


def main():
    src = strip_comments(open(SRC).read())
    fields = field_names()
    order = {n: i for i, n in enumerate(fields)}
    funcs = parse_functions(src)
    sets = analyse(funcs)
    pairs = rotation_pairs(funcs)
    lines = ["// GENERATED by tools/gen_sets.py from rh_physics.h -- do not edit.", "#pragma once", ""]
    # Parameters the fused step DERIVES instead of loading (rd_<stage> in rh_physics.h: the stage's own evaluation of what the setup
    # kernel rt_params_soil computed from the primaries): rd_writes[stage] = the planes, rd_reads[stage] = the primaries it needs.
    rd = {name[3:]: sets.pop(name) for name in [n for n in sets if n.startswith("rd_")]}
    rd.pop("all", None)
    rd["rt_subsurface_runoff_lateral"] = rd["rt_subsurface_runoff"]   # (the oneD stage in the same position)
    rd_writes = {rt: w for rt, (m, w) in rd.items()}
    rd_reads = {rt: m - w for rt, (m, w) in rd.items()}
    derived = set().union(*rd_writes.values())
    # State a LAZY step derives instead of loading (rl_<stage>: k / h of root zone and subsoil from the water contents the previous step
    # left): never loaded by the lazy kernels, not stored by the sparse ones.
    rl = {name[3:]: sets.pop(name) for name in [n for n in sets if n.startswith("rl_")]}
    rl_writes = {rt: w for rt, (m, w) in rl.items()}
    rl_reads = {rt: m - w for rt, (m, w) in rl.items()}
    lazy_derived = set().union(*rl_writes.values())
    for name in sorted(sets):
        mention, write = sets[name]
        unknown = (mention | write) - set(fields)
        if unknown:
            sys.exit(f"{name}: unknown fields {sorted(unknown)}")
        ld = sorted(mention, key=order.get)
        st = sorted(write, key=order.get)
        lines.append(f"// {name}: loads {len(ld)} planes, stores {len(st)} planes")
        lines.append(f"#define RH_SET_LOAD_{name}(X) " + " ".join(f"X({n})" for n in ld))
        lines.append(f"#define RH_SET_STORE_{name}(X) " + " ".join(f"X({n})" for n in st))
        lines.append("")
    # Pure outputs of the fused step per model (tools/liveness.py): planes no step ever needs from memory.  k_step<..., SPARSE>
    # (every step of an rh_run_steps call but the last) does not store them.  A plane counts only if it is a pure output of BOTH
    # pipelines the kernel holds (with and without the monthly surface parameters), is not the source of a deferred tau -> taum1
    # copy whose X_m1 the step reads (the lazy kernel fills that register from the X PLANE), and is no X_m1 plane itself.
    analyser = liveness.Analyser(funcs)
    rotated_all = {xm1 for xm1, _ in pairs}

    def sparse_of(stages, lazy=False):
        if lazy:   # the lazy kernels' own sequence: rl_<stage> in front of the stages that have one
            stages = [x for rt in stages for x in (["rl_" + rt] if rt in rl else []) + [rt]]
        pure, summ = liveness.pure_outputs(analyser, stages, pairs)
        alias_src = {x for xm1, x in pairs if xm1 in summ.ue}
        return (pure - rotated_all) - alias_src

    # never_read: what NO variant of the fused kernel reads (rh_plane_is_pure_output: the planes the tests poison); sparse: what the
    # sparse (lazy) kernels do not store -- the state a lazy step derives itself is not stored, what it derives it from is
    never_read = {"svat": sparse_of(SEQUENCES["step"]) & sparse_of(SEQUENCES["step_monthly"]),
                  "oned": sparse_of(SEQUENCES["step_lateral"]) & sparse_of(SEQUENCES["step_lateral_monthly"])}
    sparse = {"svat": sparse_of(SEQUENCES["step"], True) & sparse_of(SEQUENCES["step_monthly"], True),
              "oned": sparse_of(SEQUENCES["step_lateral"], True) & sparse_of(SEQUENCES["step_lateral_monthly"], True)}
    for model in sparse:
        if not lazy_derived <= sparse[model]:
            sys.exit(f"{model}: state derived by the lazy kernels is read before that: {sorted(lazy_derived - sparse[model])}")
    for seq, stages in SEQUENCES.items():
        resident, all_m, all_w = set(), set(), set()
        lines.append(f"// sequence {seq}: " + " -> ".join(stages))
        # where each rotation can happen: after the last stage that assigns X or mentions X_m1 (the final
        # after_timestep stage, which holds the copies themselves, does not count), not before X is loaded
        last = len(stages) - 1
        rot_at = {}
        for xm1, x in pairs:
            r = -1
            for i, rt in enumerate(stages[:last]):
                mention, write = sets[rt]
                if x in write or xm1 in mention:
                    r = i
            first = next((i for i, rt in enumerate(stages) if x in sets[rt][0]), last)
            rot_at[(xm1, x)] = max(r, first)
        moved = {xm1 for (xm1, x), r in rot_at.items() if r < last}
        rot_need = {}   # stage -> fields whose register value the ROT of that stage stores
        for (xm1, x), r in rot_at.items():
            if r < last:
                rot_need.setdefault(r, set()).add(x)
        # Eviction before the register-heavy stages (EVICT_BEFORE): a plane that is resident, not touched by the
        # heavy stage and needed again later is dropped (stored first if its register holds a value memory does
        # not have yet) and re-loaded at its next mention.  Costs 8 bytes of traffic per plane and column, frees
        # two VGPRs through the stage where the compiler would otherwise spill.
        recs, dirty = [], set()
        for i, rt in enumerate(stages):
            mention, write = sets[rt]
            need = mention | rot_need.get(i, set())
            if rt in EVICT_BEFORE and i > 0:
                later = set().union(*[sets[r][0] | rot_need.get(j, set())
                                      for j, r in enumerate(stages) if j > i])
                ev = {f for f in resident if f not in need and f in later}
                recs[-1]["st"] |= ev & dirty
                dirty -= ev
                resident -= ev
            later_w = set().union(*[sets[r][1] for r in stages[i + 1:]]) if i + 1 < len(stages) else set()
            ld = need - resident
            st = (write - later_w) - (moved if i == last else set())
            resident |= need
            dirty = (dirty | write) - st
            all_m |= mention
            all_w |= write
            recs.append(dict(rt=rt, ld=ld, st=set(st), rot=rot_need.get(i, set())))
        # Lazy rotation (the fused kernel when the previous operation was a complete step, so that X_m1 == X holds for
        # every rotation pair at the start): X_m1 planes are neither loaded nor stored.  Where the step reads an X_m1, the
        # register is filled from the register of X (ALIAS) at the first stage that needs X_m1 -- or, if X is assigned
        # earlier than that, at the first stage that assigns X, while its register still holds the loaded value.
        if EVICT_BEFORE:
            sys.exit("the lazy-rotation sets are not generated together with RH_EVICT")
        rotated = {xm1 for xm1, _ in pairs}
        alias_at = {}
        for xm1, x in pairs:
            need = next((i for i, rt in enumerate(stages[:last]) if xm1 in sets[rt][0]), None)
            if need is None:
                continue
            first_write = next((i for i, rt in enumerate(stages) if x in sets[rt][1]), last)
            alias_at.setdefault(min(need, first_write), []).append((xm1, x))
        # a stage that derives parameters (rd_<stage>) is the first to mention each of them (or comes before it) -- the derived value then
        # lives about as long as the loaded one did --, no stage assigns one, and the primaries it reads are loaded with the stage's own planes at the latest
        for i, rt in enumerate(stages):
            for f in sorted(rd_writes.get(rt, ())):
                first = next(j for j, r in enumerate(stages) if f in sets[r][0])
                if first < i:   # (theta_fc is derived a stage early: tew needs it)
                    sys.exit(f"sequence {seq}: {f} is derived by rd_{rt} (stage {i}) but mentioned before, by {stages[first]} (stage {first})")
        if derived & all_w:
            sys.exit(f"sequence {seq}: derived parameters are assigned by a stage: {sorted(derived & all_w)}")
        resident_l = set()
        for i, rec in enumerate(recs):
            mention, write = sets[rec["rt"]]
            need_l = ((mention - rotated - lazy_derived) | {x for _, x in alias_at.get(i, [])} | rd_reads.get(rec["rt"], set())
                      | (rl_reads.get(rec["rt"], set()) - rotated))
            rec["lld"] = need_l - resident_l
            resident_l |= need_l
            rec["lst"] = rec["st"] - rotated
            rec["alias"] = sorted(alias_at.get(i, []), key=lambda p: order.get(p[0]))
            rec["sst"] = rec["lst"] - sparse["oned" if "lateral" in seq else "svat"]
            rec["kst"] = rec["lst"] - rec["sst"]   # what the sparse kernel leaves out: stored after all where DevState::keep asks for the plane
        for rec in recs:
            rt = rec["rt"]
            for kind, key in (("ROT", "rot"), ("LOAD", "ld"), ("STORE", "st"), ("LLOAD", "lld"), ("LSTORE", "lst"), ("SSTORE", "sst"), ("KSTORE", "kst")):
                lines.append(f"#define RH_SEQ_{seq}_{kind}_{rt}(X) " + " ".join(f"X({n})" for n in sorted(rec[key], key=order.get)))
            lines.append(f"#define RH_SEQ_{seq}_ALIAS_{rt}(A) " + " ".join(f"A({xm1}, {x})" for xm1, x in rec["alias"]))
        # the fused kernel samples the summary bits of the next step's predicates (roger_hip.hip, k_step): prec and ta
        # after rt_select_pet, swe and swe_top after rt_snow -- nothing later may assign them
        for fld, after in (("prec", "rt_select_pet"), ("ta", "rt_select_pet"), ("swe", "rt_snow"), ("swe_top", "rt_snow")):
            late = [rt for rt in stages[stages.index(after) + 1:] if fld in sets[rt][1]]
            if late:
                sys.exit(f"sequence {seq}: {fld} is assigned after {after} by {late}; move the summary sampling")
        ref_m, ref_w = sets[SEQUENCE_CHECK[seq]]
        if all_m != ref_m or all_w != ref_w:
            sys.exit(f"sequence {seq} does not match {SEQUENCE_CHECK[seq]}")
        lines.append("")
    # Sparse stores of the device-driven ROUTED step (three passes: routed_a2 [_monthly], routed_b, routed_c_after, the gathers folded
    # into the second and third): a plane may be left out by a pass if the whole step only produces it (flow analysis over the
    # concatenated passes, the gathers as assignments of q_*_in from the neighbours' q_*_out), no LATER pass of the same step loads it
    # (the passes hand their intermediates over through memory) and it is not what the neighbours gather.
    gather_src = {"q_sur_out", "q_sub_out"}
    rfuncs = dict(funcs)
    rfuncs["rt_gather_surface"] = dict(body="c.q_sur_in = c.q_sur_out + c.flow_dir_topo + c.maskCatch;", refpos=[], colpos=[0])
    rfuncs["rt_gather_subsurface"] = dict(body="c.q_sub_in = c.q_sub_out + c.flow_dir_topo + c.maskCatch;", refpos=[], colpos=[0])
    ranalyser = liveness.Analyser(rfuncs)
    routed_passes = ("routed_b", "routed_c_after")

    def routed_pure(first):
        seq = (PLAIN_SEQUENCES[first] + ["rt_gather_surface"] + PLAIN_SEQUENCES["routed_b"] + ["rt_gather_subsurface"]
               + PLAIN_SEQUENCES["routed_c_after"])
        summ = ranalyser.sequence(seq)
        return summ.deff - summ.ue

    routed_sparse = (routed_pure("routed_a2") & routed_pure("routed_a2_monthly")) - gather_src
    mention_of = lambda names: set().union(*[sets[r][0] for r in names])   # noqa: E731
    routed_skip = {
        "routed_a2": routed_sparse - mention_of(PLAIN_SEQUENCES["routed_b"] + PLAIN_SEQUENCES["routed_c_after"]),
        "routed_a2_monthly": routed_sparse - mention_of(PLAIN_SEQUENCES["routed_b"] + PLAIN_SEQUENCES["routed_c_after"]),
        "routed_b": routed_sparse - mention_of(PLAIN_SEQUENCES["routed_c_after"]),
        "routed_c_after": routed_sparse,
    }
    for seq, stages in PLAIN_SEQUENCES.items():
        lines.append(f"// sequence {seq}: " + " -> ".join(stages))
        resident, all_m, all_w = set(), set(), set()
        for i, rt in enumerate(stages):
            mention, write = sets[rt]
            later_w = set().union(*[sets[r][1] for r in stages[i + 1:]]) if i + 1 < len(stages) else set()
            ld, st = mention - resident, write - later_w
            resident |= mention
            all_m |= mention
            all_w |= write
            lines.append(f"#define RH_SEQ_{seq}_LOAD_{rt}(X) " + " ".join(f"X({n})" for n in sorted(ld, key=order.get)))
            lines.append(f"#define RH_SEQ_{seq}_STORE_{rt}(X) " + " ".join(f"X({n})" for n in sorted(st, key=order.get)))
            if seq in routed_skip:
                lines.append(f"#define RH_SEQ_{seq}_SSTORE_{rt}(X) " + " ".join(f"X({n})" for n in sorted(st - routed_skip[seq], key=order.get)))
        if seq.startswith("routed_a2"):
            # k_routed_a2 samples the summary bits of the next step's predicates like the fused kernel: prec / ta after rt_select_pet, swe /
            # swe_top after rt_snow -- no later stage of this pass and no stage of the passes behind it may assign them
            behind = [r for q in ("routed_b", "routed_c_after", "routed_c") for r in PLAIN_SEQUENCES[q]] + ["rt_after_timestep_oned"]
            for fld, after in (("prec", "rt_select_pet"), ("ta", "rt_select_pet"), ("swe", "rt_snow"), ("swe_top", "rt_snow")):
                late = [r for r in stages[stages.index(after) + 1:] + behind if fld in sets[r][1]]
                if late:
                    sys.exit(f"sequence {seq}: {fld} is assigned after {after} by {late}; move the summary sampling")
        if seq in PLAIN_CHECK:
            ref_m, ref_w = sets[PLAIN_CHECK[seq]]
            if all_m != ref_m or all_w != ref_w:
                sys.exit(f"sequence {seq} does not match {PLAIN_CHECK[seq]}")
        lines.append("")
    lines.append(f"// planes the routed step only produces ({len(routed_sparse)}): left out by the last pass that assigns them unless a later pass loads them")
    lines.append("#define RH_SPARSE_FIELDS_ROUTED(X) " + " ".join(f"X({n})" for n in sorted(routed_sparse, key=order.get)))
    for model in ("svat", "oned"):
        lines.append(f"// planes the sparse fused {model} step does not store ({len(sparse[model])}): produced only, or derived by the next lazy step itself")
        lines.append(f"#define RH_SPARSE_FIELDS_{model.upper()}(X) " + " ".join(f"X({n})" for n in sorted(sparse[model], key=order.get)))
        lines.append(f"// planes NO variant of the fused {model} step reads ({len(never_read[model])})")
        lines.append(f"#define RH_NEVER_READ_FIELDS_{model.upper()}(X) " + " ".join(f"X({n})" for n in sorted(never_read[model], key=order.get)))
    lines.append("")
    # Parameter planes of the fused step: read by a stage, assigned by none (static) or by the monthly surface parameters only.  A wave
    # whose 64 columns hold ONE value of such a plane reads one element instead of 512 bytes (DevState::pmask, k_param_mask): bit b of
    # the wave's word.  A monthly plane keeps its bit only while the inputs of rt_params_surface are uniform over the wave as well.
    m_all, w_mon, w_other = set(), set(), set()
    for seq, stages in SEQUENCES.items():
        for rt in stages:
            mention, write = sets[rt]
            m_all |= mention
            if rt == "rt_params_surface":
                w_mon |= write
            else:
                w_other |= write
    static = m_all - w_mon - w_other
    monthly = (w_mon - w_other) & m_all
    monthly_inputs = (sets["rt_params_surface"][0] - sets["rt_params_surface"][1]) - {"ta"}   # (ta feeds swe_top_tot only, a state)
    if not monthly_inputs <= static:
        sys.exit(f"rt_params_surface reads planes that are no static parameters: {sorted(monthly_inputs - static)}")
    if not derived <= static:
        sys.exit(f"derived parameters that are not static parameters of the step: {sorted(derived - static)}")
    bits = sorted(static | monthly, key=order.get)
    if len(bits) > 61:
        sys.exit(f"{len(bits)} parameter planes do not fit the wave's 64-bit word (bit 63: derived parameters valid, bit 62: every column in the catchment)")
    lines.append(f"// parameter planes of the fused step ({len(static)} static, {len(monthly)} assigned by the monthly surface parameters only): X(name, bit)")
    lines.append("#define RH_PARAM_BITS(X) " + " ".join(f"X({n}, {b})" for b, n in enumerate(bits)))
    lines.append("#define RH_PARAM_MONTHLY(X) " + " ".join(f"X({n})" for n in sorted(monthly, key=order.get)))
    lines.append("#define RH_PARAM_MONTHLY_INPUTS(X) " + " ".join(f"X({n})" for n in sorted(monthly_inputs, key=order.get)))
    for model, seq in (("SVAT", "step"), ("ONED", "step_lateral")):   # what the step loads of them unless the month changes
        loaded = set().union(*[sets[rt][0] for rt in SEQUENCES[seq]]) & (static | monthly)
        lines.append(f"#define RH_PARAM_LOADED_{model}(X) " + " ".join(f"X({n})" for n in sorted(loaded, key=order.get)))
    lines.append(f"// parameters a stage of the fused step derives from the primaries instead of loading them ({len(derived)})")
    lines.append("#define RH_DERIVED_FIELDS(X) " + " ".join(f"X({n})" for n in sorted(derived, key=order.get)))
    lines.append("#define RH_LAZY_DERIVED_FIELDS(X) " + " ".join(f"X({n})" for n in sorted(lazy_derived, key=order.get)))
    lines.append("#define RH_DERIVE_PRIMARIES(X) " + " ".join(f"X({n})" for n in sorted(set().union(*rd_reads.values()), key=order.get)))
    lines.append("")
    # the rotation pairs themselves (materialising the X_m1 planes after lazy steps): X(x) for every c.x_m1 = c.x
    lines.append("// tau -> taum1 rotation of after_timestep (h_rotate): " + str(len(pairs)) + " pairs")
    lines.append("#define RH_ROTATION_FIELDS(X) " + " ".join(f"X({x})" for _, x in sorted(pairs, key=lambda p: order.get(p[1]))))
    for xm1, x in pairs:
        if xm1 != x + "_m1":
            sys.exit(f"rotation pair {xm1} = {x} does not follow the <name>_m1 convention")
    lines.append("")
    txt = "\n".join(lines)
    if not os.path.exists(OUT) or open(OUT).read() != txt:
        open(OUT, "w").write(txt)
    if "-v" in sys.argv:
        for name in sorted(sets):
            print(name, len(sets[name][0]), len(sets[name][1]))


if __name__ == "__main__":
    main()
