#!/usr/bin/env python3
"""Memory-instruction census of the fused kernel from its gfx950 ISA: what ONE column (one lane) loads and stores per launch.

    python tools/isa_census.py            # prints the table and rewrites roger_amd/csrc/rh_step_bytes.json

The kernel is compiled with -DRH_CENSUS (the tail, which one wavefront of the whole grid runs, is left out), so every
global_load / global_store of the kernel is per-column traffic: the kernel loads no plane twice and the sub-step loops of the
infiltration stages touch registers only.  The compiler drops the loads of planes the step assigns before reading them, so these
counts -- not the plane sets of tools/gen_sets.py, which list every plane a stage mentions -- are the bytes the kernel requests:
its algorithmic (compulsory) traffic, priced by bench.py's roofline.  It is a static count: an instruction that the compiler
duplicated into two mutually exclusive paths counts twice, so the figure is an upper bound a few per cent above the measured HBM
traffic (PMC: 1 714 B against 1 768 B for the lazy SVAT step)."""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from roger_amd.build import CSRC, FLAGS, HIPCC  # noqa: E402

OUT = os.path.join(CSRC, "rh_step_bytes.json")
WIDTH = {"ubyte": 1, "sbyte": 1, "ushort": 2, "sshort": 2, "short": 2, "byte": 1, "dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}


def census(extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([HIPCC] + FLAGS + ["-DRH_CENSUS"] + list(extra) + ["-S", "--cuda-device-only", os.path.join(CSRC, "roger_hip.hip"), "-o", os.path.join(tmp, "k.s")],
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(os.path.join(tmp, "k.s")).read()
    out = {}
    for fn in re.split(r"\n(?=_Z\w+:)", txt):
        # the staged passes of the routed step (settings.enable_routing_1D): k_routed_a / _b / _c / _c_after (host-driven), and the
        # device-driven ones k_routed_a2<SPARSE>, k_routed_bg<SPARSE>, k_routed_cg<AFTER, SPARSE> (the gathers folded in: their eight
        # neighbour reads of q_out, flow direction and mask count as loads of the pass)
        mr = re.match(r"_Z\d+k_(routed_[a-z0-9_]+?)(?:ILb([01])E(?:Lb([01])E)?Ev)?5ArenaP8DevState\w*:", fn)
        if mr:
            name = mr.group(1)
            if mr.group(3) is not None:      # k_routed_cg<AFTER, SPARSE>
                name += ("_after" if mr.group(2) == "1" else "") + ("_sparse" if mr.group(3) == "1" else "")
            elif mr.group(2) is not None:    # <SPARSE>
                name += "_sparse" if mr.group(2) == "1" else ""
            body = fn[: fn.find(".Lfunc_end")]   # (the whole function: the kernel has early exits)
            ld = sum(WIDTH[w] for op, w in re.findall(r"^\s+(?:global|buffer|flat)_(load|store)_(\w+)", body, re.M) if op == "load")
            st = sum(WIDTH[w] for op, w in re.findall(r"^\s+(?:global|buffer|flat)_(load|store)_(\w+)", body, re.M) if op == "store")
            out["mode0_" + name] = {"load_bytes": ld, "store_bytes": st, "in_loops": 0}
            continue
        m = re.match(r"_Z6k_stepILi(\d)ELb([01])ELb([01])ELb([01])ELb([01])EEv", fn)
        if not m or m.group(5) == "1":   # (the KEEP variants store what an accumulator asks for on top: counted by the caller, not here)
            continue
        body = fn[: fn.find(".Lfunc_end")]   # (the whole function: the kernel has early exits)
        ld = st = 0
        for op, w in re.findall(r"^\s+(?:global|buffer|flat)_(load|store)_(\w+)", body, re.M):
            if op == "load":
                ld += WIDTH[w]
            else:
                st += WIDTH[w]
        in_loop = 0
        mode, lateral, lazy, sparse = int(m.group(1)), m.group(2) == "1", m.group(3) == "1", m.group(4) == "1"
        out[f"mode{mode}_{'oned' if lateral else 'svat'}_{'sparse' if sparse else ('lazy' if lazy else 'eager')}"] = {"load_bytes": ld, "store_bytes": st, "in_loops": in_loop}
    return out


if __name__ == "__main__":
    # The lazy kernels read the parameter planes through the wave's word (DevState::pmask): counted with the word as a compile-time
    # constant -- bit 63 set (the derived parameters are not loaded: what runs after a setup whose planes are what the setup kernels
    # computed), all zero (every parameter loaded), and with the parameter planes' loads compiled out (what is left is state).
    rec = census(["-DRH_CENSUS_PMASK=0x8000000000000000ull"])
    plain = census()
    sets_txt = open(os.path.join(CSRC, "rh_sets.inc")).read()
    fields_txt = open(os.path.join(REPO, "include", "rh_fields.def")).read()
    ints = set(re.findall(r"RH_FIELD\((\w+),\s*I32", fields_txt))
    listed = lambda macro: re.findall(r"X\((\w+)\)", re.search(rf"#define {macro}\(X\) (.*)", sets_txt).group(1))   # noqa: E731
    derived = set(listed("RH_DERIVED_FIELDS"))
    for k, v in rec.items():
        if v["load_bytes"] != plain[k]["load_bytes"]:
            v["load_bytes_all_parameters_loaded"] = plain[k]["load_bytes"]
            # the parameter planes among the loads (tools/gen_sets.py: RH_PARAM_LOADED_*), the derived ones left out
            names = listed("RH_PARAM_LOADED_ONED" if "oned" in k else "RH_PARAM_LOADED_SVAT")
            v["parameter_load_bytes"] = sum(4 if n in ints else 8 for n in names if n not in derived)
    for k in sorted(rec):
        print(f"{k:28s} loads {rec[k]['load_bytes']:5d} B  stores {rec[k]['store_bytes']:5d} B  total {rec[k]['load_bytes'] + rec[k]['store_bytes']:5d} B per column and launch"
              )
    # the non-monthly pipeline (mode 0) is what a step executes unless the month changes; mode 2 holds both pipelines
    keep = {k[6:]: {kk: vv for kk, vv in v.items() if kk != "in_loops"} for k, v in rec.items() if k.startswith("mode0_")}
    keep["_note"] = ("bytes one column loads / stores per fused step (k_step, non-monthly pipeline), counted from the gfx950 ISA by tools/isa_census.py; "
                     "lazy / sparse variants: load_bytes with the derived parameters not loaded (the wave's bit 63 set), "
                     "load_bytes_all_parameters_loaded with it clear, parameter_load_bytes = the part of load_bytes that is parameter planes "
                     "(read as ONE element per wave and plane where the wave's columns hold one value)")
    txt = json.dumps(keep, indent=1, sort_keys=True) + "\n"
    if not os.path.exists(OUT) or open(OUT).read() != txt:
        open(OUT, "w").write(txt)
