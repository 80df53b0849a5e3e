"""Plane sets of the fused step, read from the generated csrc/rh_sets.inc (tools/gen_sets.py) and include/rh_fields.def:
how many bytes per column one launch of `k_step` must load and store -- its algorithmic (compulsory) HBM traffic, every
plane the step needs read once and every plane it assigns written once.  bench.py prices the kernel with these figures;
tests/test_abi.py pins them."""
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_SETS = os.path.join(_PKG, "csrc", "rh_sets.inc")
_FIELDS = os.path.join(os.path.dirname(_PKG), "include", "rh_fields.def")


def _plane_bytes():
    out = {}
    for name, typ, lv in re.findall(r"RH_FIELD\((\w+),\s*(\w+),\s*(\d)\)", open(_FIELDS).read()):
        out[name] = 4 if typ == "I32" else 8
        if lv == "2":
            out[name + "_m1"] = out[name]
    return out


def _planes(txt, seq, kind):
    out = []
    for line in re.findall(rf"#define RH_SEQ_{seq}_{kind}_\w+\(\w\)(.*)", txt):
        out += re.findall(r"\w\((\w+)(?:, \w+)?\)", line)
    return out


def step_plane_bytes(lateral=False, lazy=True, monthly=False):
    """(bytes loaded, bytes stored) per column by one fused step: `lazy` = the tau -> taum1 copies are deferred (the kernel
    neither stores the 30 X_m1 planes nor loads the 11 it reads: device-driven stepping), otherwise the eager kernel."""
    seq = "step" + ("_lateral" if lateral else "") + ("_monthly" if monthly else "")
    txt, size = open(_SETS).read(), _plane_bytes()
    if lazy:
        ld, st = _planes(txt, seq, "LLOAD"), _planes(txt, seq, "LSTORE")
    else:
        ld, st = _planes(txt, seq, "LOAD"), _planes(txt, seq, "STORE") + [x + "_m1" for x in _planes(txt, seq, "ROT")]
    return sum(size[p] for p in ld), sum(size[p] for p in st)


def pass_plane_bytes(seq):
    """(bytes loaded, bytes stored) per column by one of the staged passes of the routed step: seq = "routed_a", "routed_b", "routed_c",
    "routed_c_after" (tools/gen_sets.py PLAIN_SEQUENCES; roger_hip.hip k_routed_*)."""
    txt, size = open(_SETS).read(), _plane_bytes()
    ld, st = _planes(txt, seq, "LOAD"), _planes(txt, seq, "STORE")
    if not ld:
        raise KeyError(f"no sequence {seq} in rh_sets.inc")
    return sum(size[p] for p in ld), sum(size[p] for p in st)


def routine_union_bytes(lateral=False):
    """(bytes read, bytes written) per column by the union of the distinct variables the step's routines read / assign, X_m1
    levels included -- the method of SURVEY.md section 8(d) (2 779 B for the SVAT step from the reference's own read / write
    sets), applied to this package's restatement of the routines."""
    txt, size = open(_SETS).read(), _plane_bytes()
    rt = "rt_step_lateral" if lateral else "rt_step"
    ld = re.search(rf"#define RH_SET_LOAD_{rt}\(X\)(.*)", txt).group(1)
    st = re.search(rf"#define RH_SET_STORE_{rt}\(X\)(.*)", txt).group(1)
    return (sum(size[p] for p in re.findall(r"X\((\w+)\)", ld)), sum(size[p] for p in re.findall(r"X\((\w+)\)", st)))


if __name__ == "__main__":
    for lat in (False, True):
        for lazy in (True, False):
            print("oneD" if lat else "SVAT", "lazy" if lazy else "eager", step_plane_bytes(lat, lazy), sum(step_plane_bytes(lat, lazy)))
        print("  routine union", routine_union_bytes(lat), sum(routine_union_bytes(lat)))
    for seq in ("routed_a", "routed_b", "routed_c", "routed_c_after"):
        print(seq, pass_plane_bytes(seq), sum(pass_plane_bytes(seq)))
