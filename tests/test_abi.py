"""The C-ABI library loads and exports every entry point include/roger_hip.h declares; the
generated load/store sets are up to date.  No compute calls (CPU-only suite)."""
import ctypes
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header="roger_hip.h"):
    txt = open(os.path.join(REPO, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rh_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from roger_amd import _native
    from roger_amd.build import build_native

    build_native()
    lib = ctypes.CDLL(_native.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 40
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, missing
    # the binding declares the same set
    assert sorted(_native.DECLARED_SYMBOLS) == declared
    # include/roger_hip_sas.h
    declared_sas = _declared_functions("roger_hip_sas.h")
    assert len(declared_sas) >= 19
    missing = [f for f in declared_sas if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(_native.SAS_DECLARED_SYMBOLS) == declared_sas


def test_sas_array_registry_matches_def():
    from roger_amd import _native

    txt = open(os.path.join(REPO, "include", "rh_sas_arrays.def")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    arrays = re.findall(r"RH_SAS_ARRAY\((\w+),\s*(\w+),\s*(\w+)\)", txt)
    lib = _native.load()
    assert lib.rh_sas_num_arrays() == len(arrays)
    for i, (name, kind, _when) in enumerate(arrays):
        assert lib.rh_sas_array_name(i).decode() == name
        assert lib.rh_sas_array_index(name.encode()) == i
        assert bool(lib.rh_sas_array_is_int(i)) == (kind == "MASK")
    assert lib.rh_sas_array_index(b"no_such_array") == -1
    assert set(_native.DAILY_INPUTS) == {n for n, k, _ in arrays if k == "DAILY"}


def test_plane_registry_matches_fields_def():
    from roger_amd import _native

    txt = open(os.path.join(REPO, "include", "rh_fields.def")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    fields = re.findall(r"RH_FIELD\((\w+),\s*(\w+),\s*(\d)\)", txt)
    expect = []
    for name, typ, lv in fields:
        expect.append((name, typ == "I32"))
        if lv == "2":
            expect.append((name + "_m1", typ == "I32"))
    assert _native.plane_table() == expect
    lib = _native.load()
    assert lib.rh_plane_index(b"theta_rz_m1") == [n for n, _ in expect].index("theta_rz_m1")
    assert lib.rh_plane_index(b"no_such_plane") == -1


def test_generated_sets_are_current():
    inc = os.path.join(REPO, "roger_amd", "csrc", "rh_sets.inc")
    before = open(inc).read()
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_sets.py")], check=True)
    assert open(inc).read() == before, "rh_sets.inc is stale: run tools/gen_sets.py"
    # the fused step stores what the reference writes per step: 153 planes = 1224 B, plus `prec`,
    # which the summary path selects inside the fused kernel instead of in a pass of its own
    m = re.search(r"// rt_step: loads (\d+) planes, stores (\d+) planes", before)
    assert m and int(m.group(2)) == 154
    # lazy rotation: over the stages of a sequence the lazy sets store the same planes but the 30 X_m1 of the rotation
    # pairs, load none of them, and fill exactly the X_m1 that the step reads from X's register
    rot = set(re.search(r"#define RH_ROTATION_FIELDS\(X\) (.*)", before).group(1).replace("X(", "").replace(")", "").split())
    assert len(rot) == 30
    for seq in ("step", "step_monthly", "step_lateral", "step_lateral_monthly"):
        def planes(kind):
            out = []
            for line in re.findall(rf"#define RH_SEQ_{seq}_{kind}_\w+\(\w\)(.*)", before):
                out += re.findall(r"\w\((\w+)(?:, \w+)?\)", line)
            return out
        eager = set(planes("STORE")) | {x + "_m1" for x in planes("ROT")}
        lazy = set(planes("LSTORE"))
        assert eager - lazy == {x + "_m1" for x in rot}, seq
        assert not {p for p in planes("LLOAD") if p in {x + "_m1" for x in rot}}, seq
        assert set(planes("LLOAD")) <= set(planes("LOAD")) and len(planes("LLOAD")) == len(set(planes("LLOAD"))), seq
        assert set(planes("ALIAS")) <= {x + "_m1" for x in rot} and len(planes("ALIAS")) == 11, seq


def test_no_gpu_means_loud_failure():
    import torch

    from roger_amd import _native

    if torch.cuda.is_available():
        import pytest

        pytest.skip("a GPU is present")
    import pytest

    with pytest.raises(_native.NativeError, match="no CPU fallback"):
        _native.Context(2, 2)
    with pytest.raises(_native.NativeError, match="no CPU fallback"):
        _native.SasContext(4, 40)


def test_integration_stub_names_every_config_field():
    """INTEGRATION.md shows the ctypes structs a maintainer would add on the reference's side: they must carry every field of rh_config /
    rh_sas_config in the header's order (a shorter struct would let rh_default_config write past it)."""
    import os
    import re

    from roger_amd import _native

    doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    for cls, start in ((_native.RhConfig, "class RhConfig(C.Structure)"), (_native.RhSasConfig, "class RhSasConfig(C.Structure)")):
        block = doc[doc.index(start):]
        block = block[:block.index("\n\n")]
        names = re.findall(r'"(\w+)"', block)
        assert names == [n for n, _ in cls._fields_], (cls.__name__, names)
