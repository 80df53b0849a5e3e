"""roger_amd.nc4lite: the output files of the diagnostics are netCDF-4 (HDF5 with netCDF's conventions: dimension scales,
`_Netcdf4Dimid`, `DIMENSION_LIST` / `REFERENCE_LIST`, unlimited `Time`), as the reference writes them through h5netcdf
(roger/io_tools/netcdf.py:22-72, 121-165).  Read back by the module's own reader everywhere, and taken apart with h5py's
dimension-scale API where an interpreter with h5py exists (the build container)."""
import json
import os
import subprocess

import numpy as np
import pytest

from roger_amd import h5lite, nc4lite

CONDA_PY = "/opt/conda/bin/python3.9"


def _sample(path, nt=150):
    nx, ny = 5, 4
    rng = np.random.default_rng(3)
    x, y, t = np.arange(nx) * 25.0 + 12.5, np.arange(ny) * 25.0, np.arange(nt, dtype=np.float64)
    prec, aged = rng.random((nt, ny, nx)), rng.random((nt, 3, ny, nx))
    nc4lite.write(str(path), {"x": nx, "y": ny, "Time": None, "ages": 3}, {
        "x": (("x",), x, {"long_name": "x", "units": "m"}), "y": (("y",), y, {"long_name": "y", "units": "m"}),
        "Time": (("Time",), t, {"long_name": "Time", "units": "days", "time_origin": "2018-01-01 00:00:00"}),
        "prec": (("Time", "y", "x"), prec, {"_FillValue": np.float64(-9999.0), "long_name": "prec", "units": "mm/dt"}),
        "tt_q_ss": (("Time", "ages", "y", "x"), aged, {"long_name": "tt_q_ss", "units": ""})},
        {"setup_identifier": "SVAT", "comment": "First timestep (t=0) contains initial values."})
    return dict(x=x, y=y, Time=t, prec=prec, tt_q_ss=aged)


@pytest.mark.parametrize("nt", [0, 1, 64, 65, 150, 5000])   # no record; one leaf of the chunk B-tree full / split; two and three levels
def test_round_trip(tmp_path, nt):
    ref = _sample(tmp_path / "a.nc", nt)
    rec = nc4lite.read(str(tmp_path / "a.nc"))
    assert rec["dimensions"] == {"x": 5, "y": 4, "Time": None, "ages": 3}
    assert rec["attributes"]["setup_identifier"] == "SVAT"
    dims, arr, attrs = rec["variables"]["prec"]
    assert dims == ("Time", "y", "x") and attrs == {"_FillValue": -9999.0, "long_name": "prec", "units": "mm/dt"}
    np.testing.assert_array_equal(arr, ref["prec"])
    assert rec["variables"]["tt_q_ss"][0] == ("Time", "ages", "y", "x") and "ages" not in rec["variables"]
    assert rec["variables"]["Time"][2]["time_origin"] == "2018-01-01 00:00:00"
    plain = h5lite.read_root(str(tmp_path / "a.nc"))       # the generic HDF5 reader sees the same datasets
    for k, v in ref.items():
        np.testing.assert_array_equal(plain[k], v, err_msg=k)


def test_bad_input_is_refused(tmp_path):
    with pytest.raises(ValueError, match="unknown dimension"):
        nc4lite.write(str(tmp_path / "b.nc"), {"x": 2}, {"v": (("z",), np.zeros(2), {})})
    with pytest.raises(ValueError, match="dimension x has 2"):
        nc4lite.write(str(tmp_path / "b.nc"), {"x": 2}, {"v": (("x",), np.zeros(3), {})})
    with pytest.raises(ValueError, match="first axis"):
        nc4lite.write(str(tmp_path / "b.nc"), {"x": 2, "Time": None}, {"v": (("x", "Time"), np.zeros((2, 3)), {})})


_H5PY_CHECK = r"""
import json, sys, warnings
warnings.filterwarnings("ignore")
import h5py, numpy as np
import h5py.h5ds as h5ds
f = h5py.File(sys.argv[1], "r")
out = {"root_attrs": {k: (v.decode() if isinstance(v, bytes) else v) for k, v in f.attrs.items()}, "vars": {}}
for k in f:
    d = f[k]
    rec = {"shape": list(d.shape), "maxshape": [m for m in d.maxshape], "chunks": list(d.chunks) if d.chunks else None, "dtype": str(d.dtype),
           "is_scale": bool(h5ds.is_scale(d.id)), "dims": [[s.name for s in dim.values()] for dim in d.dims] if not h5ds.is_scale(d.id) else None,
           "sum": float(np.asarray(d[...], dtype=np.float64).sum()) if d.size else 0.0}
    for a in ("NAME", "CLASS", "units", "long_name", "time_origin"):
        if a in d.attrs:
            v = d.attrs[a]
            rec[a] = v.decode() if isinstance(v, bytes) else v
    if "_Netcdf4Dimid" in d.attrs:
        rec["dimid"] = int(d.attrs["_Netcdf4Dimid"])
    if "REFERENCE_LIST" in d.attrs:
        rec["referenced_by"] = sorted((f[r[0]].name, int(r[1])) for r in d.attrs["REFERENCE_LIST"])
    if "_FillValue" in d.attrs:
        rec["fill"] = float(d.attrs["_FillValue"])
    out["vars"][k] = rec
p = f["prec"]
out["attached"] = [bool(h5ds.is_attached(p.id, f[n].id, ax)) for ax, n in enumerate(("Time", "y", "x"))]
print(json.dumps(out))
"""


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="needs an interpreter with h5py (build container only)")
def test_h5py_sees_a_netcdf4_file(tmp_path):
    ref = _sample(tmp_path / "c.nc", 150)
    r = subprocess.run([CONDA_PY, "-c", _H5PY_CHECK, str(tmp_path / "c.nc")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    v = got["vars"]
    assert got["root_attrs"]["setup_identifier"] == "SVAT" and got["attached"] == [True, True, True]
    assert v["Time"]["is_scale"] and v["Time"]["maxshape"] == [None] and v["Time"]["NAME"] == "Time" and v["Time"]["units"] == "days"
    assert [v[k]["dimid"] for k in ("x", "y", "Time", "ages")] == [0, 1, 2, 3]
    assert v["ages"]["NAME"].startswith("This is a netCDF dimension but not a netCDF variable.") and v["ages"]["shape"] == [3]
    assert v["prec"]["dims"] == [["/Time"], ["/y"], ["/x"]] and v["prec"]["maxshape"] == [None, 4, 5] and v["prec"]["chunks"] == [1, 4, 5]
    assert v["prec"]["fill"] == -9999.0 and v["prec"]["units"] == "mm/dt" and v["prec"]["dtype"] == "float64"
    assert v["tt_q_ss"]["dims"] == [["/Time"], ["/ages"], ["/y"], ["/x"]]
    assert v["x"]["referenced_by"] == [["/prec", 2], ["/tt_q_ss", 3]] and v["Time"]["referenced_by"] == [["/prec", 0], ["/tt_q_ss", 0]]
    for k in ("prec", "tt_q_ss", "x", "Time"):
        assert abs(v[k]["sum"] - float(ref[k].sum())) <= 1e-9 * max(1.0, abs(float(ref[k].sum()))), k
