"""roger_amd.h5lite, the small HDF5 writer / reader behind the restart files where h5py is not installed: what it writes is read
back bit for bit, and -- in the build container, whose second interpreter has h5py -- h5py reads what it writes and it reads what
h5py writes the way the reference does (chunked, gzip, boolean masks; roger/restart.py:32-67)."""
import os
import subprocess

import numpy as np
import pytest

from roger_amd import h5lite

H5PY_PYTHON = "/opt/conda/bin/python3.9"


def _groups():
    rng = np.random.default_rng(0)
    core = {"S_rz": rng.normal(size=(7, 6, 2)), "lu_id": rng.integers(0, 20, (7, 6)).astype(np.int32), "itt": np.array(17, dtype=np.int64),
            "dt": np.array(1.0 / 6), "maskCatch": rng.random((7, 6)) > 0.3, "f32": rng.normal(size=(5,)).astype(np.float32),
            "empty": np.zeros((0, 3))}
    core.update({f"v{i:03d}": rng.normal(size=(3, 4)) for i in range(300)})   # a group as large as the model's "core"
    return {"core": core, "rate": {"prec": rng.normal(size=(7, 6))}}


def test_round_trip(tmp_path):
    g = _groups()
    h5lite.write(tmp_path / "a.h5", g)
    back = h5lite.read(tmp_path / "a.h5")
    assert set(back) == set(g)
    for gn in g:
        assert set(back[gn]) == set(g[gn])
        for k, v in g[gn].items():
            v = np.asarray(v)
            want = v.astype(np.int8) if v.dtype == np.bool_ else v
            assert back[gn][k].dtype == want.dtype and back[gn][k].shape == want.shape, k
            np.testing.assert_array_equal(back[gn][k], want, err_msg=k)
    with pytest.raises(ValueError):
        (tmp_path / "junk").write_bytes(b"not hdf5" * 20)
        h5lite.read(tmp_path / "junk")


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py here")
def test_against_h5py(tmp_path):
    g = _groups()
    h5lite.write(tmp_path / "mine.h5", g)
    np.savez(tmp_path / "expect.npz", S_rz=g["core"]["S_rz"], lu_id=g["core"]["lu_id"], mask=g["core"]["maskCatch"])
    script = f"""
import h5py, numpy as np
e = np.load(r"{tmp_path}/expect.npz")
with h5py.File(r"{tmp_path}/mine.h5", "r") as f:                       # h5py reads what h5lite wrote
    assert sorted(f) == ["core", "rate"] and len(f["core"]) == 307
    assert np.array_equal(f["core/S_rz"][...], e["S_rz"]) and np.array_equal(f["core/lu_id"][...], e["lu_id"])
    assert f["core/itt"][()] == 17 and f["core/dt"][()] == 1.0 / 6 and f["core/empty"].shape == (0, 3)
    assert np.array_equal(f["core/maskCatch"][...].astype(bool), e["mask"])
with h5py.File(r"{tmp_path}/theirs.h5", "w") as f:                     # and writes the way roger/restart.py:32-67 does
    grp = f.require_group("core")
    for key, a, chunks in (("S_rz", e["S_rz"], (4, 3, 1)), ("lu_id", e["lu_id"], (4, 3)), ("maskCatch", e["mask"], (4, 3))):
        grp.require_dataset(key, a.shape, a.dtype, exact=True, chunks=chunks, compression="gzip", compression_opts=1)
        grp[key][...] = a
    grp.require_dataset("itt", (), "int64", exact=True)
    grp["itt"][()] = 42
    for i in range(60):
        grp.create_dataset(f"x{{i}}", data=np.full(3, float(i)))
    f.require_group("rate").create_dataset("prec", data=e["S_rz"][..., 0], chunks=(2, 2), compression="gzip", shuffle=True)
"""
    r = subprocess.run([H5PY_PYTHON, "-c", script], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    back = h5lite.read(tmp_path / "theirs.h5")
    np.testing.assert_array_equal(back["core"]["S_rz"], g["core"]["S_rz"])
    np.testing.assert_array_equal(back["core"]["lu_id"], g["core"]["lu_id"])
    np.testing.assert_array_equal(back["core"]["maskCatch"].astype(bool), g["core"]["maskCatch"])
    assert int(back["core"]["itt"]) == 42 and len(back["core"]) == 64
    np.testing.assert_array_equal(back["core"]["x37"], np.full(3, 37.0))
    np.testing.assert_array_equal(back["rate"]["prec"], g["core"]["S_rz"][..., 0])


NC4 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "netcdf4")


def test_reads_the_reference_s_netcdf4_output():
    """A netCDF-4 file written by the reference itself (doc/_data/SVAT.maximum.nc: h5netcdf 1.3.0 / hdf5 1.14.3 -- version-2 object
    headers, 25 links in a fractal heap, chunked + gzip data, dimension scales) comes back variable by variable as h5py reads it
    (tests/golden/make_netcdf4_fixture.py wrote the expectation)."""
    want = np.load(os.path.join(NC4, "SVAT.maximum.expected.npz"))
    got = h5lite.read_root(os.path.join(NC4, "SVAT.maximum.nc"))
    assert h5lite.is_hdf5(os.path.join(NC4, "SVAT.maximum.nc"))
    assert set(got) == set(want.files) and len(got) == 25
    for k in want.files:
        assert got[k].shape == want[k].shape and got[k].dtype == want[k].dtype, k
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert got["Time"].shape == (366,) and got["S_snow"].shape == (366, 1, 1) if "S_snow" in got else True


def test_reads_every_netcdf4_file_the_reference_ships():
    """Build container only: all six output files under the reference's doc/_data against h5py."""
    ref = "/root/reference/doc/_data"
    if not (os.path.isdir(ref) and os.path.exists(H5PY_PYTHON)):
        pytest.skip("needs the reference tree and the interpreter with h5py")
    code = "\n".join([
        "import h5py, numpy as np, sys, glob, os",
        "for f in sorted(glob.glob(sys.argv[1] + '/*.nc')):",
        "    with h5py.File(f, 'r') as h:",
        "        np.savez(os.path.join(sys.argv[2], os.path.basename(f) + '.npz'), **{k: np.asarray(v) for k, v in h.items() if isinstance(v, h5py.Dataset)})",
    ])
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([H5PY_PYTHON, "-W", "ignore", "-c", code, ref, tmp], check=True)
        files = sorted(f for f in os.listdir(ref) if f.endswith(".nc"))
        assert len(files) >= 6
        for f in files:
            want, got = np.load(os.path.join(tmp, f + ".npz")), h5lite.read_root(os.path.join(ref, f))
            assert set(got) == set(want.files), f
            for k in want.files:
                assert got[k].dtype == want[k].dtype, (f, k)
                np.testing.assert_array_equal(got[k], want[k], err_msg=f"{f}: {k}")


def test_forcing_from_nc(monkeypatch):
    """roger_amd.forcing.forcing_from_nc on a forcing.nc with write_forcing's variables in h5netcdf's HDF5 structures (fixture written
    with h5py, track_order groups and dimension scales: tests/golden/make_netcdf4_fixture.py) gives what forcing_from_txt gives for the
    text files it was made from (the tutorial's first 30 days)."""
    import builtins

    from roger_amd.forcing import forcing_from_nc, forcing_from_txt

    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == "h5py":
            raise ImportError("h5py hidden: the h5lite path is under test")
        return real_import(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_h5py)
    F = forcing_from_nc(os.path.join(NC4, "forcing_like.nc"))
    want = forcing_from_txt(os.path.join(os.path.dirname(NC4), "tutorial_input"))
    assert set(F) == set(want)
    for k in want:
        assert F[k].dtype == want[k].dtype and F[k].shape == want[k].shape, k
        np.testing.assert_array_equal(F[k], want[k], err_msg=k)
    ten = forcing_from_nc(os.path.join(NC4, "forcing_like.nc"), ndays=10)
    assert ten["PREC"].size == 1440 and np.array_equal(ten["DOY"], want["DOY"][:1440])
