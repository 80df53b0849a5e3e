"""Reading the output files of roger_amd.diagnostics in tests: netCDF-4 through roger_amd.nc4lite (h5py / netCDF4 are not part of the
image), with the small part of scipy.io.netcdf_file's interface the tests use (`f.dimensions`, `f.variables[name][:]`, `.shape`,
`.dimensions`, `.units` ... as bytes like scipy returns them)."""
from roger_amd import h5lite, nc4lite


class _Var:
    def __init__(self, dims, a, attrs):
        self._a = a
        self.shape = a.shape
        self.dimensions = dims
        for k, v in attrs.items():
            setattr(self, k, v.encode() if isinstance(v, str) else v)

    def __getitem__(self, key):
        return self._a[key]


class _File:
    def __init__(self, path):
        assert h5lite.is_hdf5(str(path)), f"{path} is not an HDF5 / netCDF-4 file"
        rec = nc4lite.read(str(path))
        self.dimensions = rec["dimensions"]
        self.variables = {k: _Var(d, a, at) for k, (d, a, at) in rec["variables"].items()}
        for k, v in rec["attributes"].items():
            setattr(self, k, v.encode() if isinstance(v, str) else v)


    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def netcdf_file(path, mode="r", mmap=False):
    assert mode == "r"
    return _File(path)
