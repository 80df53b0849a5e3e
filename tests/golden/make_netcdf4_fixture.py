#!/usr/bin/env python3
"""Expected contents of the netCDF-4 fixture tests/golden/netcdf4/SVAT.maximum.nc (an output file of the reference itself, shipped
under doc/_data: written through h5netcdf 1.3.0 / hdf5 1.14.3), read with h5py -- what roger_amd.h5lite has to return for it.  Also
writes forcing_like.nc: the tutorial's 30-day forcing in the HDF5 structures h5netcdf gives `write_forcing`'s file (roger/tools/setup.py:
565-626: groups with creation-order tracking, dimension scales), for roger_amd.forcing.forcing_from_nc.

Run with an interpreter that has h5py (the build container's second one):   /opt/conda/bin/python3.9 tests/golden/make_netcdf4_fixture.py"""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
with h5py.File(os.path.join(HERE, "netcdf4", "SVAT.maximum.nc"), "r") as f:
    for k, v in f.items():
        if isinstance(v, h5py.Dataset):
            out[k] = np.asarray(v)
np.savez_compressed(os.path.join(HERE, "netcdf4", "SVAT.maximum.expected.npz"), **out)
print("SVAT.maximum.nc:", len(out), "variables")

if len(sys.argv) > 1:   # the forcing arrays, as an .npz written by the main interpreter (forcing_from_txt on tests/golden/tutorial_input)
    F = np.load(sys.argv[1])
    n = F["PREC"].size
    with h5py.File(os.path.join(HERE, "netcdf4", "forcing_like.nc"), "w", track_order=True) as f:
        f.attrs["title"] = "Meteorological forcing"
        dims = {}
        for name, size in (("x", 1), ("y", 1), ("Time", n), ("scalar", 1)):
            dims[name] = f.create_dataset(name, (size,), dtype="f4" if name == "Time" else "i8", track_order=True)
            dims[name].make_scale(name)
        dims["Time"][:] = np.arange(n, dtype="f4") / 6.0
        for k in ("PREC", "TA", "PET"):
            v = f.create_dataset(k, (1, 1, n), dtype="f4", track_order=True)
            v[0, 0, :] = F[k].astype("f4")
            for i, dname in enumerate(("x", "y", "Time")):
                v.dims[i].attach_scale(dims[dname])
        for k in ("YEAR", "MONTH", "DOY"):
            v = f.create_dataset(k, (n,), dtype="i8", track_order=True)
            v[:] = F[k]
            v.dims[0].attach_scale(dims["Time"])
        v = f.create_dataset("dt", (n,), dtype="f4", track_order=True)
        v[:] = 600
    print("forcing_like.nc:", n, "slots")
