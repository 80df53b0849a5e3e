"""Checkpoint / restart of a model state on the hip backend, the counterpart of roger/restart.py
(`write_restart`, `read_restart`): every variable of the registry and the time-stepping scalars.

The reference writes an HDF5 file with one dataset per `write_to_restart` variable under the group "core"
(roger/restart.py:32-67, 129-174); h5py is not part of this build's environment, so the container here is a
NumPy `.npz` archive with the same variable names and the reference's array shapes (ghost frame included).
A restart file written by one decomposition can be read by the same decomposition only (each rank writes its
own chunk, `<stem>.<rank>.npz` when more than one process runs).
"""
import numpy as np



def _path(path, rank, world):
    path = str(path)
    stem = path[:-4] if path.endswith(".npz") else path
    return f"{stem}.{rank}.npz" if world > 1 else f"{stem}.npz"


def _rank_world():
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def write_restart(state, path):
    """Download every device-backed variable and write it with the scalars; returns the file name."""
    vs = state.variables
    out = {}
    for key, var in state.var_meta.items():
        if var.dims is None:
            out["scalar__" + key] = np.asarray(getattr(vs, key))
        else:
            out[key] = np.asarray(getattr(vs, key))
    fname = _path(path, *_rank_world())
    np.savez_compressed(fname, **out)
    return fname


def read_restart(state, path):
    """Assign every variable found in the file (shape-checked by RogerVariables like any assignment); the device
    copies are refreshed before the next native call."""
    vs = state.variables
    with np.load(_path(path, *_rank_world())) as z, vs.unlock():
        for key in z.files:
            name = key[len("scalar__"):] if key.startswith("scalar__") else key
            if name not in state.var_meta:
                raise KeyError(f"restart file holds {name}, which this model does not have")
            val = z[key]
            if key.startswith("scalar__"):
                val = val.item() if val.ndim == 0 else val
            if name in ("tau", "taup1", "taum1"):
                continue
            setattr(vs, name, val)
