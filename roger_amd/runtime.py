"""Process-level runtime settings, mirroring roger/runtime.py.

Same names and semantics as the reference (`runtime_settings.update(backend=..., device=...,
float_type=..., num_proc=...)`, `ROGER_<NAME>` environment overrides, locking once the core is
imported), with one backend value: "hip".  The reference accepts ("numpy", "jax")
(roger/backend.py:3); INTEGRATION.md shows the one-line change that adds "hip" there.
"""
import os

BACKENDS = ("hip",)
DEVICES = ("gpu",)
FLOAT_TYPES = ("float64",)  # BASELINE configs are float64 (benchmarks/benchmark_base.py:11)
INT_TYPES = ("int64", "int32")


def _choice(choices):
    def validate(v):
        v = v.lower() if isinstance(v, str) else v
        if v not in choices:
            raise ValueError(f"must be one of {choices}")
        return v

    return validate


def _two_ints(v):
    return (int(v[0]), int(v[1]))


def _bool(v):
    return v.lower() in {"1", "true", "on"} if isinstance(v, str) else bool(v)


_AVAILABLE = {
    # name: (parser, default, read_from_env)
    "backend": (_choice(BACKENDS), "hip", True),
    "device": (_choice(DEVICES), "gpu", True),
    "float_type": (_choice(FLOAT_TYPES), "float64", True),
    "int_type": (_choice(INT_TYPES), "int32", True),
    "num_proc": (_two_ints, (1, 1), False),
    "profile_mode": (_bool, False, True),
    "loglevel": (str, "info", True),
    "force_overwrite": (_bool, False, True),   # roger/runtime.py:120-121
    "diskless_mode": (_bool, False, True),
    "monitor_water_balance": (_bool, False, True),
}


class RuntimeSettings:
    """roger/runtime.py:125-200"""

    def __init__(self, **kwargs):
        object.__setattr__(self, "__locked__", False)
        for name, (parser, default, from_env) in _AVAILABLE.items():
            if name in kwargs:
                val = kwargs[name]
            elif from_env:
                val = os.environ.get(f"ROGER_{name.upper()}", default)
            else:
                val = default
            object.__setattr__(self, name, parser(val))

    def update(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    def __setattr__(self, attr, val):
        if attr == "__locked__":
            return object.__setattr__(self, attr, val)
        if getattr(self, "__locked__", False):
            raise RuntimeError("Runtime settings cannot be modified after importing core modules")
        if attr not in _AVAILABLE:
            raise AttributeError(f"Unknown runtime setting {attr}")
        object.__setattr__(self, attr, _AVAILABLE[attr][0](val))

    def __repr__(self):
        return "RuntimeSettings(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in _AVAILABLE) + ")"


class RuntimeState:
    """roger/runtime.py:203-231: rank / size of the process group (torch.distributed, if initialised)."""

    @property
    def proc_rank(self):
        import torch.distributed as dist

        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @property
    def proc_num(self):
        import torch.distributed as dist

        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @property
    def proc_idx(self):
        from . import distributed, runtime_settings

        return distributed.proc_rank_to_index(self.proc_rank, runtime_settings.num_proc)

    @property
    def backend_module(self):
        import numpy

        return numpy  # host-side array module for user hooks
