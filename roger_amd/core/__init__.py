"""Process kernels of the hip backend: importing this package locks the runtime settings, as
importing roger.core does in the reference (roger/core/__init__.py:42-44)."""
from .. import runtime_settings as _rs

if not _rs.__locked__:
    _rs.__locked__ = True
