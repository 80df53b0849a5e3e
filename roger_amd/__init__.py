"""roger_amd -- MI355X-native ("hip") backend for RoGeR's per-cell SVAT time step.

Public API mirrors the reference package (roger/__init__.py:22-100): runtime_settings,
runtime_state, roger_routine, roger_kernel, KernelOutput, RogerSetup, RogerState -- lazily
created, so `runtime_settings.update(backend="hip", ...)` can run before anything else is
imported.
"""
import logging as _logging

__version__ = "0.1.0"

_lazy = {}


def __getattr__(name):
    if name == "runtime_settings":
        if name not in _lazy:
            from .runtime import RuntimeSettings

            _lazy[name] = RuntimeSettings()
        return _lazy[name]
    if name == "runtime_state":
        if name not in _lazy:
            from .runtime import RuntimeState

            _lazy[name] = RuntimeState()
        return _lazy[name]
    if name == "logger":
        if name not in _lazy:
            _lazy[name] = _logging.getLogger("roger_amd")
        return _lazy[name]
    if name in ("roger_routine", "roger_kernel"):
        from . import routines

        return getattr(routines, name)
    if name in ("KernelOutput", "RogerState"):
        from . import state

        return getattr(state, name)
    if name == "RogerSetup":
        from .roger import RogerSetup

        return RogerSetup
    raise AttributeError(f"module 'roger_amd' has no attribute {name!r}")
