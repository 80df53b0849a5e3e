"""`@roger_routine` / `@roger_kernel`, mirroring roger/routines.py for the hip backend.

A routine takes the RogerState first, returns nothing, and may assign to `state.variables`
(unlocked for its duration).  A kernel must not mutate the state and returns a KernelOutput.
There is no tracing compiler here: a kernel whose name is registered in NATIVE_KERNELS is
replaced by its native entry point (the Python body is not executed) and returns a
DeviceResult; any other kernel is user code and runs on the host against the lazily
synchronised host mirrors.
"""
import functools
import inspect

from .state import DeviceResult, RogerState

# kernel function name -> (C entry point, variables it writes or None for "any")
NATIVE_KERNELS = {
    "after_timestep_kernel": ("rh_after_timestep", None),          # models/svat/svat.py:187
    "calc_parameters_surface_kernel": ("rh_params_surface", None),  # core/surface.py:74
}


def _find_state(args, kwargs):
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, RogerState):
            return a
    return None


def run_native(state, entry, writes=None):
    vs = state.variables
    if vs._probe is not None:   # a hook is being probed (roger_amd/hooks.py): the call is recorded, nothing runs
        from .hooks import record_kernel_call

        return record_kernel_call(state, entry, entry)
    vs.flush_to_device()
    state.backend_context.call(entry)
    vs.mark_device_newer(writes)
    return DeviceResult(writes or ())


def roger_routine(function=None, *, dist_safe=True, local_variables=()):
    """roger/routines.py:118-176.  dist_safe / local_variables are accepted for signature
    compatibility; with one process per GPU every routine runs on every rank on its own slab."""

    def inner(fn):
        narg = 1 if "self" in inspect.signature(fn).parameters else 0
        if narg >= len(inspect.signature(fn).parameters):
            raise TypeError("roger routines must take at least one argument")

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            state = args[narg] if len(args) > narg else None
            if not isinstance(state, RogerState):
                raise TypeError(f"Argument {narg} to this roger routine must be a RogerState object")
            timer = state.profile_timers[fn.__qualname__]
            out = None
            if state._variables is not None:
                with state.variables.unlock(), timer:
                    out = fn(*args, **kwargs)
            else:
                with timer:
                    out = fn(*args, **kwargs)
            if out is not None:
                import warnings

                warnings.warn(f"Routine {fn.__qualname__} returned object of type {type(out)}. "
                              "Return objects are silently dropped.")

        wrapper._is_roger_routine = True
        return wrapper

    return inner(function) if function is not None else inner


def roger_kernel(function=None, *, static_args=()):
    """roger/routines.py:239-272"""

    def inner(fn):
        params = inspect.signature(fn).parameters
        allowed = (inspect.Parameter.POSITIONAL_ONLY, inspect.Parameter.POSITIONAL_OR_KEYWORD)
        if any(p.kind not in allowed for p in params.values()):
            raise ValueError(f"roger kernels do not support *args, **kwargs, or keyword-only parameters ({fn.__name__})")
        for sa in ((static_args,) if isinstance(static_args, str) else static_args):
            if sa not in params:
                raise ValueError(f'roger kernel {fn.__name__} has no argument "{sa}", but it is given in static_args')

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            state = _find_state(args, kwargs)
            native = NATIVE_KERNELS.get(fn.__name__)
            # the native substitutes are the SVAT / oneD kernels; in an offline-transport run a kernel of that
            # name is the transport model's own (models/svat_oxygen18: tau -> taum1 of the snow signal) and runs on the host
            if native is not None and state is not None and not state.settings.enable_offline_transport:
                return run_native(state, *native)
            if state is not None and state._variables is not None and state._variables._probe is not None:
                from .hooks import record_kernel_call   # a host kernel inside a probed hook: recorded (the hook is the host's then)

                return record_kernel_call(state, fn.__name__, None)
            if state is not None and state._variables is not None:
                with state.variables.unlock():
                    return fn(*args, **kwargs)
            return fn(*args, **kwargs)

        wrapper._is_roger_kernel = True
        return wrapper

    return inner(function) if function is not None else inner


def is_roger_routine(func):
    func = getattr(func, "__func__", func)
    return bool(getattr(func, "_is_roger_routine", False))
