"""Host-side operator set for user hooks, mirroring roger/core/operators.py:150-166 (NumPy
semantics: `update` returns a modified copy, arrays are immutable values)."""
import numpy

from .. import runtime_settings

if runtime_settings.backend != "hip":
    raise ValueError(f"Unrecognized backend {runtime_settings.backend}")


class Index:
    __slots__ = ()

    @staticmethod
    def __getitem__(key):
        return key


def _writable_copy(arr):
    out = numpy.array(arr, copy=True)
    out.flags.writeable = True
    return out


def update(arr, at, to):
    out = _writable_copy(arr)
    out[at] = to
    out.flags.writeable = False
    return out


def update_add(arr, at, to):
    out = _writable_copy(arr)
    out[at] += to
    out.flags.writeable = False
    return out


def update_multiply(arr, at, to):
    out = _writable_copy(arr)
    out[at] *= to
    out.flags.writeable = False
    return out


def for_loop(lower, upper, body_fun, init_val):
    val = init_val
    for i in range(lower, upper):
        val = body_fun(i, val)
    return val


def where(*args):
    return numpy.where(*args)


def flush():
    """Fence asynchronous device work (reference: block_until_ready, operators.py:137-145)."""
    pass


at = Index()
numpy.seterr(all="ignore")
