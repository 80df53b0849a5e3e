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


def _whole(at, ndim):
    key = at if isinstance(at, tuple) else (at,)
    return len(key) == ndim and all(isinstance(k, slice) and k == slice(None) for k in key)


def update(arr, at, to):
    """`arr.at[at].set(to)` as a value (roger/core/operators.py:21-46: copy, assign).  Assigning ONE small value to the whole array --
    the benchmark setups' `vs.prec_day = update(vs.prec_day, at[:, :, :], vs.PREC[npx.newaxis, npx.newaxis, i:i + 144])`,
    benchmarks/SVAT_benchmark.py:162-170 -- yields a read-only BROADCAST VIEW of a private copy of that value instead of nx * ny
    copies of it: the same values at every index, immutable like every array here, and `RogerVariables.flush_to_device` sees from
    the zero strides that all columns share the series (at 10^6 columns the three day arrays are 3.5 GB that nobody needs)."""
    arr = numpy.asarray(arr)
    if arr.ndim >= 2 and _whole(at, arr.ndim):
        val = numpy.asarray(to)
        if val.size * 64 <= arr.size:
            try:
                return numpy.broadcast_to(numpy.array(val, dtype=arr.dtype), arr.shape)   # (read-only by construction)
            except ValueError:
                pass   # not broadcastable: let the assignment below raise numpy's error
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
