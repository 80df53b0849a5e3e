"""roger_amd/csrc/rh_pow.h: the power function of the fused kernels (the library's pow, inlined a dozen times per column and step, was
half of the kernel's arithmetic).  The header is plain C with IEEE +, *, / and fma only, so gcc's compilation of it on the host has the
bits of the device's: the accuracy is measured HERE against the C library's pow (correctly rounded in nearly all cases) -- never more
than 1 ulp off over the domains the physics uses, as numpy's pow is against glibc's; exact where the result is exactly representable;
the library's pow for everything outside the short path --, and the device is compared with the host bit for bit (`-m gpu`)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SRC = r"""
#include <stdbool.h>
#include "rh_pow.h"
void host_rh_pow(const double *x, const double *y, double *out, unsigned char *short_path, long n) {
    for (long i = 0; i < n; ++i) {
        bool ok;
        (void)rh_pow_core(x[i], y[i], &ok);
        short_path[i] = ok;
        out[i] = rh_pow(x[i], y[i]);
    }
}
"""


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("rh_pow")
    (d / "h.c").write_text(_SRC)
    so = d / "libhost_rh_pow.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-I", os.path.join(REPO, "roger_amd", "csrc"), str(d / "h.c"), "-o", str(so), "-lm"],
                   check=True)
    lib = C.CDLL(str(so))

    def f(x, y):
        x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        out, sp = np.empty_like(x), np.empty(x.size, dtype=np.uint8)
        lib.host_rh_pow(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), sp.ctypes.data_as(C.c_void_p), C.c_long(x.size))
        return out, sp.astype(bool)

    return f


def _arguments(n, seed=5):
    """The domains of rh_physics.h: (theta / theta_sat) ** (-m_bc) and ** (1 / lambda_bc), Salvucci's terms, the cube root of the
    macropore front, ground cover / drainage exponents, and a wide log-uniform sweep."""
    rng = np.random.default_rng(seed)
    k = n // 5
    x = np.concatenate([rng.random(k), 1e-6 + 3 * rng.random(k), 1e4 * rng.random(k) * rng.random(k), np.exp(40 * (rng.random(k) - 0.5)), rng.random(n - 4 * k)])
    y = np.concatenate([-15 + 30 * rng.random(k), -10 * rng.random(k), np.full(k, 1.0 / 3), 20 * (rng.random(k) - 0.5),
                        rng.choice([1.5, 0.887, 2.0 / 3, 0.5], n - 4 * k)])
    return x, y


def _ulps(got, ref):
    with np.errstate(all="ignore"):
        return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_within_one_ulp_of_the_c_librarys_pow(host):
    x, y = _arguments(4_000_000)
    got, short = host(x, y)
    ref = np.power(x, y)        # (numpy's pow: itself within 1 ulp of glibc's; the bound below is against both being off in opposite directions)
    assert short.all()
    import math

    ref_c = np.array([math.pow(a, b) for a, b in zip(x[:200000], y[:200000])])     # glibc's, on a sample
    u = _ulps(got[:200000], ref_c)
    assert u.max() <= 1.0, (u.max(), x[np.argmax(u)], y[np.argmax(u)])
    assert (u > 0).mean() < 0.10                                                     # (numpy's own pow differs from glibc's in 5 % of arguments)
    assert _ulps(got, ref).max() <= 2.0


def test_exact_where_the_result_is_representable(host):
    x = np.array([1, 2, 4, 0.25, 8, 9, 0, 0, 5, 5, 0.1, 3, 1.5, 10, 2, 1e300, 1e-300])
    y = np.array([7.3, 3, 0.5, -0.5, 1 / 3, 0.5, 0.333, 0, 0, 1, 1, 2, 2, -2, -1074 + 1074, 1, 1])
    got, _ = host(x, y)
    import math

    np.testing.assert_array_equal(got, [math.pow(a, b) for a, b in zip(x, y)])
    assert got[4] == 2.0 and got[6] == 0.0 and got[7] == 1.0


def test_everything_else_goes_to_the_library(host):
    x = np.array([-8.0, -2.0, np.inf, np.nan, 1e-320, 0.0, -0.0, 2.0, 2.0, 1e200, 0.5])
    y = np.array([1 / 3, 2.0, 0.5, 1.0, 0.5, -1.0, 3.0, np.nan, np.inf, 5.0, 1e6])
    got, short = host(x, y)
    import math

    def cpow(a, b):
        try:
            return math.pow(a, b)
        except (OverflowError, ValueError):
            return float(np.power(np.float64(a), np.float64(b)))

    with np.errstate(all="ignore"):
        ref = np.array([cpow(a, b) for a, b in zip(x, y)])
    assert list(short) == [False, False, False, False, False, True, False, False, False, False, False]
    np.testing.assert_array_equal(got, ref)


@pytest.mark.gpu
def test_device_has_the_hosts_bits(host):
    from roger_amd import _native as N

    x, y = _arguments(2_000_000, seed=11)
    x = np.concatenate([x, [-8.0, 0.0, -0.0, np.inf, np.nan, 1e-320, 2.0, 4.0, 0.25]])
    y = np.concatenate([y, [1 / 3, 1 / 3, 3.0, 0.5, 1.0, 0.5, np.nan, 0.5, -0.5]])
    dev = N.selftest_pow(x, y)
    ref, short = host(x, y)
    same = (dev == ref) | (np.isnan(dev) & np.isnan(ref))
    assert same[short].all(), "the short path differs between host and device"
    # outside the short path both call their library's pow: equal up to the libraries' last bit
    with np.errstate(all="ignore"):
        assert (same | (_ulps(dev, ref) <= 1))[~short].all()
