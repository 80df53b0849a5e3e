// rh_pow.h -- x ** y for the physics of rh_physics.h: the library's pow is 230 instructions, inlined a dozen times per column and step
// it is HALF of the fused kernel's arithmetic.  This one is ~80: double-double log2 (atanh series on
// (m - 1) / (m + 1), m in [sqrt(1/2), sqrt(2))), the product with y in double-double, exp2 by a degree-13 polynomial.  Every operation is
// an IEEE +, *, / or fma, so the function has THE SAME BITS on the host and on the device (compiled with -ffp-contract=off): its accuracy
// is established on the host against glibc's pow (tests/test_rh_pow.py: <= 1 ulp over 10^7 arguments of the domains the physics uses,
// exact where the result is exactly representable), and the device is checked bit for bit against the host (rh_selftest_pow).
// Domain of the short path: x = +0, or x positive and normal with |y * log2 x| < 1000, y finite; anything else (negative or subnormal
// x, NaN, infinities, results near the exponent limits) goes to the library's pow.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef RH_POW_FN
#define RH_POW_FN static inline
#endif

// The polynomial coefficients sit in constant memory on the device so that they reach the FMAs as scalar-register operands (one
// v_fma_f64 per Horner step); written as literals each step costs two v_mov_b32 besides (43 of the function's 141 vector instructions).
#ifndef RH_POW_CONST
#define RH_POW_CONST static const
#endif
RH_POW_CONST double RH_POW_ATANH_C[12] = {2.0 / 25.0, 2.0 / 23.0, 2.0 / 21.0, 2.0 / 19.0, 2.0 / 17.0, 2.0 / 15.0, 2.0 / 13.0, 2.0 / 11.0, 2.0 / 9.0,
                                          2.0 / 7.0,  2.0 / 5.0,  2.0 / 3.0};
RH_POW_CONST double RH_POW_EXP_C[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                                        1.0 / 5040.0,       1.0 / 720.0,       1.0 / 120.0,      1.0 / 24.0,      1.0 / 6.0,      0.5};
RH_POW_CONST double RH_POW_K[4] = {1.4426950408889634, 2.0355273740931033e-17, 0.6931471805599453, 2.3190468138462996e-17};   // log2(e) and ln 2, head and tail

RH_POW_FN double rh_pow_bits_to_double(uint64_t b) {
    double d;
    memcpy(&d, &b, 8);
    return d;
}
RH_POW_FN uint64_t rh_pow_double_to_bits(double d) {
    uint64_t b;
    memcpy(&b, &d, 8);
    return b;
}

// the short path; `ok` = the argument pair is inside its domain (otherwise the value returned is meaningless)
RH_POW_FN double rh_pow_core(double x, double y, bool *ok) {
    const uint64_t bx = rh_pow_double_to_bits(x);
    const int ex = (int)((bx >> 52) & 0x7ff);
    const bool zero = bx == 0;                                             // +0
    const bool regular = (bx >> 63) == 0 && ex >= 1 && ex <= 2046;         // positive, normal, finite
    // x = 2^e * m, m in [sqrt(1/2), sqrt(2))
    double m = rh_pow_bits_to_double((bx & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    int e = ex - 1023;
    const bool upper = m > 1.4142135623730951;
    m = upper ? m * 0.5 : m;
    e = upper ? e + 1 : e;
    // s = (m - 1) / (m + 1) as s_hi + s_lo
    const double a = m - 1.0;                       // exact
    const double bh = m + 1.0;
    const double bl = m - (bh - 1.0);               // exact: m + 1 = bh + bl
    const double rb = 1.0 / bh;
    const double sh = a * rb;
    const double sl = (__builtin_fma(-sh, bh, a) - sh * bl) * rb;
    // ln m = 2 atanh(s) = 2 s + 2 s^3 (1/3 + s^2/5 + s^4/7 + ...), |s| <= 0.1716
    const double p = sh * sh;
    double q = RH_POW_ATANH_C[0];
#pragma unroll
    for (int k = 1; k < 12; ++k) q = __builtin_fma(q, p, RH_POW_ATANH_C[k]);
    const double lh = 2.0 * sh;                                    // head of ln m (exact doubling)
    const double ll = __builtin_fma(sh * p, q, 2.0 * sl);          // tail
    // log2 x = e + (lh + ll) * log2(e), log2(e) = L2E_H + L2E_L
    const double L2E_H = RH_POW_K[0], L2E_L = RH_POW_K[1];
    const double ph = lh * L2E_H;
    const double pl = __builtin_fma(lh, L2E_H, -ph) + (lh * L2E_L + ll * L2E_H);
    const double ed = (double)e;
    const double zh = ed + ph;                                     // |ph| <= 0.5: e == 0 or |e| >= 1 > |ph|
    const double zl = ((ed - zh) + ph) + pl;
    // y * log2 x in double-double
    const double th = y * zh;
    const double tl = __builtin_fma(y, zh, -th) + y * zl;
    // 2^(th + tl) = 2^n * exp(f ln 2), n = rint(th), f = (th - n) + tl
    const double n = rint(th);
    const double f = (th - n) + tl;
    const double LN2_H = RH_POW_K[2], LN2_L = RH_POW_K[3];
    const double u = __builtin_fma(f, LN2_H, f * LN2_L);
    double r = RH_POW_EXP_C[0];                                    // 1/13! ... 1/2!
#pragma unroll
    for (int k = 1; k < 12; ++k) r = __builtin_fma(r, u, RH_POW_EXP_C[k]);
    r = __builtin_fma(r, u, 1.0);
    r = __builtin_fma(r, u, 1.0);
    const bool inrange = (th > -1000.0) && (th < 1000.0);          // (false for NaN: y or the product not finite)
    const int64_t ni = inrange ? (int64_t)n : 0;
    const double res = rh_pow_bits_to_double(rh_pow_double_to_bits(r) + ((uint64_t)ni << 52));   // r in [0.70, 1.42): scaling by 2^n is exact
    // x = +0: 0 for y > 0, 1 for y == 0, +inf for y < 0 (NaN y: outside the domain)
    const double at_zero = y > 0 ? 0.0 : (y == 0 ? 1.0 : INFINITY);
    *ok = zero ? (y == y) : (regular && inrange);
    return zero ? at_zero : res;
}

RH_POW_FN double rh_pow(double x, double y) {
    bool ok;
    const double v = rh_pow_core(x, y, &ok);
    return ok ? v : pow(x, y);
}
