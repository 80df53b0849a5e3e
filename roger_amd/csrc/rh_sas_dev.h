// rh_sas_dev.h -- device-side building blocks shared by the SAS kernels: rh_sas_kernels.h (deterministic solver) and
// rh_sas_solvers_impl.h (explicit Euler / RK4 solvers).  Everything here is inlined into the kernels of the including
// translation unit; constants and the one out-of-line function are `static` so that the two units do not clash at link time.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "roger_hip.h"
#include "roger_hip_sas.h"

#define SAS_DEV __device__ __forceinline__
// the workgroup barrier of the cross-wave reductions (-DRH_SAS_NO_BARRIER: timing experiments only, the results are then wrong)
#ifdef RH_SAS_NO_BARRIER
#define SAS_SYNC() ((void)0)
#else
#define SAS_SYNC() __syncthreads()
#endif

// (SA / S) ** k of the power-law SAS function, the hot spot of the kernel: 5 * substeps * (ages + 1)
// evaluations per column and day.  The device library's general pow() costs ~230 VALU instructions
// here (measured: 2/3 of the kernel's instruction stream).  The argument range is narrow -- 0 < SA <= S,
// k finite -- so (SA / S)**k = 2**(k * (log2 SA - log2 S)) is evaluated directly in ~50 instructions,
// and the division goes away as well (log2 S is computed once per sub-step):
//   sas_log2:  x = m * 2**e, m in [sqrt(1/2), sqrt(2));  s = (m - 1) / (m + 1);
//              ln m = s * (2 + z * (2/3 + 2/5 z + ... + 2/19 z**8)), z = s*s <= 0.02944 (next term < 2.4e-17 rel.)
//   sas_exp2:  y = n + r, |r| <= 1/2;  2**r = exp(r ln 2) by its Taylor series to degree 13 (remainder < 4e-18);
//              result = ldexp(., n)
// Error: the rounding of the logarithms dominates, ~|log2 SA| * 2**-53 * k * ln 2 relative, i.e. < 1e-14 * k for
// SA / S > 1e-21; SA == S gives exactly 1 (Omega(S) = 1).  RH_SAS_POW=0 selects the library pow(SA / S, k).
#ifndef RH_SAS_POW
#define RH_SAS_POW 3
#endif
// Division by a divisor that is uniform over many quotients (flux * h inside the sub-step loop): the compiler's
// IEEE division is  rcp -> two Newton steps on the reciprocal -> q0 = a * r -> e = fma(-d, q0, a) -> fma(e, r, q0)
// wrapped in v_div_scale / v_div_fixup for operands near the exponent limits.  With the refined reciprocal hoisted
// out of the loop a quotient costs three instructions instead of twelve and has the same bits as `a / d` whenever
// no scaling is needed (d and a / d within ~1e+-290, true for millimetres per day).
struct UDiv {
    double d, r;
};
SAS_DEV UDiv udiv_prepare(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    return UDiv{d, r};
}
SAS_DEV double udiv(double a, const UDiv &u) {
    const double q0 = a * u.r;
    return __builtin_fma(__builtin_fma(-u.d, q0, a), u.r, q0);
}

// Square root of r in [2^-700, 1] -- the quotient SA / S of the power law with exponent 0.5 / 1.5.  The compiler's sqrt wraps this very
// sequence (v_rsq_f64, Goldschmidt step, two residual corrections: correctly rounded) in a rescaling for arguments below 2^-767 and a
// class test for 0 / inf / NaN, twenty instructions in all; inside the range neither can trigger, ten remain.  r == 0 never reaches the
// result (the caller selects 0 for SA <= 0).  Checked against the host's sqrt through rh_sas_selftest_pow (tests/test_hip_sas.py).
SAS_DEV double sqrt_unit(double r) {
    const double y = __builtin_amdgcn_rsq(r);
    double g = r * y, h = y * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    g = __builtin_fma(__builtin_fma(-g, g, r), h, g);
    return __builtin_fma(__builtin_fma(-g, g, r), h, g);
}

// min(a, b) as the one instruction it is.  fmin() makes the compiler quiet a possible signalling NaN in an operand that comes straight from
// memory (a v_max_f64 x, x in front of every v_min_f64 of the sub-step loop); the age vectors hold no signalling NaNs -- every value was
// produced by arithmetic, the NaN markers of empty classes are quiet.
SAS_DEV double min_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// r ** 0.2 for r in [2^-127, 1] (below, the single-precision log2 r is too coarse a start: 2^-17 absolute at 2^-128) -- the quotient SA / S under the benchmark's exponent of soil evaporation and capillary rise
// (benchmarks/SVATOXYGEN18_benchmark.py:129-132) -- as a fifth root: 31 instructions instead of the 46 of exp2(k * log2 .), and within
// 0.52 ulp of the true power where that path is 5 - 10 ulp wide.
//   y0 = 2^(0.2 log2 r) in single precision (v_log_f32 / v_exp_f32: 20 bits);
//   ONE Halley step on y^5 = r,  y1 = y0 + y0 (r - y0^5) / (3 y0^5 + 2 r)  (cubic: 20 -> 60 bits), with y0^5 as a double-double product
//   so that the residual r - y0^5 (5e-6 r) keeps twelve digits;
//   the exponent is the DOUBLE 0.2 = 1/5 + 1.11e-17, not 1/5: r^0.2 = r^(1/5) (1 + 1.11e-17 ln r), which is up to an ulp for small r and is
//   added to the correction before the one final rounding.
// r == 1 gives exactly 1.  *e2 = the binary exponent of r (the caller's range check); r <= 0 or NaN give NaN or garbage, clipped by the caller.
SAS_DEV double pow_fifth(double r, int *e2) {
    const int e = __builtin_amdgcn_frexp_exp(r);
    const float mf = (float)__builtin_amdgcn_frexp_mant(r);          // [0.5, 1)
    const float L = __builtin_amdgcn_logf(mf) + (float)e;            // log2 r
    const double y0 = (double)__builtin_amdgcn_exp2f(0.2f * L);
    const double p2h = y0 * y0, p2l = __builtin_fma(y0, y0, -p2h);
    const double p4h = p2h * p2h, p4l = __builtin_fma(p2h + p2h, p2l, __builtin_fma(p2h, p2h, -p4h));
    const double p5h = p4h * y0, p5l = __builtin_fma(p4l, y0, __builtin_fma(p4h, y0, -p5h));
    const double d = (r - p5h) - p5l;                                // r - y0^5 (the first difference is exact)
    const double den = __builtin_fma(3.0, p5h, r + r);
    double rc = __builtin_amdgcn_rcp(den);                            // (v_rcp_f64 holds single precision: one Newton step, the correction
    rc = __builtin_fma(rc, __builtin_fma(-den, rc, 1.0), rc);        //  needs twelve digits)
    const double c2 = (double)L * 7.6954795931166195e-18;            // (0.2 - 1/5) * ln 2 * log2 r
    *e2 = e;
    return y0 + __builtin_fma(y0, c2, (y0 * d) * rc);
}

// Polynomial coefficients live in constant memory so that they reach the FMAs as scalar-register
// operands (one v_fma_f64 per Horner step); as immediates each step costs a 64-bit v_mov besides.
#include "rh_sas_tables.inc"
// RH_SAS_LOG: 1 = table-assisted log2 (64-entry table of {1/c, log2 c} in LDS, degree-8 log2(1 + r)); 0 = the
// table-free version (s = (m - 1) / (m + 1), odd series to s^19)
#ifndef RH_SAS_LOG
#define RH_SAS_LOG 1
#endif
static __constant__ double2 SAS_LOG_T[64] = {RH_SAS_LOG_TABLE};
static __constant__ double SAS_LOG1P_C[8] = {RH_SAS_LOG1P_COEF};
static __constant__ double SAS_LOG_C[9] = {2.0 / 19.0, 2.0 / 17.0, 2.0 / 15.0, 2.0 / 13.0, 2.0 / 11.0, 2.0 / 9.0, 2.0 / 7.0, 2.0 / 5.0, 2.0 / 3.0};
static __constant__ double SAS_EXP_C[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,
                                     1.0 / 40320.0,      1.0 / 5040.0,      1.0 / 720.0,      1.0 / 120.0,     1.0 / 24.0,
                                     1.0 / 6.0,          0.5};
struct PowConsts {
    double lc[9], ec[12];
    const double2 *logt;  // the log2 table (LDS copy in the step kernel)
};
SAS_DEV PowConsts load_pow_consts(const double2 *logt) {
    PowConsts c;
    c.logt = logt;
#if RH_SAS_LOG == 1
#pragma unroll
    for (int i = 0; i < 8; ++i) c.lc[i] = SAS_LOG1P_C[i];
    c.lc[8] = 0.0;
#else
#pragma unroll
    for (int i = 0; i < 9; ++i) c.lc[i] = SAS_LOG_C[i];
#endif
#pragma unroll
    for (int i = 0; i < 12; ++i) c.ec[i] = SAS_EXP_C[i];
    return c;
}
#if RH_SAS_LOG == 1
// log2 x = e + log2 c_i + log2(1 + r):  x = m * 2^e with m in [1, 2), i = the top six mantissa bits, c_i the centre of
// that sixty-fourth, r = m / c_i - 1 by one fma on the tabulated reciprocal (|r| <= 1/128; the table's log2 c_i is
// -log2 of that very reciprocal, so the split is exact), log2(1 + r) by its series to r^8.
SAS_DEV double sas_log2(const PowConsts &C, double x) {
    const int e = __builtin_amdgcn_frexp_exp(x) - 1;
    const double m = __builtin_amdgcn_frexp_mant(x) * 2.0;  // [1, 2)
    const int i = (int)((unsigned)(__double_as_longlong(m) >> 46) & 63u);
    const double2 t = C.logt[i];
    const double r = __builtin_fma(m, t.x, -1.0);
    double p = C.lc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) p = __builtin_fma(p, r, C.lc[k]);
    return ((double)e + t.y) + p * r;
}
#else
SAS_DEV double sas_log2(const PowConsts &C, double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0, d = m + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = __builtin_fma(__builtin_fma(-d, s, f), r, s);
    const double z = s * s;
    double p = C.lc[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) p = __builtin_fma(p, z, C.lc[i]);
    const double lnm = s * __builtin_fma(p, z, 2.0);
    return __builtin_fma(lnm, 1.44269504088896340736, (double)e);
}
#endif
// 2**y.  No range clamp is needed: v_cvt_i32_f64 saturates and v_ldexp_f64 under/overflows to 0 / inf.
SAS_DEV double sas_exp2(const PowConsts &C, double y) {
    const double n = __builtin_rint(y);
    const double w = (y - n) * 0.69314718055994530942;
    double q = C.ec[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) q = __builtin_fma(q, w, C.ec[i]);
    q = __builtin_fma(q, w, 1.0);
    q = __builtin_fma(q, w, 1.0);
    return ldexp(q, (int)n);
}
// (x / S) ** k for 0 < x <= S; log2S = sas_log2(S)
SAS_DEV double sas_pow_ratio(const PowConsts &C, double x, double S, double log2S, double k) {
#if RH_SAS_POW == 0
    return pow(x / S, k);
#else
    return sas_exp2(C, k * (sas_log2(C, x) - log2S));
#endif
}

enum SasArr {
#define RH_SAS_ARRAY(name, kind, when) SA_##name,
#include "rh_sas_arrays.def"
#undef RH_SAS_ARRAY
    SA_COUNT
};
enum SasKind { K_AGE, K_NAGE, K_CELL, K_DAILY, K_PARAM, K_MASK };
enum SasWhen { W_ALWAYS, W_STATS, W_DIAG, W_ANION };

struct SasArgs {
    int64_t n;
    int64_t day_off;  // row of the daily inputs * n
    int ages, substeps, stages, stats, diag, tracer;
    double vsmow, dmin, dmax;
    int *unsupported;  // device flag: a column asked for a SAS family this kernel does not implement
    void *a[SA_COUNT];
#ifdef RH_SAS_PHASES   // measurement builds: cycles per phase of the day, summed over the columns (thread 0 of every column)
    unsigned long long *phases;
#endif
};

// ---------------------------------------------------------------------------------------------
// workgroup primitives over the blocked age layout
// ---------------------------------------------------------------------------------------------
template <int W>
struct Blk {
    const double2 *logt;   // LDS copy of the log2 table
    int tid, lane, wave;
    unsigned phase;        // alternates the double-buffered LDS scratch; one barrier per use
    double (*red)[W][8];   // [2][W][8]
    double (*xch)[W][2];   // [2][W][2]
    double *park;          // LDS parking area of the eight-class shapes ([2][8][W * 64] doubles), else null
    const double *scal;    // LDS copy of the column's scalars of the day (SasScal)
#ifdef RH_SAS_PHASES
    unsigned long long t_last;
#endif
};
#ifdef RH_SAS_PHASES
#define SAS_PH(B, P, k)                                                              \
    do {                                                                             \
        if ((B).tid == 0) {                                                          \
            const unsigned long long t_ = clock64();                                 \
            atomicAdd(&(P).phases[k], t_ - (B).t_last);                              \
            (B).t_last = t_;                                                         \
        }                                                                            \
    } while (0)
#else
#define SAS_PH(B, P, k) ((void)0)
#endif
// The column's scalars of the day -- the five fluxes, the three infiltration terms, the input signal, seven SAS parameters per flux --
// fetched in ONE batch of independent loads at the start of the kernel and kept in LDS: read where they are needed, each is a dependent
// round trip to HBM in front of a branch (flux > 0?  which family?), two or three per flux, while both waves of the column wait.
enum SasScal { SC_FLUX = 0, SC_INF = 5, SC_CIN = 8, SC_PAR = 9, SC_COUNT = SC_PAR + 5 * 7 };
// the column's scalars (SasScal) in two parts, so that the loads of the column's state can be requested in between and both round trips
// to HBM overlap: sas_fetch_scalars requests them (independent loads), sas_publish_scalars stores them to LDS; the caller's barrier
// makes them visible
SAS_DEV void sas_fetch_scalars(const SasArgs &P, double (&v)[SC_COUNT]) {
    const int64_t c = blockIdx.x, dc = P.day_off + c;
#pragma unroll
    for (int f = 0; f < 5; ++f) v[SC_FLUX + f] = ((const double *)P.a[SA_evap_soil + f])[dc];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[SC_INF + k] = ((const double *)P.a[SA_inf_mat_rz + k])[dc];
    v[SC_CIN] = ((const double *)P.a[SA_C_in])[dc];
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
        for (int i = 0; i < 7; ++i) v[SC_PAR + 7 * f + i] = ((const double *)P.a[SA_sas_params_evap_soil + f])[c * 8 + i];
}
SAS_DEV void sas_publish_scalars(const double (&v)[SC_COUNT], double *s_scal) {
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < SC_COUNT; ++k) s_scal[k] = v[k];
    }
}

// Cross-lane moves as DPP (data-parallel primitive) modifiers on VALU moves instead of LDS-crossbar
// shuffles: a DPP move costs one VALU issue, a ds_bpermute a round trip through the LDS pipeline, and
// the scans below are dependent chains of them.  gfx9 controls: row_shr:n = 0x110 + n (shift inside a
// row of 16 lanes), wave_shr:1 = 0x138, row_bcast:15 = 0x142 (lane 15 of a row to the next row),
// row_bcast:31 = 0x143 (lane 31 to rows 2 and 3).  Lanes without a source keep `ident`.
template <int CTRL, int ROW_MASK>
SAS_DEV double dpp_move(double ident, double v) {
    const unsigned long long iv = __double_as_longlong(ident), sv = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)iv, (int)(unsigned)sv, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(iv >> 32), (int)(unsigned)(sv >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// the same with 0 for the lanes without a source (bound_ctrl: the hardware supplies the zero, no register has to be cleared first);
// only for controls that write every row (row mask 0xf)
template <int CTRL>
SAS_DEV double dpp_move0(double v) {
    const unsigned long long sv = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)sv, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(sv >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// value of the previous lane; lane 0 gets `first`
SAS_DEV double lane_prev(double v, double first) { return dpp_move<0x138, 0xf>(first, v); }
SAS_DEV double lane_prev0(double v) { return dpp_move0<0x138>(v); }   // ... gets 0
// inclusive prefix sum over the 64 lanes (earlier lanes + own)
SAS_DEV double wave_scan_sum(double v) {
    v = dpp_move0<0x111>(v) + v;
    v = dpp_move0<0x112>(v) + v;
    v = dpp_move0<0x114>(v) + v;
    v = dpp_move0<0x118>(v) + v;
    v = dpp_move<0x142, 0xa>(0.0, v) + v;
    v = dpp_move<0x143, 0xc>(0.0, v) + v;
    return v;
}
SAS_DEV double lane63(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// max over the 64 lanes, in every lane
SAS_DEV double wave_max(double v) {
    const double ninf = -INFINITY;
    v = fmax(dpp_move<0x111, 0xf>(ninf, v), v);
    v = fmax(dpp_move<0x112, 0xf>(ninf, v), v);
    v = fmax(dpp_move<0x114, 0xf>(ninf, v), v);
    v = fmax(dpp_move<0x118, 0xf>(ninf, v), v);
    v = fmax(dpp_move<0x142, 0xa>(ninf, v), v);
    v = fmax(dpp_move<0x143, 0xc>(ninf, v), v);
    return lane63(v);
}
// sum over the 64 lanes, in every lane
SAS_DEV double wave_sum(double v) { return lane63(wave_scan_sum(v)); }

// Parking: two age vectors of a thread leave the register file for LDS while the kernel works on the other compartment (the eight-class
// shapes: the state of a column is 64 registers per thread, the sub-step loop needs only the StorAge it works on).  Every thread reads back
// what it wrote itself -- no barrier; slot [a][j][tid], so the lanes of a wave touch consecutive words.  The other shapes (E != 8) have no
// parking area: both calls do nothing and the vectors stay in registers.
template <int W, int E>
SAS_DEV void park2(const Blk<W> &B, const double (&a)[E], const double (&b)[E]) {
    if (E < 8) return;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        B.park[(0 * E + j) * (W * 64) + B.tid] = a[j];
        B.park[(1 * E + j) * (W * 64) + B.tid] = b[j];
    }
}
template <int W, int E>
SAS_DEV void unpark2(const Blk<W> &B, double (&a)[E], double (&b)[E]) {
    if (E < 8) return;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        a[j] = B.park[(0 * E + j) * (W * 64) + B.tid];
        b[j] = B.park[(1 * E + j) * (W * 64) + B.tid];
    }
}

// value of the previous thread (thread 0: `first`), two values per call
template <int W>
SAS_DEV void blk_prev2(Blk<W> &B, double a, double b, double a0, double b0, double &pa, double &pb) {
    pa = lane_prev(a, a0);
    pb = lane_prev(b, b0);
    if (W > 1) {
        const int buf = B.phase++ & 1;
        if (B.lane == 63) {
            B.xch[buf][B.wave][0] = a;
            B.xch[buf][B.wave][1] = b;
        }
        SAS_SYNC();
        if (B.lane == 0 && B.wave > 0) {
            pa = B.xch[buf][B.wave - 1][0];
            pb = B.xch[buf][B.wave - 1][1];
        }
    }
    if (B.tid == 0) {
        pa = a0;
        pb = b0;
    }
}

// sums of N <= 8 per-thread values over the workgroup, result in every thread
template <int W, int N>
SAS_DEV void blk_sum(Blk<W> &B, double (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
    if (W > 1) {
        const int buf = B.phase++ & 1;
        if (B.lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) B.red[buf][B.wave][i] = v[i];
        }
        SAS_SYNC();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = B.red[buf][0][i];
            for (int w = 1; w < W; ++w) s = s + B.red[buf][w][i];
            v[i] = s;
        }
    }
}
template <int W>
SAS_DEV double blk_max(Blk<W> &B, double v) {
    v = wave_max(v);
    if (W > 1) {
        const int buf = B.phase++ & 1;
        if (B.lane == 0) B.red[buf][B.wave][0] = v;
        SAS_SYNC();
        v = B.red[buf][0][0];
        for (int w = 1; w < W; ++w) v = fmax(v, B.red[buf][w][0]);
    }
    return v;
}

// Cumulative sum over the age axis (calc_SA :343-359, the cumsums of calc_tt :456-468).
//   hi[j] = cumulative value at the upper edge of the thread's j-th age class
//   lo    = cumulative value at the lower edge of its first class
//   *ptop = the value at the top of the stored water, if asked for: hi of the LAST age class THAT MOVES THE SUM
//           (non-empty and not absorbed by rounding), which is also written into hi of every class above it.  It stands for `npx.max(SA, axis=-1)`: a sequential
//           cumsum of non-negative terms is non-decreasing, its maximum is its last element, and every class above
//           the last non-empty one repeats that element EXACTLY.  The SAS functions rely on it: Omega jumps to 1
//           where SA == S (by 1 - exp(-a) for the exponential family; a kumaraswami exponent < 1 turns a one-ulp
//           gap into 1e-9).  The parallel scan is only consistent inside a thread, so the equality is restored
//           explicitly instead of taking a maximum over the lanes.
// Construction: loc = running sum inside the thread, wexc = exclusive wave scan of the thread totals,
// pw = running sum of the totals of the preceding waves; hi[j] = pw + (wexc + loc[j]), lo = pw + wexc.
// fl(x + .) is monotone, so hi is non-decreasing in j for non-negative input and hi[j] == hi[j-1] (or lo)
// exactly where the input is 0.
// EXACT_TOP: restore the exact equality above the last class that moves the sum (two more wave reductions: needed
// where Omega is discontinuous or infinitely steep at S); otherwise *ptop = hi of class `top_k` = ages - 1.
template <int W, int E, bool EXACT_TOP>
SAS_DEV void blk_cumsum(Blk<W> &B, const double (&v)[E], double (&hi)[E], double &lo, double *ptop, int base, int top_k) {
    double loc[E];
    loc[0] = v[0];
#pragma unroll
    for (int j = 1; j < E; ++j) loc[j] = loc[j - 1] + v[j];
    const double winc = wave_scan_sum(loc[E - 1]);
    const double wexc = lane_prev0(winc);
    double u[E];
#pragma unroll
    for (int j = 0; j < E; ++j) u[j] = wexc + loc[j];
    double ktop = -1.0, utop = 0.0;  // the wave's top class (as a double: exact for indices) and its u
    if (ptop && EXACT_TOP) {
        double kmine = -1.0, umine = 0.0;
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (u[j] != (j == 0 ? wexc : u[j > 0 ? j - 1 : 0])) {  // the class moves the cumulative sum: a residue of
                kmine = (double)(base + j);                         // 1e-17 mm under 100 mm is absorbed, as in the
                umine = u[j];                                       // reference's sequential cumsum
            }
        ktop = wave_max(kmine);
        utop = wave_max(kmine == ktop && ktop >= 0 ? umine : -INFINITY);  // exactly one lane holds class ktop
    } else if (ptop) {
        // (the classes above top_k = ages - 1 hold exact zeros -- padding, which no flux can fill --, so inside the thread that owns top_k
        //  the running sum repeats itself bit for bit from there on: its LAST value is the value at top_k, no selection by top_k % E)
        const int top_thread = top_k / E;  // uniform
        const double mine = u[E - 1];
        const unsigned long long b = __double_as_longlong(mine);
        const int src = top_thread & 63;
        const unsigned lo32 = __builtin_amdgcn_readlane((int)(unsigned)b, src), hi32 = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
        utop = __longlong_as_double(((unsigned long long)hi32 << 32) | lo32);  // meaningful in the owning wave
        ktop = (B.wave == (top_thread >> 6)) ? (double)top_k : -1.0;
    }
    if (W == 1) {
        const double S = (ktop >= 0 ? utop : 0.0);
#pragma unroll
        for (int j = 0; j < E; ++j) hi[j] = (ptop && EXACT_TOP && (double)(base + j) >= ktop) ? S : u[j];
        lo = wexc;
        if (ptop) *ptop = S;
        return;
    }
    const int buf = B.phase++ & 1;
    if (B.lane == 63) {
        B.red[buf][B.wave][0] = winc;
        B.red[buf][B.wave][1] = ktop;
        B.red[buf][B.wave][2] = utop;
    }
    SAS_SYNC();
    double pw = 0.0, mine = 0.0, S = 0.0, kglob = -1.0;
    for (int w = 0; w < W; ++w) {
        if (w == B.wave) mine = pw;
        if (ptop && B.red[buf][w][1] > kglob) {  // the owner's own hi = its prefix + its u
            kglob = B.red[buf][w][1];
            S = pw + B.red[buf][w][2];
        }
        pw = pw + B.red[buf][w][0];
    }
#pragma unroll
    for (int j = 0; j < E; ++j) hi[j] = (ptop && EXACT_TOP && (double)(base + j) >= kglob) ? S : mine + u[j];
    lo = mine + wexc;
    if (ptop) *ptop = S;
}

// ---------------------------------------------------------------------------------------------
// per-column physics
// ---------------------------------------------------------------------------------------------
// conc_to_delta :328-340
SAS_DEV double conc_to_delta(const SasArgs &P, double conc) {
    const double d = 1000. * (conc / (P.vsmow * (1. - conc)) - 1.);
    return ((d < P.dmin) || (d > P.dmax)) ? NAN : d;
}

template <int E>
struct Dist {  // what the age statistics need of one flux
    double tt[E], TT_hi[E], TT_lo;
};

// Backward travel time distribution of one outgoing flux, calc_tt :362-509, with the SAS families `uniform`
// (code 1), `dirac` (2), `kumaraswami` (3, 31-37), `exponential` (51) and `power` (6, 61, 62) of core/sas.py.
// The reference adds the masked results of all six families; every family contributes exact zeros for the
// codes of the others, so the sum is the selected one.
enum SasFamily { FAM_NONE, FAM_UNIFORM, FAM_DIRAC, FAM_KUMARASWAMI, FAM_EXPONENTIAL, FAM_POWER, FAM_GAMMA };

// Regularised lower incomplete gamma function P(a, x) = scipy.special.gammainc(a, x) (the gamma SAS family, sas.py:153):
// power series for x < a + 1, continued fraction of Q = 1 - P (modified Lentz) otherwise; lgam = lgamma(a).
static __device__ __attribute__((noinline)) double sas_gammainc(double a, double x, double lgam) {
    if (!(x > 0) || !(a > 0)) return 0.0;
    const double lead = exp(a * log(x) - x - lgam);
    if (x < a + 1) {
        double ap = a, del = 1 / a, sum = del;
        for (int n = 0; n < 2000; ++n) {
            ap += 1;
            del *= x / ap;
            sum += del;
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        return sum * lead;
    }
    const double tiny = 1e-300;
    double b = x + 1 - a, c = 1 / tiny, d = 1 / b, h = d;
    for (int i = 1; i < 2000; ++i) {
        const double an = -(double)i * ((double)i - a);
        b += 2;
        d = an * d + b;
        if (fabs(d) < tiny) d = tiny;
        c = b + an / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1) < 1e-16) break;
    }
    return 1 - lead * h;
}

// Omega, the cumulative SAS function of one flux, at the upper edges of the thread's age classes (core/sas.py; called from calc_tt
// :362-509 and calc_TT_num :860-907): SA_hi = cumulative StorAge at those edges (already masked), Smax = its value at the top of the
// stored water (blk_cumsum's *ptop).  Om_edge0 = Omega at SA[0] = 0 is only written by the family that can make it non-zero (a dirac
// with a negative threshold).  One instantiation per family, selected per column (uniform over the workgroup).
template <int W, int E, int FAM>
SAS_DEV void sas_omega(Blk<W> &B, const PowConsts &C, const double (&p)[7], const double (&SA_hi)[E], double Smax, double mk, int base, int A,
                       double (&Om)[E], double &Om_edge0) {
    const double code = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6];
    constexpr bool uniform = FAM == FAM_UNIFORM, power = FAM == FAM_POWER, dirac = FAM == FAM_DIRAC;
    constexpr bool kumaraswami = FAM == FAM_KUMARASWAMI, expo = FAM == FAM_EXPONENTIAL, gamma = FAM == FAM_GAMMA;
    if (uniform) {
        const double S = Smax * 1.0 * mk;
        const double lam = 1 / S * 1.0 * mk;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            double o = (SA_hi[j] < S ? (SA_hi[j] > 0 ? lam * SA_hi[j] : 0.) : 1.) * 1.0 * mk;
            if (base + j == A - 1) o = 1 * mk;  // Omega[..., -1] = 1, sas.py:30-33
            Om[j] = (S <= 0 ? 0 : o) * mk;
        }
    } else if (power) {
        const double S = Smax * mk;
        double k = p1;
        if (code != 6) {  // storage-dependent exponent, sas.py:205-226
            double S_rel = (S - p5) / (p6 - p5) * mk;
            S_rel = (S_rel < 0 ? 0 : S_rel);
            S_rel = (S_rel > 1 ? 1 : S_rel);
            if (code == 61) k = p3 + ((1 - S_rel) * p4);
            if (code == 62) k = p3 + (S_rel * p4);
        }
        // Exponents with a closed form -- the benchmark's own: 0.5 for transpiration, 1.5 for percolation
        // (benchmarks/SVATOXYGEN18_benchmark.py:129-138) -- go through a correctly rounded square root of the true quotient
        // SA / S (exactly 1 at the top edge, as in the reference's (SA / S) ** k) instead of exp2(k * log2 .).  The exponent
        // and S are uniform over the column, so the variants are branches of the whole workgroup, not selects per class.
        const int kmode = (k == 0.5) ? 1 : ((k == 1.5) ? 2 : ((k == 1.0) ? 3 : ((k == 0.2) ? 4 : 0)));
        if (S <= 0) {   // Omega = where(S <= 0, 0, .): nothing to evaluate
#pragma unroll
            for (int j = 0; j < E; ++j) Om[j] = 0.0 * mk;
        } else if (kmode != 0) {
            // one straight-line loop per variant (the variant is uniform over the workgroup): with the selection inside the class loop
            // the compiler kept a branch per class and the eight independent square roots ran one after the other
            const UDiv by_S = udiv_prepare(S);
            double r[E];
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = udiv(SA_hi[j], by_S);
            if (kmode == 1) {
#pragma unroll
                for (int j = 0; j < E; ++j) r[j] = sqrt_unit(r[j]);
            } else if (kmode == 2) {
#pragma unroll
                for (int j = 0; j < E; ++j) r[j] = r[j] * sqrt_unit(r[j]);
            } else if (kmode == 4) {
                double y[E];
                bool tiny = false;   // a quotient in (0, 2^-127): below the range of the fifth-root path (the residue of a residue; residues are ~1e-20)
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    int e2;
                    y[j] = pow_fifth(r[j], &e2);
                    tiny |= e2 < -126;
                }
                if (__builtin_expect(__ballot(tiny) != 0, 0)) {
                    const double log2S = sas_log2(C, S);
#pragma unroll
                    for (int j = 0; j < E; ++j) y[j] = SA_hi[j] > 0 ? sas_pow_ratio(C, SA_hi[j], S, log2S, k) : 0.0;
                }
#pragma unroll
                for (int j = 0; j < E; ++j) r[j] = y[j];
            }
            // where(x > 0, where(x <= S, v, 1), 0) without a compare: x <= 0 leaves a NaN (0 * inf from the reciprocal square root) or, with
            // exponent 1, a quotient <= 0, and max(., 0) returns the 0 (maxNum: the operand that is a number); v > 1 exactly where x > S.
            // (. * mk) * mk == . * mk bit for bit: the mask is 0 or 1 (the reference's maskCatch is a bool; sas_body normalises it)
#pragma unroll
            for (int j = 0; j < E; ++j) Om[j] = fmin(fmax(r[j], 0.), 1.) * mk;
        } else {
        const double log2S = sas_log2(C, S);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double x = SA_hi[j];
            // evaluated for every class and selected afterwards: straight-line code lets the E independent
            // evaluations interleave (a NaN from x <= 0 is discarded by the select)
            const double v = sas_pow_ratio(C, x, S, log2S, k);
            Om[j] = (x > 0 ? (x <= S ? v : 1.) : 0.) * mk;   // ((. * 1.0 * mk) * mk: the mask is 0 or 1)
        }
        }
    } else if (dirac) {  // piston flow, sas.py:43-64: the edge index (vs.nages) against the age threshold p1
        const double S = Smax * mk;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double o = ((double)(base + j + 1) <= p1 ? 0 : 1) * mk;
            Om[j] = (S <= 0 ? 0 : o) * 1.0 * mk;
        }
        Om_edge0 = (S <= 0 ? 0 : (0.0 <= p1 ? 0 : 1) * mk) * 1.0 * mk;
    } else if (kumaraswami) {  // sas.py:67-147; the device library's pow: two per class, accuracy before speed
        const double S = Smax * mk;
        double S_rel = (S - p5) / (p6 - p5) * mk;
        S_rel = (S_rel < 0 ? 0 : S_rel);
        S_rel = (S_rel > 1 ? 1 : S_rel);
        const double up = p3 + (S_rel * p4), down = p3 + ((1 - S_rel) * p4);
        double a = p1, b = p2;
        if (code == 31) { a = 1; b = up; }
        if (code == 32) { a = down; b = 1; }
        if (code == 33) { a = 1; b = down; }
        if (code == 34) { a = up; b = 1; }
        if (code == 35) { a = down; b = up; }
        if (code == 36) a = down;
        if (code == 37) b = up;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double x = SA_hi[j];
            const double f = 1 - pow(1 - pow(x / S, a), b);
            const double o = (S >= 0 ? (x > 0 ? (x < S ? f : 1.) : 0.) : (x > 0 ? f : 0.)) * 1.0 * mk;
            Om[j] = (S <= 0 ? 0 : o) * mk;
        }
    } else if (expo) {  // sas.py:168-190, code 51
        const double S = Smax * mk;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double x = SA_hi[j];
            const double o = (x > 0 ? (x < S ? 1 - exp(p1 * (-1) * (x / S)) : 1.) : 0.) * mk;
            Om[j] = (S <= 0 ? 0 : o) * mk;
        }
    } else if (gamma) {  // sas.py:139-163, code 4: the regularised gammainc divided by Gamma(a) once more; 0 at SA == S
        const double S = Smax * 1.0 * mk;
        const double lgam = lgamma(p1);
        const double G = exp(lgam);
        for (int j = 0; j < E; ++j) {
            const double x = SA_hi[j];
            const double o = (x > 0 ? (x < S ? sas_gammainc(p1, p2 * x / S, lgam) / G : 0.) : 0) * 1.0 * mk;
            Om[j] = (S <= 0 ? 0 : o) * mk;
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) Om[j] = 0.0;
    }
}

// The SAS families with library calls inside (kumaraswami: two pow per class, exponential, gamma) as a function of their own.  Inlined
// into the loop over sub-steps and fluxes, the constants of ALL of them would be hoisted out of that loop together and held in
// registers for the whole kernel (measured: 150 spilled VGPRs with a single age class per thread); the benchmark's power law, the
// uniform and the dirac family stay inline.  The arrays cross the call through scratch memory, on this path only.
template <int E>
__device__ __attribute__((noinline)) void omega_library_families(int fam, const double *pr, const double *SA_hi, double Smax, double mk, int base,
                                                                  int A, double *Om) {
    Blk<1> B{};
    PowConsts C{};
    double p[7], x[E], o[E], edge0 = 0.0;
    for (int i = 0; i < 7; ++i) p[i] = pr[i];
#pragma unroll
    for (int j = 0; j < E; ++j) x[j] = SA_hi[j];
    if (fam == FAM_KUMARASWAMI) sas_omega<1, E, FAM_KUMARASWAMI>(B, C, p, x, Smax, mk, base, A, o, edge0);
    else if (fam == FAM_EXPONENTIAL) sas_omega<1, E, FAM_EXPONENTIAL>(B, C, p, x, Smax, mk, base, A, o, edge0);
    else sas_omega<1, E, FAM_GAMMA>(B, C, p, x, Smax, mk, base, A, o, edge0);
#pragma unroll
    for (int j = 0; j < E; ++j) Om[j] = o[j];
}

template <int W, int E>
SAS_DEV void age_stats(Blk<W> &B, const SasArgs &P, int64_t cell, int base, const double (&cdf_hi)[E], double cdf_lo,
                       const double (&dens)[E], double *const (&dst6)[6], bool skip10_90);
// The age statistics of a flux's travel time distribution are formed as soon as the distribution exists when the whole day runs
// in one launch (they depend on nothing later); keeping tt / TT alive until the storage stage cost 36 registers across four fluxes,
// which the compiler spilled.  With the stages in launches of their own they come back from the diagnostics arrays (load_dist).
SAS_DEV bool stats_now(const SasArgs &P) { return P.stats && (P.stages & RH_SAS_STORAGE); }

// calc_age_percentile :9-56 for the five percentiles at once + the mean age.
//   cdf_hi / cdf_lo: cumulative distribution at the upper edges of the thread's classes / lower edge of its first
//   dens: the distribution itself.  dst6: arrays of the 6 statistics; skip10_90: leave rt10 / rt90 unassigned.
template <int W, int E>
SAS_DEV void age_stats(Blk<W> &B, const SasArgs &P, int64_t cell, int base, const double (&cdf_hi)[E], double cdf_lo,
                       const double (&dens)[E], double *const (&dst6)[6], bool skip10_90) {
    const int A = P.ages;
    const double Q[5] = {0.1, 0.25, 0.5, 0.75, 0.9};
    // number of classes with cdf <= q, per percentile: counted with ballots (a compare per class and percentile, the population
    // counts on the scalar unit) and summed over the waves through LDS, instead of five floating-point block sums
    int cnt5[5] = {0, 0, 0, 0, 0};
    double v[2] = {0, 0};
    double mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const bool in = base + j < A;
#pragma unroll
        for (int q = 0; q < 5; ++q) cnt5[q] += __popcll(__ballot(in && (cdf_hi[j] <= Q[q])));
        if (in) {
            v[0] += dens[j];
            v[1] += (double)(base + j + 1) * dens[j];
            mx = fmax(mx, cdf_hi[j]);
        }
    }
    // the five counts, the two sums and the maximum cross the waves in ONE exchange (eight slots per wave, one barrier)
    v[0] = wave_sum(v[0]);
    v[1] = wave_sum(v[1]);
    mx = wave_max(mx);
    if (W > 1) {
        const int buf = B.phase++ & 1;
        if (B.lane == 0) {
#pragma unroll
            for (int q = 0; q < 5; ++q) B.red[buf][B.wave][q] = (double)cnt5[q];
            B.red[buf][B.wave][5] = v[0];
            B.red[buf][B.wave][6] = v[1];
            B.red[buf][B.wave][7] = mx;
        }
        SAS_SYNC();
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double c = B.red[buf][0][q];
            for (int w = 1; w < W; ++w) c += B.red[buf][w][q];
            cnt5[q] = (int)c;
        }
        v[0] = B.red[buf][0][5];
        v[1] = B.red[buf][0][6];
        mx = B.red[buf][0][7];
        for (int w = 1; w < W; ++w) {
            v[0] = v[0] + B.red[buf][w][5];
            v[1] = v[1] + B.red[buf][w][6];
            mx = fmax(mx, B.red[buf][w][7]);
        }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        if (skip10_90 && (q == 0 || q == 4)) continue;
        double *dst = dst6[q] + cell;
        const int cnt = cnt5[q];  // number of classes with cdf <= q; the crossing is in class `cnt`
        if (!(mx > 0)) {
            if (B.tid == 0) *dst = NAN;
        } else if (cnt <= 0) {
            if (B.tid == 0) *dst = 1.0;
        } else if (cnt >= A) {
            if (B.tid == 0) *dst = (double)A;
        } else if (cnt >= base && cnt < base + E) {
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (base + j == cnt) {
                    const double x1 = cdf_hi[j], x0 = (j == 0 ? cdf_lo : cdf_hi[j > 0 ? j - 1 : 0]);
                    const double y0 = (double)cnt, y1 = (double)(cnt + 1);  // ages are 1-based
                    const double slope = (y1 - y0) / (x1 - x0);
                    *dst = (x1 == x0) ? y0 : slope * (Q[q] - x0) + y0;
                }
        }
    }
    if (B.tid == 0) dst6[5][cell] = (v[0] > 0 ? v[1] : NAN);
}
// ... with the six arrays given by the registry index of the first
template <int W, int E>
SAS_DEV void age_stats(Blk<W> &B, const SasArgs &P, int64_t cell, int base, const double (&cdf_hi)[E], double cdf_lo,
                       const double (&dens)[E], int first_arr, bool skip10_90) {
    double *dst6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) dst6[q] = (double *)P.a[first_arr + q];
    age_stats<W, E>(B, P, cell, base, cdf_hi, cdf_lo, dens, dst6, skip10_90);
}

template <int W, int E>
SAS_DEV void residence_stats(Blk<W> &B, const SasArgs &P, int64_t cell, int base, const double (&sa)[E], double mk, int first_arr,
                             bool skip10_90) {
    // RT = SA / max(SA), rt = diff(RT): calculate_age_statistics_root_zone/subsoil/soil :155-312
    double SA_hi[E], SA_lo, mx;
    blk_cumsum<W, E, false>(B, sa, SA_hi, SA_lo, &mx, base, P.ages - 1);
    mx *= mk;
    double RT_hi[E], rt[E];
    const UDiv by_mx = udiv_prepare(mx);
    const double RT_lo = (mx > 0 ? udiv(SA_lo * mk, by_mx) : 0);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        RT_hi[j] = (mx > 0 ? udiv(SA_hi[j] * mk, by_mx) : 0);
        rt[j] = RT_hi[j] - (j == 0 ? RT_lo : RT_hi[j > 0 ? j - 1 : 0]);
    }
    age_stats<W, E>(B, P, cell, base, RT_hi, RT_lo, rt, first_arr, skip10_90);
}

// calc_ageing_sa (core/transport.py:623-652) and calc_ageing_msa (:655-680): shift by one class, merge the oldest
template <int W, int E>
SAS_DEV void ageing_anion(Blk<W> &B, int A, int base, double (&sa)[E], double (&msa)[E]) {
    double p_sa, p_msa;
    blk_prev2<W>(B, sa[E - 1], msa[E - 1], 0.0, 0.0, p_sa, p_msa);
    double n_sa[E], n_msa[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int k = base + j;
        n_sa[j] = (j == 0 ? p_sa : sa[j > 0 ? j - 1 : 0]);
        n_msa[j] = (j == 0 ? p_msa : msa[j > 0 ? j - 1 : 0]);
        if (k == 0) {
            n_sa[j] = 0;
            n_msa[j] = 0;
        }
        if (k == A - 1) {
            n_sa[j] += sa[j];
            n_sa[j] = (n_sa[j] < 1e-8 ? 0 : n_sa[j]);
            n_msa[j] += msa[j];
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        sa[j] = n_sa[j];
        msa[j] = n_msa[j];
    }
}

// Ageing by one day: calc_ageing_sa_msa_iso_kernel :780-805 -> calc_ageing_msa_iso :682-739.
template <int W, int E>
SAS_DEV void ageing(Blk<W> &B, int A, int base, double (&sa)[E], double (&msa)[E]) {
    double p_sa, p_msa;
    blk_prev2<W>(B, sa[E - 1], msa[E - 1], 0.0, 0.0, p_sa, p_msa);
    double n_sa[E], n_msa[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int k = base + j;
        n_sa[j] = (j == 0 ? p_sa : sa[j > 0 ? j - 1 : 0]);
        n_msa[j] = (j == 0 ? p_msa : msa[j > 0 ? j - 1 : 0]);
        if (k == 0) {
            n_sa[j] = 0;
            n_msa[j] = 0;
        }
        if (k == A - 1) {  // merge the oldest water
            const double sam1 = sa[j], msam1 = msa[j];
            const double tot = n_sa[j] + sam1;
            const double v = (tot > 0 ? msam1 * (sam1 / tot) + n_msa[j] * (n_sa[j] / tot) : 0);
            n_msa[j] = (v != v) ? 0 : v;
            n_sa[j] += sam1;
            n_sa[j] = (n_sa[j] < 1e-8 ? 0 : n_sa[j]);
            n_msa[j] = (n_sa[j] <= 0 ? NAN : n_msa[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        sa[j] = n_sa[j];
        msa[j] = n_msa[j];
    }
}

// the whole day (deterministic: the stages in args.stages) in one launch on `stream`, per kernel family:
// launchers of the kernel translation units (host functions)
int rh_sas_launch_det_iso(hipStream_t stream, const SasArgs &args, unsigned n_cells, int nages, bool e4);
int rh_sas_launch_det_anion(hipStream_t stream, const SasArgs &args, unsigned n_cells, int nages, bool e4);
int rh_sas_launch_euler_iso(hipStream_t stream, const SasArgs &args);
int rh_sas_launch_euler_anion(hipStream_t stream, const SasArgs &args);
int rh_sas_launch_rk4_iso(hipStream_t stream, const SasArgs &args);
int rh_sas_launch_rk4_anion(hipStream_t stream, const SasArgs &args);
