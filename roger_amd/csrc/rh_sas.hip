// rh_sas.hip -- SAS / oxygen-18 transport step (deterministic solver) for gfx950, and its C ABI
// (include/roger_hip_sas.h).
//
// One workgroup per soil column.  The age axis is laid out blocked over the workgroup: thread t
// owns the E consecutive age classes [t * E, (t + 1) * E) of every age vector in registers, so the
// whole day -- 2 inflows, 5 outgoing fluxes with `substeps` sub-steps each, storage concentrations,
// age statistics, ageing -- runs on one read and one write of the four state vectors
// (sa_rz, msa_rz, sa_ss, msa_ss): 8 * ages * 8 bytes per column and day.  The arithmetic is
// dominated by the power-law SAS function, 5 * substeps * (ages + 1) `pow` per column and day
// (3 * 10^4 at ages = 1000), which makes this kernel fp64-ALU bound, not HBM bound.
//
// Every formula below is a per-element restatement of the reference's array expressions (file:line
// in the comments; roger/core/transport.py unless said otherwise) with the reference's operation
// order; no FMA contraction (-ffp-contract=off).  Two deliberate differences, both at rounding
// level: prefix sums are block scans instead of sequential `cumsum`s, sums over ages are tree
// reductions instead of numpy's pairwise sums.  The scan is built so that what the algorithm is
// sensitive to still holds exactly: cumulative values are non-decreasing for non-negative input and
// an empty age class contributes an exact 0 to every difference of cumulative values.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "roger_hip.h"
#include "roger_hip_sas.h"

#define SAS_DEV __device__ __forceinline__

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

// Polynomial coefficients live in constant memory so that they reach the FMAs as scalar-register
// operands (one v_fma_f64 per Horner step); as immediates each step costs a 64-bit v_mov besides.
#include "rh_sas_tables.inc"
// RH_SAS_LOG: 1 = table-assisted log2 (64-entry table of {1/c, log2 c} in LDS, degree-8 log2(1 + r)); 0 = the
// table-free version (s = (m - 1) / (m + 1), odd series to s^19)
#ifndef RH_SAS_LOG
#define RH_SAS_LOG 1
#endif
__constant__ double2 SAS_LOG_T[64] = {RH_SAS_LOG_TABLE};
__constant__ double SAS_LOG1P_C[8] = {RH_SAS_LOG1P_COEF};
__constant__ double SAS_LOG_C[9] = {2.0 / 19.0, 2.0 / 17.0, 2.0 / 15.0, 2.0 / 13.0, 2.0 / 11.0, 2.0 / 9.0, 2.0 / 7.0, 2.0 / 5.0, 2.0 / 3.0};
__constant__ double SAS_EXP_C[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,
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

static const char *const SAS_NAMES[] = {
#define RH_SAS_ARRAY(name, kind, when) #name,
#include "rh_sas_arrays.def"
#undef RH_SAS_ARRAY
};
static const unsigned char SAS_KIND[] = {
#define RH_SAS_ARRAY(name, kind, when) K_##kind,
#include "rh_sas_arrays.def"
#undef RH_SAS_ARRAY
};
static const unsigned char SAS_WHEN[] = {
#define RH_SAS_ARRAY(name, kind, when) W_##when,
#include "rh_sas_arrays.def"
#undef RH_SAS_ARRAY
};

struct SasArgs {
    int64_t n;
    int64_t day_off;  // row of the daily inputs * n
    int ages, substeps, stages, stats, diag, tracer;
    double vsmow, dmin, dmax;
    int *unsupported;  // device flag: a column asked for a SAS family this kernel does not implement
    void *a[SA_COUNT];
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
};

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
// value of the previous lane; lane 0 gets `first`
SAS_DEV double lane_prev(double v, double first) { return dpp_move<0x138, 0xf>(first, v); }
// inclusive prefix sum over the 64 lanes (earlier lanes + own)
SAS_DEV double wave_scan_sum(double v) {
    v = dpp_move<0x111, 0xf>(0.0, v) + v;
    v = dpp_move<0x112, 0xf>(0.0, v) + v;
    v = dpp_move<0x114, 0xf>(0.0, v) + v;
    v = dpp_move<0x118, 0xf>(0.0, v) + v;
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
        __syncthreads();
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
        __syncthreads();
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
        __syncthreads();
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
    const double wexc = lane_prev(winc, 0.0);
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
        const int top_thread = top_k / E, top_j = top_k % E;  // uniform
        double mine = u[0];
#pragma unroll
        for (int j = 1; j < E; ++j) mine = (j == top_j) ? u[j] : mine;
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
    __syncthreads();
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
__device__ __attribute__((noinline)) double sas_gammainc(double a, double x, double lgam) {
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
// One instantiation per family, selected per column (uniform over the workgroup) by calc_tt below: the benchmark's
// power law keeps its register budget (3 waves/SIMD without spills) whatever the other families need.
template <int W, int E, int FAM>
SAS_DEV void calc_tt_family(Blk<W> &B, const SasArgs &P, const double *p, double flux, const double (&sa)[E], double mk, int base,
                            double (&tt)[E]) {
    const int A = P.ages;
    const double h = 1 / (double)P.substeps;
    const double fh = flux * h;
    if (!(fh > 0)) {
        // :440-443: tti = where(flux * h > 0, ., 0) in every sub-step -> TT = 0 -> tt = 0 (:496-499);
        // the SAS evaluation cannot change that, skip it
#pragma unroll
        for (int j = 0; j < E; ++j) tt[j] = 0.0;
        return;
    }
    const double code = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6];  // read once: the loop below stores nothing, but the compiler cannot know
    constexpr bool uniform = FAM == FAM_UNIFORM, power = FAM == FAM_POWER, dirac = FAM == FAM_DIRAC;
    constexpr bool kumaraswami = FAM == FAM_KUMARASWAMI, expo = FAM == FAM_EXPONENTIAL, gamma = FAM == FAM_GAMMA;
    const PowConsts C = load_pow_consts(B.logt);
    const UDiv by_fh = udiv_prepare(fh);
    double Om_edge0 = 0.0;  // Omega at SA[0] = 0: 0 for every family but a dirac with a negative threshold
    double san[E], ttn[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        san[j] = sa[j];
        ttn[j] = 0.0;
    }
    for (int it = 0; it < P.substeps; ++it) {
        double SA_hi[E], SA_lo, Smax;
        blk_cumsum<W, E, (FAM == FAM_KUMARASWAMI || FAM == FAM_EXPONENTIAL || FAM == FAM_GAMMA)>(B, san, SA_hi, SA_lo, &Smax, base, A - 1);
        if (it == 0) {  // the first sub-step sees SA = calc_SA(sa) * maskCatch, the later ones cumsum(san) (:456-459)
#pragma unroll
            for (int j = 0; j < E; ++j) SA_hi[j] *= mk;
            Smax *= mk;
        }
        double Om[E];
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
            const int kmode = (k == 0.5) ? 1 : ((k == 1.5) ? 2 : ((k == 1.0) ? 3 : 0));
            if (S <= 0) {   // Omega = where(S <= 0, 0, .): nothing to evaluate
#pragma unroll
                for (int j = 0; j < E; ++j) Om[j] = 0.0 * mk;
            } else if (kmode != 0) {
                const UDiv by_S = udiv_prepare(S);
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const double x = SA_hi[j];
                    const double r = udiv(x, by_S);
                    double v;
                    if (kmode == 1) v = sqrt_unit(r);
                    else if (kmode == 2) v = r * sqrt_unit(r);
                    else v = r;
                    const double o = (x > 0 ? fmin(v, 1.) : 0.) * 1.0 * mk;   // x <= S ? v : 1, and v > 1 exactly where x > S
                    Om[j] = o * mk;
                }
            } else {
            const double log2S = sas_log2(C, S);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const double x = SA_hi[j];
                // evaluated for every class and selected afterwards: straight-line code lets the E independent
                // evaluations interleave (a NaN from x <= 0 is discarded by the select)
                const double v = sas_pow_ratio(C, x, S, log2S, k);
                const double o = (x > 0 ? (x <= S ? v : 1.) : 0.) * 1.0 * mk;
                Om[j] = o * mk;
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
        double Om_lo, unused;
        blk_prev2<W>(B, Om[E - 1], 0.0, Om_edge0, 0.0, Om_lo, unused);
        double tti[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double d = Om[j] - (j == 0 ? Om_lo : Om[j - 1]);
            double t = fmax(d, 0.0);                                            // :430-433  where(d >= 0, d, 0)
            const double q = fmin(flux * t * h, san[j]);                        // :435-438  where(flux t h > san, san, flux t h)
            t = udiv(q, by_fh);                                                 // :440-443: q / (flux * h), fh > 0 here
            san[j] = san[j] + -t * flux * h;                                    // :445-448
            tti[j] = t;
        }
        // :461-468.  The reference accumulates TTn += cumsum(tti) and takes diff(TTn / N) afterwards (:482-490);
        // diff(cumsum(.)) is the identity, so the sub-step distributions are accumulated directly (the
        // reference's own `ttn`).  Differs from the round trip through the cumulative sums by ~1e-16 absolute
        // and saves one block scan per sub-step.
#pragma unroll
        for (int j = 0; j < E; ++j) ttn[j] += tti[j];
    }
    const UDiv by_N = udiv_prepare((double)P.substeps), by_flux = udiv_prepare(flux);   // (flux > 0 here: fh > 0)
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const double t = udiv(ttn[j], by_N);                                      // :482-490
        const double q = (flux * t > sa[j] ? sa[j] : flux * t);                   // :493-496
        tt[j] = udiv(q, by_flux);                                                 // :497-499  where(flux > 0, q / flux, 0)
    }
}
template <int W, int E>
SAS_DEV void calc_tt(Blk<W> &B, const SasArgs &P, const double *p, double flux, const double (&sa)[E], double mk, int base,
                     double (&tt)[E]) {
    const double code = p[0];
    if (code == 6 || code == 61 || code == 62) calc_tt_family<W, E, FAM_POWER>(B, P, p, flux, sa, mk, base, tt);
    else if (code == 1) calc_tt_family<W, E, FAM_UNIFORM>(B, P, p, flux, sa, mk, base, tt);
    else if (code == 3 || (code >= 31 && code <= 37)) calc_tt_family<W, E, FAM_KUMARASWAMI>(B, P, p, flux, sa, mk, base, tt);
    else if (code == 2) calc_tt_family<W, E, FAM_DIRAC>(B, P, p, flux, sa, mk, base, tt);
    else if (code == 51) calc_tt_family<W, E, FAM_EXPONENTIAL>(B, P, p, flux, sa, mk, base, tt);
#ifndef RH_SAS_NO_GAMMA  // (experiments: the kernel without the gamma family's code)
    else if (code == 4) calc_tt_family<W, E, FAM_GAMMA>(B, P, p, flux, sa, mk, base, tt);
#endif
    else {
        // 52, the exponential with reversed age order (sas.py:186-190): Omega DEcreases from 1 to 0 along the age axis,
        // calc_tt clips every difference to 0 (:430-433) -- no water is selected, like Omega = 0.  Any other code is
        // unknown to the reference's families (all masked out: Omega = 0 as well) and is reported.
        if (code != 52 && B.tid == 0 && flux * (1 / (double)P.substeps) > 0) *P.unsupported = 1;
        calc_tt_family<W, E, FAM_NONE>(B, P, p, flux, sa, mk, base, tt);
    }
}

template <int W, int E>
SAS_DEV void age_stats(Blk<W> &B, const SasArgs &P, int64_t cell, int base, const double (&cdf_hi)[E], double cdf_lo,
                       const double (&dens)[E], int first_arr, bool skip10_90);
// The age statistics of a flux's travel time distribution are formed as soon as the distribution exists when the whole day runs
// in one launch (they depend on nothing later); keeping tt / TT alive until the storage stage cost 36 registers across four fluxes,
// which the compiler spilled.  With the stages in launches of their own they come back from the diagnostics arrays (load_dist).
SAS_DEV bool stats_now(const SasArgs &P) { return P.stats && (P.stages & RH_SAS_STORAGE); }

// One outgoing flux: SA, tt, TT, mtt, C, C_iso, the sink's isotope mixing, update_sa.
// calc_evaporation/transpiration_transport_iso_kernel (core/evapotranspiration.py:653-719, 831-901),
// calc_percolation_rz/ss_transport_iso_kernel (core/subsurface_runoff.py:1531-1626, 1753-1820),
// calc_capillary_rise_rz_transport_iso_kernel (core/capillary_rise.py:404-500).
template <int W, int E, bool SINK, bool KEEP>
SAS_DEV void outflux(Blk<W> &B, const SasArgs &P, int64_t cell, int f, double (&sa)[E], double (&msa)[E], double (&sa_sink)[E],
                     double (&msa_sink)[E], double mk, int base, Dist<E> &keep) {
    const int A = P.ages;
    const double flux = ((const double *)P.a[SA_evap_soil + f])[P.day_off + cell];
    const double *p = (const double *)P.a[SA_sas_params_evap_soil + f] + cell * 8;
    double tt[E];
    calc_tt<W, E>(B, P, p, flux, sa, mk, base, tt);
    double mtt[E], s[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        tt[j] *= mk;
        mtt[j] = (tt[j] > 0 ? msa[j] : 0) * mk;  // calc_mtt :565-596 with alpha = 1
        s[0] += mtt[j] * tt[j];
        s[1] += tt[j];
    }
    if (P.diag || (KEEP && stats_now(P))) {  // TT[1:] = cumsum(tt)
        double TT_hi[E], TT_lo;
        blk_cumsum<W, E, false>(B, tt, TT_hi, TT_lo, nullptr, base, 0);
        if (KEEP && stats_now(P)) age_stats<W, E>(B, P, cell, base, TT_hi, TT_lo, tt, f == 1 ? SA_tt10_transp : SA_tt10_q_ss, false);
        if (P.diag) {
            double *o_tt = (double *)P.a[SA_tt_evap_soil + f] + cell * A;
            double *o_mtt = (double *)P.a[SA_mtt_evap_soil + f] + cell * A;
            double *o_TT = (double *)P.a[SA_TT_evap_soil + f] + cell * (A + 1);
            if (B.tid == 0) o_TT[0] = 0.0;
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (base + j < A) {
                    o_tt[base + j] = tt[j];
                    o_mtt[base + j] = mtt[j];
                    o_TT[base + j + 1] = TT_hi[j];
                }
        }
    }
    blk_sum<W, 2>(B, s);
    if (B.tid == 0) {  // calc_conc_iso_flux :512-535
        double conc = (s[1] > 0 ? s[0] / s[1] : NAN);
        conc = (conc != 0 ? conc : NAN);
        const double C = conc * mk;
        ((double *)P.a[SA_C_evap_soil + f])[cell] = C;
        ((double *)P.a[SA_C_iso_evap_soil + f])[cell] = conc_to_delta(P, C) * mk;
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (SINK) {
            const double add = tt[j] * flux;
            const UDiv by_tot = udiv_prepare(add + sa_sink[j]);   // two quotients by one divisor (udiv: the bits of `/`)
            msa_sink[j] = (add + sa_sink[j] > 0
                               ? msa_sink[j] * udiv(sa_sink[j], by_tot) + mtt[j] * udiv(add, by_tot)
                               : msa_sink[j]) * mk;
        }
        double v = sa[j] + -flux * tt[j];  // update_sa :599-619
        v = ((v > -1e-5) && (v < 0)) ? 0 : v;
        sa[j] = v * mk;
        if (SINK) sa_sink[j] += tt[j] * flux * mk;
        msa[j] = (sa[j] <= 0 ? 0 : msa[j]) * mk;
    }
}

// Infiltration into age class 0: calc_infiltration_rz_transport_iso_kernel (core/infiltration.py:2218-2346)
// and calc_infiltration_ss_transport_iso_kernel (:2441-2512).  which: 0 matrix -> rz, 1 pf -> rz, 2 pf -> ss.
template <int W, int E>
SAS_DEV void inflow(Blk<W> &B, const SasArgs &P, int64_t cell, int which, double (&sa)[E], double (&msa)[E], double mk, int base) {
    const double inf = ((const double *)P.a[SA_inf_mat_rz + which])[P.day_off + cell];
    const double C_in = ((const double *)P.a[SA_C_in])[P.day_off + cell];
    if (B.tid == 0) {
        const double C = (inf > 0 ? C_in : 0) * mk;
        ((double *)P.a[SA_C_inf_mat_rz + which])[cell] = C;
        ((double *)P.a[SA_C_iso_inf_mat_rz + which])[cell] = conc_to_delta(P, C) * mk;
    }
    // tt is 1 in age class 0 and 0 elsewhere.  For the other classes the mixing formula reduces to msa * (sa / sa) + 0 = msa (sa > 0)
    // or msa (sa <= 0): the identity, bit for bit (a NaN marker stays a NaN) -- only the thread that owns class 0 computes
    if (base == 0) {
        const double ttk = (inf > 0 ? 1 : 0) * mk;
        const double mttk = (inf > 0 ? C_in : 0) * mk;
        msa[0] = (inf * ttk + sa[0] > 0 ? msa[0] * (sa[0] / (ttk * inf + sa[0])) + mttk * ((ttk * inf) / (inf * ttk + sa[0]))
                                        : msa[0]) * mk;
        sa[0] += inf * mk;
    }
    if (mk != 1.0) {   // (the reference multiplies every class by maskCatch)
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (base + j != 0) msa[j] = msa[j] * mk;
    }
}

// calc_age_percentile :9-56 for the five percentiles at once + the mean age.
//   cdf_hi / cdf_lo: cumulative distribution at the upper edges of the thread's classes / lower edge of its first
//   dens: the distribution itself.  dst: arrays of the 6 statistics; skip10_90: leave rt10 / rt90 unassigned.
template <int W, int E>
SAS_DEV void age_stats(Blk<W> &B, const SasArgs &P, int64_t cell, int base, const double (&cdf_hi)[E], double cdf_lo,
                       const double (&dens)[E], int first_arr, bool skip10_90) {
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
    if (W > 1) {
        const int buf = B.phase++ & 1;
        if (B.lane == 0) {
#pragma unroll
            for (int q = 0; q < 5; ++q) B.red[buf][B.wave][q] = (double)cnt5[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double c = B.red[buf][0][q];
            for (int w = 1; w < W; ++w) c += B.red[buf][w][q];
            cnt5[q] = (int)c;
        }
    }
    blk_sum<W, 2>(B, v);
    mx = blk_max<W>(B, mx);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        if (skip10_90 && (q == 0 || q == 4)) continue;
        double *dst = (double *)P.a[first_arr + q] + cell;
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
    if (B.tid == 0) ((double *)P.a[first_arr + 5])[cell] = (v[0] > 0 ? v[1] : NAN);
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

// ---------------------------------------------------------------------------------------------
// bromide: the reference's anion kernels.  msa is solute mass by age; a flux takes
// mtt = msa / sa * alpha * tt * flux, clipped to [0, msa] (calc_mtt, core/transport.py:583-596).
// ---------------------------------------------------------------------------------------------

// TT = cumsum(tt) for the age statistics (KEEP) and the diagnostics arrays; mtt may be null (soil evaporation)
template <int W, int E, bool KEEP>
SAS_DEV void record_dist(Blk<W> &B, const SasArgs &P, int64_t cell, int f, int base, const double (&tt)[E], const double *mtt,
                         Dist<E> &keep) {
    if (!(P.diag || (KEEP && stats_now(P)))) return;
    const int A = P.ages;
    double TT_hi[E], TT_lo;
    blk_cumsum<W, E, false>(B, tt, TT_hi, TT_lo, nullptr, base, 0);
    if (KEEP && stats_now(P)) age_stats<W, E>(B, P, cell, base, TT_hi, TT_lo, tt, f == 1 ? SA_tt10_transp : SA_tt10_q_ss, false);
    if (P.diag) {
        double *o_tt = (double *)P.a[SA_tt_evap_soil + f] + cell * A;
        double *o_mtt = (double *)P.a[SA_mtt_evap_soil + f] + cell * A;
        double *o_TT = (double *)P.a[SA_TT_evap_soil + f] + cell * (A + 1);
        if (B.tid == 0) o_TT[0] = 0.0;
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (base + j < A) {
                o_tt[base + j] = tt[j];
                if (mtt) o_mtt[base + j] = mtt[j];
                o_TT[base + j + 1] = TT_hi[j];
            }
    }
}

// One outgoing flux of the anion kernels.  WATER: calc_evaporation_transport_kernel (core/evapotranspiration.py:620-650),
// the solute stays behind.  Otherwise calc_transpiration_transport_anion_kernel (:905-985),
// calc_percolation_rz/ss_transport_anion_kernel (core/subsurface_runoff.py:1630-1716, 1823-1893),
// calc_capillary_rise_rz_transport_anion_kernel (core/capillary_rise.py:503-590).
template <int W, int E, bool SINK, bool KEEP, bool WATER>
SAS_DEV void outflux_anion(Blk<W> &B, const SasArgs &P, int64_t cell, int f, double alpha, double (&sa)[E], double (&msa)[E],
                           double (&sa_sink)[E], double (&msa_sink)[E], double mk, int base, Dist<E> &keep) {
    const double flux = ((const double *)P.a[SA_evap_soil + f])[P.day_off + cell];
    const double *p = (const double *)P.a[SA_sas_params_evap_soil + f] + cell * 8;
    double tt[E];
    calc_tt<W, E>(B, P, p, flux, sa, mk, base, tt);
    double mtt[E], s[1] = {0.0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        tt[j] *= mk;
        if (!WATER) {
            double m = (sa[j] > 0 ? msa[j] / sa[j] : 0) * alpha * tt[j] * flux;
            m = (m <= 0 ? 0 : m);
            m = (m > msa[j] ? msa[j] : m);
            mtt[j] = m * mk;
            s[0] += mtt[j];
        }
    }
    record_dist<W, E, KEEP>(B, P, cell, f, base, tt, WATER ? nullptr : mtt, keep);
    if (!WATER) {
        blk_sum<W, 1>(B, s);
        if (B.tid == 0) {
            ((double *)P.a[SA_C_evap_soil + f])[cell] = (flux > 0 ? s[0] / flux : 0) * mk;
            ((double *)P.a[SA_M_evap_soil + f])[cell] = s[0] * mk;
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        double v = sa[j] + -flux * tt[j];  // update_sa :599-619
        v = ((v > -1e-5) && (v < 0)) ? 0 : v;
        sa[j] = v * mk;
        if (!WATER) msa[j] += -mtt[j] * mk;
        if (SINK) {
            msa_sink[j] += mtt[j] * mk;
            sa_sink[j] += tt[j] * flux * mk;
        }
    }
}

// calc_infiltration_rz_transport_anion_kernel (core/infiltration.py:2350-2424): matrix and preferential-flow
// infiltration join age class 0 in one addition; calc_infiltration_ss_transport_anion_kernel (:2516-2566).
template <int W, int E>
SAS_DEV void inflow_anion(Blk<W> &B, const SasArgs &P, int64_t cell, bool subsoil, double (&sa)[E], double (&msa)[E], double mk,
                          int base) {
    const double C_in = ((const double *)P.a[SA_C_in])[P.day_off + cell];
    double d_sa, d_msa;
    if (!subsoil) {
        const double im = ((const double *)P.a[SA_inf_mat_rz])[P.day_off + cell];
        const double ip = ((const double *)P.a[SA_inf_pf_rz])[P.day_off + cell];
        const double C0 = (im > 0 ? C_in : 0) * mk, C1 = (ip > 0 ? C_in : 0) * mk;
        const double M0 = C0 * im * mk, M1 = C1 * ip * mk;
        if (B.tid == 0) {
            ((double *)P.a[SA_C_inf_mat_rz])[cell] = C0;
            ((double *)P.a[SA_C_inf_pf_rz])[cell] = C1;
            ((double *)P.a[SA_M_inf_mat_rz])[cell] = M0;
            ((double *)P.a[SA_M_inf_pf_rz])[cell] = M1;
        }
        d_sa = im + ip * mk;
        d_msa = M0 + M1 * mk;
    } else {
        const double ip = ((const double *)P.a[SA_inf_pf_ss])[P.day_off + cell];
        const double C2 = (ip > 0 ? C_in : 0) * mk;
        const double M2 = C2 * ip * mk;
        if (B.tid == 0) {
            ((double *)P.a[SA_C_inf_pf_ss])[cell] = C2;
            ((double *)P.a[SA_M_inf_pf_ss])[cell] = M2;
        }
        d_sa = ip * mk;
        d_msa = M2 * mk;
    }
#pragma unroll
    for (int j = 0; j < E; ++j)
        if (base + j == 0) {
            sa[j] += d_sa;
            msa[j] += d_msa;
        }
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

// tt / TT of one flux back from the diagnostics arrays (age statistics in a launch of their own)
template <int E>
SAS_DEV void load_dist(const SasArgs &P, int64_t cell, int base, int f, Dist<E> &D) {
    const int A = P.ages;
    const double *g_tt = (const double *)P.a[SA_tt_evap_soil + f] + cell * A;
    const double *g_TT = (const double *)P.a[SA_TT_evap_soil + f] + cell * (A + 1);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const bool in = base + j < A;
        D.tt[j] = in ? g_tt[base + j] : 0.0;
        D.TT_hi[j] = g_TT[in ? base + j + 1 : A];
    }
    D.TT_lo = g_TT[base < A ? base : A];
}

// RH_SAS_WAVES > 0: register budget for that many waves per SIMD (experiments; 0 = compiler's choice)
#ifndef RH_SAS_WAVES
#define RH_SAS_WAVES 3
#endif
#if RH_SAS_WAVES > 0
#define SAS_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RH_SAS_WAVES, RH_SAS_WAVES)))
#else
#define SAS_OCCUPANCY
#endif
// Eight age classes per thread (the <2, 8> shape for ages <= 1024, RH_SAS_E8): the scans and lane exchanges of a sub-step are paid
// once per thread, so twice the classes per thread halve their share; the state then needs the register budget of 2 waves per SIMD.
#define SAS_OCCUPANCY_E8 __attribute__((amdgpu_waves_per_eu(2, 2)))
template <int W, int E, bool ANION>
__device__ __forceinline__ void sas_body(const SasArgs &P);
template <int W, int E, bool ANION>
__global__ __launch_bounds__(W * 64) SAS_OCCUPANCY void k_sas(const SasArgs P) {
    sas_body<W, E, ANION>(P);
}
template <int W, bool ANION>
__global__ __launch_bounds__(W * 64) SAS_OCCUPANCY_E8 void k_sas8(const SasArgs P) {
    sas_body<W, 8, ANION>(P);
}
template <int W, int E, bool ANION>
__device__ __forceinline__ void sas_body(const SasArgs &P) {
    __shared__ double s_red[2][W][8];
    __shared__ double s_xch[2][W][2];
    __shared__ double2 s_logt[64];
    if (threadIdx.x < 64) s_logt[threadIdx.x] = SAS_LOG_T[threadIdx.x];
    __syncthreads();
    Blk<W> B;
    B.logt = s_logt;
    B.tid = threadIdx.x;
    B.lane = threadIdx.x & 63;
    B.wave = threadIdx.x >> 6;
    B.phase = 0;
    B.red = s_red;
    B.xch = s_xch;
    const int64_t cell = blockIdx.x;
    const int A = P.ages;
    const int base = B.tid * E;
    const double mk = (double)((const int *)P.a[SA_maskCatch])[cell];

    double sa_rz[E], msa_rz[E], sa_ss[E], msa_ss[E];
    {
        const double *g0 = (const double *)P.a[SA_sa_rz] + cell * A, *g1 = (const double *)P.a[SA_msa_rz] + cell * A;
        const double *g2 = (const double *)P.a[SA_sa_ss] + cell * A, *g3 = (const double *)P.a[SA_msa_ss] + cell * A;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const bool in = base + j < A;
            sa_rz[j] = in ? g0[base + j] : 0.0;
            msa_rz[j] = in ? g1[base + j] : 0.0;
            sa_ss[j] = in ? g2[base + j] : 0.0;
            msa_ss[j] = in ? g3[base + j] : 0.0;
        }
    }
    Dist<E> d_transp, d_q_ss;
    bool have_transp = false, have_q_ss = false;
    const bool stats = P.stats && (P.stages & RH_SAS_STORAGE);

    // order of svat_transport_model_deterministic :949-991
    if constexpr (ANION) {
        const double alpha_q = ((const double *)P.a[SA_alpha_q])[cell];
        if (P.stages & RH_SAS_INF_RZ) inflow_anion<W, E>(B, P, cell, false, sa_rz, msa_rz, mk, base);
        if (P.stages & RH_SAS_EVAP) {   // water only -- but the virtual tracer leaves with it at alpha = 1
            if (P.tracer == RH_SAS_TRACER_VIRTUAL)
                outflux_anion<W, E, false, false, false>(B, P, cell, 0, 1.0, sa_rz, msa_rz, sa_rz, msa_rz, mk, base, d_transp);
            else
                outflux_anion<W, E, false, false, true>(B, P, cell, 0, 0.0, sa_rz, msa_rz, sa_rz, msa_rz, mk, base, d_transp);
        }
        if (P.stages & RH_SAS_TRANSP) {
            // crop solute uptake stops if the root zone holds more than 80 % of saturation: evapotranspiration.py:932-939
            const int lu = ((const int *)P.a[SA_lu_id])[cell];
            double S[1] = {0.0};
#pragma unroll
            for (int j = 0; j < E; ++j) S[0] += sa_rz[j];
            blk_sum<W, 1>(B, S);
            const bool stop = (lu > 500) && (lu < 599) && (S[0] >= 0.8 * ((const double *)P.a[SA_S_sat_rz])[cell]);
            const double alpha = (stop ? 0 : ((const double *)P.a[SA_alpha_transp])[cell]) * mk;
            outflux_anion<W, E, false, true, false>(B, P, cell, 1, alpha, sa_rz, msa_rz, sa_rz, msa_rz, mk, base, d_transp);
            have_transp = true;
        }
        if (P.stages & RH_SAS_Q_RZ)
            outflux_anion<W, E, true, false, false>(B, P, cell, 2, alpha_q, sa_rz, msa_rz, sa_ss, msa_ss, mk, base, d_transp);
        if (P.stages & RH_SAS_INF_SS) inflow_anion<W, E>(B, P, cell, true, sa_ss, msa_ss, mk, base);
        if (P.stages & RH_SAS_Q_SS) {
            outflux_anion<W, E, false, true, false>(B, P, cell, 3, alpha_q, sa_ss, msa_ss, sa_ss, msa_ss, mk, base, d_q_ss);
            have_q_ss = true;
        }
        if (P.stages & RH_SAS_CPR)
            outflux_anion<W, E, true, false, false>(B, P, cell, 4, alpha_q, sa_ss, msa_ss, sa_rz, msa_rz, mk, base, d_transp);
    } else {
    if (P.stages & RH_SAS_INF_RZ) {
        inflow<W, E>(B, P, cell, 0, sa_rz, msa_rz, mk, base);
        inflow<W, E>(B, P, cell, 1, sa_rz, msa_rz, mk, base);
    }
    if (P.stages & RH_SAS_EVAP) outflux<W, E, false, false>(B, P, cell, 0, sa_rz, msa_rz, sa_rz, msa_rz, mk, base, d_transp);
    if (P.stages & RH_SAS_TRANSP) {
        outflux<W, E, false, true>(B, P, cell, 1, sa_rz, msa_rz, sa_rz, msa_rz, mk, base, d_transp);
        have_transp = true;
    }
    if (P.stages & RH_SAS_Q_RZ) outflux<W, E, true, false>(B, P, cell, 2, sa_rz, msa_rz, sa_ss, msa_ss, mk, base, d_transp);
    if (P.stages & RH_SAS_INF_SS) inflow<W, E>(B, P, cell, 2, sa_ss, msa_ss, mk, base);
    if (P.stages & RH_SAS_Q_SS) {
        outflux<W, E, false, true>(B, P, cell, 3, sa_ss, msa_ss, sa_ss, msa_ss, mk, base, d_q_ss);
        have_q_ss = true;
    }
    if (P.stages & RH_SAS_CPR) outflux<W, E, true, false>(B, P, cell, 4, sa_ss, msa_ss, sa_rz, msa_rz, mk, base, d_transp);
    }

    if (P.stages & RH_SAS_STORAGE) {
        // calc_root_zone_transport_iso_kernel (core/root_zone.py:189-217), calc_subsoil_transport_iso_kernel
        // (core/subsoil.py:159-188), calculate_soil_transport_iso_kernel (core/soil.py:1036-1090)
        double sa_s[E], msa_s[E];
        double s[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < E; ++j) {
            sa_rz[j] = (sa_rz[j] < 1e-8 ? 0 : sa_rz[j]);
            sa_ss[j] = (sa_ss[j] < 1e-8 ? 0 : sa_ss[j]);
            sa_s[j] = sa_rz[j] + sa_ss[j] * mk;
            if constexpr (ANION) {
                // calc_root_zone/subsoil_transport_anion_kernel (core/root_zone.py:221-258, subsoil.py:186-223),
                // calculate_soil_transport_anion_kernel (core/soil.py:1094-1142): M = nansum(msa), C = M / sum(sa)
                msa_rz[j] = (sa_rz[j] <= 0 ? 0 : msa_rz[j]);
                msa_ss[j] = (sa_ss[j] <= 0 ? 0 : msa_ss[j]);
                msa_s[j] = msa_rz[j] + msa_ss[j] * mk;
                s[0] += (msa_rz[j] != msa_rz[j]) ? 0 : msa_rz[j];
                s[2] += (msa_ss[j] != msa_ss[j]) ? 0 : msa_ss[j];
                s[4] += (msa_s[j] != msa_s[j]) ? 0 : msa_s[j];
            } else {
                const double tot = sa_rz[j] + sa_ss[j];
                const UDiv by_tot = udiv_prepare(tot);
                const double v = (tot > 0 ? msa_rz[j] * udiv(sa_rz[j], by_tot) + msa_ss[j] * udiv(sa_ss[j], by_tot) : 0);
                msa_s[j] = (v != v) ? 0 : v;
                s[0] += msa_rz[j] * sa_rz[j];
                s[2] += msa_ss[j] * sa_ss[j];
                s[4] += msa_s[j] * sa_s[j];
            }
            s[1] += sa_rz[j];
            s[3] += sa_ss[j];
            s[5] += sa_s[j];
        }
        blk_sum<W, 6>(B, s);
        if (B.tid == 0) {
            for (int k = 0; k < 3; ++k) {
                if constexpr (ANION) {
                    const double M = s[2 * k] * mk;
                    ((double *)P.a[SA_M_rz + k])[cell] = M;
                    ((double *)P.a[SA_C_rz + k])[cell] = (s[2 * k + 1] > 0 ? M / s[2 * k + 1] : 0);
                } else {  // calc_conc_iso_storage :538-562
                    const double C = (s[2 * k + 1] > 0 ? s[2 * k] / s[2 * k + 1] : 0) * mk;
                    ((double *)P.a[SA_C_rz + k])[cell] = C;
                    ((double *)P.a[SA_C_iso_rz + k])[cell] = conc_to_delta(P, C) * mk;
                }
            }
        }
        if (P.diag) {
            double *o0 = (double *)P.a[SA_sa_s] + cell * A, *o1 = (double *)P.a[SA_msa_s] + cell * A;
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (base + j < A) {
                    o0[base + j] = sa_s[j];
                    o1[base + j] = msa_s[j];
                }
        }
        if (stats) {  // calculate_age_statistics_* :59-312
            // stages run one launch at a time: the distributions come back from the diagnostics arrays
            if (!have_transp) {
                Dist<E> d;
                load_dist<E>(P, cell, base, 1, d);
                age_stats<W, E>(B, P, cell, base, d.TT_hi, d.TT_lo, d.tt, SA_tt10_transp, false);
            }
            if (!have_q_ss) {
                Dist<E> d;
                load_dist<E>(P, cell, base, 3, d);
                age_stats<W, E>(B, P, cell, base, d.TT_hi, d.TT_lo, d.tt, SA_tt10_q_ss, false);
            }
            // the reference never assigns rt10 / rt90 of root zone and subsoil (:181-196, :232-247)
            residence_stats<W, E>(B, P, cell, base, sa_rz, mk, SA_rt10_rz, true);
            residence_stats<W, E>(B, P, cell, base, sa_ss, mk, SA_rt10_ss, true);
            residence_stats<W, E>(B, P, cell, base, sa_s, mk, SA_rt10_s, false);
        }
    }

    if (P.stages & RH_SAS_RESCALE) {
        // rescale_sa_msa_iso_soil_kernel, core/soil.py:1250-1395 (no maskCatch on sa and C here, as in the reference)
        const double S_rz_init = ((const double *)P.a[SA_S_rz_init])[cell], S_ss_init = ((const double *)P.a[SA_S_ss_init])[cell];
        double t[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < E; ++j) {
            t[0] += sa_rz[j];
            t[1] += sa_ss[j];
        }
        blk_sum<W, 2>(B, t);
        double sa_s[E], msa_s[E], s[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const bool in = base + j < A;
            sa_rz[j] = in ? S_rz_init * (sa_rz[j] / t[0]) : 0.0;
            sa_ss[j] = in ? S_ss_init * (sa_ss[j] / t[1]) : 0.0;
            sa_s[j] = sa_rz[j] + sa_ss[j];
            if constexpr (ANION) {
                if (P.tracer != RH_SAS_TRACER_BROMIDE) {  // rescale_sa_msa_anion_soil_kernel, chloride / virtual tracer (core/soil.py:1507-1640):
                    msa_rz[j] *= S_rz_init / t[0];         // the solute is scaled with the water
                    msa_ss[j] *= S_ss_init / t[1];
                } else {  // bromide (:1399-1506): the soil starts free of it
                    msa_rz[j] = 0;
                    msa_ss[j] = 0;
                }
            }
            const double tot = sa_rz[j] + sa_ss[j];
            const double v = (tot > 0 ? msa_rz[j] * (sa_rz[j] / tot) + msa_ss[j] * (sa_ss[j] / tot) : 0);
            msa_s[j] = ((v != v) || (base + j == 0)) ? 0 : v;
            if constexpr (ANION) {
                if (P.tracer != RH_SAS_TRACER_BROMIDE) {   // C = sum(msa) / sum(sa), msa_s = msa_rz + msa_ss
                    msa_s[j] = msa_rz[j] + msa_ss[j];
                    s[0] += msa_rz[j];
                    s[2] += msa_ss[j];
                    s[4] += msa_s[j];
                    s[1] += sa_rz[j];
                    s[3] += sa_ss[j];
                    s[5] += sa_s[j];
                    continue;
                }
            }
            s[0] += msa_rz[j] * sa_rz[j];
            s[1] += sa_rz[j];
            s[2] += msa_ss[j] * sa_ss[j];
            s[3] += sa_ss[j];
            s[4] += msa_s[j] * sa_s[j];
            s[5] += sa_s[j];
        }
        blk_sum<W, 6>(B, s);
        if (B.tid == 0) {
            for (int k = 0; k < 3; ++k) {
                double C = (s[2 * k + 1] > 0 ? s[2 * k] / s[2 * k + 1] : 0);
                if (ANION && P.tracer != RH_SAS_TRACER_BROMIDE) C = s[2 * k] / s[2 * k + 1];   // unguarded, M_* untouched
                ((double *)P.a[SA_C_rz + k])[cell] = C;
                if constexpr (ANION) {
                    if (P.tracer == RH_SAS_TRACER_BROMIDE) ((double *)P.a[SA_M_rz + k])[cell] = 0.0;
                } else {
                    ((double *)P.a[SA_C_iso_rz + k])[cell] = conc_to_delta(P, C) * mk;
                }
            }
        }
        if (P.diag) {
            double *o0 = (double *)P.a[SA_sa_s] + cell * A, *o1 = (double *)P.a[SA_msa_s] + cell * A;
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (base + j < A) {
                    o0[base + j] = sa_s[j];
                    o1[base + j] = msa_s[j];
                }
        }
    }

    if (P.stages & RH_SAS_AGEING) {
        if constexpr (ANION) {
            ageing_anion<W, E>(B, A, base, sa_rz, msa_rz);
            ageing_anion<W, E>(B, A, base, sa_ss, msa_ss);
        } else {
            ageing<W, E>(B, A, base, sa_rz, msa_rz);
            ageing<W, E>(B, A, base, sa_ss, msa_ss);
        }
    }

    {
        double *g0 = (double *)P.a[SA_sa_rz] + cell * A, *g1 = (double *)P.a[SA_msa_rz] + cell * A;
        double *g2 = (double *)P.a[SA_sa_ss] + cell * A, *g3 = (double *)P.a[SA_msa_ss] + cell * A;
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (base + j < A) {
                g0[base + j] = sa_rz[j];
                g1[base + j] = msa_rz[j];
                g2[base + j] = sa_ss[j];
                g3[base + j] = msa_ss[j];
            }
    }
}

__global__ void k_selftest_div(const double *a, const double *d, double *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = udiv(a[i], udiv_prepare(d[i]));
}
__global__ void k_selftest_pow(const double *x, const double *k, double *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const PowConsts C = load_pow_consts(SAS_LOG_T);
    if (i >= n) return;
    // the same dispatch as the power-law branch of calc_tt_family: exponents 0.5 / 1.5 / 1 through the square root
    if (k[i] == 0.5) out[i] = sqrt_unit(x[i]);
    else if (k[i] == 1.5) out[i] = x[i] * sqrt_unit(x[i]);
    else if (k[i] == 1.0) out[i] = x[i];
    else out[i] = sas_pow_ratio(C, x[i], 1.0, 0.0, k[i]);
}

// ---------------------------------------------------------------------------------------------
// host side: context and C ABI
// ---------------------------------------------------------------------------------------------
struct rh_sas_ctx {
    rh_sas_config cfg;
    hipStream_t stream;
    bool own_stream;
    void *arr[SA_COUNT];
    int64_t elems[SA_COUNT];
    int *unsupported;
    bool timing;
    std::vector<hipEvent_t> events;
    size_t ev_used;
    std::string err;
};
static std::string g_sas_create_err;

static int sfail(rh_sas_ctx *ctx, int code, const std::string &msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_sas_create_err = msg;
    return code;
}
#define SHIPCHK(ctx, call)                                                                                      \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) return sfail(ctx, RH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

static int64_t sas_elems(const rh_sas_config &c, int a) {
    const int when = SAS_WHEN[a];
    if (when == W_STATS && !c.age_statistics) return 0;
    if (when == W_DIAG && !c.keep_distributions) return 0;
    if (when == W_ANION && c.tracer == RH_SAS_TRACER_OXYGEN18) return 0;
    switch (SAS_KIND[a]) {
    case K_AGE: return c.n_cells * c.ages;
    case K_NAGE: return c.n_cells * (c.ages + 1);
    case K_CELL: return c.n_cells;
    case K_DAILY: return c.n_cells * c.forcing_days;
    case K_PARAM: return c.n_cells * 8;
    case K_MASK: return c.n_cells;
    }
    return 0;
}

template <int W, int E>
static void launch_sas(rh_sas_ctx *ctx, const SasArgs &args) {
    if (ctx->cfg.tracer != RH_SAS_TRACER_OXYGEN18)
        hipLaunchKernelGGL((k_sas<W, E, true>), dim3((unsigned)ctx->cfg.n_cells), dim3(W * 64), 0, ctx->stream, args);
    else
        hipLaunchKernelGGL((k_sas<W, E, false>), dim3((unsigned)ctx->cfg.n_cells), dim3(W * 64), 0, ctx->stream, args);
}

template <int W>
static void launch_sas8(rh_sas_ctx *ctx, const SasArgs &args) {
    if (ctx->cfg.tracer != RH_SAS_TRACER_OXYGEN18)
        hipLaunchKernelGGL((k_sas8<W, true>), dim3((unsigned)ctx->cfg.n_cells), dim3(W * 64), 0, ctx->stream, args);
    else
        hipLaunchKernelGGL((k_sas8<W, false>), dim3((unsigned)ctx->cfg.n_cells), dim3(W * 64), 0, ctx->stream, args);
}

extern "C" {

void rh_sas_default_config(rh_sas_config *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->n_cells = 1;
    cfg->ages = 1000;         // benchmarks/SVATOXYGEN18_benchmark.py:28-44
    cfg->substeps = 1;        // settings.sas_solver_substeps default, roger/settings.py:120
    cfg->forcing_days = 1;
    cfg->vsmow = 2005.2e-6;   // roger/settings.py:76-78
    cfg->d18O_min = -20;
    cfg->d18O_max = 0;
}

const char *rh_sas_last_error(const rh_sas_ctx *ctx) { return ctx ? ctx->err.c_str() : g_sas_create_err.c_str(); }
int rh_sas_num_arrays(void) { return SA_COUNT; }
const char *rh_sas_array_name(int a) { return (a >= 0 && a < SA_COUNT) ? SAS_NAMES[a] : nullptr; }
int rh_sas_array_is_int(int a) { return (a >= 0 && a < SA_COUNT) ? (SAS_KIND[a] == K_MASK) : -1; }
int rh_sas_array_index(const char *name) {
    if (!name) return -1;
    for (int a = 0; a < SA_COUNT; ++a)
        if (std::strcmp(SAS_NAMES[a], name) == 0) return a;
    return -1;
}
int64_t rh_sas_array_elems(const rh_sas_ctx *ctx, int a) { return (ctx && a >= 0 && a < SA_COUNT) ? ctx->elems[a] : 0; }

int rh_sas_create(const rh_sas_config *cfg, rh_sas_ctx **out) {
    if (!cfg || !out) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: null argument");
    if (cfg->n_cells <= 0 || cfg->n_cells > 0x7fffffffLL) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: n_cells out of range");
    if (cfg->ages < 2 || cfg->ages + 1 > RH_SAS_MAX_NAGES)
        return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: ages must be in [2, RH_SAS_MAX_NAGES - 1]");
    if (cfg->substeps < 1 || cfg->forcing_days < 1) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: substeps and forcing_days must be >= 1");
    if (cfg->tracer < RH_SAS_TRACER_OXYGEN18 || cfg->tracer > RH_SAS_TRACER_VIRTUAL)
        return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: tracer must be RH_SAS_TRACER_OXYGEN18, _BROMIDE, _CHLORIDE or _VIRTUAL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return sfail(nullptr, RH_ERR_NODEVICE, "rh_sas_create: no HIP device visible (this backend has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: device ordinal out of range");
    SHIPCHK(nullptr, hipSetDevice(cfg->device));
    rh_sas_ctx *ctx = new (std::nothrow) rh_sas_ctx();
    if (!ctx) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: out of host memory");
    ctx->cfg = *cfg;
    ctx->stream = nullptr;
    ctx->own_stream = false;
    ctx->unsupported = nullptr;
    ctx->timing = false;
    ctx->ev_used = 0;
    for (int a = 0; a < SA_COUNT; ++a) {
        ctx->arr[a] = nullptr;
        ctx->elems[a] = 0;
    }
    auto bail = [&](hipError_t e, const char *what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        rh_sas_destroy(ctx);
        return sfail(nullptr, RH_ERR_HIP, msg);
    };
    hipError_t e;
    if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) return bail(e, "hipStreamCreate");
    ctx->own_stream = true;
    for (int a = 0; a < SA_COUNT; ++a) {
        const int64_t ne = sas_elems(*cfg, a);
        if (!ne) continue;
        const size_t bytes = (size_t)ne * (SAS_KIND[a] == K_MASK ? sizeof(int32_t) : sizeof(double));
        if ((e = hipMalloc(&ctx->arr[a], bytes)) != hipSuccess) return bail(e, "hipMalloc(SAS array)");
        if ((e = hipMemsetAsync(ctx->arr[a], 0, bytes, ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
        ctx->elems[a] = ne;
    }
    if ((e = hipMalloc((void **)&ctx->unsupported, sizeof(int))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemsetAsync(ctx->unsupported, 0, sizeof(int), ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
    {   // maskCatch defaults to 1 (roger/variables.py)
        std::vector<int32_t> ones((size_t)cfg->n_cells, 1);
        if ((e = hipMemcpyAsync(ctx->arr[SA_maskCatch], ones.data(), ones.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return bail(e, "hipMemcpy(maskCatch)");
        if (cfg->tracer != RH_SAS_TRACER_OXYGEN18) {   // alpha_transp, alpha_q: initial=1 (roger/variables.py:5377-5405)
            std::vector<double> one((size_t)cfg->n_cells, 1.0);
            for (int a : {SA_alpha_transp, SA_alpha_q})
                if ((e = hipMemcpyAsync(ctx->arr[a], one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
                    return bail(e, "hipMemcpy(alpha)");
        }
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    }
    *out = ctx;
    return RH_OK;
}

void rh_sas_destroy(rh_sas_ctx *ctx) {
    if (!ctx) return;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->events) (void)hipEventDestroy(ev);
    for (int a = 0; a < SA_COUNT; ++a)
        if (ctx->arr[a]) (void)hipFree(ctx->arr[a]);
    if (ctx->unsupported) (void)hipFree(ctx->unsupported);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rh_sas_set_stream(rh_sas_ctx *ctx, void *hip_stream) {
    if (!ctx) return RH_ERR_ARG;
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) SHIPCHK(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return RH_OK;
}

int rh_sas_sync(rh_sas_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    int bad = 0;
    SHIPCHK(ctx, hipMemcpyAsync(&bad, ctx->unsupported, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (bad)
        return sfail(ctx, RH_ERR_STATE,
                     "a column's sas_params select a code that is none of the reference's SAS families (1, 2, 3, 31-37, 4, 51, 52, 6, 61, 62)");
    return RH_OK;
}

static int sas_check(rh_sas_ctx *ctx, int a, size_t bytes, const void *host, const char *who) {
    if (!ctx) return RH_ERR_ARG;
    if (a < 0 || a >= SA_COUNT) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": unknown array id");
    if (!ctx->arr[a]) return sfail(ctx, RH_ERR_STATE, std::string(who) + ": array " + SAS_NAMES[a] + " is not held by this context (age_statistics / keep_distributions)");
    const size_t want = (size_t)ctx->elems[a] * (SAS_KIND[a] == K_MASK ? sizeof(int32_t) : sizeof(double));
    if (bytes != want) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": size mismatch for array " + SAS_NAMES[a]);
    if (!host) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": null host pointer");
    return RH_OK;
}

int rh_sas_upload(rh_sas_ctx *ctx, int a, const void *host, size_t bytes) {
    const int rc = sas_check(ctx, a, bytes, host, "rh_sas_upload");
    if (rc) return rc;
    SHIPCHK(ctx, hipMemcpyAsync(ctx->arr[a], host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}

int rh_sas_download(rh_sas_ctx *ctx, int a, void *host, size_t bytes) {
    const int rc = sas_check(ctx, a, bytes, host, "rh_sas_download");
    if (rc) return rc;
    SHIPCHK(ctx, hipMemcpyAsync(host, ctx->arr[a], bytes, hipMemcpyDeviceToHost, ctx->stream));
    return rh_sas_sync(ctx);
}

static int sas_check_cells(rh_sas_ctx *ctx, int a, int64_t first, int64_t cnt, size_t bytes, const void *host, const char *who,
                           size_t *offset) {
    if (!ctx) return RH_ERR_ARG;
    if (a < 0 || a >= SA_COUNT) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": unknown array id");
    if (!ctx->arr[a]) return sfail(ctx, RH_ERR_STATE, std::string(who) + ": array " + SAS_NAMES[a] + " is not held by this context");
    if (SAS_KIND[a] == K_DAILY) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": daily inputs are (forcing_days, n_cells); use the whole-array call");
    if (first < 0 || cnt <= 0 || first + cnt > ctx->cfg.n_cells) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": cell range out of bounds");
    const size_t row = (size_t)(ctx->elems[a] / ctx->cfg.n_cells) * (SAS_KIND[a] == K_MASK ? sizeof(int32_t) : sizeof(double));
    if (bytes != row * (size_t)cnt) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": size mismatch for array " + SAS_NAMES[a]);
    if (!host) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": null host pointer");
    *offset = row * (size_t)first;
    return RH_OK;
}

int rh_sas_upload_cells(rh_sas_ctx *ctx, int a, int64_t first_cell, int64_t n_cells, const void *host, size_t bytes) {
    size_t off;
    const int rc = sas_check_cells(ctx, a, first_cell, n_cells, bytes, host, "rh_sas_upload_cells", &off);
    if (rc) return rc;
    SHIPCHK(ctx, hipMemcpyAsync((char *)ctx->arr[a] + off, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}

int rh_sas_download_cells(rh_sas_ctx *ctx, int a, int64_t first_cell, int64_t n_cells, void *host, size_t bytes) {
    size_t off;
    const int rc = sas_check_cells(ctx, a, first_cell, n_cells, bytes, host, "rh_sas_download_cells", &off);
    if (rc) return rc;
    SHIPCHK(ctx, hipMemcpyAsync(host, (const char *)ctx->arr[a] + off, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return rh_sas_sync(ctx);
}

int rh_sas_set_daily_from_device(rh_sas_ctx *ctx, int a, int64_t day_row, const double *dev_src) {
    if (!ctx) return RH_ERR_ARG;
    if (a < 0 || a >= SA_COUNT || SAS_KIND[a] != K_DAILY) return sfail(ctx, RH_ERR_ARG, "rh_sas_set_daily_from_device: not a daily input array");
    if (day_row < 0 || day_row >= ctx->cfg.forcing_days || !dev_src) return sfail(ctx, RH_ERR_ARG, "rh_sas_set_daily_from_device: bad row or null source");
    SHIPCHK(ctx, hipMemcpyAsync((double *)ctx->arr[a] + day_row * ctx->cfg.n_cells, dev_src, (size_t)ctx->cfg.n_cells * sizeof(double),
                                hipMemcpyDeviceToDevice, ctx->stream));
    return RH_OK;
}

void *rh_sas_array_device_ptr(rh_sas_ctx *ctx, int a) { return (ctx && a >= 0 && a < SA_COUNT) ? ctx->arr[a] : nullptr; }

int rh_sas_stages(rh_sas_ctx *ctx, int64_t day, int stages) {
    if (!ctx) return RH_ERR_ARG;
    if (day < 0) return sfail(ctx, RH_ERR_ARG, "rh_sas_stages: negative day");
    if ((stages & ~(RH_SAS_ALL | RH_SAS_RESCALE)) || !stages) return sfail(ctx, RH_ERR_ARG, "rh_sas_stages: bad stage mask");
    const rh_sas_config &c = ctx->cfg;
    if ((stages & RH_SAS_STORAGE) && c.age_statistics && !c.keep_distributions &&
        (!(stages & RH_SAS_TRANSP) || !(stages & RH_SAS_Q_SS)))
        return sfail(ctx, RH_ERR_STATE,
                     "rh_sas_stages: age statistics need the transpiration and percolation stages in the same launch, or keep_distributions");
    SasArgs args;
    args.n = c.n_cells;
    args.day_off = (day % c.forcing_days) * c.n_cells;
    args.ages = c.ages;
    args.substeps = c.substeps;
    args.stages = stages;
    args.stats = c.age_statistics ? 1 : 0;
    args.diag = c.keep_distributions ? 1 : 0;
    args.tracer = c.tracer;
    args.vsmow = c.vsmow;
    args.dmin = c.d18O_min;
    args.dmax = c.d18O_max;
    args.unsupported = ctx->unsupported;
    for (int a = 0; a < SA_COUNT; ++a) args.a[a] = ctx->arr[a];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->timing) {
        if (ctx->ev_used + 2 > ctx->events.size()) {
            for (int k = 0; k < 2; ++k) {
                hipEvent_t ev;
                SHIPCHK(ctx, hipEventCreate(&ev));
                ctx->events.push_back(ev);
            }
        }
        ev0 = ctx->events[ctx->ev_used];
        ev1 = ctx->events[ctx->ev_used + 1];
        ctx->ev_used += 2;
        SHIPCHK(ctx, hipEventRecord(ev0, ctx->stream));
    }
    // smallest workgroup whose blocked layout covers the ages + 1 edges: waves x classes per thread
    const int nages = c.ages + 1;
    // eight classes per thread from 257 age classes on (9.63 against 10.17 ms per day at 10^5 columns x 1000 ages); RH_SAS_E4=1: the
    // four-class shapes for comparison
    static const bool e4 = std::getenv("RH_SAS_E4") != nullptr;
    if (!e4 && nages > 256 && nages <= 4096) {
        if (nages <= 512) launch_sas8<1>(ctx, args);
        else if (nages <= 1024) launch_sas8<2>(ctx, args);
        else if (nages <= 2048) launch_sas8<4>(ctx, args);
        else launch_sas8<8>(ctx, args);
    } else
    if (nages <= 64) launch_sas<1, 1>(ctx, args);
    else if (nages <= 128) launch_sas<1, 2>(ctx, args);
    else if (nages <= 256) launch_sas<1, 4>(ctx, args);
    else if (nages <= 512) launch_sas<2, 4>(ctx, args);
    else if (nages <= 1024) launch_sas<4, 4>(ctx, args);
    else if (nages <= 2048) launch_sas<8, 4>(ctx, args);
    else launch_sas<16, 4>(ctx, args);
    SHIPCHK(ctx, hipGetLastError());
    if (ctx->timing) SHIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
    return RH_OK;
}

int rh_sas_step(rh_sas_ctx *ctx, int64_t day) { return rh_sas_stages(ctx, day, RH_SAS_ALL); }

int rh_sas_run_days(rh_sas_ctx *ctx, int64_t day0, int64_t ndays) {
    if (!ctx) return RH_ERR_ARG;
    if (ndays < 0) return sfail(ctx, RH_ERR_ARG, "rh_sas_run_days: negative ndays");
    for (int64_t d = 0; d < ndays; ++d) {
        const int rc = rh_sas_stages(ctx, day0 + d, RH_SAS_ALL);
        if (rc) return rc;
    }
    return RH_OK;
}

static int selftest2(const double *x, const double *k, double *out, int64_t n, int which);
int rh_sas_selftest_pow(const double *x, const double *k, double *out, int64_t n) { return selftest2(x, k, out, n, 0); }
int rh_sas_selftest_div(const double *a, const double *d, double *out, int64_t n) { return selftest2(a, d, out, n, 1); }
static int selftest2(const double *x, const double *k, double *out, int64_t n, int which) {
    if (!x || !k || !out || n <= 0) return RH_ERR_ARG;
    double *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * 3 * sizeof(double)) != hipSuccess) return RH_ERR_HIP;
    int rc = RH_OK;
    if (hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, k, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        rc = RH_ERR_HIP;
    if (rc == RH_OK) {
        if (which == 0)
            hipLaunchKernelGGL(k_selftest_pow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, d + n, d + 2 * n, n);
        else
            hipLaunchKernelGGL(k_selftest_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, d + n, d + 2 * n, n);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(out, d + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = RH_ERR_HIP;
    }
    (void)hipFree(d);
    return rc;
}

int rh_sas_enable_timing(rh_sas_ctx *ctx, int on) {
    if (!ctx) return RH_ERR_ARG;
    ctx->timing = on != 0;
    ctx->ev_used = 0;
    return RH_OK;
}

int rh_sas_timing_summary(rh_sas_ctx *ctx, double *total_ms, int64_t *launches) {
    if (!ctx || !total_ms || !launches) return RH_ERR_ARG;
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0;
    for (size_t k = 0; k + 1 < ctx->ev_used; k += 2) {
        float ms = 0;
        SHIPCHK(ctx, hipEventElapsedTime(&ms, ctx->events[k], ctx->events[k + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)(ctx->ev_used / 2);
    return RH_OK;
}

}  // extern "C"
