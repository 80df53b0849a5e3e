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
#include "rh_sas_dev.h"

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
    const double pr[7] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6]};  // read once: the loop below stores nothing, but the compiler cannot know
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
        sas_omega<W, E, FAM>(B, C, pr, SA_hi, Smax, mk, base, A, Om, Om_edge0);
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
// Sixteen age classes per thread, ONE wavefront per column (ages <= 1023): no workgroup barrier and no exchange through LDS at all, the
// scans are the wave's DPP scan plus a running sum over the thread's own classes; the state of a column then takes the register file of
// a whole SIMD (1 wave/SIMD, 512 registers: VGPRs + AGPRs).  -DRH_SAS_EXPERIMENT_E16 builds ONLY this shape (experiments).
#define SAS_OCCUPANCY_E16 __attribute__((amdgpu_waves_per_eu(1, 1)))
template <bool ANION>
__global__ __launch_bounds__(64) SAS_OCCUPANCY_E16 void k_sas16(const SasArgs P) {
    sas_body<1, 16, ANION>(P);
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
    if (cfg->solver < RH_SAS_SOLVER_DETERMINISTIC || cfg->solver > RH_SAS_SOLVER_RK4)
        return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: solver must be RH_SAS_SOLVER_DETERMINISTIC, _EULER or _RK4");
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
    if (c.solver != RH_SAS_SOLVER_DETERMINISTIC && (stages & RH_SAS_ALL)) {
        // the explicit solvers evaluate all fluxes of a sub-step on one state: the day cannot be cut into stages
        if (stages != RH_SAS_ALL)
            return sfail(ctx, RH_ERR_ARG, "rh_sas_stages: with an explicit solver the day runs in one launch (RH_SAS_ALL); only RH_SAS_RESCALE may run on its own");
        const int rc = rh_sas_launch_solver(c.solver, ctx->stream, args);
        if (rc) return sfail(ctx, rc, "rh_sas_stages: unknown solver");
        SHIPCHK(ctx, hipGetLastError());
        if (ctx->timing) SHIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
        return RH_OK;
    }
    // smallest workgroup whose blocked layout covers the ages + 1 edges: waves x classes per thread
    const int nages = c.ages + 1;
    // eight classes per thread from 257 age classes on (9.63 against 10.17 ms per day at 10^5 columns x 1000 ages); RH_SAS_E4=1: the
    // four-class shapes for comparison
#ifdef RH_SAS_EXPERIMENT_E16
    if (nages > 1024) return sfail(ctx, RH_ERR_ARG, "RH_SAS_EXPERIMENT_E16: ages <= 1023 only");
    if (ctx->cfg.tracer != RH_SAS_TRACER_OXYGEN18) hipLaunchKernelGGL((k_sas16<true>), dim3((unsigned)ctx->cfg.n_cells), dim3(64), 0, ctx->stream, args);
    else hipLaunchKernelGGL((k_sas16<false>), dim3((unsigned)ctx->cfg.n_cells), dim3(64), 0, ctx->stream, args);
    SHIPCHK(ctx, hipGetLastError());
    if (ctx->timing) SHIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
    return RH_OK;
#else
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
#endif
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
