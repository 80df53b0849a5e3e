// rh_sas_kernels.h -- the deterministic SAS kernels (one day of transport per column) and their launch; included by the two translation
// units rh_sas_det_iso.hip / rh_sas_det_anion.hip, which instantiate the shapes for the isotope and for the anion kernels (compiled side
// by side: each takes minutes).
#pragma once
#include <cstdlib>

#include "rh_sas_dev.h"

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
        if constexpr (FAM == FAM_KUMARASWAMI || FAM == FAM_EXPONENTIAL || FAM == FAM_GAMMA)
            omega_library_families<E>(FAM, pr, SA_hi, Smax, mk, base, A, Om);   // out of line: rh_sas_dev.h
        else
            sas_omega<W, E, FAM>(B, C, pr, SA_hi, Smax, mk, base, A, Om, Om_edge0);
        double Om_lo, unused;
        blk_prev2<W>(B, Om[E - 1], 0.0, Om_edge0, 0.0, Om_lo, unused);
        // Written stage by stage over the thread's classes, not class by class: the E chains are independent, and in the class-by-class
        // form the compiler emitted them one after the other through a single temporary (eleven dependent fp64 operations each).
        double tq[E], tti[E];
#pragma unroll
        for (int j = 0; j < E; ++j) tq[j] = fmax(Om[j] - (j == 0 ? Om_lo : Om[j - 1]), 0.0);   // :430-433  where(d >= 0, d, 0)
#pragma unroll
        for (int j = 0; j < E; ++j) tq[j] = flux * tq[j];
#pragma unroll
        for (int j = 0; j < E; ++j) tq[j] = min_raw(tq[j] * h, san[j]);                            // :435-438  where(flux t h > san, san, flux t h)
#pragma unroll
        for (int j = 0; j < E; ++j) tti[j] = tq[j] * by_fh.r;                                   // :440-443: q / (flux * h), fh > 0 here
#pragma unroll
        for (int j = 0; j < E; ++j) tq[j] = __builtin_fma(-by_fh.d, tti[j], tq[j]);             //   (udiv, its three steps stage by stage)
#pragma unroll
        for (int j = 0; j < E; ++j) tti[j] = __builtin_fma(tq[j], by_fh.r, tti[j]);
#pragma unroll
        for (int j = 0; j < E; ++j) tq[j] = -tti[j] * flux;
#pragma unroll
        for (int j = 0; j < E; ++j) san[j] = san[j] + tq[j] * h;                                // :445-448
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
        if (code != 52 && B.tid == 0 && flux * (1 / (double)P.substeps) > 0) atomicOr(P.unsupported, 1);
        calc_tt_family<W, E, FAM_NONE>(B, P, p, flux, sa, mk, base, tt);
    }
}


// The array of flux f (0 .. 4) out of a group of five consecutive registry entries: a chain of selects over constant indices on the scalar
// unit (an index computed at run time into the kernel's argument block would move the whole block into scratch memory).
SAS_DEV double *flux_arr(const SasArgs &P, int first, int f) {
    void *q = P.a[first];
    q = (f == 1) ? P.a[first + 1] : q;
    q = (f == 2) ? P.a[first + 2] : q;
    q = (f == 3) ? P.a[first + 3] : q;
    q = (f == 4) ? P.a[first + 4] : q;
    return (double *)q;
}

// One outgoing flux: SA, tt, TT, mtt, C, C_iso, the sink's isotope mixing, update_sa.
// calc_evaporation/transpiration_transport_iso_kernel (core/evapotranspiration.py:653-719, 831-901),
// calc_percolation_rz/ss_transport_iso_kernel (core/subsurface_runoff.py:1531-1626, 1753-1820),
// calc_capillary_rise_rz_transport_iso_kernel (core/capillary_rise.py:404-500).
// sa / msa: the compartment the flux leaves (the ACTIVE one, in registers); sa_o / msa_o: the other compartment, which receives the
// water if `sink` and is otherwise untouched -- with the eight-class shapes it waits in the parking area (it is parked when this is
// called and when it returns).  ONE instance serves the five fluxes (the caller's loop over f; sink, keep and the arrays of the flux are
// run-time values): with an instance per flux the kernel was 540 KB of code and the compiler kept the addresses of five fluxes' outputs
// alive at once.  keep: the age statistics of the travel time distribution (transpiration, percolation of the subsoil).
template <int W, int E>
SAS_DEV void outflux(Blk<W> &B, const SasArgs &P, int64_t cell, int f, bool sink, bool keep, double (&sa)[E], double (&msa)[E],
                     double (&sa_o)[E], double (&msa_o)[E], double mk, int base) {
    const int A = P.ages;
    const double flux = B.scal[SC_FLUX + f];
    const double *p = B.scal + SC_PAR + 7 * f;
    double tt[E];
    calc_tt<W, E>(B, P, p, flux, sa, mk, base, tt);
    SAS_PH(B, P, 2 + 2 * f);
    double mtt[E], s[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        tt[j] *= mk;
        mtt[j] = (tt[j] > 0 ? msa[j] : 0) * mk;  // calc_mtt :565-596 with alpha = 1
        s[0] += mtt[j] * tt[j];
        s[1] += tt[j];
    }
    if (P.diag || (keep && stats_now(P))) {  // TT[1:] = cumsum(tt)
        double TT_hi[E], TT_lo;
        blk_cumsum<W, E, false>(B, tt, TT_hi, TT_lo, nullptr, base, 0);
        if (keep && stats_now(P)) {
            double *dst[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) dst[q] = (double *)(f == 1 ? P.a[SA_tt10_transp + q] : P.a[SA_tt10_q_ss + q]);
            age_stats<W, E>(B, P, cell, base, TT_hi, TT_lo, tt, dst, false);
        }
        if (P.diag) {
            double *o_tt = flux_arr(P, SA_tt_evap_soil, f) + cell * A;
            double *o_mtt = flux_arr(P, SA_mtt_evap_soil, f) + cell * A;
            double *o_TT = flux_arr(P, SA_TT_evap_soil, f) + cell * (A + 1);
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
        flux_arr(P, SA_C_evap_soil, f)[cell] = C;
        flux_arr(P, SA_C_iso_evap_soil, f)[cell] = conc_to_delta(P, C) * mk;
    }
    if (sink) {
        unpark2<W, E>(B, sa_o, msa_o);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double add = tt[j] * flux;
            const UDiv by_tot = udiv_prepare(add + sa_o[j]);   // two quotients by one divisor (udiv: the bits of `/`)
            msa_o[j] = (add + sa_o[j] > 0 ? msa_o[j] * udiv(sa_o[j], by_tot) + mtt[j] * udiv(add, by_tot) : msa_o[j]) * mk;
            sa_o[j] += tt[j] * flux * mk;
        }
        park2<W, E>(B, sa_o, msa_o);
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        double v = sa[j] + -flux * tt[j];  // update_sa :599-619
        v = ((v > -1e-5) && (v < 0)) ? 0 : v;
        sa[j] = v * mk;
        msa[j] = (sa[j] <= 0 ? 0 : msa[j]) * mk;
    }
    SAS_PH(B, P, 3 + 2 * f);
}

// Infiltration into age class 0: calc_infiltration_rz_transport_iso_kernel (core/infiltration.py:2218-2346)
// and calc_infiltration_ss_transport_iso_kernel (:2441-2512).  which: 0 matrix -> rz, 1 pf -> rz, 2 pf -> ss.
template <int W, int E>
SAS_DEV void inflow(Blk<W> &B, const SasArgs &P, int64_t cell, int which, double (&sa)[E], double (&msa)[E], double mk, int base) {
    const double inf = B.scal[SC_INF + which];
    const double C_in = B.scal[SC_CIN];
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
template <int W, int E>
SAS_DEV void record_dist(Blk<W> &B, const SasArgs &P, int64_t cell, int f, bool keep, int base, const double (&tt)[E], const double (&mtt)[E],
                         bool with_mtt) {
    if (!(P.diag || (keep && stats_now(P)))) return;
    const int A = P.ages;
    double TT_hi[E], TT_lo;
    blk_cumsum<W, E, false>(B, tt, TT_hi, TT_lo, nullptr, base, 0);
    if (keep && stats_now(P)) {
        double *dst[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dst[q] = (double *)(f == 1 ? P.a[SA_tt10_transp + q] : P.a[SA_tt10_q_ss + q]);
        age_stats<W, E>(B, P, cell, base, TT_hi, TT_lo, tt, dst, false);
    }
    if (P.diag) {
        double *o_tt = flux_arr(P, SA_tt_evap_soil, f) + cell * A;
        double *o_mtt = flux_arr(P, SA_mtt_evap_soil, f) + cell * A;
        double *o_TT = flux_arr(P, SA_TT_evap_soil, f) + cell * (A + 1);
        if (B.tid == 0) o_TT[0] = 0.0;
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (base + j < A) {
                o_tt[base + j] = tt[j];
                if (with_mtt) o_mtt[base + j] = mtt[j];
                o_TT[base + j + 1] = TT_hi[j];
            }
    }
}

// One outgoing flux of the anion kernels.  water: calc_evaporation_transport_kernel (core/evapotranspiration.py:620-650),
// the solute stays behind.  Otherwise calc_transpiration_transport_anion_kernel (:905-985),
// calc_percolation_rz/ss_transport_anion_kernel (core/subsurface_runoff.py:1630-1716, 1823-1893),
// calc_capillary_rise_rz_transport_anion_kernel (core/capillary_rise.py:503-590).  One instance for the five fluxes, as outflux above.
template <int W, int E>
SAS_DEV void outflux_anion(Blk<W> &B, const SasArgs &P, int64_t cell, int f, bool sink, bool keep, bool water, double alpha,
                           double (&sa)[E], double (&msa)[E], double (&sa_o)[E], double (&msa_o)[E], double mk, int base) {
    const double flux = B.scal[SC_FLUX + f];
    const double *p = B.scal + SC_PAR + 7 * f;
    double tt[E];
    calc_tt<W, E>(B, P, p, flux, sa, mk, base, tt);
    double mtt[E], s[1] = {0.0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        tt[j] *= mk;
        mtt[j] = 0.0;
        if (!water) {
            double m = (sa[j] > 0 ? msa[j] / sa[j] : 0) * alpha * tt[j] * flux;
            m = (m <= 0 ? 0 : m);
            m = (m > msa[j] ? msa[j] : m);
            mtt[j] = m * mk;
            s[0] += mtt[j];
        }
    }
    record_dist<W, E>(B, P, cell, f, keep, base, tt, mtt, !water);
    if (!water) {
        blk_sum<W, 1>(B, s);
        if (B.tid == 0) {
            flux_arr(P, SA_C_evap_soil, f)[cell] = (flux > 0 ? s[0] / flux : 0) * mk;
            flux_arr(P, SA_M_evap_soil, f)[cell] = s[0] * mk;
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        double v = sa[j] + -flux * tt[j];  // update_sa :599-619
        v = ((v > -1e-5) && (v < 0)) ? 0 : v;
        sa[j] = v * mk;
        if (!water) msa[j] += -mtt[j] * mk;
    }
    if (sink) {
        unpark2<W, E>(B, sa_o, msa_o);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            msa_o[j] += mtt[j] * mk;
            sa_o[j] += tt[j] * flux * mk;
        }
        park2<W, E>(B, sa_o, msa_o);
    }
}

// calc_infiltration_rz_transport_anion_kernel (core/infiltration.py:2350-2424): matrix and preferential-flow
// infiltration join age class 0 in one addition; calc_infiltration_ss_transport_anion_kernel (:2516-2566).
template <int W, int E>
SAS_DEV void inflow_anion(Blk<W> &B, const SasArgs &P, int64_t cell, bool subsoil, double (&sa)[E], double (&msa)[E], double mk,
                          int base) {
    const double C_in = B.scal[SC_CIN];
    double d_sa, d_msa;
    if (!subsoil) {
        const double im = B.scal[SC_INF + 0];
        const double ip = B.scal[SC_INF + 1];
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
        const double ip = B.scal[SC_INF + 2];
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
#ifndef RH_SAS_E8_WAVES
#define RH_SAS_E8_WAVES 2
#endif
#define SAS_OCCUPANCY_E8 __attribute__((amdgpu_waves_per_eu(RH_SAS_E8_WAVES, RH_SAS_E8_WAVES)))
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
#ifndef RH_SAS_E16_WAVES
#define RH_SAS_E16_WAVES 1
#endif
#define SAS_OCCUPANCY_E16 __attribute__((amdgpu_waves_per_eu(RH_SAS_E16_WAVES, RH_SAS_E16_WAVES)))
template <bool ANION>
__global__ __launch_bounds__(64) SAS_OCCUPANCY_E16 void k_sas16(const SasArgs P) {
    sas_body<1, 16, ANION>(P);
}
template <int W, int E, bool ANION>
__device__ __forceinline__ void sas_body(const SasArgs &P) {
    __shared__ double s_red[2][W][8];
    __shared__ double s_xch[2][W][2];
    __shared__ double2 s_logt[64];
    __shared__ double s_park[E >= 8 ? 2 * E * W * 64 : 1];   // two age vectors per thread (park2): the shapes with eight classes per thread or more
    __shared__ double s_scal[SC_COUNT];
    if (threadIdx.x < 64) s_logt[threadIdx.x] = SAS_LOG_T[threadIdx.x];
    Blk<W> B;
#ifdef RH_SAS_PHASES
    B.t_last = clock64();
#endif
    B.scal = s_scal;
    B.park = (E >= 8) ? s_park : nullptr;
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
    // the column's scalars and its state are requested together (one round trip to HBM, not two), the scalars go to LDS behind it
    double scal_regs[SC_COUNT];
    sas_fetch_scalars(P, scal_regs);
    // the catchment mask: a bool in the reference (variables.py:462-470) -- 0 or 1, which makes (x * mk) * mk == x * mk exactly
    const double mk = ((const int *)P.a[SA_maskCatch])[cell] != 0 ? 1.0 : 0.0;

    // ca / cma: the ACTIVE compartment (the one the current flux leaves), co / cmo: the other one -- which, with the eight-class shapes,
    // waits in the parking area while the fluxes run (its registers are free).  Root zone first (evaporation, transpiration, percolation
    // into the subsoil), then the compartments change places (subsoil percolation, capillary rise into the root zone).
    double ca[E], cma[E], co[E], cmo[E];
    {
        const double *g0 = (const double *)P.a[SA_sa_rz] + cell * A, *g1 = (const double *)P.a[SA_msa_rz] + cell * A;
        const double *g2 = (const double *)P.a[SA_sa_ss] + cell * A, *g3 = (const double *)P.a[SA_msa_ss] + cell * A;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const bool in = base + j < A;
            ca[j] = in ? g0[base + j] : 0.0;
            cma[j] = in ? g1[base + j] : 0.0;
            co[j] = in ? g2[base + j] : 0.0;
            cmo[j] = in ? g3[base + j] : 0.0;
        }
    }
    sas_publish_scalars(scal_regs, s_scal);
    __syncthreads();
    park2<W, E>(B, co, cmo);
    SAS_PH(B, P, 0);
    const bool have_transp = P.stages & RH_SAS_TRANSP, have_q_ss = P.stages & RH_SAS_Q_SS;
    const bool stats = P.stats && (P.stages & RH_SAS_STORAGE);

    // order of svat_transport_model_deterministic :949-991; ONE instance of the flux code, run five times
    double alpha_q = 0.0;
    if constexpr (ANION) {
        alpha_q = ((const double *)P.a[SA_alpha_q])[cell];
        if (P.stages & RH_SAS_INF_RZ) inflow_anion<W, E>(B, P, cell, false, ca, cma, mk, base);
    } else {
        if (P.stages & RH_SAS_INF_RZ) {
            inflow<W, E>(B, P, cell, 0, ca, cma, mk, base);
            inflow<W, E>(B, P, cell, 1, ca, cma, mk, base);
        }
    }
    SAS_PH(B, P, 1);
#pragma nounroll
    for (int f = 0; f < 5; ++f) {
        if (f == 3) {   // the subsoil becomes the active compartment
            if constexpr (E >= 8) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    double *s0 = &B.park[(0 * E + j) * (W * 64) + B.tid], *s1 = &B.park[(1 * E + j) * (W * 64) + B.tid];
                    const double t0 = *s0, t1 = *s1;
                    *s0 = ca[j];
                    *s1 = cma[j];
                    ca[j] = t0;
                    cma[j] = t1;
                }
            } else {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const double t0 = ca[j], t1 = cma[j];
                    ca[j] = co[j];
                    cma[j] = cmo[j];
                    co[j] = t0;
                    cmo[j] = t1;
                }
            }
            if (P.stages & RH_SAS_INF_SS) {
                if constexpr (ANION) inflow_anion<W, E>(B, P, cell, true, ca, cma, mk, base);
                else inflow<W, E>(B, P, cell, 2, ca, cma, mk, base);
            }
        }
        // RH_SAS_EVAP, _TRANSP, _Q_RZ = 1 << (1 + f); RH_SAS_Q_SS, _CPR = 1 << (2 + f)
        if (!(P.stages & (f < 3 ? (2 << f) : (4 << f)))) continue;
        const bool sink = (f == 2) || (f == 4), keep = (f == 1) || (f == 3);
        if constexpr (ANION) {
            // soil evaporation takes water only -- but the virtual tracer leaves with it at alpha = 1
            const bool water = (f == 0) && (P.tracer != RH_SAS_TRACER_VIRTUAL);
            double alpha = (f == 0) ? 1.0 : alpha_q;
            if (f == 1) {
                // crop solute uptake stops if the root zone holds more than 80 % of saturation: evapotranspiration.py:932-939
                const int lu = ((const int *)P.a[SA_lu_id])[cell];
                double S[1] = {0.0};
#pragma unroll
                for (int j = 0; j < E; ++j) S[0] += ca[j];
                blk_sum<W, 1>(B, S);
                const bool stop = (lu > 500) && (lu < 599) && (S[0] >= 0.8 * ((const double *)P.a[SA_S_sat_rz])[cell]);
                alpha = (stop ? 0 : ((const double *)P.a[SA_alpha_transp])[cell]) * mk;
            }
            outflux_anion<W, E>(B, P, cell, f, sink, keep, water, alpha, ca, cma, co, cmo, mk, base);
        } else {
            outflux<W, E>(B, P, cell, f, sink, keep, ca, cma, co, cmo, mk, base);
        }
    }
    unpark2<W, E>(B, co, cmo);
    double (&sa_rz)[E] = co, (&msa_rz)[E] = cmo, (&sa_ss)[E] = ca, (&msa_ss)[E] = cma;   // (after the change of places at f == 3)

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
        SAS_PH(B, P, 12);
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
            park2<W, E>(B, msa_rz, msa_ss);   // (the statistics read the StorAges only)
            residence_stats<W, E>(B, P, cell, base, sa_rz, mk, SA_rt10_rz, true);
            residence_stats<W, E>(B, P, cell, base, sa_ss, mk, SA_rt10_ss, true);
            residence_stats<W, E>(B, P, cell, base, sa_s, mk, SA_rt10_s, false);
            unpark2<W, E>(B, msa_rz, msa_ss);
        }
        SAS_PH(B, P, 13);
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

    {
        // The classes above ages - 1 are padding of the register layout: no flux can fill them, and blk_cumsum takes the top of the stored
        // water from the last partial sum of the thread that owns class ages - 1 BECAUSE they hold exact zeros.  Checked once per column
        // and day behind the fluxes (the ageing below shifts the oldest class into the padding, which is never stored); rh_sas_sync
        // reports a violation.
        bool padded_water = false;
#pragma unroll
        for (int j = 0; j < E; ++j) padded_water |= (base + j >= A) && ((sa_rz[j] != 0.0) || (sa_ss[j] != 0.0));
        if (padded_water) atomicOr(P.unsupported, 2);
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
    SAS_PH(B, P, 14);
}


#ifdef RH_SAS_DET_ANION   // this unit's kernels: 0 = isotopes (oxygen-18, deuterium), 1 = anions (bromide, chloride, virtual tracer)
// the whole day (or the stages in args.stages) in one launch; returns RH_ERR_ARG if the age axis fits no shape
int RH_SAS_DET_NAME(hipStream_t stream, const SasArgs &args, unsigned n_cells, int nages, bool e4) {
    constexpr bool AN = RH_SAS_DET_ANION != 0;
#define RH_L4(W, E) hipLaunchKernelGGL((k_sas<W, E, AN>), dim3(n_cells), dim3(W * 64), 0, stream, args)
#define RH_L8(W) hipLaunchKernelGGL((k_sas8<W, AN>), dim3(n_cells), dim3(W * 64), 0, stream, args)
#ifdef RH_SAS_EXPERIMENT_E16
    if (nages > 1024) return RH_ERR_ARG;
    hipLaunchKernelGGL((k_sas16<AN>), dim3(n_cells), dim3(64), 0, stream, args);
    return RH_OK;
#else
    // eight classes per thread from 257 age classes on (9.63 against 10.17 ms per day at 10^5 columns x 1000 ages); RH_SAS_E4=1: the
    // four-class shapes for comparison
    if (!e4 && nages > 256 && nages <= 4096) {
        if (nages <= 512) RH_L8(1);
        else if (nages <= 1024) RH_L8(2);
        else if (nages <= 2048) RH_L8(4);
        else RH_L8(8);
    } else if (nages <= 64) RH_L4(1, 1);
    else if (nages <= 128) RH_L4(1, 2);
    else if (nages <= 256) RH_L4(1, 4);
    else if (nages <= 512) RH_L4(2, 4);
    else if (nages <= 1024) RH_L4(4, 4);
    else if (nages <= 2048) RH_L4(8, 4);
    else RH_L4(16, 4);
    return RH_OK;
#endif
#undef RH_L4
#undef RH_L8
}
#endif
