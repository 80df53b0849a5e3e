// rh_sas_solvers.hip -- the explicit solvers of the SAS / oxygen-18 transport step (settings.sas_solver = "Euler") for gfx950.
//
// The reference's explicit Euler scheme (svat_transport_model_euler, roger/core/transport.py:2064-2414, driven by
// calculate_storage_selection :3220-3262) splits the day into `substeps` sub-steps of length h = 1 / substeps.  In each of them
//   1. the infiltration of the sub-step joins age class 0 of root zone and subsoil (:2071-2145),
//   2. the travel time distributions of ALL five outgoing fluxes are evaluated on the StorAge as it stands (calc_TT_num :860-907,
//      calc_TT_num_nonneg :911-945: differences of Omega times the flux, limited to the water of the class, normalised to 1),
//   3. both StorAges take their net change in one update, the water that moves between them (percolation, capillary rise) mixing
//      its isotope signal into the receiving class (:2266-2310),
//   4. root_zone / subsoil / soil storages are formed (core/root_zone.py:189-217, subsoil.py:159-188, soil.py:1036-1090: classes
//      below 1e-8 mm are emptied).
// Concentrations of the fluxes, the age statistics and the diagnostics arrays are those of the LAST sub-step (each sub-step overwrites
// them in the reference); the day ends with the ageing.
//
// Same layout as the deterministic kernel (rh_sas.hip): one workgroup per column, thread t owns E consecutive age classes of the four
// state vectors in registers for the whole day -- one read and one write of the state per column and day.  The five fluxes run
// through ONE call site of the SAS function (a loop over the fluxes, uniform branches) to keep the code small.
#include "rh_sas_dev.h"

// The f-th of five consecutive arrays of the kernel argument.  Indexing `P.a[first + f]` with a run-time f would make the compiler copy
// the whole argument block into scratch memory; a chain of selects over constant indices stays in scalar registers.
SAS_DEV void *arr5(const SasArgs &P, int first, int f) {
    void *r = P.a[first];
    r = (f == 1) ? P.a[first + 1] : r;
    r = (f == 2) ? P.a[first + 2] : r;
    r = (f == 3) ? P.a[first + 3] : r;
    r = (f == 4) ? P.a[first + 4] : r;
    return r;
}

// Mixing of an addition (dsa1 of water carrying dmsa1) into an age class, transport.py:2122-2137, 2276-2291.  The root zone's formula
// keeps the old signal only where it is positive (`& (msa > 0)`), the subsoil's does not: both as the reference has them.
template <bool NEED_POS>
SAS_DEV double euler_mix(double msa, double sa, double dsa1, double dmsa1) {
    const double tot = dsa1 + sa;
    const UDiv by_tot = udiv_prepare(tot);
    const double a = ((tot > 0) && (!NEED_POS || (msa > 0))) ? msa * udiv(sa, by_tot) : 0;
    const double b = (tot > 0) ? dmsa1 * udiv(dsa1, by_tot) : 0;
    const double m = a + b;
    return ((dsa1 > 0) && (m <= 0)) ? dmsa1 : m;
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

// tt of one flux on the cumulative StorAge of its source: calc_TT_num + calc_TT_num_nonneg + the clipped differences (:2187-2199).
//   SA_hi: cumulative StorAge (masked) at the upper edges of the thread's classes; sa: the StorAge itself (masked)
template <int W, int E>
SAS_DEV void euler_tt(Blk<W> &B, const SasArgs &P, const double *p, double flux_h, const double (&SA_hi)[E], const double (&sa)[E],
                      double Smax, double mk, int base, double (&tt)[E]) {
    const int A = P.ages;
    if (!(flux_h > 0)) {  // TTq = where(flux <= 0, 0, .) :893-896 -> every difference 0 -> nothing is selected
#pragma unroll
        for (int j = 0; j < E; ++j) tt[j] = 0.0;
        return;
    }
    const double pr[7] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6]};
    const double code = pr[0];
    const PowConsts C = load_pow_consts(B.logt);   // (here, not at the top of the kernel: the coefficients live in scalar registers)
    double Om[E], Om_edge0 = 0.0;
    if (code == 6 || code == 61 || code == 62) sas_omega<W, E, FAM_POWER>(B, C, pr, SA_hi, Smax, mk, base, A, Om, Om_edge0);
    else if (code == 1) sas_omega<W, E, FAM_UNIFORM>(B, C, pr, SA_hi, Smax, mk, base, A, Om, Om_edge0);
    else if (code == 2) sas_omega<W, E, FAM_DIRAC>(B, C, pr, SA_hi, Smax, mk, base, A, Om, Om_edge0);
    else if (code == 3 || (code >= 31 && code <= 37) || code == 51 || code == 4) {
        const int fam = (code == 51) ? FAM_EXPONENTIAL : ((code == 4) ? FAM_GAMMA : FAM_KUMARASWAMI);
        double x[E], o[E], q[7];   // copies of their own: what crosses the call lives in scratch memory
#pragma unroll
        for (int j = 0; j < E; ++j) x[j] = SA_hi[j];
#pragma unroll
        for (int i = 0; i < 7; ++i) q[i] = pr[i];
        omega_library_families<E>(fam, q, x, Smax, mk, base, A, o);
#pragma unroll
        for (int j = 0; j < E; ++j) Om[j] = o[j];
    } else {
        // 52 (the exponential with reversed age order, sas.py:186-190) selects nothing: its Omega decreases along the age axis, every
        // difference is negative and `where(ttq_nonneg > 0, ., 0)` (:931-934) leaves 0.  Any other code is none of the reference's families.
        if (code != 52 && B.tid == 0) *P.unsupported = 1;
#pragma unroll
        for (int j = 0; j < E; ++j) tt[j] = 0.0;
        return;
    }
    double Om_lo, unused;
    blk_prev2<W>(B, Om[E - 1], 0.0, Om_edge0, 0.0, Om_lo, unused);
    double nn[E], s[1] = {0.0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        // :920-923  diff(SA).  The reference's sequential cumsum returns the class itself up to one rounding of SA, and exactly 0 for an
        // empty class; the differences of the block scan are only that consistent inside a thread (one ulp of either sign across
        // lanes, which the limiter below would turn into a selected 1e-14 mm of an EMPTY class): the class itself is used.
        const double sa_d = sa[j];
        const double ttq = (Om[j] - (j == 0 ? Om_lo : Om[j > 0 ? j - 1 : 0])) * flux_h;     // :924-927
        const double v = (sa_d + ttq < 0) ? -sa_d : ttq;                                     // :928-930
        nn[j] = (base + j < A) ? v : 0.0;
        s[0] += nn[j];
    }
    blk_sum<W, 1>(B, s);
    const UDiv by_s = udiv_prepare(s[0]);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const double t = (nn[j] > 0) ? udiv(nn[j], by_s) : 0.0;   // :931-934; TT = cumsum(.), tt = where(diff(TT) >= 0, diff(TT), 0): the round
        tt[j] = (t >= 0) ? t : 0.0;                                // trip through the cumulative sum is the identity up to ~1e-16 absolute
    }
}

// root_zone / subsoil / soil storages of the isotope model and, with `outputs`, their concentrations, the residence time statistics and
// the diagnostics arrays (core/root_zone.py:189-217, subsoil.py:159-188, soil.py:1036-1090; transport.py:155-312)
template <int W, int E>
SAS_DEV void storages_iso(Blk<W> &B, const SasArgs &P, int64_t cell, int base, double (&sa_rz)[E], const double (&msa_rz)[E], double (&sa_ss)[E],
                          const double (&msa_ss)[E], double mk, bool outputs) {
    const int A = P.ages;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        sa_rz[j] = (sa_rz[j] < 1e-8 ? 0 : sa_rz[j]);
        sa_ss[j] = (sa_ss[j] < 1e-8 ? 0 : sa_ss[j]);
    }
    if (!outputs) return;
    double sa_s[E], msa_s[E];
    double s[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        sa_s[j] = sa_rz[j] + sa_ss[j] * mk;
        const double tot = sa_rz[j] + sa_ss[j];
        const UDiv by_tot = udiv_prepare(tot);
        const double v = (tot > 0 ? msa_rz[j] * udiv(sa_rz[j], by_tot) + msa_ss[j] * udiv(sa_ss[j], by_tot) : 0);
        msa_s[j] = (v != v) ? 0 : v;
        s[0] += msa_rz[j] * sa_rz[j];
        s[1] += sa_rz[j];
        s[2] += msa_ss[j] * sa_ss[j];
        s[3] += sa_ss[j];
        s[4] += msa_s[j] * sa_s[j];
        s[5] += sa_s[j];
    }
    blk_sum<W, 6>(B, s);
    if (B.tid == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // calc_conc_iso_storage :538-562
            const double Cs = (s[2 * k + 1] > 0 ? s[2 * k] / s[2 * k + 1] : 0) * mk;
            ((double *)P.a[SA_C_rz + k])[cell] = Cs;
            ((double *)P.a[SA_C_iso_rz + k])[cell] = conc_to_delta(P, Cs) * mk;
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
    if (P.stats) {  // the reference never assigns rt10 / rt90 of root zone and subsoil (:181-196, :232-247)
        residence_stats<W, E>(B, P, cell, base, sa_rz, mk, SA_rt10_rz, true);
        residence_stats<W, E>(B, P, cell, base, sa_ss, mk, SA_rt10_ss, true);
        residence_stats<W, E>(B, P, cell, base, sa_s, mk, SA_rt10_s, false);
    }
}

// One sub-step of length h.  LAST: the last of the day, which also forms everything the reference overwrites in every sub-step --
// concentrations of the fluxes, distributions, statistics.  An instantiation of its own rather than a flag: inside the loop over the
// sub-steps the compiler hoists the addresses of all those output arrays out of the loop and spills them (measured: 85 VGPRs, with
// their reloads in every sub-step).
template <int W, int E, bool LAST>
SAS_DEV void euler_substep(Blk<W> &B, const SasArgs &P, int64_t cell, int base, double h, double mk, double im, double ip, double is, double C_in,
                           double (&sa_rz)[E], double (&msa_rz)[E], double (&sa_ss)[E], double (&msa_ss)[E]) {
    const int A = P.ages;
    // 1. upper boundary condition :2071-2145.  tt_inf is 1 in age class 0 and 0 elsewhere: for the other classes the mixing reduces
    //    to msa * (sa / sa) where the class holds water (and, in the root zone, a positive signal), 0 otherwise
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (base + j == 0) {
            const double t0 = (im > 0 ? 1 : 0) * mk, t1 = (ip > 0 ? 1 : 0) * mk, t2 = (is > 0 ? 1 : 0) * mk;
            const double m0 = (im > 0 ? C_in : 0) * mk, m1 = (ip > 0 ? C_in : 0) * mk, m2 = (is > 0 ? C_in : 0) * mk;
            const double dsa_rz = (im * t0 + ip * t1) * h, dsa_ss = (is * t2) * h;
            const double dmsa_rz1 = ((m0 != m0) ? 0 : m0) * (dsa_rz > 0 ? ((im * t0 * h) / dsa_rz) : 0) +
                                    ((m1 != m1) ? 0 : m1) * (dsa_rz > 0 ? ((ip * t1 * h) / dsa_rz) : 0);
            const double dmsa_ss1 = ((m2 != m2) ? 0 : m2) * (dsa_ss > 0 ? ((is * t2 * h) / dsa_ss) : 0);
            msa_rz[j] = euler_mix<true>(msa_rz[j], sa_rz[j], dsa_rz, dmsa_rz1);
            msa_ss[j] = euler_mix<false>(msa_ss[j], sa_ss[j], dsa_ss, dmsa_ss1);
            sa_rz[j] += dsa_rz;
            sa_ss[j] += dsa_ss;
            msa_rz[j] = (sa_rz[j] <= 0) ? 0 : msa_rz[j];
            msa_ss[j] = (sa_ss[j] <= 0) ? 0 : msa_ss[j];
        } else {
            msa_rz[j] = ((sa_rz[j] > 0) && (msa_rz[j] > 0)) ? msa_rz[j] : 0;
            msa_ss[j] = (sa_ss[j] > 0) ? msa_ss[j] : 0;
        }
    }
    if (LAST && B.tid == 0) {   // :2324-2335, delta_fluxes_svat :3660-3697
        const double inf[3] = {im, ip, is};
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const double Ci = (inf[w] > 0 ? C_in : NAN) * mk;
            ((double *)P.a[SA_C_inf_mat_rz + w])[cell] = Ci;
            ((double *)P.a[SA_C_iso_inf_mat_rz + w])[cell] = conc_to_delta(P, Ci) * mk;
        }
    }
    // 2. + 3. the five fluxes (evap_soil, transp, q_rz from the root zone; q_ss, cpr_rz from the subsoil): flux * tt per class on
    //    SA = calc_SA(sa) * maskCatch of the source (:2147-2155), formed when the loop reaches the storage's first flux (neither
    //    StorAge changes before the update below).  The equality of the cumulative sum above the top of the stored water is restored
    //    exactly (blk_cumsum): the differences of Omega there are exact zeros, as with the reference's sequential cumsum
    double e_ev[E], e_tr[E], e_qrz[E], e_qss[E], e_cpr[E];
    double SA[E], sa_src[E], S_top = 0.0;
#pragma unroll 1
    for (int f = 0; f < 5; ++f) {
        const bool from_ss = f >= 3;
        if (f == 0 || f == 3) {
            double lo_unused;
#pragma unroll
            for (int j = 0; j < E; ++j) sa_src[j] = from_ss ? sa_ss[j] : sa_rz[j];
            blk_cumsum<W, E, true>(B, sa_src, SA, lo_unused, &S_top, base, A - 1);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                SA[j] *= mk;
                sa_src[j] *= mk;
            }
            S_top *= mk;
        }
        const double flux = ((const double *)arr5(P, SA_evap_soil, f))[P.day_off + cell];
        const double *p = (const double *)arr5(P, SA_sas_params_evap_soil, f) + cell * 8;
        double tt[E];
        euler_tt<W, E>(B, P, p, flux * h, SA, sa_src, S_top, mk, base, tt);
        if constexpr (LAST) {   // concentrations, distributions and their statistics are those of the last sub-step
            double mtt[E], s[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < E; ++j) {
                mtt[j] = (tt[j] > 0 ? (from_ss ? msa_ss[j] : msa_rz[j]) : 0);   // calc_mtt :565-596, isotopes
                s[0] += mtt[j] * tt[j];
                s[1] += tt[j];
            }
            blk_sum<W, 2>(B, s);
            if (B.tid == 0) {   // calc_conc_iso_flux :512-535
                double conc = (s[1] > 0 ? s[0] / s[1] : NAN);
                conc = (conc != 0 ? conc : NAN);
                const double Cf = conc * mk;
                ((double *)arr5(P, SA_C_evap_soil, f))[cell] = Cf;
                ((double *)arr5(P, SA_C_iso_evap_soil, f))[cell] = conc_to_delta(P, Cf) * mk;
            }
            const bool want_stats = P.stats && (f == 1 || f == 3);
            if (P.diag || want_stats) {
                double TT_hi[E], TT_lo;
                blk_cumsum<W, E, false>(B, tt, TT_hi, TT_lo, nullptr, base, 0);
                if (want_stats) age_stats<W, E>(B, P, cell, base, TT_hi, TT_lo, tt, f == 1 ? SA_tt10_transp : SA_tt10_q_ss, false);
                if (P.diag) {
                    double *o_tt = (double *)arr5(P, SA_tt_evap_soil, f) + cell * A;
                    double *o_mtt = (double *)arr5(P, SA_mtt_evap_soil, f) + cell * A;
                    double *o_TT = (double *)arr5(P, SA_TT_evap_soil, f) + cell * (A + 1);
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
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double e = flux * tt[j];
            if (f == 0) e_ev[j] = e;
            else if (f == 1) e_tr[j] = e;
            else if (f == 2) e_qrz[j] = e;
            else if (f == 3) e_qss[j] = e;
            else e_cpr[j] = e;
        }
    }

    // 4. update of both StorAges :2266-2310
#pragma unroll
    for (int j = 0; j < E; ++j) {
        double dsa_rz = (e_cpr[j] - e_ev[j] - e_tr[j] - e_qrz[j]) * h;
        dsa_rz = (sa_rz[j] + dsa_rz < 0) ? -sa_rz[j] : dsa_rz;
        double dsa_ss = (e_qrz[j] - e_cpr[j] - e_qss[j]) * h;
        dsa_ss = (sa_ss[j] + dsa_ss < 0) ? -sa_ss[j] : dsa_ss;
        // the water that changes storage carries the signal of its class: mtt = msa of the source where tt > 0, and the weight
        // (flux * tt * h) / dsa1 of the single contribution is x / x = 1
        const double dsa_rz1 = e_cpr[j] * h;
        const double dmsa_rz1 = (dsa_rz1 > 0) ? ((msa_ss[j] != msa_ss[j]) ? 0 : msa_ss[j]) : 0;
        const double dsa_ss1 = e_qrz[j] * h;
        const double dmsa_ss1 = (dsa_ss1 > 0) ? ((msa_rz[j] != msa_rz[j]) ? 0 : msa_rz[j]) : 0;
        const double n_rz = euler_mix<true>(msa_rz[j], sa_rz[j], dsa_rz1, dmsa_rz1);
        const double n_ss = euler_mix<false>(msa_ss[j], sa_ss[j], dsa_ss1, dmsa_ss1);
        sa_rz[j] += dsa_rz;
        sa_ss[j] += dsa_ss;
        msa_rz[j] = (sa_rz[j] <= 0) ? 0 : n_rz;
        msa_ss[j] = (sa_ss[j] <= 0) ? 0 : n_ss;
    }
    // 5. storages (and, after the last sub-step, everything that is derived from them)
    storages_iso<W, E>(B, P, cell, base, sa_rz, msa_rz, sa_ss, msa_ss, mk, LAST);
}

// register budget: waves per SIMD the kernel is compiled for.  Measured at 10^5 columns x 1000 ages x 6 sub-steps: 2 waves (256
// registers, nothing spilled) 19.0 ms per day, 3 waves 20.7 ms, 4 waves 25.3 ms -- the reloads cost more than the occupancy brings
#ifndef RH_EULER_WAVES
#define RH_EULER_WAVES 2
#endif
template <int W, int E>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(RH_EULER_WAVES, RH_EULER_WAVES))) void k_sas_euler(const SasArgs P) {
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
    const double h = 1 / (double)P.substeps;   // settings.h (benchmarks/SVATOXYGEN18_benchmark.py:30-31)
    const double im = ((const double *)P.a[SA_inf_mat_rz])[P.day_off + cell], ip = ((const double *)P.a[SA_inf_pf_rz])[P.day_off + cell];
    const double is = ((const double *)P.a[SA_inf_pf_ss])[P.day_off + cell], C_in = ((const double *)P.a[SA_C_in])[P.day_off + cell];

    for (int it = 0; it + 1 < P.substeps; ++it) euler_substep<W, E, false>(B, P, cell, base, h, mk, im, ip, is, C_in, sa_rz, msa_rz, sa_ss, msa_ss);
    euler_substep<W, E, true>(B, P, cell, base, h, mk, im, ip, is, C_in, sa_rz, msa_rz, sa_ss, msa_ss);

    ageing<W, E>(B, A, base, sa_rz, msa_rz);
    ageing<W, E>(B, A, base, sa_ss, msa_ss);
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

template <int W, int E>
static void launch_euler(hipStream_t stream, const SasArgs &args) {
    hipLaunchKernelGGL((k_sas_euler<W, E>), dim3((unsigned)args.n), dim3(W * 64), 0, stream, args);
}

// The whole day of an explicit solver in one launch; the smallest workgroup whose blocked layout covers the age classes.
int rh_sas_launch_solver(int solver, hipStream_t stream, const SasArgs &args) {
    if (solver != RH_SAS_SOLVER_EULER) return RH_ERR_ARG;
    const int nages = args.ages + 1;
    if (nages <= 64) launch_euler<1, 1>(stream, args);
    else if (nages <= 256) launch_euler<1, 4>(stream, args);
    else if (nages <= 512) launch_euler<2, 4>(stream, args);
    else if (nages <= 1024) launch_euler<4, 4>(stream, args);
    else if (nages <= 2048) launch_euler<8, 4>(stream, args);
    else launch_euler<8, 8>(stream, args);   // (sixteen waves would leave 128 registers per thread: 700 spilled)
    return RH_OK;
}
