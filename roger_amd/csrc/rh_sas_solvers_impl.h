// rh_sas_solvers_impl.h -- the explicit solvers of the SAS / oxygen-18 transport step (settings.sas_solver = "Euler", "RK4") for gfx950.
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
// Same layout as the deterministic kernel (rh_sas_kernels.h): one workgroup per column, thread t owns E consecutive age classes of the four
// state vectors in registers for the whole day -- one read and one write of the state per column and day.  The five fluxes run
// through ONE call site of the SAS function (a loop over the fluxes, uniform branches) to keep the code small.
#pragma once
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

// tt of one flux on the cumulative StorAge of its source: calc_TT_num + calc_TT_num_nonneg + the clipped differences (:2187-2199).
//   SA_hi: cumulative StorAge (masked) at the upper edges of the thread's classes; sa: its differences, diff(SA) (one_flux)
template <int W, int E>
SAS_DEV void euler_tt(Blk<W> &B, const SasArgs &P, const double *p, double flux_h, const double (&SA_hi)[E], const double (&sa)[E],
                      double Smax, double mk, int base, bool nonneg, double (&tt)[E]) {
    const int A = P.ages;
    const bool no_flux = !(flux_h > 0);   // TTq = where(flux <= 0, 0, .) :893-896: every difference of Omega is 0
    if (no_flux && nonneg) {              // ... and with non-negative classes nothing is selected
#pragma unroll
        for (int j = 0; j < E; ++j) tt[j] = 0.0;
        return;
    }
    const double pr[7] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6]};
    const double code = pr[0];
    const PowConsts C = load_pow_consts(B.logt);   // (here, not at the top of the kernel: the coefficients live in scalar registers)
    double Om[E], Om_edge0 = 0.0;
    if (no_flux) {
        // a trial StorAge of RK4 with a negative class: the limiter below selects `-sa` of it although no water leaves (as the reference does)
#pragma unroll
        for (int j = 0; j < E; ++j) Om[j] = 0.0;
    } else if (code == 6 || code == 61 || code == 62) sas_omega<W, E, FAM_POWER>(B, C, pr, SA_hi, Smax, mk, base, A, Om, Om_edge0);
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
        // difference is negative and `where(ttq_nonneg > 0, ., 0)` (:931-934) leaves 0 -- evaluated as Omega = 0.  Any other code is none
        // of the reference's families.
        if (code != 52 && B.tid == 0) atomicOr(P.unsupported, 1);
#pragma unroll
        for (int j = 0; j < E; ++j) Om[j] = 0.0;
    }
    double Om_lo, unused;
    blk_prev2<W>(B, Om[E - 1], 0.0, Om_edge0, 0.0, Om_lo, unused);
    double nn[E], s[1] = {0.0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const double sa_d = sa[j];                                                          // :920-923  diff(SA), see one_flux
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

// ... of the anion kernels (core/root_zone.py:221-258, subsoil.py:186-223, soil.py:1094-1142): classes without water drop their solute,
// M = nansum(msa), C = M / sum(sa)
template <int W, int E>
SAS_DEV void storages_anion(Blk<W> &B, const SasArgs &P, int64_t cell, int base, double (&sa_rz)[E], double (&msa_rz)[E], double (&sa_ss)[E],
                            double (&msa_ss)[E], double mk, bool outputs) {
    const int A = P.ages;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        sa_rz[j] = (sa_rz[j] < 1e-8 ? 0 : sa_rz[j]);
        sa_ss[j] = (sa_ss[j] < 1e-8 ? 0 : sa_ss[j]);
        msa_rz[j] = (sa_rz[j] <= 0 ? 0 : msa_rz[j]);
        msa_ss[j] = (sa_ss[j] <= 0 ? 0 : msa_ss[j]);
    }
    if (!outputs) return;
    double sa_s[E], msa_s[E];
    double s[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        sa_s[j] = sa_rz[j] + sa_ss[j] * mk;
        msa_s[j] = msa_rz[j] + msa_ss[j] * mk;
        s[0] += (msa_rz[j] != msa_rz[j]) ? 0 : msa_rz[j];
        s[1] += sa_rz[j];
        s[2] += (msa_ss[j] != msa_ss[j]) ? 0 : msa_ss[j];
        s[3] += sa_ss[j];
        s[4] += (msa_s[j] != msa_s[j]) ? 0 : msa_s[j];
        s[5] += sa_s[j];
    }
    blk_sum<W, 6>(B, s);
    if (B.tid == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double M = s[2 * k] * mk;
            ((double *)P.a[SA_M_rz + k])[cell] = M;
            ((double *)P.a[SA_C_rz + k])[cell] = (s[2 * k + 1] > 0 ? M / s[2 * k + 1] : 0);
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
    if (P.stats) {
        residence_stats<W, E>(B, P, cell, base, sa_rz, mk, SA_rt10_rz, true);
        residence_stats<W, E>(B, P, cell, base, sa_ss, mk, SA_rt10_ss, true);
        residence_stats<W, E>(B, P, cell, base, sa_s, mk, SA_rt10_s, false);
    }
}

// What the reference overwrites in every sub-step and the day keeps from the last one, for one flux: its concentration (:2336-2357,
// calc_conc_iso_flux :512-535, delta_fluxes_svat :3660-3697), the age statistics of transpiration and percolation, the diagnostics arrays
//   ANION: the solute the flux takes along (anion_mtt) comes in as mtt_in, its concentration is sum(mtt) / (flux * h) (:2391-2406); the soil
//   evaporation's solute and concentration are never assigned
template <int W, int E, bool ANION>
SAS_DEV void flux_outputs(Blk<W> &B, const SasArgs &P, int64_t cell, int base, int f, const double (&tt)[E], const double (&msa_rz)[E],
                          const double (&msa_ss)[E], double mk, const double (&mtt_in)[E], double flux_h) {
    const int A = P.ages;
    const bool from_ss = f >= 3;
    double mtt[E], s[2] = {0.0, 0.0};
    if constexpr (ANION) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            mtt[j] = mtt_in[j];
            s[0] += mtt[j];
        }
        if (f > 0) {
            blk_sum<W, 2>(B, s);
            if (B.tid == 0) ((double *)arr5(P, SA_C_evap_soil, f))[cell] = (flux_h > 0 ? s[0] / flux_h : 0) * mk;
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            mtt[j] = (tt[j] > 0 ? (from_ss ? msa_ss[j] : msa_rz[j]) : 0);   // calc_mtt :565-596, isotopes
            s[0] += mtt[j] * tt[j];
            s[1] += tt[j];
        }
        blk_sum<W, 2>(B, s);
        if (B.tid == 0) {
            double conc = (s[1] > 0 ? s[0] / s[1] : NAN);
            conc = (conc != 0 ? conc : NAN);
            const double Cf = conc * mk;
            ((double *)arr5(P, SA_C_evap_soil, f))[cell] = Cf;
            ((double *)arr5(P, SA_C_iso_evap_soil, f))[cell] = conc_to_delta(P, Cf) * mk;
        }
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
                    if (!ANION || f > 0) o_mtt[base + j] = mtt[j];
                    o_TT[base + j + 1] = TT_hi[j];
                }
        }
    }
}

// The five per-flux arrays of a sub-step: separate local arrays, declared by FIVE(name).  The flux index is a run-time value (one call
// site of the SAS function for all five); the arrays are addressed through chains of selects over constant indices so that they stay
// in registers (as members of one struct the compiler turns the chain back into an indexed access to scratch memory).
#define FIVE(name) double name##_ev[E], name##_tr[E], name##_qrz[E], name##_qss[E], name##_cpr[E]
#define FIVE_ARGS(name) name##_ev, name##_tr, name##_qrz, name##_qss, name##_cpr
#define SET5(name, f, j, v)              \
    do {                                 \
        const double v_ = (v);           \
        if ((f) == 0) name##_ev[j] = v_;       \
        else if ((f) == 1) name##_tr[j] = v_;  \
        else if ((f) == 2) name##_qrz[j] = v_; \
        else if ((f) == 3) name##_qss[j] = v_; \
        else name##_cpr[j] = v_;               \
    } while (0)
template <int E>
SAS_DEV double get5(int f, int j, const double (&ev)[E], const double (&tr)[E], const double (&qrz)[E], const double (&qss)[E],
                    const double (&cpr)[E]) {
    double v = ev[j];
    v = (f == 1) ? tr[j] : v;
    v = (f == 2) ? qrz[j] : v;
    v = (f == 3) ? qss[j] : v;
    v = (f == 4) ? cpr[j] : v;
    return v;
}

// calc_mtt, anion branch (:583-596): the solute flux f takes from the classes of its source, msa / sa * alpha * tt * (flux * h) clipped to
// [0, msa]; the soil evaporation (f = 0) takes none
template <int E>
SAS_DEV void anion_mtt(const SasArgs &P, int64_t cell, int f, double flux_h, const double (&tt)[E], const double (&sa_rz)[E], const double (&msa_rz)[E],
                       const double (&sa_ss)[E], const double (&msa_ss)[E], double (&mtt)[E]) {
    const bool from_ss = f >= 3;
    const double *pa = (f == 1) ? (const double *)P.a[SA_alpha_transp] : (const double *)P.a[SA_alpha_q];   // (constant indices: arr5)
    const double alpha = pa[cell];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const double sa = from_ss ? sa_ss[j] : sa_rz[j], msa = from_ss ? msa_ss[j] : msa_rz[j];
        double m = (sa > 0 ? msa / sa : 0) * alpha * tt[j] * flux_h;
        m = (m <= 0 ? 0 : m);
        m = (m > msa ? msa : m);
        mtt[j] = (f == 0) ? 0.0 : m;
    }
}

// tt of flux f (0..4: evap_soil, transp, q_rz from the root zone; q_ss, cpr_rz from the subsoil; called with f ascending) on the StorAges
// s_rz / s_ss, the flux scaled by `scale` [/ 2]: SA = cumsum(s) [* maskCatch] of the source is formed when the loop over the fluxes
// reaches the storage's first flux (neither StorAge changes inside that loop) and kept in SA / sa_src / S_top.  The equality of the
// cumulative sum above the top of the stored water is restored exactly (blk_cumsum): the differences of Omega there are exact zeros,
// as with the reference's sequential cumsum.  masked: the state itself (non-negative classes); otherwise a trial StorAge of RK4.
// Returns the flux of the day.
template <int W, int E>
SAS_DEV double one_flux(Blk<W> &B, const SasArgs &P, int64_t cell, int base, int f, double scale, bool halve, double mk, bool masked,
                        const double (&s_rz)[E], const double (&s_ss)[E], double (&SA)[E], double (&sa_src)[E], double &S_top, double (&tt)[E]) {
    const bool from_ss = f >= 3;
    if (f == 0 || f == 3) {
        const double m = masked ? mk : 1.0;
        double lo, raw[E];
#pragma unroll
        for (int j = 0; j < E; ++j) raw[j] = from_ss ? s_ss[j] : s_rz[j];
        blk_cumsum<W, E, true>(B, raw, SA, lo, &S_top, base, P.ages - 1);
        // diff(SA) of calc_TT_num_nonneg (:920-923) as the reference's sequential cumsum returns it: fl(SA[k] + sa[k]) - SA[k], which is the
        // class up to one rounding and exactly 0 for a class that the cumulative sum absorbs (empty, or a residue of 1e-17 mm of either
        // sign under 100 mm).  Formed on the thread's own lower edge: the plain differences of the block scan are only that consistent
        // inside a thread (one ulp of either sign across lanes, which the limiter would turn into a selected 1e-14 mm of an empty
        // class, or -- a negative residue in a trial StorAge of RK4 under a flux of 0 -- into the whole distribution).
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double edge = (j == 0) ? lo : SA[j > 0 ? j - 1 : 0];
            sa_src[j] = (edge + raw[j]) - edge;
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {
            SA[j] *= m;
            sa_src[j] *= m;
        }
        S_top *= m;
        if (!masked) {
            // a trial StorAge of the Runge-Kutta scheme may hold negative classes (its third stage limits the change by
            // `sarkn - dsarkn < 0`): the cumulative sum is then not monotone and max(SA) (over all nages edges, SA[0] = 0 included) is
            // not its top value
            double mx = 0.0;
#pragma unroll
            for (int j = 0; j < E; ++j) mx = fmax(mx, SA[j]);
            S_top = blk_max<W>(B, mx);
        }
    }
    const double flux = B.scal[SC_FLUX + f];          // (the column's scalars of the day wait in LDS: sas_load_scalars)
    const double *p = B.scal + SC_PAR + 7 * f;
    euler_tt<W, E>(B, P, p, halve ? flux * scale / 2 : flux * scale, SA, sa_src, S_top, mk, base, masked, tt);
    return flux;
}

// One sub-step of length h of the explicit Euler scheme (svat_transport_model_euler :2064-2414) or, RK4, of the Runge-Kutta scheme
// (svat_transport_model_rk4 :1139-2047).
// LAST: the last of the day, which also forms everything the reference overwrites in every sub-step -- concentrations of the fluxes,
// distributions, statistics.  An instantiation of its own rather than a flag: inside the loop over the sub-steps the compiler hoists
// the addresses of all those output arrays out of the loop and spills them (measured: 85 VGPRs, with their reloads in every sub-step).
//
// RK4 for the isotopes: the intermediate signals (msarkn, mttrkn) never reach the result -- the final mtt is calc_mtt on the state after
// the infiltration (:1878-1896) -- so only the trial StorAges are followed:
//   stage 1 on the state, fluxes * h;      the trial StorAge moves by (net outflow) * h       (:1388-1466, isotope branch)
//   stage 2 on that,      fluxes * h / 2;  it moves by (net outflow) * h / 2                  (:1552-1580)
//   stage 3 on that,      fluxes * h / 2;  it moves by (net outflow) * h / 2, limited by `sarkn - dsarkn < 0` as written (:1700-1728)
//   stage 4 on that,      fluxes * h;      tt = (tt1 + 2 tt2 + 2 tt3 + tt4) / 6 (:1835-1855), then Euler's update with it (:1898-1941)
//
// ANION (bromide, chloride, virtual tracer: msa is solute mass by age): the water side is the same up to the trial StorAges of RK4, which
// move by h / 2 after the first AND the second evaluation and not at all after the third (:1432-1446, 1581-1595, 1729-1751 -- the trial
// solute that changes there never reaches the result either); the infiltration adds C_in * h per infiltration flux to age class 0
// whatever infiltrates (:2100-2112, 2150-2170); the fluxes take msa / sa * alpha * tt * flux * h (calc_mtt's anion branch) and the update
// adds the differences unmixed, refused where they would turn a class negative (:2308-2332); RK4's root zone -- as written -- GAINS the
// soil evaporation and does not receive the capillary rise (:1941-1946).
template <int W, int E, bool LAST, bool RK4, bool ANION>
SAS_DEV void explicit_substep(Blk<W> &B, const SasArgs &P, int64_t cell, int base, double h, double mk, double im, double ip, double is,
                              double C_in, double (&sa_rz)[E], double (&msa_rz)[E], double (&sa_ss)[E], double (&msa_ss)[E]) {
    // 1. upper boundary condition :2071-2145 (:1146-1220).  tt_inf is 1 in age class 0 and 0 elsewhere: for the other classes the mixing
    //    reduces to msa * (sa / sa) where the class holds water (and, in the root zone, a positive signal), 0 otherwise
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if constexpr (ANION) {
            if (base + j == 0) {
                const double t0 = (im > 0 ? 1 : 0) * mk, t1 = (ip > 0 ? 1 : 0) * mk, t2 = (is > 0 ? 1 : 0) * mk;
                const double m = C_in * mk, mm = (m != m) ? 0 : m;
                sa_rz[j] += (im * t0 + ip * t1) * h;
                sa_ss[j] += (is * t2) * h;
                msa_rz[j] += mm * h + mm * h;
                msa_ss[j] += mm * h;
            }
            continue;
        }
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
    if (LAST && B.tid == 0) {   // :2324-2335, delta_fluxes_svat :3660-3697; anions :2379-2390
        const double inf[3] = {im, ip, is};
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            if constexpr (ANION) {
                ((double *)P.a[SA_C_inf_mat_rz + w])[cell] = (inf[w] * h > 0 ? C_in : 0) * mk;
            } else {
                const double Ci = (inf[w] > 0 ? C_in : NAN) * mk;
                ((double *)P.a[SA_C_inf_mat_rz + w])[cell] = Ci;
                ((double *)P.a[SA_C_iso_inf_mat_rz + w])[cell] = conc_to_delta(P, Ci) * mk;
            }
        }
    }
    // 2. + 3. flux * tt per class of the five fluxes [, the solute they take]
    FIVE(e);
    FIVE(m);   // (ANION only; m_ev stays unused)
    double SA[E], sa_src[E], S_top = 0.0;
    if constexpr (!RK4) {
#pragma unroll 1
        for (int f = 0; f < 5; ++f) {
            double tt[E], mtt[E];
            const double flux = one_flux<W, E>(B, P, cell, base, f, h, false, mk, true, sa_rz, sa_ss, SA, sa_src, S_top, tt);
            if constexpr (ANION) {
                anion_mtt<E>(P, cell, f, flux * h, tt, sa_rz, msa_rz, sa_ss, msa_ss, mtt);
#pragma unroll
                for (int j = 0; j < E; ++j) SET5(m, f, j, mtt[j]);
            }
            if constexpr (LAST) flux_outputs<W, E, ANION>(B, P, cell, base, f, tt, msa_rz, msa_ss, mk, mtt, flux * h);
#pragma unroll
            for (int j = 0; j < E; ++j) SET5(e, f, j, flux * tt[j]);
        }
    } else {
        double s_rz[E], s_ss[E];
        FIVE(acc);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            s_rz[j] = sa_rz[j];
            s_ss[j] = sa_ss[j];
        }
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
            const bool half = (stage == 1 || stage == 2);
            const double w = half ? 2.0 : 1.0;
#pragma unroll 1
            for (int f = 0; f < 5; ++f) {
                double tt[E];
                const double flux = one_flux<W, E>(B, P, cell, base, f, h, half, mk, stage == 0, s_rz, s_ss, SA, sa_src, S_top, tt);
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    SET5(acc, f, j, stage == 0 ? tt[j] : get5<E>(f, j, FIVE_ARGS(acc)) + w * tt[j]);   // (tt1 + 2 * tt2 + 2 * tt3 + tt4), left to right
                    SET5(e, f, j, flux * tt[j]);
                }
            }
            if (stage < (ANION ? 2 : 3)) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    double d_rz = (e_cpr[j] - e_ev[j] - e_tr[j] - e_qrz[j]) * h;
                    double d_ss = (e_qrz[j] - e_cpr[j] - e_qss[j]) * h;
                    if (stage > 0 || ANION) {
                        d_rz = d_rz / 2;
                        d_ss = d_ss / 2;
                    }
                    const double t_rz = (stage == 2) ? s_rz[j] - d_rz : s_rz[j] + d_rz;
                    const double t_ss = (stage == 2) ? s_ss[j] - d_ss : s_ss[j] + d_ss;
                    d_rz = (t_rz < 0) ? -s_rz[j] : d_rz;
                    d_ss = (t_ss < 0) ? -s_ss[j] : d_ss;
                    s_rz[j] += d_rz;
                    s_ss[j] += d_ss;
                }
            }
        }
#pragma unroll 1
        for (int f = 0; f < 5; ++f) {
            const double flux = B.scal[SC_FLUX + f];
            double tt[E], mtt[E];
#pragma unroll
            for (int j = 0; j < E; ++j) tt[j] = get5<E>(f, j, FIVE_ARGS(acc)) / 6.;
            if constexpr (ANION) {   // :1881-1896: on the state after the infiltration
                anion_mtt<E>(P, cell, f, flux * h, tt, sa_rz, msa_rz, sa_ss, msa_ss, mtt);
#pragma unroll
                for (int j = 0; j < E; ++j) SET5(m, f, j, mtt[j]);
            }
            if constexpr (LAST) flux_outputs<W, E, ANION>(B, P, cell, base, f, tt, msa_rz, msa_ss, mk, mtt, flux * h);
#pragma unroll
            for (int j = 0; j < E; ++j) SET5(e, f, j, flux * tt[j]);
        }
    }

    // 4. update of both StorAges :2266-2310 (:1898-1941); anions :2308-2332 (:1941-1966)
    if constexpr (ANION) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            double dsa_rz = RK4 ? (e_ev[j] - e_tr[j] - e_qrz[j]) * h : (e_cpr[j] - e_ev[j] - e_tr[j] - e_qrz[j]) * h;
            dsa_rz = (sa_rz[j] + dsa_rz < 0) ? -sa_rz[j] : dsa_rz;
            double dsa_ss = (e_qrz[j] - e_cpr[j] - e_qss[j]) * h;
            dsa_ss = (sa_ss[j] + dsa_ss < 0) ? -sa_ss[j] : dsa_ss;
#define RH_NZ(x) (((x) != (x)) ? 0 : (x))
            double dmsa_rz = RH_NZ(m_cpr[j]) - RH_NZ(m_tr[j]) - RH_NZ(m_qrz[j]);
            double dmsa_ss = RH_NZ(m_qrz[j]) - RH_NZ(m_cpr[j]) - RH_NZ(m_qss[j]);
#undef RH_NZ
            dmsa_rz = (msa_rz[j] + dmsa_rz < 0) ? 0 : dmsa_rz;
            dmsa_ss = (msa_ss[j] + dmsa_ss < 0) ? 0 : dmsa_ss;
            sa_rz[j] += dsa_rz;
            sa_ss[j] += dsa_ss;
            msa_rz[j] += dmsa_rz;
            msa_ss[j] += dmsa_ss;
        }
        storages_anion<W, E>(B, P, cell, base, sa_rz, msa_rz, sa_ss, msa_ss, mk, LAST);
        return;
    }
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
template <int W, int E, bool RK4, bool ANION>
__device__ __forceinline__ void explicit_body(const SasArgs &P);
template <int W, int E, bool ANION>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(RH_EULER_WAVES, RH_EULER_WAVES))) void k_sas_euler(const SasArgs P) {
    explicit_body<W, E, false, ANION>(P);
}
template <int W, int E, bool ANION>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(RH_EULER_WAVES, RH_EULER_WAVES))) void k_sas_rk4(const SasArgs P) {
    explicit_body<W, E, true, ANION>(P);
}
template <int W, int E, bool RK4, bool ANION>
__device__ __forceinline__ void explicit_body(const SasArgs &P) {
    __shared__ double s_red[2][W][8];
    __shared__ double s_xch[2][W][2];
    __shared__ double2 s_logt[64];
    __shared__ double s_scal[SC_COUNT];
    if (threadIdx.x < 64) s_logt[threadIdx.x] = SAS_LOG_T[threadIdx.x];
    Blk<W> B;
    B.logt = s_logt;
    B.tid = threadIdx.x;
    B.lane = threadIdx.x & 63;
    B.wave = threadIdx.x >> 6;
    B.phase = 0;
    B.red = s_red;
    B.xch = s_xch;
    B.park = nullptr;
    B.scal = s_scal;
    const int64_t cell = blockIdx.x;
    const int A = P.ages;
    const int base = B.tid * E;
    double scal_regs[SC_COUNT];        // the column's scalars and its state are requested together, the scalars go to LDS behind it
    sas_fetch_scalars(P, scal_regs);
    const double mk = ((const int *)P.a[SA_maskCatch])[cell] != 0 ? 1.0 : 0.0;   // a bool in the reference: 0 or 1 (sas_omega relies on it)

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
    sas_publish_scalars(scal_regs, s_scal);
    __syncthreads();
    const double h = 1 / (double)P.substeps;   // settings.h (benchmarks/SVATOXYGEN18_benchmark.py:30-31)
    const double im = B.scal[SC_INF + 0], ip = B.scal[SC_INF + 1], is = B.scal[SC_INF + 2], C_in = B.scal[SC_CIN];

    for (int it = 0; it + 1 < P.substeps; ++it)
        explicit_substep<W, E, false, RK4, ANION>(B, P, cell, base, h, mk, im, ip, is, C_in, sa_rz, msa_rz, sa_ss, msa_ss);
    explicit_substep<W, E, true, RK4, ANION>(B, P, cell, base, h, mk, im, ip, is, C_in, sa_rz, msa_rz, sa_ss, msa_ss);

    if constexpr (ANION) {
        ageing_anion<W, E>(B, A, base, sa_rz, msa_rz);
        ageing_anion<W, E>(B, A, base, sa_ss, msa_ss);
    } else {
        ageing<W, E>(B, A, base, sa_rz, msa_rz);
        ageing<W, E>(B, A, base, sa_ss, msa_ss);
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

#ifdef RH_SOLVER_RK4   // this unit's kernels: RH_SOLVER_RK4 0 / 1 (Euler / RK4), RH_SOLVER_ANION 0 / 1
// The whole day of an explicit solver in one launch; the smallest workgroup whose blocked layout covers the age classes.
int RH_SOLVER_NAME(hipStream_t stream, const SasArgs &args) {
    constexpr bool AN = RH_SOLVER_ANION != 0;
    const dim3 grid((unsigned)args.n);
#if RH_SOLVER_RK4
#define RH_LX(W, E) hipLaunchKernelGGL((k_sas_rk4<W, E, AN>), grid, dim3(W * 64), 0, stream, args)
#else
#define RH_LX(W, E) hipLaunchKernelGGL((k_sas_euler<W, E, AN>), grid, dim3(W * 64), 0, stream, args)
#endif
    const int nages = args.ages + 1;
    if (nages <= 64) RH_LX(1, 1);
    else if (nages <= 256) RH_LX(1, 4);
    else if (nages <= 512) RH_LX(2, 4);
    else if (nages <= 1024) RH_LX(4, 4);
    else if (nages <= 2048) RH_LX(8, 4);
    else RH_LX(8, 8);   // (sixteen waves would leave 128 registers per thread: 700 spilled)
#undef RH_LX
    return RH_OK;
}
#endif
