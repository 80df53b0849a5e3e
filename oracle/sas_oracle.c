/*
 * oracle/sas_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference NumPy backend's offline oxygen-18 transport step with
 * StorAge-selection (SAS) functions, deterministic solver (RoGeR,
 * roger/core/transport.py:949-991 `svat_transport_model_deterministic`, without write_output), and of the
 * same step for bromide (tracer = 1: the anion kernels, solute mass by age instead of a concentration).
 * One soil column at a time; age vectors are plain arrays.  It is the checker of the HIP SAS
 * kernel and the cpu_baseline of the SAS bench; the product never includes or calls it.
 *
 * Pinned by tests/golden/sas_<case>.npz, produced by the reference itself
 * (tests/golden/make_golden_sas.py), replayed in tests/test_oracle_sas.py.
 *
 * numpy semantics restated: `cumsum` is a running sum, `sum` over the contiguous age axis is
 * pairwise (oc_np_sum), `max` propagates NaN, `where` evaluates both branches.
 * SAS families: uniform (code 1), dirac (2), kumaraswami (3, 31-37), exponential (51) and power law (6, 61, 62)
 * -- roger/core/sas.py; every family contributes exactly 0 for the codes of the others, so the reference's sum
 * over all families is the selected one.  Also gamma (4; the incomplete gamma function by series / continued
 * fraction instead of scipy's cephes code) and the exponential with reversed age order (52).
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct oc_sas {
    int64_t n, ages, substeps;
    double vsmow, d18O_min, d18O_max;
    const int32_t *maskCatch;
    /* age-resolved state, (n, ages): [.., tau] of the reference */
    double *sa_rz, *msa_rz, *sa_ss, *msa_ss;
    /* daily fluxes set by the set_forcing hook, (n) */
    const double *inf_mat_rz, *inf_pf_rz, *inf_pf_ss, *evap_soil, *transp, *q_rz, *q_ss, *cpr_rz, *C_in;
    /* vs.sas_params_<flux>, (n, 8); order: evap_soil, transp, q_rz, q_ss, cpr_rz */
    const double *sas_params[5];
    /* outputs per outgoing flux, same order: tt, mtt (n, ages), TT (n, ages+1), C, C_iso (n) */
    double *tt[5], *mtt[5], *TT[5], *C[5], *C_iso[5];
    /* infiltration signals (n): inf_mat_rz, inf_pf_rz, inf_pf_ss */
    double *C_inf[3], *C_iso_inf[3];
    /* storages */
    double *sa_s, *msa_s;                                  /* (n, ages) */
    double *C_rz, *C_ss, *C_s, *C_iso_rz, *C_iso_ss, *C_iso_s; /* (n) */
    /* age statistics (n) or NULL: 10/25/50/75/90 percentile and mean of transp, q_ss, rz, ss, s */
    double *stats[5][6];
    /* soil.rescale_SA after the warm-up run */
    const double *S_rz_init, *S_ss_init;
    /* tracer: 0 oxygen-18 (msa = concentration by age), 1 bromide, 2 chloride, 3 virtual tracer (anion kernels: msa =
     * solute mass by age; bromide and chloride differ in soil.rescale_SA only, the virtual tracer is chloride that also
     * leaves with the soil evaporation) */
    int64_t tracer;
    const double *alpha_transp, *alpha_q, *S_sat_rz; /* (n) partition coefficients, saturation storage of the root zone */
    const int32_t *lu_id;                            /* (n) land use: 500 < lu_id < 599 is a crop */
    double *M[5];                                    /* (n) solute mass of the outgoing fluxes; [0] (evap_soil) unused */
    double *M_inf[3], *M_rz, *M_ss, *M_s;            /* (n) */
    /* settings.sas_solver: 0 "deterministic" (calc_tt's sub-stepped solve per flux), 1 "Euler", 2 "RK4" (all fluxes of a sub-step from
     * one StorAge, transport.py:2064-2414, 1139-2047; `substeps` sub-steps of h = 1 / substeps, calculate_storage_selection :3220-3300) */
    int64_t solver;
} oc_sas;

/* numpy pairwise add.reduce (see svat_oracle.c) */
static double np_pairwise(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
    }
}
static double np_sum(const double *a, int64_t n) { return 0.0 + np_pairwise(a, n); }
static double np_max(const double *a, int64_t n) {
    double m = a[0];
    for (int64_t i = 1; i < n; ++i) {
        if (isnan(a[i])) return a[i];
        if (a[i] > m) m = a[i];
    }
    return m;
}

/* transport.py:315-340 */
static double conc_to_delta(const oc_sas *P, double conc) {
    double d = 1000. * (conc / (P->vsmow * (1. - conc)) - 1.);
    return ((d < P->d18O_min) || (d > P->d18O_max)) ? NAN : d;
}

/* DIAGNOSTIC SWITCHES (environment OC_SAS_DEVICE_ORDER, default 0 = the reference's arithmetic; tools/sas_tie_causes.py): the device
 * kernel differs from the reference in the ORDER of two floating-point summations, and this reproduces either on the CPU to find out
 * which of them produces the device's residue ties:
 *   bit 0: the cumulative sums over the age axis in the association of the kernel's wave scan (one age class per lane, ages <= 63:
 *          the shape the small golden cases run in) instead of numpy's running sum;
 *   bit 1: the sub-step distributions accumulated directly (ttn += tti) instead of diff(cumsum(tti)) accumulated and differenced. */
static int oc_device_order(void) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("OC_SAS_DEVICE_ORDER");
        v = e ? atoi(e) : 0;
    }
    return v;
}
/* inclusive prefix sum over 64 lanes as rh_sas_dev.h wave_scan_sum forms it: row_shr 1, 2, 4, 8 inside rows of 16 lanes, then lane 15
 * of a row into rows 1 and 3, lane 31 into rows 2 and 3 */
static void oc_wave_scan(double *v) {
    double t[64];
    for (int sh = 1; sh <= 8; sh <<= 1) {
        for (int i = 0; i < 64; ++i) t[i] = ((i & 15) >= sh ? v[i - sh] : 0.0) + v[i];
        memcpy(v, t, sizeof(t));
    }
    for (int i = 0; i < 64; ++i) t[i] = (((i >> 4) & 1) ? v[(i & ~15) - 1] : 0.0) + v[i];
    memcpy(v, t, sizeof(t));
    for (int i = 0; i < 64; ++i) t[i] = ((i >> 5) ? v[31] : 0.0) + v[i];
    memcpy(v, t, sizeof(t));
}
/* SA[0] = 0, SA[1:] = cumsum(sa): transport.py:343-359 */
static void calc_SA(double *SA, const double *sa, int64_t ages) {
    SA[0] = 0;
    if ((oc_device_order() & 1) && ages <= 63) {   /* k_sas<1, 1>: hi = (inclusive scan of the lane before) + own value */
        double w[64];
        for (int i = 0; i < 64; ++i) w[i] = (i < ages ? sa[i] : 0.0);
        oc_wave_scan(w);
        for (int64_t k = 0; k < ages; ++k) SA[k + 1] = (k == 0 ? 0.0 : w[k - 1]) + sa[k];
        return;
    }
    double acc = 0;
    for (int64_t k = 0; k < ages; ++k) {
        acc = (k == 0) ? sa[0] : acc + sa[k];
        SA[k + 1] = acc;
    }
}

/* Regularised lower incomplete gamma function P(a, x) -- what scipy.special.gammainc returns (the gamma SAS family,
 * sas.py:153).  scipy's implementation (cephes igam / igamc with Temme's expansion for large a) is not restated: P is a
 * mathematical function, evaluated here by its power series for x < a + 1 and by the continued fraction of Q = 1 - P
 * (modified Lentz) otherwise, both to double precision; pinned against the reference's outputs by the golden case
 * sas_gamma_a40. */
static double oc_gammainc(double a, double x) {
    if (!(x > 0) || !(a > 0)) return 0.0;
    const double lead = exp(a * log(x) - x - lgamma(a));
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

/* Omega(S_T) over the nages points of SA; sas.py.  `p` is the column's 8 parameters (a private
 * copy: the storage-dependent variants 61/62 rewrite p[1], sas.py:219-226). */
static void sas_omega(double *Om, const double *SA, int64_t nages, double *p, double mk) {
    const double code = p[0];
    const double Smax = np_max(SA, nages);
    for (int64_t k = 0; k < nages; ++k) Om[k] = 0.0;
    if (code == 1) { /* uniform, sas.py:5-40 */
        const double S = Smax * 1.0 * mk;
        const double lam = 1 / S * 1.0 * mk;
        for (int64_t k = 0; k < nages; ++k) Om[k] = (SA[k] < S ? (SA[k] > 0 ? lam * SA[k] : 0.) : 1.) * 1.0 * mk;
        Om[nages - 1] = 1 * mk;
        for (int64_t k = 0; k < nages; ++k) Om[k] = (S <= 0 ? 0 : Om[k]) * mk;
    } else if (code == 6 || code == 61 || code == 62) { /* power, sas.py:191-240 */
        const double S = Smax * mk;
        double S_rel = (S - p[5]) / (p[6] - p[5]) * mk;
        S_rel = (S_rel < 0 ? 0 : S_rel);
        S_rel = (S_rel > 1 ? 1 : S_rel);
        if (code == 61) p[1] = p[3] + ((1 - S_rel) * p[4]);
        if (code == 62) p[1] = p[3] + (S_rel * p[4]);
        for (int64_t k = 0; k < nages; ++k) Om[k] = (SA[k] > 0 ? (SA[k] <= S ? pow(SA[k] / S, p[1]) : 1.) : 0.) * 1.0 * mk;
        for (int64_t k = 0; k < nages; ++k) Om[k] = (S <= 0 ? 0 : Om[k]) * mk;
    } else if (code == 2) { /* dirac (piston flow), sas.py:43-64: vs.nages = 0 .. ages is the edge index */
        const double S = Smax * mk;
        for (int64_t k = 0; k < nages; ++k) Om[k] = ((double)k <= p[1] ? 0 : 1) * mk;
        for (int64_t k = 0; k < nages; ++k) Om[k] = (S <= 0 ? 0 : Om[k]) * 1.0 * mk;
    } else if (code == 3 || (code >= 31 && code <= 37)) { /* kumaraswami, sas.py:67-147 */
        const double S = Smax * mk;
        double S_rel = (S - p[5]) / (p[6] - p[5]) * mk;
        S_rel = (S_rel < 0 ? 0 : S_rel);
        S_rel = (S_rel > 1 ? 1 : S_rel);
        const double up = p[3] + (S_rel * p[4]), down = p[3] + ((1 - S_rel) * p[4]);
        if (code == 31) { p[1] = 1; p[2] = up; }
        if (code == 32) { p[1] = down; p[2] = 1; }
        if (code == 33) { p[1] = 1; p[2] = down; }
        if (code == 34) { p[1] = up; p[2] = 1; }
        if (code == 35) { p[1] = down; p[2] = up; }
        if (code == 36) p[1] = down;
        if (code == 37) p[2] = up;
        for (int64_t k = 0; k < nages; ++k) {
            const double f = 1 - pow(1 - pow(SA[k] / S, p[1]), p[2]);
            const double o = (S >= 0 ? (SA[k] > 0 ? (SA[k] < S ? f : 1.) : 0.) : (SA[k] > 0 ? f : 0.));
            Om[k] = o * 1.0 * mk;
        }
        for (int64_t k = 0; k < nages; ++k) Om[k] = (S <= 0 ? 0 : Om[k]) * mk;
    } else if (code == 51) { /* exponential with preference for young water, sas.py:168-190 */
        const double S = Smax * mk;
        for (int64_t k = 0; k < nages; ++k)
            Om[k] = (SA[k] > 0 ? (SA[k] < S ? 1 - exp(p[1] * (-1) * (SA[k] / S)) : 1.) : 0.) * mk;
        for (int64_t k = 0; k < nages; ++k) Om[k] = (S <= 0 ? 0 : Om[k]) * mk;
    } else if (code == 52) { /* exponential, age order reversed (preference for old water), sas.py:186-190: Omega then
                              * DEcreases from 1 to 0 along the age axis and calc_tt clips every difference to 0 */
        const double S = Smax * mk;
        for (int64_t k = 0; k < nages; ++k) {
            const double x = SA[nages - 1 - k];
            Om[k] = (x > 0 ? (x < S ? 1 - exp(p[1] * (-1) * (x / S)) : 1.) : 0.) * mk;
        }
        for (int64_t k = 0; k < nages; ++k) Om[k] = (S <= 0 ? 0 : Om[k]) * mk;
    } else if (code == 4) { /* gamma, sas.py:139-163: gammainc is regularised already and is divided by Gamma(a) once
                             * more; the top edge (SA == S) gets 0, not 1 */
        const double S = Smax * 1.0 * mk;
        const double G = exp(lgamma(p[1]));
        for (int64_t k = 0; k < nages; ++k)
            Om[k] = (SA[k] > 0 ? (SA[k] < S ? oc_gammainc(p[1], p[2] * SA[k] / S) / G : 0.) : 0) * 1.0 * mk;
        for (int64_t k = 0; k < nages; ++k) Om[k] = (S <= 0 ? 0 : Om[k]) * mk;
    }
}

/* backward travel time distribution of one flux: transport.py:362-509.  work: 6 * (ages + 1) doubles */
static void calc_tt(const oc_sas *P, double *tt, const double *SA, const double *sa, double flux, const double *params,
                    double mk, double *work) {
    const int64_t A = P->ages, NA = A + 1, N = P->substeps;
    double *SAn = work, *san = SAn + NA, *TTn = san + NA, *TTi = TTn + NA, *tti = TTi + NA, *TT = tti + NA;
    double p[8];
    memcpy(p, params, sizeof(p));
    memcpy(SAn, SA, sizeof(double) * NA);
    memcpy(san, sa, sizeof(double) * A);
    for (int64_t k = 0; k < NA; ++k) TTn[k] = 0;
    double ttn_direct[4096];
    if (oc_device_order() & 2) {
        if (A > 4096) abort();
        for (int64_t k = 0; k < A; ++k) ttn_direct[k] = 0;
    }
    const double h = 1.0 / (double)N;
    for (int64_t it = 0; it < N; ++it) {
        sas_omega(TTi, SAn, NA, p, mk);
        double acc = 0, cum = 0;
        for (int64_t k = 0; k < A; ++k) {
            double d = TTi[k + 1] - TTi[k];
            double t = (d >= 0 ? d : 0);
            double q = (flux * t * h > san[k] ? san[k] : flux * t * h);
            t = (flux * h > 0 ? q / (flux * h) : 0);
            tti[k] = t;
            san[k] += -t * flux * h;
            acc = (k == 0) ? san[0] : acc + san[k];
            SAn[k + 1] = acc;
            cum = (k == 0) ? t : cum + t;
            TTn[k + 1] += cum;
            if (oc_device_order() & 2) ttn_direct[k] += t;
        }
        if (oc_device_order() & 1) calc_SA(SAn, san, A);
    }
    for (int64_t k = 0; k < NA; ++k) TT[k] = TTn[k] / (double)N;
    if (oc_device_order() & 2) {   /* the kernel: tt = ttn / N directly */
        for (int64_t k = 0; k < A; ++k) {
            double t = ttn_direct[k] / (double)N;
            double q = (flux * t > sa[k] ? sa[k] : flux * t);
            tt[k] = (flux > 0 ? q / flux : 0);
        }
        return;
    }
    for (int64_t k = 0; k < A; ++k) {
        double t = TT[k + 1] - TT[k];
        double q = (flux * t > sa[k] ? sa[k] : flux * t);
        tt[k] = (flux > 0 ? q / flux : 0);
    }
}

/* outgoing flux from `sa`/`msa` (source store); if sa_sink != NULL the water joins the sink store at
 * the same age with volume-weighted mixing of the isotope signal.
 * evapotranspiration.py:653-719, 831-901; subsurface_runoff.py:1531-1626, 1753-1820;
 * capillary_rise.py:404-500 */
static void outflux(const oc_sas *P, int64_t i, int f, double flux, double *sa, double *msa, double *sa_sink,
                    double *msa_sink, double mk, double *work) {
    const int64_t A = P->ages;
    double *SA = work, *scratch = work + (A + 1);
    double *tt = P->tt[f] + i * A, *mtt = P->mtt[f] + i * A, *TT = P->TT[f] + i * (A + 1);
    calc_SA(SA, sa, A);
    for (int64_t k = 0; k <= A; ++k) SA[k] *= mk;
    calc_tt(P, tt, SA, sa, flux, P->sas_params[f] + i * 8, mk, scratch);
    for (int64_t k = 0; k < A; ++k) tt[k] *= mk;
    {
        double acc = 0;
        for (int64_t k = 0; k < A; ++k) {
            acc = (k == 0) ? tt[0] : acc + tt[k];
            TT[k + 1] = acc;
        }
    }
    for (int64_t k = 0; k < A; ++k) mtt[k] = (tt[k] > 0 ? msa[k] : 0) * mk;
    {   /* calc_conc_iso_flux :512-535 */
        double *prod = scratch;
        for (int64_t k = 0; k < A; ++k) prod[k] = mtt[k] * tt[k];
        const double s = np_sum(tt, A);
        double conc = (s > 0 ? np_sum(prod, A) / s : NAN);
        conc = (conc != 0 ? conc : NAN);
        P->C[f][i] = conc * mk;
        P->C_iso[f][i] = conc_to_delta(P, P->C[f][i]) * mk;
    }
    if (sa_sink) {
        for (int64_t k = 0; k < A; ++k) {
            const double add = tt[k] * flux;
            msa_sink[k] = (add + sa_sink[k] > 0 ? msa_sink[k] * (sa_sink[k] / (add + sa_sink[k])) + mtt[k] * (add / (add + sa_sink[k]))
                                                : msa_sink[k]) * mk;
        }
    }
    for (int64_t k = 0; k < A; ++k) { /* update_sa :599-619 */
        double v = sa[k] + -flux * tt[k];
        v = ((v > -1e-5) && (v < 0)) ? 0 : v;
        sa[k] = v * mk;
    }
    if (sa_sink)
        for (int64_t k = 0; k < A; ++k) sa_sink[k] += tt[k] * flux * mk;
    for (int64_t k = 0; k < A; ++k) msa[k] = (sa[k] <= 0 ? 0 : msa[k]) * mk;
}

/* --- anion (bromide) kernels: msa holds solute mass per age class ------------------------------------------ */

/* nansum over the age axis: NaN -> 0, then numpy's pairwise add.reduce */
static double np_nansum(const double *a, int64_t n, double *scratch) {
    for (int64_t k = 0; k < n; ++k) scratch[k] = isnan(a[k]) ? 0 : a[k];
    return np_sum(scratch, n);
}

/* outgoing flux of the anion kernels.  water_only: calc_evaporation_transport_kernel
 * (evapotranspiration.py:620-650), the solute stays behind.  Otherwise calc_transpiration_transport_anion_kernel
 * (:905-985), calc_percolation_rz/ss_transport_anion_kernel (subsurface_runoff.py:1630-1716, 1823-1893),
 * calc_capillary_rise_rz_transport_anion_kernel (capillary_rise.py:503-590) with calc_mtt's anion branch
 * (transport.py:583-596): mtt = msa / sa * alpha * tt * flux clipped to [0, msa]. */
static void outflux_anion(const oc_sas *P, int64_t i, int f, double flux, double alpha, int water_only, double *sa, double *msa,
                          double *sa_sink, double *msa_sink, double mk, double *work) {
    const int64_t A = P->ages;
    double *SA = work, *scratch = work + (A + 1);
    double *tt = P->tt[f] + i * A, *mtt = P->mtt[f] + i * A, *TT = P->TT[f] + i * (A + 1);
    calc_SA(SA, sa, A);
    for (int64_t k = 0; k <= A; ++k) SA[k] *= mk;
    calc_tt(P, tt, SA, sa, flux, P->sas_params[f] + i * 8, mk, scratch);
    for (int64_t k = 0; k < A; ++k) tt[k] *= mk;
    {
        double acc = 0;
        for (int64_t k = 0; k < A; ++k) {
            acc = (k == 0) ? tt[0] : acc + tt[k];
            TT[k + 1] = acc;
        }
    }
    if (!water_only) {
        for (int64_t k = 0; k < A; ++k) {
            double m = (sa[k] > 0 ? msa[k] / sa[k] : 0) * alpha * tt[k] * flux;
            m = (m <= 0 ? 0 : m);
            m = (m > msa[k] ? msa[k] : m);
            mtt[k] = m * mk;
        }
        const double tot = np_sum(mtt, A);
        P->C[f][i] = (flux > 0 ? tot / flux : 0) * mk;
        P->M[f][i] = tot * mk;
    }
    for (int64_t k = 0; k < A; ++k) { /* update_sa :599-619 */
        double v = sa[k] + -flux * tt[k];
        v = ((v > -1e-5) && (v < 0)) ? 0 : v;
        sa[k] = v * mk;
    }
    if (!water_only)
        for (int64_t k = 0; k < A; ++k) msa[k] += -mtt[k] * mk;
    if (sa_sink) {
        for (int64_t k = 0; k < A; ++k) msa_sink[k] += mtt[k] * mk;
        for (int64_t k = 0; k < A; ++k) sa_sink[k] += tt[k] * flux * mk;
    }
}

/* ageing of the anion kernels: calc_ageing_sa :623-652, calc_ageing_msa :655-680 (no NaN marker) */
static void ageing_anion(double *sa, double *msa, int64_t A) {
    const double sa_last = sa[A - 1], msa_last = msa[A - 1];
    for (int64_t k = A - 1; k >= 1; --k) {
        sa[k] = sa[k - 1];
        msa[k] = msa[k - 1];
    }
    sa[0] = 0;
    msa[0] = 0;
    sa[A - 1] += sa_last;
    sa[A - 1] = (sa[A - 1] < 1e-8 ? 0 : sa[A - 1]);
    msa[A - 1] += msa_last;
}

static void storages_anion(const oc_sas *P, int64_t i, double *work);
/* one day of svat_transport_model_deterministic for one column, bromide */
static void step_anion(const oc_sas *P, int64_t i, double *work) {
    const int64_t A = P->ages;
    const double mk = (double)P->maskCatch[i];
    double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    const double C_in = P->C_in[i];
    {   /* calc_infiltration_rz_transport_anion_kernel, infiltration.py:2350-2424 */
        const double im = P->inf_mat_rz[i], ip = P->inf_pf_rz[i];
        P->C_inf[0][i] = (im > 0 ? C_in : 0) * mk;
        P->C_inf[1][i] = (ip > 0 ? C_in : 0) * mk;
        P->M_inf[0][i] = P->C_inf[0][i] * im * mk;
        P->M_inf[1][i] = P->C_inf[1][i] * ip * mk;
        sa_rz[0] += im + ip * mk;
        msa_rz[0] += P->M_inf[0][i] + P->M_inf[1][i] * mk;
    }
    /* soil evaporation: water only (calc_evaporation_transport_kernel), but the virtual tracer leaves with it at alpha = 1
     * (calc_evaporation_transport_virtualtracer_kernel, evapotranspiration.py:722-791) */
    outflux_anion(P, i, 0, P->evap_soil[i], 1.0, P->tracer == 3 ? 0 : 1, sa_rz, msa_rz, NULL, NULL, mk, work);
    {   /* crop solute uptake stops above 80 % saturation: evapotranspiration.py:932-939 */
        const int crop = (P->lu_id[i] > 500) && (P->lu_id[i] < 599) && (np_sum(sa_rz, A) >= 0.8 * P->S_sat_rz[i]);
        const double alpha = (crop ? 0 : P->alpha_transp[i]) * mk;
        outflux_anion(P, i, 1, P->transp[i], alpha, 0, sa_rz, msa_rz, NULL, NULL, mk, work);
    }
    outflux_anion(P, i, 2, P->q_rz[i], P->alpha_q[i], 0, sa_rz, msa_rz, sa_ss, msa_ss, mk, work);
    {   /* calc_infiltration_ss_transport_anion_kernel, infiltration.py:2516-2566 */
        const double ip = P->inf_pf_ss[i];
        P->C_inf[2][i] = (ip > 0 ? C_in : 0) * mk;
        P->M_inf[2][i] = P->C_inf[2][i] * ip * mk;
        sa_ss[0] += ip * mk;
        msa_ss[0] += P->M_inf[2][i] * mk;
    }
    outflux_anion(P, i, 3, P->q_ss[i], P->alpha_q[i], 0, sa_ss, msa_ss, NULL, NULL, mk, work);
    outflux_anion(P, i, 4, P->cpr_rz[i], P->alpha_q[i], 0, sa_ss, msa_ss, sa_rz, msa_rz, mk, work);
    storages_anion(P, i, work);
}

/* storages of the anion kernels: root_zone.py:221-258, subsoil.py:186-223, soil.py:1094-1142 */
static void storages_anion(const oc_sas *P, int64_t i, double *work) {
    const int64_t A = P->ages;
    const double mk = (double)P->maskCatch[i];
    double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    for (int64_t k = 0; k < A; ++k) sa_rz[k] = (sa_rz[k] < 1e-8 ? 0 : sa_rz[k]);
    for (int64_t k = 0; k < A; ++k) msa_rz[k] = (sa_rz[k] <= 0 ? 0 : msa_rz[k]);
    P->M_rz[i] = np_nansum(msa_rz, A, work) * mk;
    {
        const double S = np_sum(sa_rz, A);
        P->C_rz[i] = (S > 0 ? P->M_rz[i] / S : 0);
    }
    for (int64_t k = 0; k < A; ++k) sa_ss[k] = (sa_ss[k] < 1e-8 ? 0 : sa_ss[k]);
    for (int64_t k = 0; k < A; ++k) msa_ss[k] = (sa_ss[k] <= 0 ? 0 : msa_ss[k]);
    P->M_ss[i] = np_nansum(msa_ss, A, work) * mk;
    {
        const double S = np_sum(sa_ss, A);
        P->C_ss[i] = (S > 0 ? P->M_ss[i] / S : 0);
    }
    double *sa_s = P->sa_s + i * A, *msa_s = P->msa_s + i * A;
    for (int64_t k = 0; k < A; ++k) {
        sa_s[k] = sa_rz[k] + sa_ss[k] * mk;
        msa_s[k] = msa_rz[k] + msa_ss[k] * mk;
    }
    P->M_s[i] = np_nansum(msa_s, A, work) * mk;
    {
        const double S = np_sum(sa_s, A);
        P->C_s[i] = (S > 0 ? P->M_s[i] / S : 0);
    }
}

/* infiltration into age class 0: infiltration.py:2218-2346 (rz: matrix then preferential flow),
 * :2441-2512 (ss) */
static void inflow(const oc_sas *P, int64_t i, int which, double inf, double *sa, double *msa, double mk) {
    const int64_t A = P->ages;
    const double C_in = P->C_in[i];
    P->C_inf[which][i] = (inf > 0 ? C_in : 0) * mk;
    P->C_iso_inf[which][i] = conc_to_delta(P, P->C_inf[which][i]) * mk;
    for (int64_t k = 0; k < A; ++k) {
        const double ttk = (k == 0) ? (inf > 0 ? 1 : 0) * mk : 0.0;
        const double mttk = (k == 0) ? (inf > 0 ? C_in : 0) * mk : 0.0;
        msa[k] = (inf * ttk + sa[k] > 0 ? msa[k] * (sa[k] / (ttk * inf + sa[k])) + mttk * ((ttk * inf) / (inf * ttk + sa[k]))
                                       : msa[k]) * mk;
    }
    sa[0] += inf * mk;
}

/* calc_conc_iso_storage :538-562 */
static double conc_storage(const double *sa, const double *msa, int64_t A, double *scratch) {
    for (int64_t k = 0; k < A; ++k) scratch[k] = msa[k] * sa[k];
    const double s = np_sum(sa, A);
    return (s > 0 ? np_sum(scratch, A) / s : 0);
}

/* np.interp(q, cdf, ages 1..A) with the NaN rule of calc_age_percentile, transport.py:9-56 */
static double age_percentile(const double *cdf, int64_t A, double q) {
    if (!(np_max(cdf, A) > 0)) return NAN;
    if (q < cdf[0]) return 1.0;
    if (q >= cdf[A - 1]) return (double)A;
    int64_t j = 0;
    while (j + 1 < A && !(q < cdf[j + 1])) ++j; /* largest j with cdf[j] <= q  (cdf non-decreasing) */
    const double x0 = cdf[j], x1 = cdf[j + 1];
    if (x1 == x0) return (double)(j + 1);
    const double slope = ((double)(j + 2) - (double)(j + 1)) / (x1 - x0);
    return slope * (q - x0) + (double)(j + 1);
}

/* ageing by one day: transport.py:682-739 */
static void ageing(double *sa, double *msa, int64_t A, double *scratch) {
    double *sam1 = scratch, *msam1 = scratch + A;
    memcpy(sam1, sa, sizeof(double) * A);
    memcpy(msam1, msa, sizeof(double) * A);
    for (int64_t k = 1; k < A; ++k) sa[k] = sam1[k - 1];
    for (int64_t k = 1; k < A; ++k) msa[k] = msam1[k - 1];
    msa[0] = 0;
    {
        const double tot = sa[A - 1] + sam1[A - 1];
        double v = (tot > 0 ? msam1[A - 1] * (sam1[A - 1] / tot) + msa[A - 1] * (sa[A - 1] / tot) : 0);
        msa[A - 1] = isnan(v) ? 0 : v;
    }
    sa[0] = 0;
    sa[A - 1] += sam1[A - 1];
    sa[A - 1] = (sa[A - 1] < 1e-8 ? 0 : sa[A - 1]);
    msa[A - 1] = (sa[A - 1] <= 0 ? NAN : msa[A - 1]);
}

/* soil.rescale_SA for oxygen-18: rescale_sa_msa_iso_soil_kernel, core/soil.py:1250-1395 (time level tau) */
void oc_sas_rescale(const oc_sas *P) {
    const int64_t A = P->ages;
    double *work = (double *)malloc(sizeof(double) * (A + 4));
    for (int64_t i = 0; i < P->n; ++i) {
        const double mk = (double)P->maskCatch[i];
        double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
        const double t_rz = np_sum(sa_rz, A), t_ss = np_sum(sa_ss, A);
        for (int64_t k = 0; k < A; ++k) sa_rz[k] = P->S_rz_init[i] * (sa_rz[k] / t_rz);
        for (int64_t k = 0; k < A; ++k) sa_ss[k] = P->S_ss_init[i] * (sa_ss[k] / t_ss);
        if (P->tracer == 2 || P->tracer == 3) { /* chloride, virtual tracer: rescale_sa_msa_anion_soil_kernel, core/soil.py:1507-1640 -- the solute is scaled
                               * with the water (by S_init / sum(sa) of the state before the rescaling), M_* stay */
            const double f_rz = P->S_rz_init[i] / t_rz, f_ss = P->S_ss_init[i] / t_ss;
            for (int64_t k = 0; k < A; ++k) {
                msa_rz[k] *= f_rz;
                msa_ss[k] *= f_ss;
                P->sa_s[i * A + k] = sa_rz[k] + sa_ss[k];
                P->msa_s[i * A + k] = msa_rz[k] + msa_ss[k];
            }
            P->C_rz[i] = np_sum(msa_rz, A) / np_sum(sa_rz, A);
            P->C_ss[i] = np_sum(msa_ss, A) / np_sum(sa_ss, A);
            P->C_s[i] = np_sum(P->msa_s + i * A, A) / np_sum(P->sa_s + i * A, A);
            continue;
        }
        if (P->tracer == 1) { /* bromide: rescale_sa_msa_anion_soil_kernel, core/soil.py:1399-1506 -- the soil starts free of bromide */
            for (int64_t k = 0; k < A; ++k) {
                msa_rz[k] = msa_ss[k] = P->msa_s[i * A + k] = 0;
                P->sa_s[i * A + k] = sa_rz[k] + sa_ss[k];
            }
            P->C_rz[i] = P->C_ss[i] = P->C_s[i] = 0;
            P->M_rz[i] = P->M_ss[i] = P->M_s[i] = 0;
            continue;
        }
        P->C_rz[i] = conc_storage(sa_rz, msa_rz, A, work);
        P->C_iso_rz[i] = conc_to_delta(P, P->C_rz[i]) * mk;
        P->C_ss[i] = conc_storage(sa_ss, msa_ss, A, work);
        P->C_iso_ss[i] = conc_to_delta(P, P->C_ss[i]) * mk;
        double *sa_s = P->sa_s + i * A, *msa_s = P->msa_s + i * A;
        for (int64_t k = 0; k < A; ++k) {
            sa_s[k] = sa_rz[k] + sa_ss[k];
            const double tot = sa_rz[k] + sa_ss[k];
            double v = (tot > 0 ? msa_rz[k] * (sa_rz[k] / tot) + msa_ss[k] * (sa_ss[k] / tot) : 0);
            msa_s[k] = isnan(v) ? 0 : v;
        }
        msa_s[0] = 0;
        P->C_s[i] = conc_storage(sa_s, msa_s, A, work);
        P->C_iso_s[i] = conc_to_delta(P, P->C_s[i]) * mk;
    }
    free(work);
}

/* one day of svat_transport_model_deterministic for all columns */
void oc_sas_set_num_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}
int oc_sas_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}


/* storages of the oxygen-18 model: root_zone.py:189-217, subsoil.py:159-188, soil.py:1036-1090 */
static void storages_iso(const oc_sas *P, int64_t i, double *work) {
    const int64_t A = P->ages;
    const double mk = (double)P->maskCatch[i];
    double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    double *sa_s = P->sa_s + i * A, *msa_s = P->msa_s + i * A;
    for (int64_t k = 0; k < A; ++k) sa_rz[k] = (sa_rz[k] < 1e-8 ? 0 : sa_rz[k]);
    P->C_rz[i] = conc_storage(sa_rz, msa_rz, A, work) * mk;
    P->C_iso_rz[i] = conc_to_delta(P, P->C_rz[i]) * mk;
    for (int64_t k = 0; k < A; ++k) sa_ss[k] = (sa_ss[k] < 1e-8 ? 0 : sa_ss[k]);
    P->C_ss[i] = conc_storage(sa_ss, msa_ss, A, work) * mk;
    P->C_iso_ss[i] = conc_to_delta(P, P->C_ss[i]) * mk;
    for (int64_t k = 0; k < A; ++k) {
        sa_s[k] = sa_rz[k] + sa_ss[k] * mk;
        const double tot = sa_rz[k] + sa_ss[k];
        double v = (tot > 0 ? msa_rz[k] * (sa_rz[k] / tot) + msa_ss[k] * (sa_ss[k] / tot) : 0);
        msa_s[k] = isnan(v) ? 0 : v;
    }
    P->C_s[i] = conc_storage(sa_s, msa_s, A, work) * mk;
    P->C_iso_s[i] = conc_to_delta(P, P->C_s[i]) * mk;
}

/* ---- Euler / RK4 solvers (oxygen-18, deuterium) ------------------------------------------------------------------------------- */
/* Mixing of an addition (dsa1 of water carrying dmsa1) into an age class, transport.py:2122-2137, 2276-2291.  The root zone's
 * formula keeps the old signal only where it is positive (`& (msa > 0)`), the subsoil's does not: both as the reference has them. */
static double euler_mix(double msa, double sa, double dsa1, double dmsa1, int need_pos) {
    const double tot = dsa1 + sa;
    const double a = ((tot > 0) && (!need_pos || (msa > 0))) ? msa * (sa / tot) : 0;
    const double b = (tot > 0) ? dmsa1 * (dsa1 / tot) : 0;
    double m = a + b;
    m = ((dsa1 > 0) && (m <= 0)) ? dmsa1 : m;
    return m;
}
/* calc_TT_num + calc_TT_num_nonneg + the clipped differences (transport.py:860-948, 2187-2199): the cumulative and the plain travel
 * time distribution of one flux from the cumulative StorAge SA (NA points, masked) of its source.  work: 3 * NA doubles. */
static void euler_tt(const oc_sas *P, double *TT, double *tt, const double *SA, const double *params, double flux_h, double mk, double *work) {
    const int64_t A = P->ages, NA = A + 1;
    double *Om = work, *nn = work + NA;
    double p[8];
    memcpy(p, params, sizeof(p));
    sas_omega(Om, SA, NA, p, mk);
    if (flux_h <= 0)
        for (int64_t k = 0; k < NA; ++k) Om[k] = 0;
    for (int64_t k = 0; k < A; ++k) {
        const double sa_d = SA[k + 1] - SA[k], ttq = (Om[k + 1] - Om[k]) * flux_h;
        double v = (sa_d + ttq < 0) ? -sa_d : ttq;
        v = (v == 0) ? 0.0 : v;   /* where(x == -0, 0, x) */
        nn[k] = v;
    }
    const double s = np_sum(nn, A);
    for (int64_t k = 0; k < A; ++k) nn[k] = (nn[k] > 0) ? nn[k] / s : 0;
    TT[0] = 0;
    {
        double acc = 0;
        for (int64_t k = 0; k < A; ++k) {
            acc = (k == 0) ? nn[0] : acc + nn[k];
            TT[k + 1] = acc;
        }
    }
    for (int64_t k = 0; k < A; ++k) {
        const double d = TT[k + 1] - TT[k];
        tt[k] = (d >= 0) ? d : 0;
    }
}
static double conc_iso_flux(const double *mtt, const double *tt, int64_t A, double *scratch) { /* calc_conc_iso_flux :512-535 */
    for (int64_t k = 0; k < A; ++k) scratch[k] = mtt[k] * tt[k];
    const double s = np_sum(tt, A);
    double conc = (s > 0 ? np_sum(scratch, A) / s : NAN);
    return (conc != 0 ? conc : NAN);
}
/* the upper boundary condition of a sub-step: infiltration into age class 0, transport.py:2071-2145 (Euler), 1146-1220 (RK4) */
static void euler_inflow(const oc_sas *P, int64_t i, double h) {
    const int64_t A = P->ages;
    const double mk = (double)P->maskCatch[i];
    double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    const double im = P->inf_mat_rz[i], ip = P->inf_pf_rz[i], is = P->inf_pf_ss[i], C_in = P->C_in[i];
    for (int64_t k = 0; k < A; ++k) {
        const double t0 = (k == 0) ? (im > 0 ? 1 : 0) * mk : 0, t1 = (k == 0) ? (ip > 0 ? 1 : 0) * mk : 0, t2 = (k == 0) ? (is > 0 ? 1 : 0) * mk : 0;
        const double m0 = (k == 0) ? (im > 0 ? C_in : 0) * mk : 0, m1 = (k == 0) ? (ip > 0 ? C_in : 0) * mk : 0, m2 = (k == 0) ? (is > 0 ? C_in : 0) * mk : 0;
        const double dsa_rz = (im * t0 + ip * t1) * h, dsa_ss = (is * t2) * h;
        const double dmsa_rz1 = (isnan(m0) ? 0 : m0) * (dsa_rz > 0 ? ((im * t0 * h) / dsa_rz) : 0) + (isnan(m1) ? 0 : m1) * (dsa_rz > 0 ? ((ip * t1 * h) / dsa_rz) : 0);
        const double dmsa_ss1 = (isnan(m2) ? 0 : m2) * (dsa_ss > 0 ? ((is * t2 * h) / dsa_ss) : 0);
        msa_rz[k] = euler_mix(msa_rz[k], sa_rz[k], dsa_rz, dmsa_rz1, 1);
        msa_ss[k] = euler_mix(msa_ss[k], sa_ss[k], dsa_ss, dmsa_ss1, 0);
        sa_rz[k] += dsa_rz;
        sa_ss[k] += dsa_ss;
        msa_rz[k] = (sa_rz[k] <= 0) ? 0 : msa_rz[k];
        msa_ss[k] = (sa_ss[k] <= 0) ? 0 : msa_ss[k];
    }
}
/* ... of the anion kernels, transport.py:2150-2170 (Euler), 1225-1245 (RK4): water as above; the solute arrays of the three
 * infiltration fluxes hold C_in in age class 0 whatever infiltrates (:2100-2112), and each adds it times h */
static void euler_inflow_anion(const oc_sas *P, int64_t i, double h) {
    const double mk = (double)P->maskCatch[i];
    double *sa_rz = P->sa_rz + i * P->ages, *msa_rz = P->msa_rz + i * P->ages, *sa_ss = P->sa_ss + i * P->ages, *msa_ss = P->msa_ss + i * P->ages;
    const double im = P->inf_mat_rz[i], ip = P->inf_pf_rz[i], is = P->inf_pf_ss[i];
    const double t0 = (im > 0 ? 1 : 0) * mk, t1 = (ip > 0 ? 1 : 0) * mk, t2 = (is > 0 ? 1 : 0) * mk;
    const double m = P->C_in[i] * mk, mm = isnan(m) ? 0 : m;
    sa_rz[0] += (im * t0 + ip * t1) * h;
    sa_ss[0] += (is * t2) * h;
    msa_rz[0] += mm * h + mm * h;
    msa_ss[0] += mm * h;
}
/* calc_mtt, anion branch (:583-596): msa / sa * alpha * tt * flux clipped to [0, msa]; flux_h = flux * h */
static void anion_mtt(double *mtt, const double *sa, const double *msa, const double *tt, double alpha, double flux_h, int64_t A) {
    for (int64_t k = 0; k < A; ++k) {
        double m = (sa[k] > 0 ? msa[k] / sa[k] : 0) * alpha * tt[k] * flux_h;
        m = (m <= 0 ? 0 : m);
        mtt[k] = (m > msa[k] ? msa[k] : m);
    }
}
/* the solute leaving with the four fluxes that carry it (the soil evaporation's is never assigned for the anions, :2196-2200) from the
 * state as it stands and the distributions in P->tt */
static void anion_mtt_fluxes(const oc_sas *P, int64_t i, double h) {
    const int64_t A = P->ages;
    const double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    anion_mtt(P->mtt[1] + i * A, sa_rz, msa_rz, P->tt[1] + i * A, P->alpha_transp[i], P->transp[i] * h, A);
    anion_mtt(P->mtt[2] + i * A, sa_rz, msa_rz, P->tt[2] + i * A, P->alpha_q[i], P->q_rz[i] * h, A);
    anion_mtt(P->mtt[3] + i * A, sa_ss, msa_ss, P->tt[3] + i * A, P->alpha_q[i], P->q_ss[i] * h, A);
    anion_mtt(P->mtt[4] + i * A, sa_ss, msa_ss, P->tt[4] + i * A, P->alpha_q[i], P->cpr_rz[i] * h, A);
}
/* the update of a sub-step, anion kernels: transport.py:2308-2332 (Euler) and 1941-1966 (RK4, whose root zone -- as written -- GAINS the
 * soil evaporation and does not receive the capillary rise) */
static void explicit_update_anion(const oc_sas *P, int64_t i, double h, int rk4) {
    const int64_t A = P->ages;
    double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    const double ev = P->evap_soil[i], tr = P->transp[i], qrz = P->q_rz[i], qss = P->q_ss[i], cpr = P->cpr_rz[i];
    const double *tt_ev = P->tt[0] + i * A, *tt_tr = P->tt[1] + i * A, *tt_qrz = P->tt[2] + i * A, *tt_qss = P->tt[3] + i * A, *tt_cpr = P->tt[4] + i * A;
    const double *mtt_tr = P->mtt[1] + i * A, *mtt_qrz = P->mtt[2] + i * A, *mtt_qss = P->mtt[3] + i * A, *mtt_cpr = P->mtt[4] + i * A;
#define NZ(x) (isnan(x) ? 0 : (x))
    for (int64_t k = 0; k < A; ++k) {
        double dsa_rz = rk4 ? (ev * tt_ev[k] - tr * tt_tr[k] - qrz * tt_qrz[k]) * h : (cpr * tt_cpr[k] - ev * tt_ev[k] - tr * tt_tr[k] - qrz * tt_qrz[k]) * h;
        dsa_rz = (sa_rz[k] + dsa_rz < 0) ? -sa_rz[k] : dsa_rz;
        double dsa_ss = (qrz * tt_qrz[k] - cpr * tt_cpr[k] - qss * tt_qss[k]) * h;
        dsa_ss = (sa_ss[k] + dsa_ss < 0) ? -sa_ss[k] : dsa_ss;
        double dmsa_rz = NZ(mtt_cpr[k]) - NZ(mtt_tr[k]) - NZ(mtt_qrz[k]);
        double dmsa_ss = NZ(mtt_qrz[k]) - NZ(mtt_cpr[k]) - NZ(mtt_qss[k]);
        dmsa_rz = (msa_rz[k] + dmsa_rz < 0) ? 0 : dmsa_rz;
        dmsa_ss = (msa_ss[k] + dmsa_ss < 0) ? 0 : dmsa_ss;
        sa_rz[k] += dsa_rz;
        sa_ss[k] += dsa_ss;
        msa_rz[k] += dmsa_rz;
        msa_ss[k] += dmsa_ss;
    }
#undef NZ
}
/* concentrations of a sub-step, anion kernels: transport.py:2379-2407, 2012-2040 */
static void explicit_concentrations_anion(const oc_sas *P, int64_t i, double h) {
    const int64_t A = P->ages;
    const double mk = (double)P->maskCatch[i];
    const double inf[3] = {P->inf_mat_rz[i], P->inf_pf_rz[i], P->inf_pf_ss[i]};
    const double flux[5] = {P->evap_soil[i], P->transp[i], P->q_rz[i], P->q_ss[i], P->cpr_rz[i]};
    for (int w = 0; w < 3; ++w) P->C_inf[w][i] = (inf[w] * h > 0 ? P->C_in[i] : 0) * mk;
    for (int f = 1; f < 5; ++f) P->C[f][i] = (flux[f] > 0 ? np_sum(P->mtt[f] + i * A, A) / (flux[f] * h) : 0) * mk;
}
/* order of the five outgoing fluxes in the arrays: evap_soil, transp, q_rz (root zone), q_ss, cpr_rz (subsoil) */
static const int EULER_SRC_SS[5] = {0, 0, 0, 1, 1};
/* the distributions of the five fluxes from the state as it stands; SA_rz / SA_ss: NA doubles each, work: 3 * NA */
static void euler_distributions(const oc_sas *P, int64_t i, double h, double *SA_rz, double *SA_ss, double *work) {
    const int64_t A = P->ages, NA = A + 1;
    const double mk = (double)P->maskCatch[i];
    const double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    const double flux[5] = {P->evap_soil[i], P->transp[i], P->q_rz[i], P->q_ss[i], P->cpr_rz[i]};
    calc_SA(SA_rz, sa_rz, A);
    calc_SA(SA_ss, sa_ss, A);
    for (int64_t k = 0; k < NA; ++k) {
        SA_rz[k] *= mk;
        SA_ss[k] *= mk;
    }
    for (int f = 0; f < 5; ++f) {
        double *tt = P->tt[f] + i * A, *mtt = P->mtt[f] + i * A, *TT = P->TT[f] + i * NA;
        const double *msa = EULER_SRC_SS[f] ? msa_ss : msa_rz;
        euler_tt(P, TT, tt, EULER_SRC_SS[f] ? SA_ss : SA_rz, P->sas_params[f] + i * 8, flux[f] * h, mk, work);
        if (P->tracer == 0)
            for (int64_t k = 0; k < A; ++k) mtt[k] = (tt[k] > 0) ? msa[k] : 0;   /* calc_mtt :565-596, isotopes */
    }
    if (P->tracer != 0) anion_mtt_fluxes(P, i, h);
}
/* the StorAge update of a sub-step from the distributions in P->tt / P->mtt, transport.py:2266-2310 */
static void euler_update(const oc_sas *P, int64_t i, double h) {
    const int64_t A = P->ages;
    double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    const double ev = P->evap_soil[i], tr = P->transp[i], qrz = P->q_rz[i], qss = P->q_ss[i], cpr = P->cpr_rz[i];
    const double *tt_ev = P->tt[0] + i * A, *tt_tr = P->tt[1] + i * A, *tt_qrz = P->tt[2] + i * A, *tt_qss = P->tt[3] + i * A, *tt_cpr = P->tt[4] + i * A;
    const double *mtt_qrz = P->mtt[2] + i * A, *mtt_cpr = P->mtt[4] + i * A;
    for (int64_t k = 0; k < A; ++k) {
        double dsa_rz = (cpr * tt_cpr[k] - ev * tt_ev[k] - tr * tt_tr[k] - qrz * tt_qrz[k]) * h;
        dsa_rz = (sa_rz[k] + dsa_rz < 0) ? -sa_rz[k] : dsa_rz;
        double dsa_ss = (qrz * tt_qrz[k] - cpr * tt_cpr[k] - qss * tt_qss[k]) * h;
        dsa_ss = (sa_ss[k] + dsa_ss < 0) ? -sa_ss[k] : dsa_ss;
        const double dsa_rz1 = (cpr * tt_cpr[k]) * h;
        const double dmsa_rz1 = (isnan(mtt_cpr[k]) ? 0 : mtt_cpr[k]) * (dsa_rz1 > 0 ? ((cpr * tt_cpr[k] * h) / dsa_rz1) : 0);
        const double dsa_ss1 = (qrz * tt_qrz[k]) * h;
        const double dmsa_ss1 = (isnan(mtt_qrz[k]) ? 0 : mtt_qrz[k]) * (dsa_ss1 > 0 ? ((qrz * tt_qrz[k] * h) / dsa_ss1) : 0);
        msa_rz[k] = euler_mix(msa_rz[k], sa_rz[k], dsa_rz1, dmsa_rz1, 1);
        msa_ss[k] = euler_mix(msa_ss[k], sa_ss[k], dsa_ss1, dmsa_ss1, 0);
        sa_rz[k] += dsa_rz;
        sa_ss[k] += dsa_ss;
        msa_rz[k] = (sa_rz[k] <= 0) ? 0 : msa_rz[k];
        msa_ss[k] = (sa_ss[k] <= 0) ? 0 : msa_ss[k];
    }
}
/* concentrations of the fluxes of a sub-step, transport.py:2324-2357, and delta_fluxes_svat :3660-3697 */
static void euler_concentrations(const oc_sas *P, int64_t i, double *work) {
    const int64_t A = P->ages;
    const double mk = (double)P->maskCatch[i];
    const double inf[3] = {P->inf_mat_rz[i], P->inf_pf_rz[i], P->inf_pf_ss[i]};
    for (int w = 0; w < 3; ++w) {
        P->C_inf[w][i] = (inf[w] > 0 ? P->C_in[i] : NAN) * mk;
        P->C_iso_inf[w][i] = conc_to_delta(P, P->C_inf[w][i]) * mk;
    }
    for (int f = 0; f < 5; ++f) {
        P->C[f][i] = conc_iso_flux(P->mtt[f] + i * A, P->tt[f] + i * A, A, work) * mk;
        P->C_iso[f][i] = conc_to_delta(P, P->C[f][i]) * mk;
    }
}
/* svat_transport_model_euler, transport.py:2064-2414 (oxygen-18 / deuterium): one sub-step of length h */
static void euler_substep(const oc_sas *P, int64_t i, double h, double *work) {
    const int64_t NA = P->ages + 1;
    if (P->tracer != 0) {
        euler_inflow_anion(P, i, h);
        euler_distributions(P, i, h, work, work + NA, work + 2 * NA);
        explicit_update_anion(P, i, h, 0);
        explicit_concentrations_anion(P, i, h);
        return;
    }
    euler_inflow(P, i, h);
    euler_distributions(P, i, h, work, work + NA, work + 2 * NA);
    euler_update(P, i, h);
    euler_concentrations(P, i, work);
}

/* svat_transport_model_rk4, transport.py:1139-2047 (oxygen-18 / deuterium): one sub-step of length h with four evaluations of the
 * travel time distributions.  For the isotopes the intermediate signals (msarkn, mttrkn) never reach the result -- the final mtt is
 * calc_mtt on the state after the infiltration (:1878-1896) -- so only the intermediate StorAges are followed:
 *   stage 1 on the state, fluxes * h;          the trial StorAge moves by (net outflow) * h      (:1388-1466, isotope branch)
 *   stage 2 on that,      fluxes * h / 2;      it moves by (net outflow) * h / 2                 (:1552-1580)
 *   stage 3 on that,      fluxes * h / 2;      it moves by (net outflow) * h / 2, limited by `sarkn - dsarkn < 0` as written (:1700-1728)
 *   stage 4 on that,      fluxes * h
 *   tt = (tt1 + 2 tt2 + 2 tt3 + tt4) / 6 (:1835-1855), then the update of Euler's scheme with it (:1898-1941).
 * work: 3 * NA (euler_tt) + 2 * NA (SA) + 2 * A (trial StorAges) + 5 * A (the stage's tt) doubles. */
static void rk4_substep(const oc_sas *P, int64_t i, double h, double *work) {
    const int64_t A = P->ages, NA = A + 1;
    const double mk = (double)P->maskCatch[i];
    double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
    const double flux[5] = {P->evap_soil[i], P->transp[i], P->q_rz[i], P->q_ss[i], P->cpr_rz[i]};
    double *tw = work, *SA_rz = work + 3 * NA, *SA_ss = SA_rz + NA, *s_rz = SA_ss + NA, *s_ss = s_rz + A, *tts = s_ss + A;
    double *TT = P->TT[0] + i * NA; /* scratch for the stages' cumulative distributions (overwritten at the end) */
    const int anion = P->tracer != 0;
    if (anion) euler_inflow_anion(P, i, h);
    else euler_inflow(P, i, h);
    calc_SA(SA_rz, sa_rz, A);
    calc_SA(SA_ss, sa_ss, A);
    for (int64_t k = 0; k < NA; ++k) {
        SA_rz[k] *= mk;
        SA_ss[k] *= mk;
    }
    memcpy(s_rz, sa_rz, sizeof(double) * A);
    memcpy(s_ss, sa_ss, sizeof(double) * A);
    for (int f = 0; f < 5; ++f)
        for (int64_t k = 0; k < A; ++k) P->tt[f][i * A + k] = 0;
    for (int stage = 0; stage < 4; ++stage) {
        const int half = (stage == 1 || stage == 2);
        const double w = half ? 2.0 : 1.0;
        for (int f = 0; f < 5; ++f) {
            double *t = tts + f * A;
            const double fh = half ? flux[f] * h / 2 : flux[f] * h;
            euler_tt(P, TT, t, EULER_SRC_SS[f] ? SA_ss : SA_rz, P->sas_params[f] + i * 8, fh, mk, tw);
            double *acc = P->tt[f] + i * A;
            /* (tt1 + 2 * tt2 + 2 * tt3 + tt4): left to right */
            for (int64_t k = 0; k < A; ++k) acc[k] = (stage == 0) ? t[k] : acc[k] + w * t[k];
        }
        if (stage == 3) break;
        /* the anion kernels move the trial StorAge by h / 2 after the first AND the second evaluation and not at all after the third
         * (:1432-1446, 1581-1595, 1729-1751: only the trial solute, which never reaches the result, changes there) */
        if (anion && stage == 2) continue;
        const double *t_ev = tts, *t_tr = tts + A, *t_qrz = tts + 2 * A, *t_qss = tts + 3 * A, *t_cpr = tts + 4 * A;
        for (int64_t k = 0; k < A; ++k) {
            double d_rz = (flux[4] * t_cpr[k] - flux[0] * t_ev[k] - flux[1] * t_tr[k] - flux[2] * t_qrz[k]) * h;
            double d_ss = (flux[2] * t_qrz[k] - flux[4] * t_cpr[k] - flux[3] * t_qss[k]) * h;
            if (stage > 0 || anion) {
                d_rz = d_rz / 2;
                d_ss = d_ss / 2;
            }
            if (stage == 2) { /* as written: `sarkn - dsarkn < 0` */
                d_rz = (s_rz[k] - d_rz < 0) ? -s_rz[k] : d_rz;
                d_ss = (s_ss[k] - d_ss < 0) ? -s_ss[k] : d_ss;
            } else {
                d_rz = (s_rz[k] + d_rz < 0) ? -s_rz[k] : d_rz;
                d_ss = (s_ss[k] + d_ss < 0) ? -s_ss[k] : d_ss;
            }
            s_rz[k] += d_rz;
            s_ss[k] += d_ss;
        }
        calc_SA(SA_rz, s_rz, A); /* SArkn[1:] = cumsum(sarkn): no mask from here on */
        calc_SA(SA_ss, s_ss, A);
    }
    for (int f = 0; f < 5; ++f) {
        double *tt = P->tt[f] + i * A, *mtt = P->mtt[f] + i * A, *TTf = P->TT[f] + i * NA;
        const double *msa = EULER_SRC_SS[f] ? msa_ss : msa_rz;
        for (int64_t k = 0; k < A; ++k) tt[k] = tt[k] / 6.;
        TTf[0] = 0;
        double acc = 0;
        for (int64_t k = 0; k < A; ++k) {
            acc = (k == 0) ? tt[0] : acc + tt[k];
            TTf[k + 1] = acc;
        }
        if (!anion)
            for (int64_t k = 0; k < A; ++k) mtt[k] = (tt[k] > 0) ? msa[k] : 0; /* calc_mtt :565-596, isotopes */
    }
    if (anion) {
        anion_mtt_fluxes(P, i, h);   /* :1881-1896: on the state after the infiltration */
        explicit_update_anion(P, i, h, 1);
        explicit_concentrations_anion(P, i, h);
        return;
    }
    euler_update(P, i, h);
    euler_concentrations(P, i, work);
}

/* one day of svat_transport_model_deterministic for all columns; columns are independent and run on all host threads
 * once there are enough of them (bench.py's cpu_baseline) */
void oc_sas_step(const oc_sas *P) {
    const int64_t A = P->ages, NA = A + 1;
#pragma omp parallel if (P->n >= 256)
    {
    double *work = (double *)malloc(sizeof(double) * (12 * NA + 4));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < P->n; ++i) {
        const double mk = (double)P->maskCatch[i];
        double *sa_rz = P->sa_rz + i * A, *msa_rz = P->msa_rz + i * A, *sa_ss = P->sa_ss + i * A, *msa_ss = P->msa_ss + i * A;
        double *sa_s = P->sa_s + i * A;
        if (P->tracer != 0 && P->solver == 0) {
            step_anion(P, i, work);
            goto statistics;
        }
        if (P->solver == 1 || P->solver == 2) {   /* calculate_storage_selection, Euler / RK4 branches :3220-3304: per sub-step the model, the
                                 * storages, [the age statistics: they depend on the state of the moment only, the last ones remain] */
            const double h = 1 / (double)P->substeps;
            for (int64_t it = 0; it < P->substeps; ++it) {
                if (P->solver == 1) euler_substep(P, i, h, work);
                else rk4_substep(P, i, h, work);
                if (P->tracer != 0) storages_anion(P, i, work);
                else storages_iso(P, i, work);
            }
            goto statistics;
        }
        inflow(P, i, 0, P->inf_mat_rz[i], sa_rz, msa_rz, mk);
        inflow(P, i, 1, P->inf_pf_rz[i], sa_rz, msa_rz, mk);
        outflux(P, i, 0, P->evap_soil[i], sa_rz, msa_rz, NULL, NULL, mk, work);
        outflux(P, i, 1, P->transp[i], sa_rz, msa_rz, NULL, NULL, mk, work);
        outflux(P, i, 2, P->q_rz[i], sa_rz, msa_rz, sa_ss, msa_ss, mk, work);
        inflow(P, i, 2, P->inf_pf_ss[i], sa_ss, msa_ss, mk);
        outflux(P, i, 3, P->q_ss[i], sa_ss, msa_ss, NULL, NULL, mk, work);
        outflux(P, i, 4, P->cpr_rz[i], sa_ss, msa_ss, sa_rz, msa_rz, mk, work);
        storages_iso(P, i, work);
    statistics:
        if (P->stats[0][0]) { /* transport.py:59-312 */
            static const double Q[5] = {0.1, 0.25, 0.5, 0.75, 0.9};
            const double *ttd[2] = {P->tt[1] + i * A, P->tt[3] + i * A};
            const double *TTd[2] = {P->TT[1] + i * NA + 1, P->TT[3] + i * NA + 1};
            for (int d = 0; d < 2; ++d) {
                for (int q = 0; q < 5; ++q) P->stats[d][q][i] = age_percentile(TTd[d], A, Q[q]);
                for (int64_t k = 0; k < A; ++k) work[k] = (double)(k + 1) * ttd[d][k];
                P->stats[d][5][i] = (np_sum(ttd[d], A) > 0 ? np_sum(work, A) : NAN);
            }
            /* residence time distributions of the storages: RT = SA / max(SA), :155-312 */
            const double *store[3] = {sa_rz, sa_ss, sa_s};
            for (int d = 0; d < 3; ++d) {
                double *RT = work, *rt = work + NA, *prod = rt + A;
                calc_SA(RT, store[d], A);
                for (int64_t k = 0; k < NA; ++k) RT[k] *= mk;
                const double mx = np_max(RT, NA);
                for (int64_t k = 0; k < NA; ++k) RT[k] = (mx > 0 ? RT[k] / mx : 0);
                for (int64_t k = 0; k < A; ++k) rt[k] = RT[k + 1] - RT[k];
                for (int q = 0; q < 5; ++q) {
                    /* the reference never assigns rt10/rt90 of root zone and subsoil (:181-196, :232-247) */
                    if (d < 2 && (q == 0 || q == 4)) continue;
                    P->stats[2 + d][q][i] = age_percentile(RT + 1, A, Q[q]);
                }
                for (int64_t k = 0; k < A; ++k) prod[k] = (double)(k + 1) * rt[k];
                P->stats[2 + d][5][i] = (np_sum(rt, A) > 0 ? np_sum(prod, A) : NAN);
            }
        }
        if (P->tracer != 0) {
            ageing_anion(sa_rz, msa_rz, A);
            ageing_anion(sa_ss, msa_ss, A);
            continue;
        }
        ageing(sa_rz, msa_rz, A, work);
        ageing(sa_ss, msa_ss, A, work);
    }
    free(work);
    }
}
