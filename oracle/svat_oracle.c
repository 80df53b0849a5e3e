/*
 * oracle/svat_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar, per-cell CPU restatement of the reference NumPy backend's SVAT time
 * step (RoGeR, /root/reference/roger/core/<module>.py), written from the reference's
 * array expressions with the same operation order, masks and quirks.  It is the
 * checker for the HIP kernels in roger_amd/csrc and the `cpu_baseline` ("port")
 * of bench.py.  Nothing in the product path includes, links or calls this file.
 *
 * Parity pin: tests/golden/<case>.npz hold trajectories produced by the reference
 * NumPy backend itself (tests/golden/make_golden.py); tests/test_oracle_golden.py
 * replays them through this file.
 *
 * NumPy-backend semantics restated here (roger/routines.py:307-380): inside a
 * @roger_kernel every `vs.X = ...` assignment persists, whether or not X is in
 * the returned KernelOutput.
 *
 * Conventions: `mk` = maskCatch as 0.0/1.0; bool*float products are written as
 * products (NaN*0 stays NaN, as in NumPy); npx.where -> ?: ; no fmin/fmax
 * except where the reference calls npx.fmin.
 */
#include "svat_cell.h"

#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#define OC_PAR_MIN 8192 /* cells from which the column loops fork (OpenMP) */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define B(c) ((double)((c) ? 1 : 0))

/* ------------------------------------------------------------------------ */
/* field table: name, kind, offset into oc_cell, plane index                  */
/* ------------------------------------------------------------------------ */
typedef struct {
    const char *name;
    int is_int;
    size_t off;
} oc_field_desc;

#define OC_T_F(n) {#n, 0, offsetof(oc_cell, n)},
#define OC_T_F2(n) {#n, 0, offsetof(oc_cell, n)}, {#n "_m1", 0, offsetof(oc_cell, n##_m1)},
#define OC_T_I(n) {#n, 1, offsetof(oc_cell, n)},
#define OC_T(kind, n) OC_T_##kind(n)
static const oc_field_desc OC_TABLE[] = {OC_FIELDS(OC_T)};
#define OC_NPLANES ((int)(sizeof(OC_TABLE) / sizeof(OC_TABLE[0])))

int oc_nplanes(void) { return OC_NPLANES; }
const char *oc_plane_name(int i) { return OC_TABLE[i].name; }
int oc_plane_is_int(int i) { return OC_TABLE[i].is_int; }

static void gather(oc_cell *c, void *const *planes, int64_t i) {
    for (int p = 0; p < OC_NPLANES; ++p) {
        char *dst = (char *)c + OC_TABLE[p].off;
        if (OC_TABLE[p].is_int)
            *(int32_t *)dst = ((const int32_t *)planes[p])[i];
        else
            *(double *)dst = ((const double *)planes[p])[i];
    }
}
static void scatter(const oc_cell *c, void *const *planes, int64_t i) {
    for (int p = 0; p < OC_NPLANES; ++p) {
        const char *src = (const char *)c + OC_TABLE[p].off;
        if (OC_TABLE[p].is_int)
            ((int32_t *)planes[p])[i] = *(const int32_t *)src;
        else
            ((double *)planes[p])[i] = *(const double *)src;
    }
}

/* ------------------------------------------------------------------------ */
/* numpy add.reduce over a contiguous axis (pairwise summation),             */
/* numpy/_core/src/umath/loops_utils.h.src: @TYPE@_pairwise_sum              */
/* ------------------------------------------------------------------------ */
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
double oc_np_sum(const double *a, int64_t n) { return 0.0 + np_pairwise(a, n); }

/* ======================================================================== */
/* a1  adaptive time stepping: roger/core/adaptive_time_stepping.py:22-437   */
/* ======================================================================== */
/* forcing of the current day: prec_day/ta_day/pet_day, each (n, 144) when
 * fstride==144 or one shared (144) vector when fstride==0
 * (benchmarks/SVAT_benchmark.py:162-170 broadcasts one station series).
 *
 * The routine is written in three phases so that a domain-decomposed test can combine the
 * global predicates of several ranks (bitwise OR of the returned words) between them, the way
 * the reference's MPI variant combines them by gathering to rank 0
 * (adaptive_time_stepping_dist_safe.py).  oc_adaptive_dt runs the three back to back. */
enum { W0_SWE_NOT_LE0 = 0, W0_SWE_GT0, W0_SWETOP_NOT_LE0, W0_SWETOP_GT0, W0_P_NOT_LE0, W0_P_GT0, W0_P_GTHPI,
       W0_P_NOT_LEHPI, W0_TA_NOT_GT, W0_TA_GT, W0_PGT0_TALE, W0_NOT_PLE0_TALE };
enum { W1_EV1A = 0, W1_EV1B, W1_PREC_NOT_LE0, W1_NOT_PGT0_TALE, W1_SWEM1_GT0, W1_SWE_NOT_LE0, W1_P_EQ0, W1_PM1_NE0,
       W1_P_NE0, W1_PM1_EQ0 };
#define WB(b) (1ull << (b))
#define HAS(w, b) (((w) >> (b)) & 1ull)

static int plane_index(const char *name) {
    for (int p = 0; p < OC_NPLANES; ++p)
        if (!strcmp(OC_TABLE[p].name, name)) return p;
    return -1;
}

typedef struct {
    int sel_daily, sel_hourly, sel_10min;
    int64_t dt_secs;
} adt_sel;

/* phase 1: predicates of lines 38-81 over the local cells and the forcing */
uint64_t oc_adt_pred1(void *const *planes, int64_t n, const double *prec_day, const double *ta_day, int64_t fstride,
                      const oc_settings *st) {
    const double *swe = planes[plane_index("swe")], *swe_top = planes[plane_index("swe_top")];
    uint64_t w = 0;
    int64_t nf = fstride ? n : 1;
    for (int64_t i = 0; i < nf; ++i) {
        const double *pd = prec_day + i * fstride, *td = ta_day + i * fstride;
        for (int k = 0; k < 144; ++k) {
            double p = pd[k], t = td[k];
            if (!(p <= 0)) w |= WB(W0_P_NOT_LE0);
            if (p > 0) w |= WB(W0_P_GT0);
            if (p > (double)st->hpi) w |= WB(W0_P_GTHPI);
            if (!(p <= (double)st->hpi)) w |= WB(W0_P_NOT_LEHPI);
            if (!(t > st->ta_fm)) w |= WB(W0_TA_NOT_GT);
            if (t > st->ta_fm) w |= WB(W0_TA_GT);
            if ((p > 0) && (t <= st->ta_fm)) w |= WB(W0_PGT0_TALE);
            if (!((p <= 0) && (t <= st->ta_fm))) w |= WB(W0_NOT_PLE0_TALE);
        }
    }
#pragma omp parallel for schedule(static) reduction(| : w) if (n >= OC_PAR_MIN)
    for (int64_t i = 0; i < n; ++i) {
        if (!(swe[i] <= 0)) w |= WB(W0_SWE_NOT_LE0);
        if (swe[i] > 0) w |= WB(W0_SWE_GT0);
        if (!(swe_top[i] <= 0)) w |= WB(W0_SWETOP_NOT_LE0);
        if (swe_top[i] > 0) w |= WB(W0_SWETOP_GT0);
    }
    return w;
}

static adt_sel adt_select_flags(uint64_t w, const oc_scalars *s) {
    int all_p_le0 = !HAS(w, W0_P_NOT_LE0), any_p_gt0 = HAS(w, W0_P_GT0), any_p_gthpi = HAS(w, W0_P_GTHPI);
    int all_p_lehpi = !HAS(w, W0_P_NOT_LEHPI), all_ta_gt = !HAS(w, W0_TA_NOT_GT), any_ta_gt = HAS(w, W0_TA_GT);
    int any_pgt0_tale = HAS(w, W0_PGT0_TALE), all_ple0_tale = !HAS(w, W0_NOT_PLE0_TALE);
    int all_swe_le0 = !HAS(w, W0_SWE_NOT_LE0), all_swetop_le0 = !HAS(w, W0_SWETOP_NOT_LE0);
    int snow_any = (HAS(w, W0_SWE_GT0) || HAS(w, W0_SWETOP_GT0)) && any_ta_gt;
    int cond0 = all_p_le0 && all_swe_le0 && all_swetop_le0 && all_ta_gt;
    int cond00 = any_pgt0_tale || all_ple0_tale;
    int cond1 = any_p_gthpi && any_p_gt0 && any_ta_gt;
    int cond2 = all_p_lehpi && any_p_gt0 && any_ta_gt;
    int cond3 = any_p_gthpi && any_p_gt0 && snow_any;
    int cond4 = all_p_lehpi && any_p_gt0 && snow_any;
    int cond5 = all_p_le0 && snow_any;
    int cond_time = (s->time % (24 * 60 * 60) == 0);
    adt_sel r;
    r.sel_daily = cond0 || cond00;
    r.sel_hourly = (cond2 || cond4 || cond5) && !cond1 && !cond3;
    r.sel_10min = (cond1 || cond3) && !cond2 && !cond4 && !cond5;
    /* dt_secs, lines 143-144, 166, 190 (line 144 overwrites line 143 unconditionally) */
    r.dt_secs = cond_time ? 24 * 60 * 60 : 60 * 60;
    if (r.sel_hourly) r.dt_secs = 60 * 60;
    if (r.sel_10min) r.dt_secs = 10 * 60;
    return r;
}

/* which prec/ta selection wins for predicate word 0 (the last one applied, lines 128-189): -1 none, 0 daily,
 * 1 hourly, 2 ten minutes -- for test doubles that derive word 1 from summary bits */
int oc_adt_sel_p(uint64_t word0, const oc_scalars *s) {
    adt_sel f = adt_select_flags(word0, s);
    return f.sel_10min ? 2 : (f.sel_hourly ? 1 : (f.sel_daily ? 0 : -1));
}

/* per-cell aggregates {prec, ta, pet} x {daily, hourly, 10 min}, lines 384-437 */
static double *adt_aggregates(int64_t n, const double *prec_day, const double *ta_day, const double *pet_day,
                              int64_t fstride, int64_t itd) {
    int64_t nf = fstride ? n : 1;
    double *buf = (double *)malloc(sizeof(double) * 144 * 3);
    double *agg = (double *)malloc(sizeof(double) * 9 * (size_t)nf);
    for (int64_t i = 0; i < nf; ++i) {
        const double *pd = prec_day + i * fstride, *td = ta_day + i * fstride, *ed = pet_day + i * fstride;
        double *a = agg + 9 * i;
        a[0] = oc_np_sum(pd, 144);
        {
            int cnt = 0;
            for (int k = 0; k < 144; ++k) {
                buf[k] = isnan(td[k]) ? 0.0 : td[k];
                cnt += !isnan(td[k]);
            }
            a[1] = oc_np_sum(buf, 144) / (double)cnt;
        }
        a[2] = oc_np_sum(ed, 144);
        {
            int cnt = 0;
            for (int k = 0; k < 144; ++k) {
                int in = (k >= itd) && (k < itd + 6);
                buf[k] = in ? pd[k] : 0.0;
                double tv = in ? td[k] : NAN;
                buf[144 + k] = isnan(tv) ? 0.0 : tv;
                cnt += !isnan(tv);
                buf[288 + k] = in ? ed[k] : 0.0;
            }
            a[3] = oc_np_sum(buf, 144);
            a[4] = oc_np_sum(buf + 144, 144) / (double)cnt;
            a[5] = oc_np_sum(buf + 288, 144);
        }
        {
            int64_t k = itd;
            if (k < 0) k += 144;
            if (k > 143) k = 143; /* out-of-range would raise in the reference */
            a[6] = pd[k];
            a[7] = td[k];
            a[8] = ed[k];
        }
    }
    free(buf);
    return agg;
}

/* phase 2: prec/ta selection (lines 128-189) and the predicates of lines 192-201 and of
 * calculate_infiltration (infiltration.py:2155-2167) over the local cells */
uint64_t oc_adt_select(void *const *planes, int64_t n, const double *prec_day, const double *ta_day,
                       const double *pet_day, int64_t fstride, const oc_scalars *s, const oc_settings *st,
                       uint64_t word0) {
    adt_sel f = adt_select_flags(word0, s);
    double *prec = planes[plane_index("prec")], *ta = planes[plane_index("ta")];
    const double *prec_m1 = planes[plane_index("prec_m1")];
    const double *swe = planes[plane_index("swe")], *swe_m1 = planes[plane_index("swe_m1")];
    const double *swe_top = planes[plane_index("swe_top")];
    double *agg = adt_aggregates(n, prec_day, ta_day, pet_day, fstride, s->itt_day);
    uint64_t w = 0;
#pragma omp parallel for schedule(static) reduction(| : w) if (n >= OC_PAR_MIN)
    for (int64_t i = 0; i < n; ++i) {
        const double *a = agg + 9 * (fstride ? i : 0);
        if (f.sel_daily) { prec[i] = a[0]; ta[i] = a[1]; }
        if (f.sel_hourly) { prec[i] = a[3]; ta[i] = a[4]; }
        if (f.sel_10min) { prec[i] = a[6]; ta[i] = a[7]; }
        if ((prec[i] > 0) && (ta[i] > st->ta_fm)) w |= WB(W1_EV1A);
        if (((swe[i] > 0) || (swe_top[i] > 0)) && (ta[i] > st->ta_fm)) w |= WB(W1_EV1B);
        if (!(prec[i] <= 0)) w |= WB(W1_PREC_NOT_LE0);
        if (!((prec[i] > 0) && (ta[i] <= st->ta_fm))) w |= WB(W1_NOT_PGT0_TALE);
        if (swe_m1[i] > 0) w |= WB(W1_SWEM1_GT0);
        if (!(swe[i] <= 0)) w |= WB(W1_SWE_NOT_LE0);
        if (prec[i] == 0) w |= WB(W1_P_EQ0);
        if (prec_m1[i] != 0) w |= WB(W1_PM1_NE0);
        if (prec[i] != 0) w |= WB(W1_P_NE0);
        if (prec_m1[i] == 0) w |= WB(W1_PM1_EQ0);
    }
    free(agg);
    return w;
}

/* phase 3: scalar bookkeeping (lines 192-373) and the pet/ta selection (lines 262-376) */
void oc_adt_finish(void *const *planes, int64_t n, const double *prec_day, const double *ta_day, const double *pet_day,
                   int64_t fstride, oc_scalars *s, const oc_settings *st, uint64_t word0, uint64_t word1) {
    adt_sel f = adt_select_flags(word0, s);
    int64_t dt_secs = f.dt_secs;
    double *ta = planes[plane_index("ta")], *pet = planes[plane_index("pet")], *pet_res = planes[plane_index("pet_res")];
    double *agg = adt_aggregates(n, prec_day, ta_day, pet_day, fstride, s->itt_day);
    int cond_event1 = HAS(word1, W1_EV1A) || HAS(word1, W1_EV1B);
    int cond_event2 = !HAS(word1, W1_PREC_NOT_LE0) || !HAS(word1, W1_NOT_PGT0_TALE) ||
                      (HAS(word1, W1_SWEM1_GT0) && !HAS(word1, W1_SWE_NOT_LE0));
    if (cond_event1) s->time_event0 = 0;
    if (cond_event2) s->time_event0 = s->time_event0 + dt_secs;

    /* lines 209-222 */
    int64_t te0 = s->time_event0, tm = s->time, ee = st->end_event;
    int cond6 = (te0 <= ee) && (dt_secs == 600);
    int cond7 = (te0 <= ee) && (dt_secs == 3600);
    int cond8 = (te0 <= ee) && (dt_secs == 86400);
    int cond9 = (te0 > ee) && (tm % 3600 != 0) && (dt_secs == 600);
    int cond10 = (te0 > ee) && (tm % 3600 == 0) && ((dt_secs == 600) || (dt_secs == 3600));
    int cond11 = (te0 > ee) && (tm % 86400 == 0) && (dt_secs == 86400);

    /* lines 262-368, applied in order; per-cell pet/ta selection */
    int sel[6] = {cond6, cond7, cond8, cond9, cond10, cond11};
    int which[6] = {2, 1, 0, 2, 1, 0}; /* 0 daily, 1 hourly, 2 10min */
#pragma omp parallel for schedule(static) if (n >= OC_PAR_MIN)
    for (int64_t i = 0; i < n; ++i) {
        const double *a = agg + 9 * (fstride ? i : 0);
        for (int q = 0; q < 6; ++q)
            if (sel[q]) {
                int w = which[q];
                pet[i] = a[3 * w + 2];
                ta[i] = a[3 * w + 1];
            }
    }
    /* scalar bookkeeping in the reference order; conditions were frozen above */
    double dt = s->dt;
    int64_t itt_day = s->itt_day;
    if (cond6) { s->event_id[1] = s->event_id_counter; dt = 1.0 / 6; itt_day = itt_day + 1; }
    if (cond7) { s->event_id[1] = s->event_id_counter; dt = 1; itt_day = itt_day + 6; }
    if (cond8) { dt = 24; itt_day = 0; }
    if (cond9) { s->event_id[1] = 0; dt = 1.0 / 6; dt_secs = 600; itt_day = itt_day + 1; }
    if (cond10) { s->event_id[1] = 0; dt = 1; dt_secs = 3600; itt_day = itt_day + 6; }
    if (cond11) { s->event_id[1] = 0; dt = 24; dt_secs = 86400; itt_day = 0; }
    s->dt = dt;
    s->dt_secs = dt_secs;
    s->itt_day = itt_day;
    /* lines 371-373 */
    if ((s->event_id[0] > 0) && (s->event_id[1] == 0)) s->event_id_counter += 1;
    /* line 376 */
#pragma omp parallel for schedule(static) if (n >= OC_PAR_MIN)
    for (int64_t i = 0; i < n; ++i) pet_res[i] = pet[i];
    free(agg);
}

void oc_adaptive_dt(void *const *planes, int64_t n, const double *prec_day, const double *ta_day,
                    const double *pet_day, int64_t fstride, oc_scalars *s, const oc_settings *st) {
    uint64_t w0 = oc_adt_pred1(planes, n, prec_day, ta_day, fstride, st);
    uint64_t w1 = oc_adt_select(planes, n, prec_day, ta_day, pet_day, fstride, s, st, w0);
    oc_adt_finish(planes, n, prec_day, ta_day, pet_day, fstride, s, st, w0, w1);
}

/* ======================================================================== */
/* a2  interception: roger/core/interception.py                              */
/* ======================================================================== */
static double swe_top_tot_of(double swe_top_tot, double ta, int lu, double mk) {
    /* interception.py:170-206 and surface.py:243-308 (same nine updates) */
    double v = swe_top_tot;
    v = ((ta > -1) && (lu == 10) ? 9 : v) * mk;
    v = ((ta > -1) && (lu == 11) ? 15 : v) * mk;
    v = ((ta > -1) && (lu == 12) ? 25 : v) * mk;
    v = ((ta >= -3) && (ta <= -1) && (lu == 10) ? 2.5 + 0.5 * ta * 9 : v) * mk;
    v = ((ta >= -3) && (ta <= -1) && (lu == 11) ? 2.5 + 0.5 * ta * 15 : v) * mk;
    v = ((ta >= -3) && (ta <= -1) && (lu == 12) ? 2.5 + 0.5 * ta * 25 : v) * mk;
    v = ((ta < -3) && (lu == 10) ? 18 : v) * mk;
    v = ((ta < -3) && (lu == 11) ? 30 : v) * mk;
    v = ((ta < -3) && (lu == 12) ? 50 : v) * mk;
    return v;
}

static void interception_cell(oc_cell *c, const oc_settings *st) {
    const double mk = (double)c->maskCatch;
    /* calc_rain_int_top, interception.py:7-71 */
    {
        int mask_rain = c->ta > st->ta_fm;
        c->rain_top = (mask_rain ? c->prec : 0) * mk;
        double wtmx = (10000. / (100 - st->rmax) / 100.) * c->swe_top;
        double S_tot = (c->S_int_top_tot < wtmx ? wtmx : c->S_int_top_tot) * mk;
        double free_ = (c->S_int_top < S_tot ? S_tot - c->S_int_top : 0) * mk;
        double need = c->prec * (1. - c->throughfall_coeff_top);
        int mask1 = (free_ >= need) && (c->ta > st->ta_fm) && (free_ > 0);
        int mask2 = (free_ < need) && (c->ta > st->ta_fm) && (free_ > 0);
        c->int_rain_top = 0;
        c->int_rain_top += c->prec * (1. - c->throughfall_coeff_top) * B(mask1) * mk;
        c->int_rain_top = (mask2 ? free_ : c->int_rain_top) * mk;
        c->S_int_top += c->int_rain_top * mk;
    }
    /* calc_rain_int_ground, interception.py:75-151 */
    {
        int mask_rain = c->ta > st->ta_fm;
        double rain = (c->prec - c->int_rain_top) * B(mask_rain) * mk;
        double free_ =
            ((c->S_int_ground < c->S_int_ground_tot) && (c->S_snow <= 0) ? c->S_int_ground_tot - c->S_int_ground : 0) * mk;
        double need = rain * (1. - c->throughfall_coeff_ground);
        int mask1 = (free_ >= need) && (c->ta > st->ta_fm) && (free_ > 0);
        int mask2 = (free_ < need) && (c->ta > st->ta_fm) && (free_ > 0);
        c->int_rain_ground = 0;
        c->int_rain_ground += rain * (1. - c->throughfall_coeff_ground) * B(mask1) * mk;
        c->int_rain_ground = (mask2 ? free_ : c->int_rain_ground) * mk;
        c->int_rain_ground = (c->lu_id == 599 ? 0 : c->int_rain_ground) * mk;
        c->S_int_ground += c->int_rain_ground * mk;
        c->rain_ground = (c->rain_top - c->int_rain_top - c->int_rain_ground) * mk;
        c->z0 += (c->S_snow > 0 ? 0 : c->rain_ground) * mk;
        c->prec_event_csum += (c->S_snow > 0 ? 0 : c->rain_ground) * mk;
    }
    /* calc_snow_int_top, interception.py:155-245 */
    {
        int mask_snow = c->ta <= st->ta_fm;
        c->snow_top = (mask_snow ? c->prec : 0) * mk;
        c->swe_top_tot = swe_top_tot_of(c->swe_top_tot, c->ta, c->lu_id, mk);
        double free_ = (c->swe_top >= c->swe_top_tot ? 0 : c->swe_top_tot - c->swe_top) * mk;
        double need = c->prec * (1. - c->throughfall_coeff_top);
        int mask1 = (free_ >= need) && (c->ta <= st->ta_fm) && (free_ > 0);
        int mask2 = (free_ < need) && (c->ta <= st->ta_fm) && (free_ > 0);
        c->int_snow_top = 0;
        c->int_snow_top += c->prec * (1. - c->throughfall_coeff_top) * B(mask1) * mk;
        c->int_snow_top = (mask2 ? free_ : c->int_snow_top) * mk;
        c->S_int_top += c->int_snow_top * mk;
        c->swe_top += c->int_snow_top * mk;
    }
    /* calc_snow_int_ground, interception.py:249-318 */
    {
        int mask_snow = c->ta <= st->ta_fm;
        double snow = (c->prec - c->int_snow_top) * B(mask_snow) * mk;
        double free_ = (c->S_int_ground >= c->S_int_ground_tot ? 0 : c->S_int_ground_tot - c->S_int_ground) * mk;
        double need = snow * (1. - c->throughfall_coeff_ground);
        int mask1 = (free_ >= need) && (c->ta <= st->ta_fm) && (free_ > 0);
        int mask2 = (free_ < need) && (c->ta <= st->ta_fm) && (free_ > 0);
        c->int_snow_ground = 0;
        c->int_snow_ground += snow * (1. - c->throughfall_coeff_ground) * B(mask1) * mk;
        c->int_snow_ground = (mask2 ? free_ : c->int_snow_ground) * mk;
        c->int_snow_ground = (c->lu_id == 599 ? 0 : c->int_snow_ground) * mk;
        c->S_int_ground += c->int_snow_ground * mk;
        c->swe_ground += c->int_snow_ground * mk;
        c->snow_ground = (c->snow_top - c->int_snow_top - c->int_snow_ground) * mk;
        c->prec_event_csum += c->snow_ground * mk;
    }
    /* calc_int, interception.py:322-343 */
    c->int_top = (c->int_rain_top + c->int_snow_top) * mk;
    c->int_ground = (c->int_rain_ground + c->int_snow_ground) * mk;
    c->int_prec = (c->int_rain_top + c->int_rain_ground + c->int_snow_top + c->int_snow_ground) * mk;
}

/* ======================================================================== */
/* a3  evapotranspiration: roger/core/evapotranspiration.py:9-616            */
/* ======================================================================== */
static int is_tree_lu(int lu) { return lu == 10 || lu == 11 || lu == 12 || lu == 15 || lu == 16 || lu == 17; }

static void evapotranspiration_cell(oc_cell *c, const oc_settings *st) {
    const double mk = (double)c->maskCatch;
    /* calc_evap_int_top :10-66 */
    {
        int base = (c->S_int_top <= c->S_int_top_tot) && (c->S_int_top_tot > 0) && (c->S_int_top > 0);
        int mask1 = base && (c->pet_res <= c->S_int_top);
        int mask2 = base && (c->pet_res > c->S_int_top);
        c->evap_int_top = 0;
        c->evap_int_top += c->pet_res * B(mask1) * mk;
        c->pet_res = (mask1 ? 0 : c->pet_res) * mk;
        c->evap_int_top += c->S_int_top * B(mask2) * mk;
        c->pet_res += -c->S_int_top * B(mask2) * mk;
        c->S_int_top += -c->evap_int_top * mk;
    }
    /* calc_evap_int_ground :70-134 */
    {
        int base = (c->S_int_ground <= c->S_int_ground_tot) && (c->S_int_ground_tot > 0) && (c->S_int_ground > 0);
        int mask1 = base && (c->pet_res <= c->S_int_ground);
        int mask2 = base && (c->pet_res > c->S_int_ground);
        c->evap_int_ground = 0;
        c->evap_int_ground += c->pet_res * B(mask1) * mk;
        c->pet_res = (mask1 ? 0 : c->pet_res) * mk;
        c->evap_int_ground += c->S_int_ground * B(mask2) * mk;
        c->pet_res += -c->S_int_ground * B(mask2) * mk;
        c->S_int_ground += -c->evap_int_ground * mk;
        c->evap_int = c->evap_int_ground + c->evap_int_top * mk; /* precedence as in :126-130 */
    }
    /* calc_evap_dep :138-194 */
    {
        int base = (c->S_dep > 0) && (c->pet_res > 0) && (c->prec <= 0);
        int mask1 = base && (c->S_dep <= c->pet_res);
        int mask2 = base && (c->S_dep > c->pet_res);
        c->evap_dep = 0;
        c->evap_dep += c->S_dep * B(mask1) * mk;
        c->pet_res += -c->S_dep * B(mask1) * mk;
        c->evap_dep += c->pet_res * B(mask2) * mk;
        c->pet_res = (mask2 ? 0 : c->pet_res) * mk;
        int mask3 = (c->S_dep > 0) && (c->evap_dep > 0);
        c->S_dep += -c->evap_dep * B(mask3) * mk;
    }
    /* calc_evap_sur :198-212 */
    c->evap_sur = c->evap_int_top + c->evap_int_ground + c->evap_dep * mk;
    /* calc_evap_soil :216-345 */
    {
        int mask3 = c->de <= c->rew;
        int mask4 = (c->de > c->rew) && (c->de <= c->tew);
        int mask5 = c->de > c->tew;
        c->k_stress_evap = (mask3 ? 1 : c->k_stress_evap) * mk;
        c->k_stress_evap = (mask4 ? (c->tew - c->de) / (c->tew - c->rew) : c->k_stress_evap) * mk;
        c->k_stress_evap = (mask5 ? 0 : c->k_stress_evap) * mk;
        c->evap_coeff = c->basal_evap_coeff * c->k_stress_evap * mk;
        double pevap = c->pet_res * c->evap_coeff * mk;
        c->pevap_soil = c->pet_res * c->evap_coeff * mk;
        int base = (c->S_fp_rz > 0) && (pevap > 0) && (c->swe <= 0) && (c->prec <= 0);
        int mask1 = base && (pevap <= c->S_fp_rz);
        int mask2 = base && (pevap > c->S_fp_rz);
        double evap_fp = 0;
        evap_fp += pevap * B(mask1) * mk;
        c->pet_res += -pevap * B(mask1) * mk;
        c->pet_res = (c->pet_res < 0 ? 0 : c->pet_res) * mk;
        evap_fp += c->S_fp_rz * B(mask2) * mk;
        c->pet_res += -c->S_fp_rz * B(mask2) * mk;
        c->pet_res = (c->pet_res < 0 ? 0 : c->pet_res) * mk;
        c->evap_soil = evap_fp * mk;
        c->S_fp_rz += -c->evap_soil * mk;
    }
    /* calc_transp :349-543 */
    {
        double theta_ws = st->transp_water_stress * c->theta_ufc + c->theta_pwp * mk;
        int mask_crops = (c->lu_id >= 500) && (c->lu_id < 600);
        c->k_stress_transp =
            (mask_crops ? c->k_stress_transp : ((c->theta_rz - c->theta_pwp) / (theta_ws - c->theta_pwp))) * mk;
        c->k_stress_transp = (c->k_stress_transp > 1 ? 1 : c->k_stress_transp);
        c->transp_coeff = c->basal_transp_coeff * c->k_stress_transp * mk;
        int mask_anoxia = (c->lu_id > 500) && (c->lu_id < 599) && (c->theta_rz >= 0.8 * c->theta_sat);
        {
            double r = c->S_lp_rz / c->S_ac_rz;
            double v = ((r >= 0) && (r <= 1)) ? 1 - pow(r, 1.5) : 1;
            c->transp_coeff = (mask_anoxia ? v : c->transp_coeff) * mk;
        }
        double _pt = (c->pevap_soil < c->pet ? c->pet - c->pevap_soil : 0) * mk;
        double _ptransp = (c->evap_soil < c->pet ? c->pet - c->evap_soil : 0) * mk;
        c->pt = _pt * c->basal_transp_coeff * mk;
        c->ptransp = _ptransp * c->transp_coeff * mk;
        c->ptransp = (is_tree_lu(c->lu_id) ? c->pet * c->transp_coeff : c->ptransp) * mk;
        c->ptransp_res = c->ptransp * mk;
        double transp_lp = 0, transp_fp = 0;
        int mask1 = (c->S_lp_rz > 0) && (c->ptransp_res <= c->S_lp_rz) && (c->ptransp > 0) && (c->prec <= 0);
        transp_lp += (mask1 ? c->ptransp_res : 0) * mk;
        c->ptransp_res = (mask1 ? 0 : c->ptransp_res) * mk;
        int mask2 = (c->S_lp_rz > 0) && (c->ptransp_res > c->S_lp_rz) && (c->ptransp > 0) && (c->prec <= 0);
        transp_lp += (mask2 ? c->S_lp_rz : 0) * mk;
        c->ptransp_res += (mask2 ? -c->S_lp_rz : 0) * mk;
        int mask3 = (c->S_fp_rz > 0) && (c->ptransp_res <= c->S_fp_rz) && (c->S_lp_rz <= 0) && (c->ptransp > 0) &&
                    (c->prec <= 0);
        transp_fp += (mask3 ? c->ptransp_res : 0) * mk;
        c->ptransp_res = (mask3 ? 0 : c->ptransp_res) * mk;
        int mask4 = (c->S_fp_rz > 0) && (c->ptransp_res > c->S_fp_rz) && (c->S_lp_rz <= 0) && (c->ptransp > 0) &&
                    (c->prec <= 0);
        transp_fp += (mask4 ? c->S_fp_rz : 0) * mk;
        c->ptransp_res += (mask4 ? -c->S_fp_rz : 0) * mk;
        c->ptransp_res = (c->ptransp_res < 0 ? 0 : c->ptransp_res) * mk;
        c->S_lp_rz += -transp_lp * mk;
        c->S_fp_rz += -transp_fp * mk;
        c->transp = (transp_fp + transp_lp) * mk;
    }
    /* calc_acc_evap_soil_deficit :547-560 (precedence: a + b*ratio*mk) */
    c->de += c->evap_soil + c->transp * (c->z_evap / c->z_root) * mk;
    /* calc_aet_soil :564-576, calc_aet :580-599 */
    c->aet_soil = (c->evap_soil + c->transp) * mk;
    c->aet = (c->evap_int_top + c->evap_int_ground + c->evap_dep + c->evap_soil + c->transp) * mk;
}

/* ======================================================================== */
/* a4  snow: roger/core/snow.py                                              */
/* ======================================================================== */
static void snow_cell(oc_cell *c, const oc_settings *st, double dt) {
    const double mk = (double)c->maskCatch;
    const double kw = (10000. / (100 - st->rmax) / 100.);
    /* calc_snow_accumulation :7-26 */
    {
        int mask1 = c->ta <= st->ta_fm;
        c->S_snow += c->snow_ground * B(mask1) * mk;
        c->swe += c->snow_ground * B(mask1) * mk;
    }
    /* calc_rain_on_snow :30-44 */
    {
        int mask1 = (c->swe > 0) && (c->ta > st->ta_fm);
        c->S_snow += c->rain_ground * B(mask1) * mk;
    }
    /* calc_snow_melt_int_top :48-133 */
    {
        double pot = (st->sf * (c->ta - st->ta_fm) * dt) * mk;
        int mask1 = (pot > 0) && (pot <= c->swe_top) && (c->swe_top > 0);
        int mask2 = (pot > 0) && (pot > c->swe_top) && (c->swe_top > 0);
        c->snow_melt_top = 0;
        c->snow_melt_top = (mask1 ? pot : c->snow_melt_top) * mk;
        c->snow_melt_top = (mask2 ? c->swe_top : c->snow_melt_top) * mk;
        int mask4 = (c->snow_melt_top > 0) && (c->snow_melt_top <= c->swe_top);
        int mask5 = (c->snow_melt_top > 0) && (c->snow_melt_top > c->swe_top);
        c->pet_res += -c->snow_melt_top * B(mask4) * mk;
        c->swe_top += -c->snow_melt_top * B(mask4) * mk;
        c->pet_res += -c->swe_top * B(mask5) * mk;
        c->swe_top += (mask5 ? 0 : -c->swe_top) * mk; /* :92-95, zeroes swe_top whenever !mask5 */
        c->pet_res = (c->pet_res < 0 ? 0 : c->pet_res) * mk;
        double wtmx = kw * c->swe_top;
        double q_ret = (c->S_int_top > c->S_int_top_tot ? c->S_int_top - c->swe_top : 0) * mk;
        c->snow_melt_drip =
            (q_ret > wtmx ? q_ret - wtmx
                          : ((wtmx <= 0) && (c->S_int_top_tot < c->S_int_top) ? c->S_int_top - c->S_int_top_tot : 0)) *
            mk;
        int mask6 = c->S_int_top_tot < c->S_int_top;
        c->S_snow += (mask6 ? c->snow_melt_drip : 0) * mk;
        c->S_int_top += (mask6 ? -c->snow_melt_drip : 0) * mk;
    }
    /* calc_snow_melt_ground_int :137-193 */
    {
        double pot = (st->sf * (c->ta - st->ta_fm) * dt) * mk;
        int mask1 = (pot > 0) && (pot <= c->swe_ground) && (c->swe_ground > 0);
        int mask2 = (pot > 0) && (pot > c->swe_ground) && (c->swe_ground > 0);
        c->snow_melt_ground = 0;
        c->snow_melt_ground = (mask1 ? pot : c->snow_melt_ground) * mk;
        c->snow_melt_ground = (mask2 ? c->swe_ground : c->snow_melt_ground) * mk;
        int mask4 = (c->snow_melt_ground > 0) && (c->snow_melt_ground <= c->swe_ground);
        int mask5 = (c->snow_melt_ground > 0) && (c->snow_melt_ground > c->swe_ground);
        c->pet_res += -c->snow_melt_ground * B(mask4) * mk;
        c->swe_ground += -c->snow_melt_ground * B(mask4) * mk;
        c->pet_res += -c->swe_ground * B(mask5) * mk;
        c->swe_ground += (mask5 ? 0 : -c->swe_ground) * mk;
    }
    /* calc_snow_melt :197-290 */
    {
        double pot = (st->sf * (c->ta - st->ta_fm) * dt) * mk;
        int mask1 = (pot > 0) && (pot <= c->swe) && (c->swe > 0);
        int mask2 = (pot > 0) && (pot > c->swe) && (c->swe > 0);
        c->snow_melt = 0;
        c->snow_melt = (mask1 ? pot : c->snow_melt) * mk;
        c->snow_melt = (mask2 ? c->swe : c->snow_melt) * mk;
        int mask5 = (c->snow_melt > 0) && (c->snow_melt <= c->swe);
        int mask6 = (c->snow_melt > 0) && (c->snow_melt > c->swe);
        c->pet_res += -c->snow_melt * B(mask5) * mk;
        c->swe += -c->snow_melt * B(mask5) * mk;
        c->pet_res += -c->swe * B(mask6) * mk;
        c->swe = (mask6 ? 0 : c->swe) * mk;
        c->pet_res = (c->pet_res < 0 ? 0 : c->pet_res) * mk;
        double wtmx = kw * c->swe;
        double q_ret = (c->S_snow > 0 ? c->S_snow - c->swe : 0) * mk;
        c->q_snow = 0;
        c->q_snow = (q_ret > wtmx ? q_ret - wtmx : (wtmx <= 0 ? c->S_snow : 0)) * mk;
        c->S_snow += -c->q_snow * mk;
        c->z0 += c->q_snow * mk;
        c->prec_event_csum += c->q_snow * mk;
    }
}

/* ======================================================================== */
/* a5-a9  infiltration: roger/core/infiltration.py:1-2193                    */
/* ======================================================================== */
static double calc_theta_d(const oc_cell *c, double mk) { /* :1564-1594 */
    double v = 0;
    v = (c->z_root > 0 ? (c->theta_sat - c->theta_rz) * (1 - c->sealing / 1) : v) * mk;
    v = (c->z_soil <= 0 ? 0.01 : v) * mk;
    v = (v <= 0 ? 0.01 : v) * mk;
    return v;
}
static double calc_theta_d_rel(const oc_cell *c, double mk) { /* :1598-1632 */
    double v = 0;
    v = (c->z_root > 0 ? ((c->theta_sat - c->theta_rz) / (c->theta_sat - c->theta_pwp)) * (1 - c->sealing / 1) : v) * mk;
    v = (c->z_soil <= 0 ? 0.01 : v) * mk;
    v = (v <= 0 ? 0.01 : v) * mk;
    return v;
}
static double calc_theta_d_fp(const oc_cell *c, double mk) { /* :1636-1666 */
    double v = 0;
    v = (c->z_soil > 0 ? (c->theta_fc - c->theta_rz) * (1 - c->sealing / 1) : v) * mk;
    v = (c->z_soil <= 0 ? 0.01 : v) * mk;
    v = (v <= 0 ? 0.01 : v) * mk;
    return v;
}

static void depth_shrinkage_cracks_cell(oc_cell *c) { /* :1768-1826 */
    const double mk = (double)c->maskCatch;
    double th = c->theta_rz;
    c->z_sc = (th < c->theta_4
                   ? c->z_sc_max
                   : ((th >= c->theta_4) && (th < c->theta_27) ? (th - c->theta_4) / (c->theta_27 - c->theta_4) : 0) *
                         c->z_sc_max) *
              mk;
    c->z_sc = (th < c->theta_4 ? c->z_sc_max : c->z_sc) * mk;
    c->z_sc = (th > c->theta_27 ? 0 : c->z_sc) * mk;
    c->z_sc = ((1 - c->sealing / 1) * c->z_sc) * mk;
    c->z_sc = (c->z_sc > c->z_root ? c->z_root : c->z_sc) * mk;
    c->z_sc = (c->lu_id == 13 ? 0 : c->z_sc) * mk;
}

static void set_event_vars_cell(oc_cell *c) { /* :1830-1976 */
    const double mk = (double)c->maskCatch;
    c->no_wf = 1;
    c->z_wf = c->z_wf_m1 = 0;
    c->z_wf_t0 = c->z_wf_t0_m1 = 0;
    c->z_wf_t1 = c->z_wf_t1_m1 = 0;
    c->z_wf_fc = 0;
    c->inf_mat_event_csum = 0;
    c->inf_mat_pot_event_csum = 0;
    c->inf_mp_event_csum = 0;
    c->y_mp = c->y_mp_m1 = 0;
    c->inf_sc_event_csum = 0;
    c->y_sc = c->y_sc_m1 = 0;
    double td = calc_theta_d(c, mk);
    c->theta_d = td * mk;
    double tdr = calc_theta_d_rel(c, mk);
    c->theta_d_rel = tdr * mk;
    c->theta_d_t0 = td * mk;
    c->theta_d_rel_t0 = tdr * mk;
    c->theta_d_fp = calc_theta_d_fp(c, mk) * mk;
    c->prec_event_csum = 0;
    c->t_event_csum = 0;
    c->de = 0;
}

static void start_rainfall_pause_cell(oc_cell *c) { /* :1980-1995, calc_z_wf_fc :1536-1560 */
    const double mk = (double)c->maskCatch;
    double zf = (c->theta_d_fp > 0 ? c->inf_mat_event_csum / c->theta_d_fp : c->z_wf) * mk;
    zf = (zf > c->z_soil ? c->z_soil : zf) * mk;
    int mask = (c->prec == 0) && (c->prec_m1 != 0);
    c->z_wf_fc = (mask ? zf : c->z_wf_fc) * mk;
}

static void end_rainfall_pause_cell(oc_cell *c) { /* :1999-2053 */
    const double mk = (double)c->maskCatch;
    int mask = (c->prec != 0) && (c->prec_m1 == 0);
    c->no_wf = mask ? 2 : c->no_wf;
    double td = calc_theta_d(c, mk);
    c->theta_d = (mask ? td : c->theta_d) * mk;
    double tdr = calc_theta_d_rel(c, mk);
    c->theta_d_rel = (mask ? tdr : c->theta_d_rel) * mk;
    c->z_wf_t1 = mask ? 0 : c->z_wf_t1;
    c->z_wf_t1_m1 = mask ? 0 : c->z_wf_t1_m1;
    c->prec_event_csum = mask ? 0 : c->prec_event_csum;
    c->t_event_csum = mask ? 0 : c->t_event_csum;
}

static void reset_event_vars_cell(oc_cell *c) { /* :2057-2144 */
    const double mk = (double)c->maskCatch;
    c->z_wf = c->z_wf_m1 = 0;
    c->z_wf_t0 = c->z_wf_t0_m1 = 0;
    c->z_wf_t1 = c->z_wf_t1_m1 = 0;
    c->y_mp = 0; /* only [tau], :2080-2084 */
    c->y_sc = c->y_sc_m1 = 0;
    double td = calc_theta_d(c, mk);
    c->theta_d = td * mk;
    c->theta_d_t0 = td * mk;
    c->pi_gr = 0;
    c->pi_m = 0;
    c->t_sat = 0;
    c->Fs = 0;
    c->z_sc = 0;
}

static void green_ampt_params_cell(oc_cell *c, double dt) { /* :8-48, 1670-1764 */
    const double mk = (double)c->maskCatch;
    double pi_gr = c->ks * (((c->theta_d * c->wfs) / (c->prec_event_csum + 1)) + 1);
    c->pi_gr = pi_gr * mk;
    double pi_m = c->ks * c->theta_d * c->wfs * mk;
    c->pi_m = pi_m * mk;
    /* calc_sat_time :1707-1740 (uses the freshly updated vs.pi_m / vs.pi_gr) */
    int mask1 = (c->pi_m <= c->prec_event_csum) && (c->pi_m > c->pi_gr) && (c->t_sat == 0);
    int mask2 = ((c->prec * (1 / dt) - c->ks) * c->prec_event_csum > c->ks * c->theta_d * c->wfs) &&
                (c->pi_m <= c->prec_event_csum) && (c->pi_m <= c->pi_gr) && (c->t_sat == 0);
    c->t_sat = mask1 ? c->t_event_csum - dt : c->t_sat;
    c->t_sat = mask2 ? c->t_event_csum + ((c->ks * c->theta_d * c->wfs) / (c->pi_m * (c->pi_m * -c->ks))) -
                           (dt / c->pi_m) * c->prec_event_csum
                     : c->t_sat;
    c->t_sat = c->t_sat * mk;
    /* calc_Fs :1744-1764 (uses the local pi_m) */
    double Fs = ((c->ks * c->theta_d * c->wfs) / (pi_m - c->ks)) * mk;
    Fs = (pi_m <= c->ks ? pi_m : Fs) * mk;
    c->Fs = Fs * mk;
}

static void inf_mat_cell(oc_cell *c, double dt) { /* :52-427 */
    const double mk = (double)c->maskCatch;
    int mask1 = (c->pi_m <= c->prec_event_csum) && (c->t_event_csum > c->t_sat) && (c->t_sat > 0);
    int mask2 = (c->pi_m > c->prec_event_csum) && (c->t_event_csum > c->t_sat) && (c->t_sat > 0);
    int mask3 = (c->t_sat > c->t_event_csum - dt) && (c->t_sat < c->t_event_csum);
    int mask4 = (c->pi_m > c->prec_event_csum) && (c->t_sat <= 0);
    double a = c->ks * (c->t_event_csum - c->t_sat) * mk;
    double b = c->Fs + 2 * c->theta_d * c->wfs * mk;
    double l1 = (c->z0 > c->ks * dt ? (c->ks * dt * c->wfs * c->theta_d) / (c->z0 - c->ks * dt)
                                    : (c->ks * dt * c->wfs * c->theta_d) / (c->ks * dt)) *
                mk;
    double seal = ((1 - c->sealing) / 1);
    c->inf_mat_pot = c->ks * dt;
    double rec = (c->ks * dt / 2) * (1 + (1 + 2 * b / a) / sqrt(1 + (4 * b / a) + (4 * (c->Fs_t0 * c->Fs_t0) / (a * a))));
    c->inf_mat_pot = (mask1 ? rec * seal : c->inf_mat_pot) * mk;
    c->inf_mat_pot = (mask2 ? c->ks * dt * (1 + ((c->wfs * c->theta_d) / l1)) * seal : c->inf_mat_pot) * mk;
    double pot_rec = (mask3 ? rec : 0) * mk;
    double pot_sat = (mask3 ? c->z0 * (c->t_sat - (c->t_event_csum - dt)) : 0) * mk;
    c->inf_mat_pot = (mask3 ? pot_sat + pot_rec * seal : c->inf_mat_pot) * mk;
    c->inf_mat_pot = (mask4 ? c->pi_gr * seal : c->inf_mat_pot) * mk;

    int mask7 = c->z0 < c->inf_mat_pot;
    int mask8 = c->z0 >= c->inf_mat_pot;
    c->inf_mat = (mask7 ? c->z0 : c->inf_mat) * mk;
    c->inf_mat = (mask8 ? c->inf_mat_pot : c->inf_mat) * mk;
    double room = (c->S_ac_rz + c->S_ufc_rz) - (c->S_lp_rz + c->S_fp_rz);
    c->inf_mat = (c->inf_mat > room ? room : c->inf_mat) * mk;
    c->inf_mat = (c->inf_mat < 0 ? 0 : c->inf_mat) * mk;
    c->inf_mat_event_csum += c->inf_mat * mk;
    c->inf_mat_pot_event_csum += c->inf_mat_pot * mk;

    double dz_wf = 0;
    dz_wf = (c->no_wf == 1 ? (c->inf_mat / c->theta_d_t0) : dz_wf) * mk;
    dz_wf = (c->no_wf == 2 ? c->inf_mat / c->theta_d : dz_wf) * mk;
    c->z_wf_t0 += (isfinite(dz_wf) ? dz_wf : 0) * mk;
    c->z_wf_t1 += (isfinite(dz_wf) ? dz_wf : 0) * mk;
    c->z_wf_t0 = (c->z_wf_t0 > c->z_soil ? c->z_soil : c->z_wf_t0) * mk;
    c->z_wf_t1 = (c->z_wf_t1 > c->z_soil ? c->z_soil : c->z_wf_t1) * mk;
    c->z0 += -c->inf_mat * mk;
    c->z0 = (c->z0 < 0 ? 0 : c->z0) * mk;

    double dz0 = ((c->z_wf_fc > 0) && (c->rain_ground <= 0) && (c->no_wf == 1) ? c->inf_mat_pot / c->theta_d_t0 : 0) * mk;
    c->z_wf_t0 += (isfinite(dz0) ? dz0 : 0) * mk;
    c->z_wf_t0 = ((c->z_wf_t0 > c->z_wf_fc) && (c->z_wf_fc > 0) ? c->z_wf_fc : c->z_wf_t0) * mk;
    c->z_wf_t0 = (c->z_wf_t0 > c->z_soil ? c->z_soil : c->z_wf_t0) * mk;
    double dz1 = ((c->z_wf_fc > 0) && (c->rain_ground <= 0) && (c->no_wf == 2) ? c->inf_mat_pot / c->theta_d : 0) * mk;
    c->z_wf_t1 += (isfinite(dz1) ? dz1 : 0) * mk;
    c->z_wf_t1 = ((c->z_wf_t1 > c->z_wf_fc) && (c->z_wf_fc > 0) ? c->z_wf_fc : c->z_wf_t1) * mk;
    c->z_wf_t1 = (c->z_wf_t1 > c->z_soil ? c->z_soil : c->z_wf_t1) * mk;

    int mask14 = (c->z_wf_t0 >= c->z_wf_t1) && (c->z_wf_t1 <= 0);
    int mask15 = (c->z_wf_t0 > c->z_wf_t1) && (c->z_wf_t1 > 0);
    int mask20 = (c->z_wf_t0 <= c->z_wf_t1) && (c->z_wf_t1 > 0);
    c->z_wf = (mask14 ? c->z_wf_t0 : c->z_wf) * mk;
    c->theta_d = (mask14 ? c->theta_d_t0 : c->theta_d) * mk;
    c->theta_d_rel = (mask14 ? c->theta_d_rel_t0 : c->theta_d_rel) * mk;
    c->z_wf_m1 = (mask15 ? 0 : c->z_wf_m1) * mk;
    c->z_wf = (mask15 ? c->z_wf_t1 : c->z_wf) * mk;
    c->no_wf = mask20 ? 1 : c->no_wf;
    c->z_wf = (mask20 ? c->z_wf_t0 : c->z_wf) * mk;
    c->theta_d = (mask20 ? c->theta_d_t0 : c->theta_d) * mk;
    c->theta_d_rel = (mask20 ? c->theta_d_rel_t0 : c->theta_d_rel) * mk;
    c->z_wf = (c->z_wf > c->z_soil ? c->z_soil : c->z_wf) * mk;
    c->theta_d = (c->theta_d_t1 <= 0 ? c->theta_d_t0 : c->theta_d) * mk;
}

static void inf_mp_cell(oc_cell *c, const oc_settings *st, double dt) { /* :431-1077 */
    const double mk = (double)c->maskCatch;
    /* :443-462: the second assignment of each pair wins */
    double z_wf = (c->no_wf == 2 ? 0 : c->z_wf_t1) * mk;
    double z_wf_m1 = (c->no_wf == 2 ? 0 : c->z_wf_t1_m1) * mk;
    c->lmpv_non_sat = c->lmpv - z_wf * mk;
    c->lmpv_non_sat = (c->lmpv_non_sat < 0 ? 0 : c->lmpv_non_sat) * mk;
    double dz_wf = z_wf - z_wf_m1 * mk;
    dz_wf = (z_wf >= c->lmpv ? c->lmpv_non_sat : dz_wf) * mk;
    dz_wf = (c->lmpv_non_sat <= 0 ? 0 : dz_wf) * mk;
    dz_wf = (dz_wf <= 0 ? 0 : dz_wf) * mk;
    c->lmpv_non_sat = c->lmpv - c->z_wf * mk;
    c->lmpv_non_sat = (c->lmpv_non_sat < 0 ? 0 : c->lmpv_non_sat) * mk;
    int substeps = (int)nearbyint(dt / (1.0 / 5)); /* npx.round: half-to-even */
    c->lmpv_non_sat = (substeps == 1 ? c->lmpv_non_sat + dz_wf / 1.39 : c->lmpv_non_sat) * mk;

    double y1 = 0, y2, ym1, a, b1 = 0, b2 = 0, cc = 0, pot_di = 0, di = 0, z0_di = 0, inf_mp = 0, inf_mp_pot = 0;
    double ecs, t = 0, y;
    y2 = (st->r_mp / 2) * mk;
    a = c->theta_d * (st->r_mp * st->r_mp) * mk;
    y = c->y_mp_m1 * mk;
    ym1 = c->y_mp_m1 * mk;
    ecs = c->inf_mp_event_csum * mk;
    const double k6 = sqrt(6.0) * 2; /* `6**0.5 * 2` folds before the array product */
    for (int it = 0; it < substeps; ++it) {
        z0_di = c->z0 * (c->mp_drain_area / substeps) * mk;
        t += (dt / substeps) * mk;
        cc = c->ks * c->wfs * t * mk;
        cc = (isnan(cc) ? 0 : cc) * mk;
        b1 = (k6 * sqrt(cc * (6 * cc - a))) * mk;
        b1 = (isnan(b1) ? 0 : b1) * mk;
        b2 = (st->r_mp * (c->theta_d * c->theta_d)) * (12 * cc - a + b1) * mk;
        b2 = (isnan(b2) ? 0 : b2) * mk;
        b2 = (b2 <= 0 ? 0 : b2) * mk;
        y1 = (pow(b2, 1.0 / 3) / c->theta_d) * 0.5 * mk;
        y2 = (a / pow(b2, 1.0 / 3)) * 0.5 * mk;
        y = (y1 + y2 + ym1) * mk;
        y = (y < st->r_mp ? st->r_mp : y) * mk;
        y = (y < ym1 ? ym1 : y) * mk;
        pot_di = (st->pi * (y * y - ym1 * ym1) * c->lmpv_non_sat * c->theta_d * c->dmpv * 1e-06) * mk;
        inf_mp_pot += pot_di * mk;
        di = (pot_di > z0_di ? z0_di : pot_di) * mk;
        di = (c->lmpv_non_sat == 0 ? 0 : di) * mk;
        inf_mp += di * mk;
        ecs += di * mk;
        y = st->r_mp + sqrt((ecs / (c->dmpv * c->theta_d)) / st->pi) * mk;
        y = (y < st->r_mp ? st->r_mp : y) * mk;
        t = c->theta_d / (c->ks * c->wfs * st->r_mp) *
            (pow(y, 3) / 3.0 - (y * y) * st->r_mp / 2.0 + pow(st->r_mp, 3) / 6.0) * mk;
        inf_mp = (inf_mp < 0 ? 0 : inf_mp) * mk;
        ym1 = y * mk;
    }
    (void)y1; (void)y2; (void)inf_mp_pot;
    c->y_mp = y * mk;
    c->y_mp = (isnan(c->y_mp) ? 0 : c->y_mp) * mk;
    c->inf_mp = inf_mp * mk;
    c->inf_mp = (isnan(c->inf_mp) ? 0 : c->inf_mp) * mk;

    double share = (c->lmpv_non_sat > 0 ? 1.0 - (c->lmpv - c->z_root) / c->lmpv_non_sat : 0) * mk;
    share = (c->lmpv <= c->z_root ? 1 : share) * mk;
    share = (z_wf >= c->z_root ? 0 : share) * mk;
    share = (share < 0 ? 0 : share) * mk;
    share = (share > 1 ? 1 : share) * mk;

    c->inf_mp_rz = c->inf_mp * share * mk;
    double room = (c->S_ac_rz + c->S_ufc_rz) - (c->inf_mat_rz + c->S_lp_rz + c->S_fp_rz);
    c->inf_mp_rz = ((c->inf_mp_rz > room) && (room >= 0) ? room : c->inf_mp_rz) * mk;
    c->inf_mp_rz = (room < 0 ? 0 : c->inf_mp_rz) * mk;

    c->inf_mp_ss = c->inf_mp * (1 - share) * mk;
    double room_ss = (c->S_ac_ss + c->S_ufc_ss) - (c->S_lp_ss + c->S_fp_ss);
    c->inf_mp_ss = ((c->inf_mp_ss > room_ss) && (room_ss > 0) ? room_ss : c->inf_mp_ss) * mk;
    c->inf_ss = c->inf_mp_ss * mk;
    c->S_fp_ss += c->inf_ss * mk;
    int m = c->S_fp_ss > c->S_ufc_ss;
    c->S_lp_ss += (m ? (c->S_fp_ss - c->S_ufc_ss) : 0) * mk;
    c->S_fp_ss = (m ? c->S_ufc_ss : c->S_fp_ss) * mk;
    m = c->S_lp_ss > c->S_ac_ss;
    c->inf_mp_ss += (m ? -(c->S_lp_ss - c->S_ac_ss) : 0) * mk;
    c->inf_mp_ss = (c->inf_mp_ss < 0 ? 0 : c->inf_mp_ss) * mk;
    c->S_lp_ss = (m ? c->S_ac_ss : c->S_lp_ss) * mk;
    c->inf_mp = 0;
    c->inf_mp = c->inf_mp_rz + c->inf_mp_ss * mk;
    c->inf_mp_event_csum += c->inf_mp * mk;
    c->z0 += -c->inf_mp * mk;
    c->z0 = (c->z0 < 0 ? 0 : c->z0) * mk;
}

static void inf_sc_cell(oc_cell *c, const oc_settings *st, double dt) { /* :1081-1318 */
    const double mk = (double)c->maskCatch;
    double z_wf = (c->no_wf == 2 ? 0 : c->z_wf_t1) * mk;
    double z_wf_m1 = (c->no_wf == 2 ? 0 : c->z_wf_t1_m1) * mk;
    c->z_sc_non_sat = c->z_sc - z_wf * mk;
    c->z_sc_non_sat = (c->z_sc_non_sat < 0 ? 0 : c->z_sc_non_sat) * mk;
    double dz_wf = z_wf - z_wf_m1 * mk;
    dz_wf = (z_wf >= c->z_sc ? c->z_sc_non_sat : dz_wf) * mk;
    dz_wf = (c->z_sc_non_sat <= 0 ? 0 : dz_wf) * mk;
    dz_wf = (dz_wf <= 0 ? 0 : dz_wf) * mk;
    c->z_sc_non_sat = c->z_sc - c->z_wf * mk;
    c->z_sc_non_sat = (c->z_sc_non_sat < 0 ? 0 : c->z_sc_non_sat) * mk;
    int substeps = (int)nearbyint(dt / (1.0 / 5));
    c->z_sc_non_sat = (substeps == 1 ? c->z_sc_non_sat + dz_wf / 1.39 : c->z_sc_non_sat) * mk;

    double y = c->y_sc_m1 * mk, ym1 = c->y_sc_m1 * mk, pot_di = 0, di = 0, z0_di = 0;
    double ecs = c->inf_sc_event_csum * mk, t = 0, inf_sc = 0;
    for (int it = 0; it < substeps; ++it) {
        z0_di = (c->z0 / substeps) * mk;
        t += (dt / substeps) * mk;
        y = sqrt((c->ks * c->wfs * t * 2) / c->theta_d) * mk;
        pot_di = ((c->z_sc_non_sat * c->theta_d * st->l_sc) * (y - ym1) * 1e-06) * mk;
        pot_di = (pot_di <= 0 ? 0 : pot_di) * mk;
        di = (pot_di > z0_di ? z0_di : pot_di) * mk;
        di = (c->z_sc_non_sat <= 0 ? 0 : di) * mk;
        di += di * mk; /* :1248-1252 doubles inf_sc_di */
        ecs += di * mk;
        y = (ecs / st->l_sc / 2) * mk;
        t = ((ym1 * ym1 * c->theta_d) / (c->ks * c->wfs * 2)) * mk;
        ym1 = y * mk;
    }
    c->y_sc = y * mk;
    c->inf_sc = inf_sc * mk; /* the loop never accumulates inf_sc (:1278-1283) */
    c->inf_sc_event_csum += c->inf_sc * mk;
    c->z0 += -c->inf_sc * mk;
    c->z0 = (c->z0 < 0 ? 0 : c->z0) * mk;
}

static void inf_rz_cell(oc_cell *c) { /* calc_inf_rz :1322-1417, calc_inf :1520-1532 */
    const double mk = (double)c->maskCatch;
    c->inf_mat_rz = c->inf_mat * mk;
    c->inf_sc_rz = c->inf_sc * mk;
    c->inf_rz = (c->inf_mat_rz + c->inf_mp_rz + c->inf_sc_rz) * mk;
    c->S_fp_rz += c->inf_rz * mk;
    int m = c->S_fp_rz > c->S_ufc_rz;
    c->S_lp_rz += (m ? (c->S_fp_rz - c->S_ufc_rz) : 0) * mk;
    c->S_fp_rz = (m ? c->S_ufc_rz : c->S_fp_rz) * mk;
    m = c->S_lp_rz > c->S_ac_rz;
    c->inf_mp_rz += (m ? -(c->S_lp_rz - c->S_ac_rz) : 0) * mk;
    c->inf_mp_rz = (c->inf_mp_rz < 0 ? 0 : c->inf_mp_rz) * mk;
    c->z0 += (m ? c->S_lp_rz - c->S_ac_rz : 0) * mk;
    c->S_lp_rz = (m ? c->S_ac_rz : c->S_lp_rz) * mk;
    c->inf_mp = 0;
    c->inf_mp = c->inf_mp_rz + c->inf_mp_ss * mk;
    c->inf_rz = (c->inf_mat_rz + c->inf_mp_rz + c->inf_sc_rz) * mk;
    c->inf = (c->inf_rz + c->inf_ss) * mk;
}

static void hof_sof_cell(oc_cell *c) { /* calc_hof_and_sof :1421-1476 */
    const double mk = (double)c->maskCatch;
    c->q_hof = 0;
    c->q_hof = c->z0 * mk;
    c->q_hof = (c->q_hof < 0 ? 0 : c->q_hof) * mk;
    c->q_sof = 0;
    int mask2 = ((c->S_lp_rz + c->S_fp_rz) > (c->S_ac_rz + c->S_ufc_rz)) &&
                ((c->S_lp_ss + c->S_fp_ss) >= (c->S_ac_ss + c->S_ufc_ss));
    c->q_sof = (mask2 ? (c->S_lp_rz + c->S_fp_rz) - (c->S_ac_rz + c->S_ufc_rz) : c->q_sof) * mk;
    int m = c->q_sof > 0;
    c->S_fp_rz = (m ? c->S_ufc_rz : c->S_fp_rz) * mk;
    c->S_lp_rz = (m ? c->S_ac_rz : c->S_lp_rz) * mk;
}

static void surface_runoff_cell(oc_cell *c) { /* calc_surface_runoff :1480-1516 */
    const double mk = (double)c->maskCatch;
    c->z0 += -c->q_hof * mk;
    c->z0 = (c->z0 < 0 ? 0 : c->z0) * mk;
    c->q_sur = 0;
    c->q_sur += (c->q_hof + c->q_sof) * mk;
    c->q_sur += (c->maskRiver || c->maskLake) ? c->prec : 0;
}

/* ======================================================================== */
/* a10 subsurface runoff, SVAT branch: roger/core/subsurface_runoff.py        */
/* ======================================================================== */
/* oneD model: saturated thickness per 200 mm layer, subsurface_runoff.py:51-245 */
static void z_sat_layer_cell(oc_cell *c) {
    const double mk = (double)c->maskCatch;
    double *L[8] = {&c->z_sat_layer_1, &c->z_sat_layer_2, &c->z_sat_layer_3, &c->z_sat_layer_4,
                    &c->z_sat_layer_5, &c->z_sat_layer_6, &c->z_sat_layer_7, &c->z_sat_layer_8};
    for (int k = 0; k < 8; ++k) {
        double v = (k == 0) ? c->z_sat * mk : c->z_sat - (200.0 * k) * mk;
        if (k < 7) v = (v > 200 ? 200 : v) * mk;
        v = (v <= 0 ? 0 : v) * mk;
        *L[k] = v;
    }
}

/* calc_potential_lateral_subsurface_runoff :248-372 */
static void pot_lateral_cell(oc_cell *c, const oc_settings *st, double dt) {
    const double mk = (double)c->maskCatch;
    const double dx = st->dx, r2 = st->r_mp * st->r_mp;
    const double per_len = (1 / (dx * (c->z_soil / 1000)));
    c->q_sub_mat_pot = ((c->ks * c->slope * c->z_sat * dx * 1000 * dt) * 1e-6 * per_len) * mk;
    c->q_sub_mat_pot = (c->z_sat <= 0 ? 0 : c->q_sub_mat_pot) * mk;
    const double zl[8] = {c->z_sat_layer_1, c->z_sat_layer_2, c->z_sat_layer_3, c->z_sat_layer_4,
                          c->z_sat_layer_5, c->z_sat_layer_6, c->z_sat_layer_7, c->z_sat_layer_8};
    const double vl[8] = {c->v_mp_layer_1, c->v_mp_layer_2, c->v_mp_layer_3, c->v_mp_layer_4,
                          c->v_mp_layer_5, c->v_mp_layer_6, c->v_mp_layer_7, c->v_mp_layer_8};
    double sum = 0;
    for (int k = 0; k < 8; ++k) {
        double term = zl[k] * vl[k] * dt * dx * 1000 * c->dmph * 1e-6 * r2 * st->pi * 1e-6;
        sum = (k == 0) ? term : sum + term;
    }
    c->q_sub_mp_pot = (sum * per_len) * mk;
    c->q_sub_mp_pot = (c->q_sub_mp_pot < 0 ? 0 : c->q_sub_mp_pot) * mk;
    c->q_sub_mp_pot = (c->z_sat <= 0 ? 0 : c->q_sub_mp_pot) * mk;
    c->q_sub_pot = (c->q_sub_mp_pot + c->q_sub_mat_pot) * mk;
    c->q_sub_mat_share = (c->q_sub_mat_pot / c->q_sub_pot) * mk;
    c->q_sub_mat_share = (c->q_sub_pot == 0 ? 0 : c->q_sub_mat_share) * mk;
    c->q_sub_mp_share = (c->q_sub_mp_pot / c->q_sub_pot) * mk;
    c->q_sub_mp_share = (c->q_sub_pot == 0 ? 0 : c->q_sub_mp_share) * mk;
    int mask1 = c->q_sub_pot > c->S_lp_rz + c->S_lp_ss;
    c->q_sub_pot = (mask1 ? c->S_lp_rz + c->S_lp_ss : c->q_sub_pot) * mk;
    c->q_sub_mat_pot = c->q_sub_pot * c->q_sub_mat_share * mk;
    c->q_sub_mp_pot = c->q_sub_pot * c->q_sub_mp_share * mk;
}

/* calc_lateral_subsurface_runoff_rz :375-457 */
static void lateral_rz_cell(oc_cell *c) {
    const double mk = (double)c->maskCatch;
    double share = (c->z_sat > 0 ? ((c->z_sat - (c->z_soil - c->z_root)) / c->z_sat) : 0) * mk;
    int mask1 = (c->z_sat <= c->z_soil - c->z_root) || (c->S_lp_rz <= 0);
    share = (mask1 ? 0 : share) * mk;
    share = (isnan(share) ? 0 : share) * mk;
    c->S_zsat_rz = ((c->z_sat * share) * c->theta_ac) * mk;
    c->q_sub_rz = (c->q_sub_pot * share < c->S_zsat_rz ? c->q_sub_pot * share : c->S_zsat_rz) * mk;
    c->q_sub_mat_rz = c->q_sub_rz * c->q_sub_mat_share * mk;
    c->q_sub_mp_rz = c->q_sub_rz * c->q_sub_mp_share * mk;
    c->q_sub_mp_pot_rz = c->q_sub_mp_pot * share * mk;
    c->z_sat += -c->q_sub_rz / c->theta_ac * mk;
    c->S_lp_rz += -c->q_sub_rz * mk;
}

/* calc_potential_lateral_subsurface_runoff_ss :460-515 */
static void pot_lateral_ss_cell(oc_cell *c) {
    const double mk = (double)c->maskCatch;
    double share = ((c->z_soil - c->z_root) / c->z_sat) * mk;
    int mask1 = (c->z_sat <= c->z_soil - c->z_root) || (c->S_lp_rz <= 0);
    int mask2 = c->z_sat <= 0;
    int mask3 = isnan(share);
    share = (mask1 ? 1 : share) * mk;
    share = (mask2 ? 0 : share) * mk;
    share = (mask3 ? 0 : share) * mk;
    c->q_sub_mat_pot_ss = c->q_sub_mat_pot * share * mk;
    c->q_sub_mp_pot_ss = c->q_sub_mp_pot * share * mk;
    c->q_sub_pot_ss = (c->q_sub_mat_pot_ss + c->q_sub_mp_pot_ss) * mk;
}

/* calc_lateral_subsurface_runoff_ss :518-659, calc_lateral_subsurface_runoff :662-690 */
static void lateral_ss_cell(oc_cell *c) {
    const double mk = (double)c->maskCatch;
    c->q_ss = 0;
    c->q_ss = (c->z_sat <= 0 ? c->q_pot_ss : c->q_ss) * mk;
    double tot = c->q_pot_ss + c->q_sub_pot_ss;
    double fv = (tot > 0 ? c->q_pot_ss / tot : 0) * mk;
    double fl = (tot > 0 ? c->q_sub_pot_ss / tot : 0) * mk;
    double q_ss_sat = (tot <= c->S_zsat_ss ? tot * fv : c->S_zsat_ss * fv) * mk;
    c->q_ss = (c->z_sat > 0 ? q_ss_sat : c->q_ss);
    c->q_sub_ss = 0;
    c->q_sub_ss = (tot <= c->S_zsat_ss ? tot * fl : c->S_zsat_ss * fl) * mk;
    c->q_sub_mat_ss = c->q_sub_ss * c->q_sub_mat_share * mk;
    c->q_sub_mp_ss = c->q_sub_ss * c->q_sub_mp_share * mk;
    int mask1 = c->S_lp_ss < c->q_ss;
    int mask2 = c->S_lp_ss >= c->q_ss;
    c->S_fp_ss += (mask1 ? -(c->q_ss - c->S_lp_ss) : 0) * mk;
    c->S_lp_ss = (mask1 ? 0 : c->S_lp_ss) * mk;
    c->S_lp_ss += (mask2 ? -c->q_ss : 0) * mk;
    int mask = c->z_sat > 0;
    c->S_lp_ss += (mask ? -c->q_sub_ss : 0) * mk;
    c->z_sat += -((c->q_sub_ss + c->q_ss) / c->theta_ac) * mk;
    c->z_sat = (c->z_sat < 0 ? 0 : c->z_sat) * mk;
    c->S_zsat = c->z_sat * c->theta_ac * mk;
    c->q_sub_mat = (c->q_sub_mat_rz + c->q_sub_mat_ss) * mk;
    c->q_sub_mp = (c->q_sub_mp_rz + c->q_sub_mp_ss) * mk;
    c->q_sub = (c->q_sub_rz + c->q_sub_ss) * mk;
}

static void subsurface_runoff_cell(oc_cell *c, double dt, const oc_settings *st) {
    const double mk = (double)c->maskCatch;
    const int lateral = (int)st->enable_lateral_flow;
    /* calc_rise_of_saturation_water_table :693-765 */
    {
        double lmpv_ss = c->lmpv - c->z_root * mk;
        lmpv_ss = (c->lmpv < c->z_root ? 0 : lmpv_ss) * mk;
        double z_sat_top =
            (c->S_lp_ss < c->theta_ac ? c->S_lp_ss / c->theta_ac : c->S_lp_rz + c->S_lp_ss / c->theta_ac) * mk;
        double z_nomp = (c->z_soil - c->z_root) - lmpv_ss - c->z_sat * mk;
        z_nomp = (z_nomp < 0 ? 0 : z_nomp);
        double inner =
            ((c->S_fp_ss >= c->S_ufc_ss) && (((c->S_lp_ss + 1e-6) / c->theta_ac) < (c->z_soil - c->z_root)))
                ? c->S_lp_ss / c->theta_ac
                : (((c->S_fp_rz >= c->S_ufc_rz) && (c->S_lp_ss + 1e-6 >= c->S_ac_ss))
                       ? c->S_lp_rz / c->theta_ac + c->S_lp_ss / c->theta_ac
                       : c->S_lp_ss / c->theta_ac);
        c->z_sat = (z_sat_top > z_nomp ? inner : c->S_lp_ss / c->theta_ac) * mk;
    }
    /* calc_S_zsat :7-48 */
    {
        c->S_zsat = (c->z_sat <= c->z_soil ? c->z_sat * c->theta_ac : c->z_soil * c->theta_ac) * mk;
        c->S_zsat_ss = (c->z_sat <= c->z_soil - c->z_root ? c->S_zsat : (c->z_soil - c->z_root) * c->theta_ac) * mk;
        c->S_zsat_rz = (c->z_sat > c->z_soil - c->z_root ? (c->z_sat - (c->z_soil - c->z_root)) * c->theta_ac : 0) * mk;
    }
    if (lateral) z_sat_layer_cell(c);
    /* calc_potential_percolation_rz :768-896 */
    {
        int mask1 = (c->z_wf < c->z_root) && (c->z_sat <= 0);
        int mask2 = (c->z_wf >= c->z_root) && (c->z_sat <= 0);
        int mask3 = (c->z_sat > 0) && (c->z_root < c->z_soil - c->z_sat);
        double perc = (mask1 ? c->k_rz * dt : 0) * mk;
        perc = (mask2 ? c->k_rz * dt : perc) * mk;
        double z = (c->z_soil - c->z_root) - c->z_sat;
        if (mask3) {
            double p1 = pow(z / (-c->ha * 10.2), -c->n_salv);
            double p2 = pow(-c->h_rz / -c->ha, -c->n_salv);
            perc = ((p1 - p2) / (1 + p2 + (c->n_salv - 1) * p1)) * dt * c->ks * (-1);
        }
        perc = perc * mk;
        perc = (perc < 0 ? 0 : perc) * mk;
        int lim = c->z_root_m1 < c->z_soil - c->z_sat;
        int mask4 = (perc > 0) && (c->S_lp_rz + c->S_fp_rz >= perc) && lim;
        int mask5 = (perc > 0) && (c->S_lp_rz + c->S_fp_rz < perc) && lim;
        c->q_pot_rz = 0;
        c->q_pot_rz = (mask4 ? perc : c->q_pot_rz) * mk;
        c->q_pot_rz = (mask5 ? c->S_fp_rz + c->S_lp_rz : c->q_pot_rz) * mk;
        double room = (c->S_ac_ss + c->S_ufc_ss) - (c->S_lp_ss + c->S_fp_ss);
        int mask6 = (c->q_pot_rz > 0) && (room > 0) && (c->q_pot_rz > room) && lim;
        c->q_pot_rz = (mask6 ? room : c->q_pot_rz) * mk;
        int mask7 = (c->S_lp_ss >= c->S_ac_ss - 1e-6) && (c->S_fp_ss >= c->S_ufc_ss - 1e-6);
        c->q_pot_rz = (mask7 ? 0 : c->q_pot_rz) * mk;
        int mask8 = c->z_root_m1 >= c->z_soil - c->z_sat;
        c->q_pot_rz = (mask8 ? 0 : c->q_pot_rz) * mk;
    }
    /* calc_percolation_rz :900-968 */
    {
        int mask1 = (c->S_lp_rz < c->q_pot_rz) && (c->z_sat < c->z_soil - c->z_root);
        int mask2 = (c->S_lp_rz >= c->q_pot_rz) && (c->z_sat < c->z_soil - c->z_root);
        int mask3 = c->z_sat >= c->z_soil - c->z_root;
        c->q_rz = c->q_pot_rz * mk;
        c->q_rz = (mask3 ? 0 : c->q_rz) * mk;
        c->S_fp_rz += (mask1 ? -(c->q_rz - c->S_lp_rz) : 0) * mk;
        c->S_lp_rz = (mask1 ? 0 : c->S_lp_rz) * mk;
        c->S_lp_rz += (mask2 ? -c->q_rz : 0) * mk;
        c->S_fp_ss += c->q_rz * mk;
        int m = c->S_fp_ss > c->S_ufc_ss;
        c->S_lp_ss += (m ? c->S_fp_ss - c->S_ufc_ss : 0) * mk;
        c->S_fp_ss = (m ? c->S_ufc_ss : c->S_fp_ss) * mk;
        m = c->S_lp_ss > c->S_ac_ss;
        c->q_rz += (m ? -(c->S_lp_ss - c->S_ac_ss) : 0) * mk;
        c->S_lp_rz += (m ? c->S_lp_ss - c->S_ac_ss : 0) * mk;
        c->S_lp_ss = (m ? c->S_ac_ss : c->S_lp_ss) * mk;
    }
    if (lateral) {
        pot_lateral_cell(c, st, dt);
        lateral_rz_cell(c);
        pot_lateral_ss_cell(c);
    }
    /* calc_potential_percolation_ss :971-1098 (the second perc_pot assignment replaces the first) */
    {
        double z = (c->z_gw * 1000 - c->z_soil) + ((c->z_soil - c->z_root) / 2) * mk;
        double p1 = pow(z / (-c->ha * 10.2), -c->n_salv);
        double p2 = pow(-c->h_ss / -c->ha, -c->n_salv);
        int condB = (c->z_gw <= 10) && (c->z_gw * 1000 > c->z_soil) && (c->z_sat > 0);
        double perc;
        if (condB)
            perc = fmin(fmin(c->kf * dt, c->ks_ss * dt), c->k_ss * dt);
        else
            perc = fmin(c->kf * dt, (p1 - p2) / (1 + p2 + (c->n_salv - 1) * p1) * dt * c->ks_ss * (-1));
        perc = perc * mk;
        int mask1 = (perc > 0) && (c->z_soil < c->z_gw * 1000) && (perc <= c->S_fp_ss + c->S_lp_ss);
        int mask2 = (perc > 0) && (c->z_soil < c->z_gw * 1000) && (perc > c->S_fp_ss + c->S_lp_ss);
        c->q_pot_ss = 0;
        c->q_pot_ss = (mask1 ? perc : c->q_pot_ss) * mk;
        c->q_pot_ss = (mask2 ? c->S_fp_ss + c->S_lp_ss : c->q_pot_ss) * mk;
        double cpr_pot = ((p1 - p2) / (1 + p2 + (c->n_salv - 1) * p1)) * dt * c->ks_ss * mk;
        cpr_pot = ((perc > 0) && (c->z_soil < c->z_gw * 1000) ? 0 : cpr_pot) * mk;
        int mask3 = (c->z_gw * 1000 - c->z_soil > 10000);
        cpr_pot = (mask3 ? 0 : cpr_pot) * mk;
        c->q_pot_ss = (cpr_pot > 0 ? 0 : c->q_pot_ss) * mk;
    }
    if (lateral) {
        lateral_ss_cell(c);
        return;
    }
    /* calc_percolation_ss :1101-1154 */
    {
        c->q_ss = c->q_pot_ss * mk;
        c->z_sat += (c->z_sat > 0 ? -c->q_ss / c->theta_ac : 0) * mk;
        c->z_sat = (c->z_sat < 0 ? 0 : c->z_sat) * mk;
        c->S_zsat_ss = c->z_sat * c->theta_ac * mk;
        int mask1 = c->S_lp_ss < c->q_pot_ss;
        int mask2 = c->S_lp_ss >= c->q_pot_ss;
        c->S_fp_ss += (mask1 ? -(c->q_ss - c->S_lp_ss) : 0) * mk;
        c->S_lp_ss = (mask1 ? 0 : c->S_lp_ss) * mk;
        c->S_lp_ss += (mask2 ? -c->q_ss : 0) * mk;
    }
}

/* ======================================================================== */
/* a11 capillary rise: roger/core/capillary_rise.py:7-173                    */
/* ======================================================================== */
static void capillary_rise_cell(oc_cell *c, double dt) {
    const double mk = (double)c->maskCatch;
    double z = ((c->z_root + (c->z_soil - c->z_root) / 2) - c->z_root / 2) * mk;
    double p1 = pow(z / (-c->ha * 10.2), -c->n_salv);
    double p2 = pow(-c->h_rz / -c->ha, -c->n_salv);
    c->cpr_rz = (p1 - p2) / (1 + p2 + (c->n_salv - 1) * p1) * dt * c->ks * mk;
    c->cpr_rz = (c->cpr_rz < 0 ? 0 : c->cpr_rz) * mk;
    c->cpr_rz = (isnan(c->cpr_rz) ? 0 : c->cpr_rz) * mk;
    c->cpr_rz = (c->S_lp_rz > 0 ? 0 : c->cpr_rz) * mk;
    c->cpr_rz = (c->h_rz > c->h_ss ? 0 : c->cpr_rz) * mk;
    c->cpr_rz = (c->cpr_rz > (c->S_fp_ss + c->S_lp_ss) ? c->S_fp_ss + c->S_lp_ss : c->cpr_rz) * mk;
    c->cpr_rz =
        ((c->cpr_rz > c->S_ufc_rz - c->S_fp_rz) && (c->S_ufc_rz - c->S_fp_rz > 0) ? c->S_ufc_rz - c->S_fp_rz : c->cpr_rz) *
        mk;
    int geo = (c->z_wf < c->z_root) || (c->z_sat < c->z_soil - c->z_root);
    int mask1 = (c->cpr_rz > 0) && (c->S_lp_ss <= 0) && geo;
    int mask2 = (c->cpr_rz > 0) && (c->S_lp_ss > 0) && (c->cpr_rz <= c->S_lp_ss) && geo;
    int mask3 = (c->cpr_rz > 0) && (c->S_lp_ss > 0) && (c->cpr_rz > c->S_lp_ss) && geo;
    c->S_fp_rz += (mask1 ? c->cpr_rz : 0) * mk;
    c->S_fp_ss += (mask1 ? -c->cpr_rz : 0) * mk;
    c->S_fp_rz += (mask2 ? c->cpr_rz : 0) * mk;
    c->S_lp_ss += (mask2 ? -c->cpr_rz : 0) * mk;
    c->S_fp_rz += (mask3 ? c->cpr_rz : 0) * mk;
    c->S_fp_ss += (mask3 ? -(c->cpr_rz - c->S_lp_ss) : 0) * mk;
    c->S_lp_ss = (mask3 ? 0 : c->S_lp_ss) * mk;
    int mask4 = c->S_fp_rz > c->S_ufc_rz;
    c->S_lp_rz += (mask4 ? (c->S_fp_rz - c->S_ufc_rz) : 0) * mk;
    c->S_fp_rz = (mask4 ? c->S_ufc_rz : c->S_fp_rz) * mk;
}

/* ======================================================================== */
/* a12 storage updates: surface.py:8-37, root_zone.py:7-166,                 */
/*     subsoil.py:6-137, soil.py:9-140, numerics.py:125-214                  */
/* ======================================================================== */
static void storage_cell(oc_cell *c, int64_t month_tau) {
    const double mk = (double)c->maskCatch;
    /* surface.calc_S */
    c->S_sur = (c->S_int_top + c->S_int_ground + c->S_dep + c->S_snow + c->z0) * mk;
    /* root zone */
    c->S_rz = (c->S_pwp_rz + c->S_fp_rz + c->S_lp_rz) * mk;
    c->dS_rz = (c->S_rz - c->S_rz_m1) * mk;
    c->theta_rz = ((c->S_fp_rz + c->S_lp_rz) / c->z_root + c->theta_pwp) * mk;
    if (month_tau >= 4 && month_tau <= 9) {
        double d = c->theta_irr - c->theta_rz;
        d = (d <= 0 ? 0 : d);
        c->irr_demand = d * c->z_root;
    } else {
        c->irr_demand = 0;
    }
    c->k_rz = (c->ks / (1 + pow(c->theta_rz / c->theta_sat, -c->m_bc))) * mk;
    c->h_rz = (c->ha / pow(c->theta_rz / c->theta_sat, 1 / c->lambda_bc)) * mk;
    /* subsoil */
    c->S_ss = (c->S_pwp_ss + c->S_fp_ss + c->S_lp_ss) * mk;
    c->dS_ss = (c->S_ss - c->S_ss_m1) * mk;
    c->theta_ss = ((c->S_fp_ss + c->S_lp_ss) / (c->z_soil - c->z_root) + c->theta_pwp) * mk;
    c->ks_ss = c->ks; /* calc_ks, soil compaction off */
    c->k_ss = (c->ks / (1 + pow(c->theta_ss / c->theta_sat, -c->m_bc))) * mk;
    c->h_ss = (c->ha / pow(c->theta_ss / c->theta_sat, 1 / c->lambda_bc)) * mk;
    /* soil */
    c->S_fp_s = (c->S_fp_rz + c->S_fp_ss) * mk;
    c->S_lp_s = (c->S_lp_rz + c->S_lp_ss) * mk;
    c->S_s = (c->S_pwp_s + c->S_fp_s + c->S_lp_s) * mk;
    c->dS_s = (c->S_s - c->S_s_m1) * mk;
    c->theta = ((c->S_fp_s + c->S_lp_s) / c->z_soil + c->theta_pwp) * mk;
    c->k = (c->ks / (1 + pow(c->theta / c->theta_sat, -c->m_bc))) * mk;
    c->h = (c->ha / pow(c->theta / c->theta_sat, 1 / c->lambda_bc)) * mk;
    /* numerics.calc_storage_kernel */
    c->S = c->S_sur + c->S_s * mk;
    c->dS = c->S - c->S_m1 * mk;
}

/* ======================================================================== */
/* unidirectional (D8) routing, settings.enable_routing_1D                    */
/*   surface_runoff.calc_surface_runoff_routing_1D :14-227                    */
/*   subsurface_runoff.calc_subsurface_runoff_routing_1D :1158-1437           */
/* ======================================================================== */
/* the reference's direction order of the *_d8 arrays: N, NE, E, SE, S, SW, W, NW; flow_dir_topo codes and where the water
 * goes in (x, y) index space (`at[2:-2, 1:-3, 0]` for north: y - 1; `at[3:-1, 2:-2, 2]` for east: x + 1) */
static const int D8_CODE[8] = {64, 128, 1, 2, 4, 8, 16, 32};
static const int D8_DX[8] = {0, -1, 1, 1, 0, -1, -1, -1};
static const int D8_DY[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
/* q_out of a cell: sum over the eight directions of where(flow_dir == code_d, q, 0) * maskCatch -- at most one term */
static double d8_out(double q, int flow_dir, double mk) {
    for (int d = 0; d < 8; ++d)
        if (flow_dir == D8_CODE[d]) return (q * mk) * mk;
    return 0.0 * mk;
}
/* q_in of cell (ix, iy): np.sum over the eight *_in_d8 entries, in_d8[c, d] = where(mask_d[s], out_d8[s, d], 0) * maskCatch[s] with
 * s = c - (dx_d, dy_d) an INTERIOR cell (the slices only read [2:-2, 2:-2]); a sum of 8 contiguous values is numpy's unrolled
 * pairwise block ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7)) */
/* a decomposed run (x split over ranks): the neighbour ranks' edge columns, ny values each, side 0 = the column x = -1, side 1 =
 * x = nx; NULL where the rank has no neighbour.  Set by the driver before oc_route_in (tests/oracle_context.py). */
static const double *HALO_Q[2];
static const int32_t *HALO_FD[2], *HALO_MK[2];
void oc_route_set_halo(int side, const double *q, const int32_t *flow_dir, const int32_t *mask) {
    HALO_Q[side] = q;
    HALO_FD[side] = flow_dir;
    HALO_MK[side] = mask;
}
static double d8_in(const double *q_out, const int32_t *flow_dir, const int32_t *mask, int64_t nx, int64_t ny, int64_t ix, int64_t iy) {
    double a[8];
    for (int d = 0; d < 8; ++d) {
        const int64_t sx = ix - D8_DX[d], sy = iy - D8_DY[d];
        a[d] = 0.0;
        if (sy < 0 || sy >= ny) continue;
        if (sx < 0 || sx >= nx) {
            const int side = sx < 0 ? 0 : 1;
            if (HALO_Q[side]) a[d] = (HALO_FD[side][sy] == D8_CODE[d] ? HALO_Q[side][sy] : 0.0) * (double)HALO_MK[side][sy];
            continue;
        }
        const int64_t s = sx * ny + sy;
        a[d] = (flow_dir[s] == D8_CODE[d] ? q_out[s] : 0.0) * (double)mask[s];
    }
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
static void route_surface_out_cell(oc_cell *c, const oc_settings *st, double dt_secs) {
    const double mk = (double)c->maskCatch;
    c->z0 += c->q_sof * mk;
    const double area = (c->z0 / 1000) * 0.5 * (2 * st->dx) * mk;
    const double perimeter = 2 * (c->z0 / 1000) + st->dx * mk;
    const double radius = area / perimeter * mk;
    /* surface runoff (m3/s to mm/dt) */
    c->q_sur = c->k_st * pow(c->slope, 0.5) * pow(radius, 2.0 / 3.0) * area * (dt_secs / (st->dx * st->dy * 1000)) * mk;
    c->q_sur = (c->q_sur > c->z0 ? c->z0 : c->q_sur) * mk;
    c->q_sur_out = d8_out(c->q_sur, c->flow_dir_topo, mk);
}
static void route_surface_in_cell(oc_cell *c) {
    const double mk = (double)c->maskCatch;
    c->q_sur_in = c->q_sur_in * mk;
    c->q_sur_in = (c->outer_boundary == 1 ? 0 : c->q_sur_in) * mk;
    c->z0 += -c->q_sur_out * mk;
    c->z0 += c->q_sur_in * mk;
}
static void route_subsurface_out_cell(oc_cell *c) {
    c->q_sub_out = d8_out(c->q_sub, c->flow_dir_topo, (double)c->maskCatch);
}
static void route_subsurface_in_cell(oc_cell *c) { /* :1311-1437 */
    const double mk = (double)c->maskCatch;
    const double S1_rz = c->S_fp_rz + c->S_lp_rz, S1_ss = c->S_fp_ss + c->S_lp_ss;
    c->q_sub_in = c->q_sub_in * mk;
    c->q_sub_in = (c->outer_boundary == 1 ? 0 : c->q_sub_in) * mk;
    c->z_sat += (c->q_sub_in / c->theta_ac) * mk;
    c->z_sat = (c->z_sat < 0 ? 0 : c->z_sat) * mk;
    c->S_zsat = c->z_sat * c->theta_ac * mk;
    c->S_lp_ss += c->q_sub_in * mk;
    const int over = c->S_lp_ss > c->S_ac_ss;
    c->S_lp_rz += (over ? c->S_lp_ss - c->S_ac_ss : 0) * mk;
    c->S_lp_ss = (over ? c->S_ac_ss : c->S_lp_ss) * mk;
    /* saturation overland flow */
    c->q_sof += (((c->S_lp_rz + c->S_fp_rz) > (c->S_ac_rz + c->S_ufc_rz)) ? (c->S_lp_rz + c->S_fp_rz) - (c->S_ac_rz + c->S_ufc_rz) : 0) * mk;
    c->q_sur += c->q_sof * mk;
    c->z0 += c->q_sof * mk;
    const int sof = c->q_sof > 0;
    c->S_fp_rz = (sof ? c->S_ufc_rz : c->S_fp_rz) * mk;
    c->S_lp_rz = (sof ? c->S_ac_rz : c->S_lp_rz) * mk;
    c->q_sub_in_rz = (c->S_fp_rz + c->S_lp_rz) - S1_rz;
    c->q_sub_in_ss = (c->S_fp_ss + c->S_lp_ss) - S1_ss;
}

/* a13 numerics.calc_dS_num_error :303-345 and sanity_check :979-1011 */
static int np_isclose(double a, double b, double atol, double rtol) {
    if (isfinite(a) && isfinite(b)) return fabs(a - b) <= atol + rtol * fabs(b);
    return a == b;
}
static double nan0(double x) { return isnan(x) ? 0 : x; }

static int num_error_cell(oc_cell *c, const oc_settings *st) {
    if (st->enable_lateral_flow && st->enable_routing_1D) { /* numerics.py:247-270 */
        c->dS_num_error = fabs((c->S - c->S_m1) - (c->prec - c->q_sur_out + c->q_sur_in - c->aet - c->q_ss - c->q_sub_out + c->q_sub_in));
        return 0;
    }
    if (st->enable_lateral_flow) { /* numerics.py:226-245: only dS_num_error in this branch */
        c->dS_num_error = fabs((c->S - c->S_m1) - (c->prec - c->q_sur - c->aet - c->q_ss - c->q_sub));
        return 0;
    }
    c->dS_num_error = fabs((c->S - c->S_m1) - (c->prec - c->q_sur - c->aet - c->q_ss));
    c->dS_rz_num_error = fabs((c->S_rz - c->S_rz_m1) - (c->inf_mat_rz + c->inf_mp_rz + c->inf_sc_rz + c->cpr_rz -
                                                        c->transp - c->evap_soil - c->q_rz));
    c->dS_ss_num_error = fabs((c->S_ss - c->S_ss_m1) - (c->inf_mp_ss + c->q_rz - c->q_ss - c->cpr_rz));
    return 0;
}
static int sanity_cell(const oc_cell *c, const oc_settings *st) {
    double rhs = st->enable_lateral_flow ? c->prec - c->q_sur - c->aet - c->q_ss - c->q_sub /* numerics.py:744-759 */
                                         : c->prec - c->q_sur - c->aet - c->q_ss;
    if (st->enable_lateral_flow && st->enable_routing_1D) /* numerics.py:778-798 */
        rhs = c->prec - c->q_sur_out + c->q_sur_in - c->aet - c->q_ss - c->q_sub_out + c->q_sub_in;
    int check1 = c->maskCatch ? np_isclose(c->S - c->S_m1, rhs, st->atol, st->rtol) : 1;
    int check2 = (nan0(c->S_fp_rz) > -st->atol) && (nan0(c->S_lp_rz) > -st->atol) && (nan0(c->S_fp_ss) > -st->atol) &&
                 (nan0(c->S_lp_ss) > -st->atol);
    int check3 = (nan0(c->S_fp_rz) - st->atol <= nan0(c->S_ufc_rz)) && (nan0(c->S_lp_rz) - st->atol <= nan0(c->S_ac_rz)) &&
                 (nan0(c->S_fp_ss) - st->atol <= nan0(c->S_ufc_ss)) && (nan0(c->S_lp_ss) - st->atol <= nan0(c->S_ac_ss));
    return check1 && check2 && check3;
}

/* a14 after_timestep_kernel: roger/models/svat/svat.py:187-384 */
static double snap0(double x) { return ((x > -1e-6) && (x < 0)) ? 0 : x; }
static void after_timestep_cell(oc_cell *c, int snap) {
    c->ta_m1 = c->ta;
    c->z_root_m1 = c->z_root;
    c->ground_cover_m1 = c->ground_cover;
    c->S_sur_m1 = c->S_sur;
    c->S_int_top_m1 = c->S_int_top;
    c->S_int_ground_m1 = c->S_int_ground;
    c->S_dep_m1 = c->S_dep;
    c->S_snow_m1 = c->S_snow;
    c->swe_m1 = c->swe;
    c->S_rz_m1 = c->S_rz;
    c->S_ss_m1 = c->S_ss;
    c->S_s_m1 = c->S_s;
    c->S_m1 = c->S;
    c->z_sat_m1 = c->z_sat;
    c->z_wf_m1 = c->z_wf;
    c->z_wf_t0_m1 = c->z_wf_t0;
    c->z_wf_t1_m1 = c->z_wf_t1;
    c->y_mp_m1 = c->y_mp;
    c->y_sc_m1 = c->y_sc;
    c->theta_rz_m1 = c->theta_rz;
    c->theta_ss_m1 = c->theta_ss;
    c->theta_m1 = c->theta;
    c->k_rz_m1 = c->k_rz;
    c->k_ss_m1 = c->k_ss;
    c->k_m1 = c->k;
    c->h_rz_m1 = c->h_rz;
    c->h_ss_m1 = c->h_ss;
    c->h_m1 = c->h;
    c->z0_m1 = c->z0;
    if (snap) { /* models/svat/svat.py:326-345; the oneD model's kernel has no such lines */
        c->S_fp_rz = snap0(c->S_fp_rz);
        c->S_lp_rz = snap0(c->S_lp_rz);
        c->S_fp_ss = snap0(c->S_fp_ss);
        c->S_lp_ss = snap0(c->S_lp_ss);
    }
    c->prec_m1 = c->prec;
}

/* ======================================================================== */
/* a15 monthly surface parameters: roger/core/surface.py:74-343               */
/* ======================================================================== */
typedef struct oc_luts {
    const double *ilu; /* (25,13) */
    const double *gc;  /* (25,13) */
    const double *gcm; /* (25,2)  */
    const double *rdlu; /* (25,7) */
} oc_luts;

static int lut_row(const double *lut, int ncol, int i) { /* utilities._get_row_no: first match, else 0 */
    for (int r = 0; r < 25; ++r)
        if (lut[r * ncol] == (double)i) return r;
    return 0;
}
static int in_list(int v, const int *l, int n) {
    for (int i = 0; i < n; ++i)
        if (l[i] == v) return 1;
    return 0;
}

static void params_surface_cell(oc_cell *c, const oc_luts *L, int64_t month_tau) {
    const double mk = (double)c->maskCatch;
    const int lu = c->lu_id;
    static const int trees[] = {10, 11, 12, 15, 17};
    static const int ground[] = {0, 5, 6, 7, 8, 9, 13, 98, 31, 32, 33, 40, 41, 50, 60, 98};
    static const int trees_ground[] = {10, 11, 12, 15, 16};
    static const int cc[] = {0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 98, 31, 32, 33, 40, 41, 50, 60, 90, 98};
    int m = (int)month_tau;
    /* each for_loop iteration multiplies by mk; with mk in {0,1} only the value matters */
    double v = 0;
    if (lu >= 10 && lu < 16 && in_list(lu, trees, 5)) v = L->ilu[lut_row(L->ilu, 13, lu) * 13 + m];
    v = v * mk;
    c->S_int_top_tot = v * c->c_int * mk;
    v = 0;
    if (lu >= 0 && lu < 81 && in_list(lu, ground, 16)) v = L->ilu[lut_row(L->ilu, 13, lu) * 13 + m];
    v = v * mk;
    if (lu >= 10 && lu < 16 && in_list(lu, trees_ground, 5)) v = 1;
    v = v * mk;
    c->S_int_ground_tot = v * c->c_int * mk;
    int is_cc = (lu >= 0 && lu < 81 && in_list(lu, cc, 21));
    int row = lut_row(L->gc, 13, lu);
    double gcv = L->gc[row * 13 + m], gcm1 = L->gcm[row * 2 + 1];
    v = is_cc ? gcv : 0;
    v = v * mk;
    c->ground_cover = v * mk;
    v = is_cc ? gcv / gcm1 : 0;
    v = v * mk;
    v = (c->maskRiver || c->maskLake) ? 0 : v;
    c->basal_transp_coeff = v * mk;
    v = is_cc ? 1 - ((gcv / gcm1) * gcm1) : 0;
    v = v * mk;
    /* `maskRiver | maskLake | lu_id == 0` parses as ((maskRiver | maskLake | lu_id) == 0), surface.py:230 */
    v = ((((c->maskRiver ? 1 : 0) | (c->maskLake ? 1 : 0) | lu) == 0) ? 1 : v);
    c->basal_evap_coeff = v * mk;
    c->swe_top_tot = swe_top_tot_of(c->swe_top_tot, c->ta, lu, mk);
    c->lai = log(1 / (1 - c->ground_cover)) / log(1 / 0.7) * mk;
    c->throughfall_coeff_top = ((lu == 10 || lu == 11 || lu == 12) ? (c->lai > 1 ? 0.1 : 1.0 - c->lai) : 0) * mk;
    c->throughfall_coeff_ground = ((lu >= 500 && lu < 598) ? (c->lai > 1 ? 0.1 : 1.0 - c->lai) : 0) * mk;
}

/* calc_topo_kernel: surface.py:40-71 */
static void topo_cell(oc_cell *c) {
    c->maskRiver = (c->lu_id == 20);
    c->maskLake = (c->lu_id == 14);
    c->maskCatch = (c->lu_id != 14) && (c->lu_id != 20) && (c->lu_id != 999) && c->maskCatch;
}

/* ======================================================================== */
/* a16 setup-time soil parameters and initial conditions: soil.py:143-1010,  */
/*     surface.py:398-427                                                    */
/* ======================================================================== */
static void params_soil_cell(oc_cell *c, const oc_luts *L, const oc_settings *st) {
    const double mk = (double)c->maskCatch;
    /* calc_parameters_soil_kernel :143-283 */
    c->S_ac_s = (c->z_soil * c->theta_ac) * mk;
    c->S_ufc_s = (c->z_soil * c->theta_ufc) * mk;
    c->S_pwp_s = (c->z_soil * c->theta_pwp) * mk;
    c->S_fc_s = (c->z_soil * (c->theta_ufc + c->theta_pwp)) * mk;
    c->S_sat_s = (c->z_soil * (c->theta_ac + c->theta_ufc + c->theta_pwp)) * mk;
    c->theta_sat = (c->theta_ac + c->theta_ufc + c->theta_pwp) * mk;
    c->theta_fc = (c->theta_ufc + c->theta_pwp) * mk;
    c->lambda_bc = ((log(c->theta_fc / c->theta_sat) - log(c->theta_pwp / c->theta_sat)) / (log(15850) - log(63))) * mk;
    c->ha = (pow(c->theta_pwp / c->theta_sat, 1.0 / c->lambda_bc) * (-15850)) * mk;
    c->m_bc = ((st->a_bc + st->b_bc * c->lambda_bc) / c->lambda_bc) * mk;
    c->n_salv = (st->a_bc + st->b_bc * c->lambda_bc) * mk;
    c->wfs = (((2 + 3 * c->lambda_bc) / (1 + 3 * c->lambda_bc) * c->ha / 2) * (-10)) * mk;
    c->theta_27 = (pow(c->ha / (-pow(10, 2.7)), c->lambda_bc) * c->theta_sat) * mk;
    c->theta_4 = (pow(c->ha / (-(10000.0)), c->lambda_bc) * c->theta_sat) * mk;
    c->theta_6 = (pow(c->ha / (-(1000000.0)), c->lambda_bc) * c->theta_sat) * mk;
    c->sand = (1 * (c->theta_ac / 0.24)) * mk;
    c->sand = (c->sand < 0 ? 0 : c->sand) * mk;
    c->sand = (c->sand > 1 ? 1 : c->sand) * mk;
    c->clay = (st->clay_max * (c->theta_6 - st->clay_min) / 0.3) * mk;
    c->clay = (c->clay < st->clay_min ? st->clay_min : c->clay) * mk;
    c->z_sc_max = (c->clay * 700) * mk;
    c->mp_drain_area = 1 - exp((-1) * pow(c->dmpv / 82, 0.887)) * mk;
    /* calc_parameters_root_zone_kernel :286-470 */
    int mask1 = c->theta_pwp < st->theta_rew_min;
    int mask2 = (c->theta_pwp >= st->theta_rew_min) && (c->theta_pwp <= st->theta_rew_max);
    int mask3 = c->theta_pwp > st->theta_rew_max;
    c->rew = (mask1 ? st->rew_min : c->rew) * mk;
    c->rew = (mask2 ? c->theta_pwp / st->theta_rew_max : c->rew) * mk;
    c->rew = (mask3 ? st->rew_max : c->rew) * mk;
    c->z_evap = ((c->rew / st->rew_max) * st->z_evap_max) * mk;
    c->tew = ((c->theta_fc - 0.5 * c->theta_pwp) * c->z_evap) * mk;
    {
        static const int cc[] = {0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 98, 31, 32, 33, 40, 41, 50, 60, 98};
        int lu = c->lu_id;
        double zr = c->z_root_m1; /* vs.z_root[..., 0] */
        if (lu >= 0 && lu < 61 && in_list(lu, cc, 20)) zr = L->rdlu[lut_row(L->rdlu, 7, lu) * 7 + 1];
        zr = zr * mk;
        zr = (c->maskRiver || c->maskLake) ? 0 : zr;
        zr = ((lu == 10 || lu == 11 || lu == 12 || lu == 15 || lu == 16 || lu == 17) ? 1500 : zr) * mk;
        zr = (lu == 100 ? 300 : zr) * mk;
        zr = (zr >= c->z_soil ? st->zroot_to_zsoil_max * c->z_soil : zr) * mk;
        c->z_root_m1 = zr * c->c_root;
        c->z_root = zr * c->c_root;
        int crops = (lu >= 500 && lu < 600);
        c->z_root_m1 = (crops ? 200 : c->z_root_m1) * mk;
        c->z_root = (crops ? 200 : c->z_root) * mk;
        c->z_root_m1 = (c->z_root_m1 < c->z_soil ? c->z_root_m1 : c->z_soil * 0.9);
        c->z_root = (c->z_root < c->z_soil ? c->z_root : c->z_soil * 0.9);
    }
    c->S_ac_rz = (c->theta_ac * c->z_root) * mk;
    c->S_ufc_rz = (c->theta_ufc * c->z_root) * mk;
    c->S_pwp_rz = (c->theta_pwp * c->z_root) * mk;
    c->S_sat_rz = ((c->theta_ac + c->theta_ufc + c->theta_pwp) * c->z_root) * mk;
    c->S_fc_rz = ((c->theta_ufc + c->theta_pwp) * c->z_root) * mk;
    /* calc_parameters_subsoil_kernel (no compaction) */
    double dz = c->z_soil - c->z_root;
    c->S_ac_ss = (c->theta_ac * dz) * mk;
    c->S_ufc_ss = (c->theta_ufc * dz) * mk;
    c->S_pwp_ss = (c->theta_pwp * dz) * mk;
    c->S_sat_ss = ((c->theta_ac + c->theta_ufc + c->theta_pwp) * dz) * mk;
    c->S_fc_ss = ((c->theta_ufc + c->theta_pwp) * dz) * mk;
}

/* calc_parameters_lateral_flow_kernel: soil.py:560-641.  lut_mlms is (10000, 9): slope in percent,
 * then the macropore flow velocity (m/h) of layers 8..1 */
static void params_lateral_cell(oc_cell *c, const double *mlms, int64_t nrows, int max_slope_per) {
    const double mk = (double)c->maskCatch;
    double *V[8] = {&c->v_mp_layer_1, &c->v_mp_layer_2, &c->v_mp_layer_3, &c->v_mp_layer_4,
                    &c->v_mp_layer_5, &c->v_mp_layer_6, &c->v_mp_layer_7, &c->v_mp_layer_8};
    int key = c->slope_per;
    int hit = (key >= 1) && (key <= max_slope_per);
    int64_t row = 0; /* utilities._get_row_no: first match, else row 0 */
    if (hit)
        for (int64_t r = 0; r < nrows; ++r)
            if (mlms[r * 9] == (double)key) { row = r; break; }
    for (int k = 0; k < 8; ++k) {
        double v = hit ? mlms[row * 9 + (8 - k)] * 1000 : 0.0;
        *V[k] = (v * mk) * mk;
    }
}

static void initial_conditions_cell(oc_cell *c) {
    const double mk = (double)c->maskCatch;
    /* surface.calc_initial_conditions_surface_kernel :398-414 */
    c->S_sur = (c->S_int_top + c->S_int_ground + c->S_dep + c->S_snow) * mk;
    c->S_sur_m1 = (c->S_int_top_m1 + c->S_int_ground_m1 + c->S_dep_m1 + c->S_snow_m1) * mk;
    /* root zone, soil.py:765-845 */
    c->theta_fp_rz = (c->theta_rz > c->theta_pwp ? c->theta_rz - c->theta_pwp : c->theta_fp_rz) * mk;
    c->theta_fp_rz = (c->theta_rz <= c->theta_pwp ? 0 : c->theta_fp_rz) * mk;
    c->theta_fp_rz = (c->theta_fp_rz >= c->theta_ufc ? c->theta_ufc : c->theta_fp_rz) * mk;
    c->theta_lp_rz = (c->theta_rz > c->theta_fc ? c->theta_rz - c->theta_fc : c->theta_lp_rz) * mk;
    c->theta_lp_rz = (c->theta_rz <= c->theta_fc ? 0 : c->theta_lp_rz) * mk;
    c->S_fp_rz = (c->theta_fp_rz * c->z_root) * mk;
    c->S_lp_rz = (c->theta_lp_rz * c->z_root) * mk;
    c->S_rz = c->S_rz_m1 = (c->S_pwp_rz + c->S_fp_rz + c->S_lp_rz) * mk;
    c->theta_rz = ((c->S_fp_rz + c->S_lp_rz) / c->z_root + c->theta_pwp) * mk;
    c->k_rz = (c->ks / (1 + pow(c->theta_rz / c->theta_sat, -c->m_bc))) * mk;
    c->h_rz = (c->ha / pow(c->theta_rz / c->theta_sat, 1 / c->lambda_bc)) * mk;
    /* subsoil, soil.py:848-935 */
    c->theta_fp_ss = (c->theta_ss > c->theta_pwp ? c->theta_ss - c->theta_pwp : c->theta_fp_ss) * mk;
    c->theta_fp_ss = (c->theta_ss <= c->theta_pwp ? 0 : c->theta_fp_ss) * mk;
    c->theta_fp_ss = (c->theta_fp_ss >= c->theta_ufc ? c->theta_ufc : c->theta_fp_ss) * mk;
    c->theta_lp_ss = (c->theta_ss > c->theta_fc ? c->theta_ss - c->theta_fc : c->theta_lp_ss) * mk;
    c->theta_lp_ss = (c->theta_ss <= c->theta_fc ? 0 : c->theta_lp_ss) * mk;
    c->S_fp_ss = (c->theta_fp_ss * (c->z_soil - c->z_root)) * mk;
    c->S_lp_ss = (c->theta_lp_ss * (c->z_soil - c->z_root)) * mk;
    c->S_ss = c->S_ss_m1 = (c->S_pwp_ss + c->S_fp_ss + c->S_lp_ss) * mk;
    c->theta_ss = ((c->S_fp_ss + c->S_lp_ss) / (c->z_soil - c->z_root) + c->theta_pwp) * mk;
    c->k_ss = (c->ks / (1 + pow(c->theta_ss / c->theta_sat, -c->m_bc))) * mk;
    c->h_ss = (c->ha / pow(c->theta_ss / c->theta_sat, 1 / c->lambda_bc)) * mk;
    /* soil, soil.py:742-762 */
    c->S_fp_s = (c->S_fp_rz + c->S_fp_ss) * mk;
    c->S_lp_s = (c->S_lp_rz + c->S_lp_ss) * mk;
    c->S_s = (c->S_rz + c->S_ss) * mk;
    c->S_s_m1 = (c->S_rz_m1 + c->S_ss_m1) * mk;
    c->theta = (c->S_s / c->z_soil) * mk;
    c->theta_m1 = (c->S_s_m1 / c->z_soil) * mk;
    /* calc_initial_conditions_kernel :938-948 */
    c->S = c->S_sur + c->S_s * mk;
    c->S_m1 = c->S_sur_m1 + c->S_s_m1 * mk;
}

/* ======================================================================== */
/* drivers over the SoA planes                                               */
/* ======================================================================== */
/* Columns are independent: with OpenMP (the default build) the cell loops run on all host threads once a grid is large
 * enough to pay for the fork (the cpu_baseline of bench.py; the golden-vector tests stay single-threaded).  The
 * results do not depend on the thread count: every column is computed by one thread, the flags are AND / OR reductions. */
#define OC_PRAGMA(x) _Pragma(#x)
#define FOR_CELLS_(clause, body)                                                        \
    OC_PRAGMA(omp parallel for schedule(static) if (n >= OC_PAR_MIN) clause)            \
    for (int64_t i = 0; i < n; ++i) {                                                   \
        oc_cell cell;                                                                   \
        oc_cell *c = &cell;                                                             \
        gather(c, planes, i);                                                           \
        body;                                                                           \
        scatter(c, planes, i);                                                          \
    }
#define FOR_CELLS(body) FOR_CELLS_(, body)
#define FOR_CELLS_OK(body) FOR_CELLS_(reduction(& : ok), body)

int oc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oc_set_num_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

void oc_interception(void *const *planes, int64_t n, const oc_settings *st) { FOR_CELLS(interception_cell(c, st)) }
void oc_evapotranspiration(void *const *planes, int64_t n, const oc_settings *st) {
    FOR_CELLS(evapotranspiration_cell(c, st))
}
void oc_snow(void *const *planes, int64_t n, const oc_scalars *s, const oc_settings *st) {
    FOR_CELLS(snow_cell(c, st, s->dt))
}

/* calculate_infiltration: infiltration.py:2148-2193 (host-side branching on global predicates) */
typedef struct {
    int cond1, cond2, cond3, cond4, cond5;
} inf_conds;

static inf_conds infiltration_conds_from(int any_p0, int any_pm1_n0, int any_pn0, int any_pm1_0, const oc_scalars *s) {
    inf_conds k;
    k.cond1 = (s->event_id[0] == 0) && (s->event_id[1] >= 1);
    k.cond2 = any_p0 && any_pm1_n0 && (s->event_id[0] >= 1);
    k.cond3 = any_pn0 && any_pm1_0 && (s->event_id[0] == s->event_id[1]);
    k.cond4 = (s->event_id[0] >= 1) && (s->event_id[1] == 0);
    k.cond5 = s->event_id[1] >= 1;
    return k;
}
static inf_conds infiltration_conds(void *const *planes, int64_t n, const oc_scalars *s) {
    const double *prec = planes[plane_index("prec")], *prec_m1 = planes[plane_index("prec_m1")];
    int any_p0 = 0, any_pm1_n0 = 0, any_pn0 = 0, any_pm1_0 = 0;
    for (int64_t i = 0; i < n; ++i) {
        any_p0 |= (prec[i] == 0);
        any_pm1_n0 |= (prec_m1[i] != 0);
        any_pn0 |= (prec[i] != 0);
        any_pm1_0 |= (prec_m1[i] == 0);
    }
    return infiltration_conds_from(any_p0, any_pm1_n0, any_pn0, any_pm1_0, s);
}

static void infiltration_cell(oc_cell *c, const oc_settings *st, double dt, inf_conds k) {
    if (k.cond1) {
        depth_shrinkage_cracks_cell(c);
        set_event_vars_cell(c);
    }
    if (k.cond2) start_rainfall_pause_cell(c);
    if (k.cond3) end_rainfall_pause_cell(c);
    if (k.cond5) c->t_event_csum += dt;
    green_ampt_params_cell(c, dt);
    inf_mat_cell(c, dt);
    inf_mp_cell(c, st, dt);
    inf_sc_cell(c, st, dt);
    inf_rz_cell(c);
    hof_sof_cell(c);
    if (!st->enable_routing_1D) surface_runoff_cell(c); /* infiltration.py:2189-2190 */
    if (k.cond4) reset_event_vars_cell(c);
}

void oc_infiltration(void *const *planes, int64_t n, const oc_scalars *s, const oc_settings *st) {
    inf_conds k = infiltration_conds(planes, n, s);
    FOR_CELLS(infiltration_cell(c, st, s->dt, k))
}
void oc_subsurface_runoff(void *const *planes, int64_t n, const oc_scalars *s, const oc_settings *st) {
    FOR_CELLS(subsurface_runoff_cell(c, s->dt, st))
}
/* the gathers of the routing: q_in of every cell from its eight neighbours' q_out (whole local grid; a decomposed run would need the
 * neighbour rank's edge column -- the reference itself never exchanges it, roger/core/utilities.py:79 is not called on this path) */
static void route_gather(void *const *planes, const oc_settings *st, const char *src, const char *dst) {
    const double *q_out = planes[plane_index(src)];
    double *q_in = planes[plane_index(dst)];
    const int32_t *fd = planes[plane_index("flow_dir_topo")], *mk = planes[plane_index("maskCatch")];
    for (int64_t ix = 0; ix < st->nx; ++ix)
        for (int64_t iy = 0; iy < st->ny; ++iy) q_in[ix * st->ny + iy] = d8_in(q_out, fd, mk, st->nx, st->ny, ix, iy);
}
/* which: 0 surface, 1 subsurface.  out: the per-column outflow; in: the gather (with the halo columns of oc_route_set_halo) and the
 * per-column inflow.  A driver of a decomposed run exchanges the edge columns of q_*_out in between. */
void oc_route_out(void *const *planes, int64_t n, int which, const oc_scalars *s, const oc_settings *st) {
    if (which == 0) { FOR_CELLS(route_surface_out_cell(c, st, (double)s->dt_secs)) }
    else { FOR_CELLS(route_subsurface_out_cell(c)) }
}
void oc_route_in(void *const *planes, int64_t n, int which, const oc_settings *st) {
    if (which == 0) {
        route_gather(planes, st, "q_sur_out", "q_sur_in");
        FOR_CELLS(route_surface_in_cell(c))
    } else {
        route_gather(planes, st, "q_sub_out", "q_sub_in");
        FOR_CELLS(route_subsurface_in_cell(c))
    }
}
/* the exchange of a decomposed run, called between out and in (it reads the edge columns of q_*_out and calls oc_route_set_halo) */
typedef void (*oc_exchange_fn)(int which);
static oc_exchange_fn EXCHANGE;
void oc_route_set_exchange(oc_exchange_fn fn) { EXCHANGE = fn; }
void oc_surface_routing(void *const *planes, int64_t n, const oc_scalars *s, const oc_settings *st) {
    oc_route_out(planes, n, 0, s, st);
    if (EXCHANGE) EXCHANGE(0);
    oc_route_in(planes, n, 0, st);
}
void oc_subsurface_routing(void *const *planes, int64_t n, const oc_settings *st) {
    oc_route_out(planes, n, 1, NULL, st);
    if (EXCHANGE) EXCHANGE(1);
    oc_route_in(planes, n, 1, st);
}
void oc_capillary_rise(void *const *planes, int64_t n, const oc_scalars *s) { FOR_CELLS(capillary_rise_cell(c, s->dt)) }
void oc_storage(void *const *planes, int64_t n, const oc_scalars *s) { FOR_CELLS(storage_cell(c, s->month[1])) }
int oc_num_error(void *const *planes, int64_t n, oc_scalars *s, const oc_settings *st) {
    int ok = 1;
    FOR_CELLS_OK(ok &= sanity_cell(c, st); num_error_cell(c, st))
    s->sanity_ok = ok;
    return ok;
}
void oc_after_timestep(void *const *planes, int64_t n, oc_scalars *s, const oc_settings *st) {
    FOR_CELLS(after_timestep_cell(c, !st->enable_lateral_flow))
    s->event_id[0] = s->event_id[1];
    s->year[0] = s->year[1];
    s->month[0] = s->month[1];
    s->doy[0] = s->doy[1];
}
void oc_params_surface(void *const *planes, int64_t n, const oc_scalars *s, const double *ilu, const double *gc,
                       const double *gcm, const double *rdlu) {
    oc_luts L = {ilu, gc, gcm, rdlu};
    FOR_CELLS(params_surface_cell(c, &L, s->month[1]))
}
void oc_topo(void *const *planes, int64_t n) { FOR_CELLS(topo_cell(c)) }
void oc_params_soil(void *const *planes, int64_t n, const oc_settings *st, const double *ilu, const double *gc,
                    const double *gcm, const double *rdlu) {
    oc_luts L = {ilu, gc, gcm, rdlu};
    FOR_CELLS(params_soil_cell(c, &L, st))
}
void oc_params_lateral(void *const *planes, int64_t n, const double *mlms, int64_t nrows) {
    const int32_t *sp = planes[plane_index("slope_per")];
    int mx = 0;
    for (int64_t i = 0; i < n; ++i) mx = sp[i] > mx ? sp[i] : mx;
    FOR_CELLS(params_lateral_cell(c, mlms, nrows, mx))
}
void oc_initial_conditions(void *const *planes, int64_t n) { FOR_CELLS(initial_conditions_cell(c)) }

/* Everything of a step after the adaptive time stepping, with the infiltration predicates taken
 * from predicate word 1 (global over all ranks in a decomposed run).  `core_only` != 0 stops
 * before after_timestep (the hook-preserving driver runs that separately). */
int oc_step_after_adt(void *const *planes, int64_t n, oc_scalars *s, const oc_settings *st, int monthly, int core_only,
                      uint64_t word1, const double *ilu, const double *gc, const double *gcm, const double *rdlu) {
    oc_luts L = {ilu, gc, gcm, rdlu};
    inf_conds k = infiltration_conds_from(HAS(word1, W1_P_EQ0), HAS(word1, W1_PM1_NE0), HAS(word1, W1_P_NE0),
                                          HAS(word1, W1_PM1_EQ0), s);
    int ok = 1;
    s->itt += 1;
    s->time += s->dt_secs;
    if (st->enable_routing_1D) { /* the routing couples the columns twice per step: three passes over the grid (roger/roger.py:410-447) */
        FOR_CELLS(
            if (monthly) params_surface_cell(c, &L, s->month[1]);
            interception_cell(c, st);
            evapotranspiration_cell(c, st);
            snow_cell(c, st, s->dt);
            infiltration_cell(c, st, s->dt, k);)
        oc_surface_routing(planes, n, s, st);
        FOR_CELLS(subsurface_runoff_cell(c, s->dt, st))
        oc_subsurface_routing(planes, n, st);
        FOR_CELLS_OK(
            capillary_rise_cell(c, s->dt);
            storage_cell(c, s->month[1]);
            ok &= sanity_cell(c, st);
            num_error_cell(c, st);
            if (!core_only) after_timestep_cell(c, !st->enable_lateral_flow);)
        s->sanity_ok = ok;
        if (!core_only) {
            s->event_id[0] = s->event_id[1];
            s->year[0] = s->year[1];
            s->month[0] = s->month[1];
            s->doy[0] = s->doy[1];
        }
        return ok;
    }
    FOR_CELLS_OK(
        if (monthly) params_surface_cell(c, &L, s->month[1]);
        interception_cell(c, st);
        evapotranspiration_cell(c, st);
        snow_cell(c, st, s->dt);
        infiltration_cell(c, st, s->dt, k);
        subsurface_runoff_cell(c, s->dt, st);
        capillary_rise_cell(c, s->dt);
        storage_cell(c, s->month[1]);
        ok &= sanity_cell(c, st);
        num_error_cell(c, st);
        if (!core_only) after_timestep_cell(c, !st->enable_lateral_flow);)
    s->sanity_ok = ok;
    if (!core_only) {
        s->event_id[0] = s->event_id[1];
        s->year[0] = s->year[1];
        s->month[0] = s->month[1];
        s->doy[0] = s->doy[1];
    }
    return ok;
}

/* One full SVAT time step in the order of RogerSetup.step (roger/roger.py:396-485), without the
 * user hooks (set_forcing / set_parameters) which the caller runs before.  `monthly` != 0 runs
 * calc_parameters_surface_kernel first (svat.py:115-120).  Gathers each cell once. */
int oc_step(void *const *planes, int64_t n, const double *prec_day, const double *ta_day, const double *pet_day,
            int64_t fstride, oc_scalars *s, const oc_settings *st, int monthly, const double *ilu, const double *gc,
            const double *gcm, const double *rdlu) {
    uint64_t w0 = oc_adt_pred1(planes, n, prec_day, ta_day, fstride, st);
    uint64_t w1 = oc_adt_select(planes, n, prec_day, ta_day, pet_day, fstride, s, st, w0);
    oc_adt_finish(planes, n, prec_day, ta_day, pet_day, fstride, s, st, w0, w1);
    return oc_step_after_adt(planes, n, s, st, monthly, 0, w1, ilu, gc, gcm, rdlu);
}
