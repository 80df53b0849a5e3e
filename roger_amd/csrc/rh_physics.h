// rh_physics.h -- the per-column process equations of the SVAT step as device functions.
//
// Every function updates one soil column held in registers (`Col &c`); nothing here touches
// memory.  The kernels in roger_hip.hip load exactly the planes a routine mentions and store the
// planes it assigns; those sets are extracted from this file by tools/gen_sets.py, so keep to the
// access style `c.<field>` and to the naming rt_* (routine, gets a kernel) / h_* (helper).
//
// Arithmetic contract: float64, no FMA contraction, IEEE comparisons (the build uses
// -ffp-contract=off and no fast-math).  Update order and masks follow the reference NumPy
// backend (file:line given per function, paths relative to roger/core/), including the spots
// where the reference's own expressions are not what their comments say; see DESIGN.md
// "Faithful quirks".  `mk` is maskCatch as 0.0/1.0 and is applied by multiplication because
// that is what the reference does (NaN * 0 stays NaN).
#pragma once

#include "rh_col.h"

// The power function of the physics: rh_pow (rh_pow.h: ~80 instructions instead of the library's 230, within 1 ulp of a correctly rounded
// pow, the same bits on host and device); -DRH_POW_LIBRARY selects the library's pow (A/B, and what rounds 1 - 2 ran).
#define RH_POW_FN __device__ __forceinline__
#define RH_POW_CONST static __constant__
#include "rh_pow.h"
#ifdef RH_POW_LIBRARY
#define RH_POW(x, y) pow(x, y)
#else
#define RH_POW(x, y) rh_pow(x, y)
#endif

RH_DEV double h_b(bool x) { return x ? 1.0 : 0.0; }

// ---------------------------------------------------------------------------------------------
// interception.py
// ---------------------------------------------------------------------------------------------
// maximum snow interception of conifers, interception.py:170-206 (= surface.py:243-308)
RH_DEV double h_swe_top_tot(double v, double ta, int lu, double mk) {
    const bool warm = ta > -1, mid = (ta >= -3) && (ta <= -1), cold = ta < -3;
    v = (warm && lu == 10 ? 9.0 : v) * mk;
    v = (warm && lu == 11 ? 15.0 : v) * mk;
    v = (warm && lu == 12 ? 25.0 : v) * mk;
    v = (mid && lu == 10 ? 2.5 + 0.5 * ta * 9 : v) * mk;
    v = (mid && lu == 11 ? 2.5 + 0.5 * ta * 15 : v) * mk;
    v = (mid && lu == 12 ? 2.5 + 0.5 * ta * 25 : v) * mk;
    v = (cold && lu == 10 ? 18.0 : v) * mk;
    v = (cold && lu == 11 ? 30.0 : v) * mk;
    v = (cold && lu == 12 ? 50.0 : v) * mk;
    return v;
}

// calculate_interception, interception.py:347-356
RH_DEV void rt_interception(Col &c, const Consts &K) {
    const double mk = (double)c.maskCatch;
    const bool liquid = c.ta > K.ta_fm, frozen = c.ta <= K.ta_fm;
    const double tf_top = 1. - c.throughfall_coeff_top, tf_gr = 1. - c.throughfall_coeff_ground;

    // rain, upper storage :7-71
    c.rain_top = (liquid ? c.prec : 0.0) * mk;
    const double wtmx = (10000. / (100 - K.rmax) / 100.) * c.swe_top;
    const double cap_top = (c.S_int_top_tot < wtmx ? wtmx : c.S_int_top_tot) * mk;
    double room = (c.S_int_top < cap_top ? cap_top - c.S_int_top : 0.0) * mk;
    double want = c.prec * tf_top;
    c.int_rain_top = 0.0 + c.prec * tf_top * h_b((room >= want) && liquid && (room > 0)) * mk;
    c.int_rain_top = ((room < want) && liquid && (room > 0) ? room : c.int_rain_top) * mk;
    c.S_int_top += c.int_rain_top * mk;

    // rain, lower storage :75-151
    const double rain = (c.prec - c.int_rain_top) * h_b(liquid) * mk;
    room = ((c.S_int_ground < c.S_int_ground_tot) && (c.S_snow <= 0) ? c.S_int_ground_tot - c.S_int_ground : 0.0) * mk;
    want = rain * tf_gr;
    c.int_rain_ground = 0.0 + rain * tf_gr * h_b((room >= want) && liquid && (room > 0)) * mk;
    c.int_rain_ground = ((room < want) && liquid && (room > 0) ? room : c.int_rain_ground) * mk;
    c.int_rain_ground = (c.lu_id == 599 ? 0.0 : c.int_rain_ground) * mk;
    c.S_int_ground += c.int_rain_ground * mk;
    c.rain_ground = (c.rain_top - c.int_rain_top - c.int_rain_ground) * mk;
    const double to_ground = (c.S_snow > 0 ? 0.0 : c.rain_ground) * mk;
    c.z0 += to_ground;
    c.prec_event_csum += to_ground;

    // snow, upper storage :155-245
    c.snow_top = (frozen ? c.prec : 0.0) * mk;
    c.swe_top_tot = h_swe_top_tot(c.swe_top_tot, c.ta, c.lu_id, mk);
    room = (c.swe_top >= c.swe_top_tot ? 0.0 : c.swe_top_tot - c.swe_top) * mk;
    want = c.prec * tf_top;
    c.int_snow_top = 0.0 + c.prec * tf_top * h_b((room >= want) && frozen && (room > 0)) * mk;
    c.int_snow_top = ((room < want) && frozen && (room > 0) ? room : c.int_snow_top) * mk;
    c.S_int_top += c.int_snow_top * mk;
    c.swe_top += c.int_snow_top * mk;

    // snow, lower storage :249-318
    const double snow = (c.prec - c.int_snow_top) * h_b(frozen) * mk;
    room = (c.S_int_ground >= c.S_int_ground_tot ? 0.0 : c.S_int_ground_tot - c.S_int_ground) * mk;
    want = snow * tf_gr;
    c.int_snow_ground = 0.0 + snow * tf_gr * h_b((room >= want) && frozen && (room > 0)) * mk;
    c.int_snow_ground = ((room < want) && frozen && (room > 0) ? room : c.int_snow_ground) * mk;
    c.int_snow_ground = (c.lu_id == 599 ? 0.0 : c.int_snow_ground) * mk;
    c.S_int_ground += c.int_snow_ground * mk;
    c.swe_ground += c.int_snow_ground * mk;
    c.snow_ground = (c.snow_top - c.int_snow_top - c.int_snow_ground) * mk;
    c.prec_event_csum += c.snow_ground * mk;

    // totals :322-343
    c.int_top = (c.int_rain_top + c.int_snow_top) * mk;
    c.int_ground = (c.int_rain_ground + c.int_snow_ground) * mk;
    c.int_prec = (c.int_rain_top + c.int_rain_ground + c.int_snow_top + c.int_snow_ground) * mk;
}

// ---------------------------------------------------------------------------------------------
// evapotranspiration.py:9-616
// ---------------------------------------------------------------------------------------------
// Depletes residual PET from one interception store (:10-66 / :70-134); returns the evaporation.
RH_DEV double h_evap_store(double &store, double cap, double &pet_res, double mk) {
    const bool wet = (store <= cap) && (cap > 0) && (store > 0);
    const bool partial = wet && (pet_res <= store), full = wet && (pet_res > store);
    double e = 0.0 + pet_res * h_b(partial) * mk;
    pet_res = (partial ? 0.0 : pet_res) * mk;
    e += store * h_b(full) * mk;
    pet_res += -store * h_b(full) * mk;
    store += -e * mk;
    return e;
}

RH_DEV void rt_evapotranspiration(Col &c, const Consts &K) {
    const double mk = (double)c.maskCatch;
    c.evap_int_top = h_evap_store(c.S_int_top, c.S_int_top_tot, c.pet_res, mk);
    c.evap_int_ground = h_evap_store(c.S_int_ground, c.S_int_ground_tot, c.pet_res, mk);
    c.evap_int = c.evap_int_ground + c.evap_int_top * mk;  // :126-130

    // depression storage :138-194
    {
        const bool on = (c.S_dep > 0) && (c.pet_res > 0) && (c.prec <= 0);
        const bool all = on && (c.S_dep <= c.pet_res), some = on && (c.S_dep > c.pet_res);
        c.evap_dep = 0.0 + c.S_dep * h_b(all) * mk;
        c.pet_res += -c.S_dep * h_b(all) * mk;
        c.evap_dep += c.pet_res * h_b(some) * mk;
        c.pet_res = (some ? 0.0 : c.pet_res) * mk;
        c.S_dep += -c.evap_dep * h_b((c.S_dep > 0) && (c.evap_dep > 0)) * mk;
    }
    c.evap_sur = c.evap_int_top + c.evap_int_ground + c.evap_dep * mk;  // :204-210

    // soil evaporation, FAO-56 stress :216-345
    {
        c.k_stress_evap = (c.de <= c.rew ? 1.0 : c.k_stress_evap) * mk;
        c.k_stress_evap = ((c.de > c.rew) && (c.de <= c.tew) ? (c.tew - c.de) / (c.tew - c.rew) : c.k_stress_evap) * mk;
        c.k_stress_evap = (c.de > c.tew ? 0.0 : c.k_stress_evap) * mk;
        c.evap_coeff = c.basal_evap_coeff * c.k_stress_evap * mk;
        const double pe = c.pet_res * c.evap_coeff * mk;
        c.pevap_soil = pe;
        const bool on = (c.S_fp_rz > 0) && (pe > 0) && (c.swe <= 0) && (c.prec <= 0);
        const bool some = on && (pe <= c.S_fp_rz), all = on && (pe > c.S_fp_rz);
        double e = 0.0 + pe * h_b(some) * mk;
        c.pet_res += -pe * h_b(some) * mk;
        c.pet_res = (c.pet_res < 0 ? 0.0 : c.pet_res) * mk;
        e += c.S_fp_rz * h_b(all) * mk;
        c.pet_res += -c.S_fp_rz * h_b(all) * mk;
        c.pet_res = (c.pet_res < 0 ? 0.0 : c.pet_res) * mk;
        c.evap_soil = e * mk;
        c.S_fp_rz += -c.evap_soil * mk;
    }

    // transpiration :349-543
    {
        const int lu = c.lu_id;
        const double th_ws = K.transp_water_stress * c.theta_ufc + c.theta_pwp * mk;
        const bool crop = (lu >= 500) && (lu < 600);
        c.k_stress_transp = (crop ? c.k_stress_transp : (c.theta_rz - c.theta_pwp) / (th_ws - c.theta_pwp)) * mk;
        c.k_stress_transp = (c.k_stress_transp > 1 ? 1.0 : c.k_stress_transp);
        c.transp_coeff = c.basal_transp_coeff * c.k_stress_transp * mk;
        const bool anoxia = (lu > 500) && (lu < 599) && (c.theta_rz >= 0.8 * c.theta_sat);
        if (anoxia) {  // only crops; keeps the pow off the common path
            const double r = c.S_lp_rz / c.S_ac_rz;
            c.transp_coeff = ((r >= 0) && (r <= 1) ? 1 - RH_POW(r, 1.5) : 1.0);
        }
        c.transp_coeff = c.transp_coeff * mk;
        const double pt0 = (c.pevap_soil < c.pet ? c.pet - c.pevap_soil : 0.0) * mk;
        const double pt1 = (c.evap_soil < c.pet ? c.pet - c.evap_soil : 0.0) * mk;
        c.pt = pt0 * c.basal_transp_coeff * mk;
        c.ptransp = pt1 * c.transp_coeff * mk;
        const bool tree = (lu == 10) || (lu == 11) || (lu == 12) || (lu == 15) || (lu == 16) || (lu == 17);
        c.ptransp = (tree ? c.pet * c.transp_coeff : c.ptransp) * mk;
        c.ptransp_res = c.ptransp * mk;
        const bool dry = (c.ptransp > 0) && (c.prec <= 0);
        double t_lp = 0.0, t_fp = 0.0;
        bool m = (c.S_lp_rz > 0) && (c.ptransp_res <= c.S_lp_rz) && dry;
        t_lp += (m ? c.ptransp_res : 0.0) * mk;
        c.ptransp_res = (m ? 0.0 : c.ptransp_res) * mk;
        m = (c.S_lp_rz > 0) && (c.ptransp_res > c.S_lp_rz) && dry;
        t_lp += (m ? c.S_lp_rz : 0.0) * mk;
        c.ptransp_res += (m ? -c.S_lp_rz : 0.0) * mk;
        m = (c.S_fp_rz > 0) && (c.ptransp_res <= c.S_fp_rz) && (c.S_lp_rz <= 0) && dry;
        t_fp += (m ? c.ptransp_res : 0.0) * mk;
        c.ptransp_res = (m ? 0.0 : c.ptransp_res) * mk;
        m = (c.S_fp_rz > 0) && (c.ptransp_res > c.S_fp_rz) && (c.S_lp_rz <= 0) && dry;
        t_fp += (m ? c.S_fp_rz : 0.0) * mk;
        c.ptransp_res += (m ? -c.S_fp_rz : 0.0) * mk;
        c.ptransp_res = (c.ptransp_res < 0 ? 0.0 : c.ptransp_res) * mk;
        c.S_lp_rz += -t_lp * mk;
        c.S_fp_rz += -t_fp * mk;
        c.transp = (t_fp + t_lp) * mk;
    }
    c.de += c.evap_soil + c.transp * (c.z_evap / c.z_root) * mk;  // :553-558
    c.aet_soil = (c.evap_soil + c.transp) * mk;
    c.aet = (c.evap_int_top + c.evap_int_ground + c.evap_dep + c.evap_soil + c.transp) * mk;
}

// ---------------------------------------------------------------------------------------------
// snow.py
// ---------------------------------------------------------------------------------------------
// Degree-day melt of one snow store (:56-95, :145-191, :209-246): returns the melt, debits
// pet_res.  `zero_rest` reproduces `swe += where(full, 0, -swe)` of the two interception
// stores (:92-95, :188-191), which empties the store whenever it was not already "over-melted".
RH_DEV double h_melt(double &swe, double &pet_res, double pot, double mk, bool zero_rest) {
    double melt = 0.0;
    melt = ((pot > 0) && (pot <= swe) && (swe > 0) ? pot : melt) * mk;
    melt = ((pot > 0) && (pot > swe) && (swe > 0) ? swe : melt) * mk;
    const bool part = (melt > 0) && (melt <= swe), full = (melt > 0) && (melt > swe);
    pet_res += -melt * h_b(part) * mk;
    swe += -melt * h_b(part) * mk;
    pet_res += -swe * h_b(full) * mk;
    if (zero_rest)
        swe += (full ? 0.0 : -swe) * mk;
    else
        swe = (full ? 0.0 : swe) * mk;
    return melt;
}

RH_DEV void rt_snow(Col &c, const Consts &K, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    const double kw = (10000. / (100 - K.rmax) / 100.);
    const double acc = c.snow_ground * h_b(c.ta <= K.ta_fm) * mk;  // :7-26
    c.S_snow += acc;
    c.swe += acc;
    c.S_snow += c.rain_ground * h_b((c.swe > 0) && (c.ta > K.ta_fm)) * mk;  // :30-44
    const double pot = (K.sf * (c.ta - K.ta_fm) * X.dt) * mk;

    // canopy snow :48-133
    c.snow_melt_top = h_melt(c.swe_top, c.pet_res, pot, mk, true);
    c.pet_res = (c.pet_res < 0 ? 0.0 : c.pet_res) * mk;
    {
        const double wtmx = kw * c.swe_top;
        const double q_ret = (c.S_int_top > c.S_int_top_tot ? c.S_int_top - c.swe_top : 0.0) * mk;
        const bool over = c.S_int_top_tot < c.S_int_top;
        c.snow_melt_drip = (q_ret > wtmx ? q_ret - wtmx : ((wtmx <= 0) && over ? c.S_int_top - c.S_int_top_tot : 0.0)) * mk;
        c.S_snow += (over ? c.snow_melt_drip : 0.0) * mk;
        c.S_int_top += (over ? -c.snow_melt_drip : 0.0) * mk;
    }
    // ground interception snow :137-193
    c.snow_melt_ground = h_melt(c.swe_ground, c.pet_res, pot, mk, true);
    // snow pack :197-290
    c.snow_melt = h_melt(c.swe, c.pet_res, pot, mk, false);
    c.pet_res = (c.pet_res < 0 ? 0.0 : c.pet_res) * mk;
    {
        const double wtmx = kw * c.swe;
        const double q_ret = (c.S_snow > 0 ? c.S_snow - c.swe : 0.0) * mk;
        c.q_snow = (q_ret > wtmx ? q_ret - wtmx : (wtmx <= 0 ? c.S_snow : 0.0)) * mk;
        c.S_snow += -c.q_snow * mk;
        c.z0 += c.q_snow * mk;
        c.prec_event_csum += c.q_snow * mk;
    }
}

// ---------------------------------------------------------------------------------------------
// infiltration.py:1-2193
// ---------------------------------------------------------------------------------------------
RH_DEV double h_floor01(double v, double z_soil, double mk) {  // the two clamps shared by :1582-1591 etc.
    v = (z_soil <= 0 ? 0.01 : v) * mk;
    return (v <= 0 ? 0.01 : v) * mk;
}
RH_DEV double h_theta_d(const Col &c, double mk) {  // :1564-1594
    const double v = (c.z_root > 0 ? (c.theta_sat - c.theta_rz) * (1 - c.sealing / 1) : 0.0) * mk;
    return h_floor01(v, c.z_soil, mk);
}
RH_DEV double h_theta_d_rel(const Col &c, double mk) {  // :1598-1632
    const double v =
        (c.z_root > 0 ? ((c.theta_sat - c.theta_rz) / (c.theta_sat - c.theta_pwp)) * (1 - c.sealing / 1) : 0.0) * mk;
    return h_floor01(v, c.z_soil, mk);
}
RH_DEV double h_theta_d_fp(const Col &c, double mk) {  // :1636-1666
    const double v = (c.z_soil > 0 ? (c.theta_fc - c.theta_rz) * (1 - c.sealing / 1) : 0.0) * mk;
    return h_floor01(v, c.z_soil, mk);
}

// start of an event: calc_depth_shrinkage_cracks :1768-1826 + set_event_vars :1830-1976
RH_DEV void h_event_start(Col &c, double mk) {
    const double th = c.theta_rz;
    double z = (th < c.theta_4 ? c.z_sc_max
                               : ((th >= c.theta_4) && (th < c.theta_27) ? (th - c.theta_4) / (c.theta_27 - c.theta_4) : 0.0) *
                                     c.z_sc_max) *
               mk;
    z = (th < c.theta_4 ? c.z_sc_max : z) * mk;
    z = (th > c.theta_27 ? 0.0 : z) * mk;
    z = ((1 - c.sealing / 1) * z) * mk;
    z = (z > c.z_root ? c.z_root : z) * mk;
    c.z_sc = (c.lu_id == 13 ? 0.0 : z) * mk;

    c.no_wf = 1;
    c.z_wf = 0.0; c.z_wf_m1 = 0.0;
    c.z_wf_t0 = 0.0; c.z_wf_t0_m1 = 0.0;
    c.z_wf_t1 = 0.0; c.z_wf_t1_m1 = 0.0;
    c.z_wf_fc = 0.0;
    c.inf_mat_event_csum = 0.0;
    c.inf_mat_pot_event_csum = 0.0;
    c.inf_mp_event_csum = 0.0;
    c.y_mp = 0.0; c.y_mp_m1 = 0.0;
    c.inf_sc_event_csum = 0.0;
    c.y_sc = 0.0; c.y_sc_m1 = 0.0;
    const double td = h_theta_d(c, mk), tdr = h_theta_d_rel(c, mk);
    c.theta_d = td * mk;
    c.theta_d_rel = tdr * mk;
    c.theta_d_t0 = td * mk;
    c.theta_d_rel_t0 = tdr * mk;
    c.theta_d_fp = h_theta_d_fp(c, mk) * mk;
    c.prec_event_csum = 0.0;
    c.t_event_csum = 0.0;
    c.de = 0.0;
}

// rainfall pause begins :1980-1995 (+ calc_z_wf_fc :1536-1560)
RH_DEV void h_pause_start(Col &c, double mk) {
    double zf = (c.theta_d_fp > 0 ? c.inf_mat_event_csum / c.theta_d_fp : c.z_wf) * mk;
    zf = (zf > c.z_soil ? c.z_soil : zf) * mk;
    c.z_wf_fc = ((c.prec == 0) && (c.prec_m1 != 0) ? zf : c.z_wf_fc) * mk;
}

// rainfall pause ends: second wetting front starts :1999-2053
RH_DEV void h_pause_end(Col &c, double mk) {
    const bool m = (c.prec != 0) && (c.prec_m1 == 0);
    c.no_wf = m ? 2 : c.no_wf;
    c.theta_d = (m ? h_theta_d(c, mk) : c.theta_d) * mk;
    c.theta_d_rel = (m ? h_theta_d_rel(c, mk) : c.theta_d_rel) * mk;
    c.z_wf_t1 = m ? 0.0 : c.z_wf_t1;
    c.z_wf_t1_m1 = m ? 0.0 : c.z_wf_t1_m1;
    c.prec_event_csum = m ? 0.0 : c.prec_event_csum;
    c.t_event_csum = m ? 0.0 : c.t_event_csum;
}

// end of an event :2057-2144
RH_DEV void h_event_end(Col &c, double mk) {
    c.z_wf = 0.0; c.z_wf_m1 = 0.0;
    c.z_wf_t0 = 0.0; c.z_wf_t0_m1 = 0.0;
    c.z_wf_t1 = 0.0; c.z_wf_t1_m1 = 0.0;
    c.y_mp = 0.0;  // [tau] only :2080-2084
    c.y_sc = 0.0; c.y_sc_m1 = 0.0;
    const double td = h_theta_d(c, mk);
    c.theta_d = td * mk;
    c.theta_d_t0 = td * mk;
    c.pi_gr = 0.0;
    c.pi_m = 0.0;
    c.t_sat = 0.0;
    c.Fs = 0.0;
    c.z_sc = 0.0;
}

// Green-Ampt / Peschke parameters :8-48, :1670-1764
RH_DEV void h_green_ampt(Col &c, double dt, double mk) {
    const double kdw = c.ks * c.theta_d * c.wfs;
    c.pi_gr = (c.ks * (((c.theta_d * c.wfs) / (c.prec_event_csum + 1)) + 1)) * mk;
    const double pi_m = kdw * mk;
    c.pi_m = pi_m * mk;
    const bool reached = (c.pi_m <= c.prec_event_csum) && (c.t_sat == 0);
    const bool m1 = reached && (c.pi_m > c.pi_gr);
    const bool m2 = reached && (c.pi_m <= c.pi_gr) && ((c.prec * (1 / dt) - c.ks) * c.prec_event_csum > kdw);
    double ts = m1 ? c.t_event_csum - dt : c.t_sat;
    ts = m2 ? c.t_event_csum + (kdw / (c.pi_m * (c.pi_m * -c.ks))) - (dt / c.pi_m) * c.prec_event_csum : ts;  // :1731-1735
    c.t_sat = ts * mk;
    double Fs = (kdw / (pi_m - c.ks)) * mk;
    Fs = (pi_m <= c.ks ? pi_m : Fs) * mk;
    c.Fs = Fs * mk;
}

// matrix infiltration and the dual wetting fronts :52-427
RH_DEV void h_inf_mat(Col &c, double dt, double mk) {
    const double ksdt = c.ks * dt, tsat = c.t_sat, tev = c.t_event_csum;
    const bool after = (tev > tsat) && (tsat > 0);
    const bool m1 = after && (c.pi_m <= c.prec_event_csum), m2 = after && (c.pi_m > c.prec_event_csum);
    const bool m3 = (tsat > tev - dt) && (tsat < tev);
    const bool m4 = (c.pi_m > c.prec_event_csum) && (tsat <= 0);
    const double open = ((1 - c.sealing) / 1);
    const double a = c.ks * (tev - tsat) * mk;
    const double b = c.Fs + 2 * c.theta_d * c.wfs * mk;
    const double wd = c.wfs * c.theta_d;
    const double l1 = (c.z0 > ksdt ? (ksdt * c.wfs * c.theta_d) / (c.z0 - ksdt) : (ksdt * c.wfs * c.theta_d) / ksdt) * mk;
    const double rec = (ksdt / 2) * (1 + (1 + 2 * b / a) / sqrt(1 + (4 * b / a) + (4 * (c.Fs_t0 * c.Fs_t0) / (a * a))));
    double pot = ksdt;
    pot = (m1 ? rec * open : pot) * mk;
    pot = (m2 ? ksdt * (1 + (wd / l1)) * open : pot) * mk;
    const double rec3 = (m3 ? rec : 0.0) * mk;
    const double sat3 = (m3 ? c.z0 * (tsat - (tev - dt)) : 0.0) * mk;
    pot = (m3 ? sat3 + rec3 * open : pot) * mk;
    pot = (m4 ? c.pi_gr * open : pot) * mk;
    c.inf_mat_pot = pot;

    double inf = (c.z0 < pot ? c.z0 : c.inf_mat) * mk;
    inf = (c.z0 >= pot ? pot : inf) * mk;
    const double space = (c.S_ac_rz + c.S_ufc_rz) - (c.S_lp_rz + c.S_fp_rz);
    inf = (inf > space ? space : inf) * mk;
    inf = (inf < 0 ? 0.0 : inf) * mk;
    c.inf_mat = inf;
    c.inf_mat_event_csum += inf * mk;
    c.inf_mat_pot_event_csum += pot * mk;

    double dz = 0.0;
    dz = (c.no_wf == 1 ? inf / c.theta_d_t0 : dz) * mk;
    dz = (c.no_wf == 2 ? inf / c.theta_d : dz) * mk;
    dz = (isfinite(dz) ? dz : 0.0) * mk;
    double w0 = c.z_wf_t0 + dz, w1 = c.z_wf_t1 + dz;
    w0 = (w0 > c.z_soil ? c.z_soil : w0) * mk;
    w1 = (w1 > c.z_soil ? c.z_soil : w1) * mk;
    c.z0 += -inf * mk;
    c.z0 = (c.z0 < 0 ? 0.0 : c.z0) * mk;

    // fronts keep moving during a rainfall pause
    const bool pause = (c.z_wf_fc > 0) && (c.rain_ground <= 0);
    double d0 = (pause && (c.no_wf == 1) ? pot / c.theta_d_t0 : 0.0) * mk;
    w0 += (isfinite(d0) ? d0 : 0.0) * mk;
    w0 = ((w0 > c.z_wf_fc) && (c.z_wf_fc > 0) ? c.z_wf_fc : w0) * mk;
    w0 = (w0 > c.z_soil ? c.z_soil : w0) * mk;
    double d1 = (pause && (c.no_wf == 2) ? pot / c.theta_d : 0.0) * mk;
    w1 += (isfinite(d1) ? d1 : 0.0) * mk;
    w1 = ((w1 > c.z_wf_fc) && (c.z_wf_fc > 0) ? c.z_wf_fc : w1) * mk;
    w1 = (w1 > c.z_soil ? c.z_soil : w1) * mk;
    c.z_wf_t0 = w0;
    c.z_wf_t1 = w1;

    const bool m14 = (w0 >= w1) && (w1 <= 0), m15 = (w0 > w1) && (w1 > 0), m20 = (w0 <= w1) && (w1 > 0);
    c.z_wf = (m14 ? w0 : c.z_wf) * mk;
    c.theta_d = (m14 ? c.theta_d_t0 : c.theta_d) * mk;
    c.theta_d_rel = (m14 ? c.theta_d_rel_t0 : c.theta_d_rel) * mk;
    c.z_wf_m1 = (m15 ? 0.0 : c.z_wf_m1) * mk;
    c.z_wf = (m15 ? w1 : c.z_wf) * mk;
    c.no_wf = m20 ? 1 : c.no_wf;
    c.z_wf = (m20 ? w0 : c.z_wf) * mk;
    c.theta_d = (m20 ? c.theta_d_t0 : c.theta_d) * mk;
    c.theta_d_rel = (m20 ? c.theta_d_rel_t0 : c.theta_d_rel) * mk;
    c.z_wf = (c.z_wf > c.z_soil ? c.z_soil : c.z_wf) * mk;
    c.theta_d = (c.theta_d_t1 <= 0 ? c.theta_d_t0 : c.theta_d) * mk;  // :411-416
}

RH_DEV bool h_same(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// Length of the front that macropores / cracks still reach, :443-518 and :1092-1167.  The
// reference builds z_wf(_m1) twice and the second assignment wins, so only the second front
// (z_wf_t1) enters.
RH_DEV double h_open_length(double len, double z_wf_tau, const Col &c, int substeps, double mk) {
    const double zw = (c.no_wf == 2 ? 0.0 : c.z_wf_t1) * mk;
    const double zw_m1 = (c.no_wf == 2 ? 0.0 : c.z_wf_t1_m1) * mk;
    double open0 = len - zw * mk;
    open0 = (open0 < 0 ? 0.0 : open0) * mk;
    double dz = zw - zw_m1 * mk;
    dz = (zw >= len ? open0 : dz) * mk;
    dz = (open0 <= 0 ? 0.0 : dz) * mk;
    dz = (dz <= 0 ? 0.0 : dz) * mk;
    double open = len - z_wf_tau * mk;
    open = (open < 0 ? 0.0 : open) * mk;
    return (substeps == 1 ? open + dz / 1.39 : open) * mk;
}

// macropore infiltration :431-1077.  Sub-stepped radial wetting front; the reference always runs
// `substeps` iterations (1 / 5 / 120 for dt = 10 min / 1 h / 24 h).  Here a lane leaves the loop
// as soon as one iteration maps its loop state onto itself bit for bit: the body is a pure
// function of that state, so every further iteration would reproduce it.  On dry days (z0 = 0)
// that happens after the second iteration; the wavefront runs until its last lane is done.
RH_DEV void h_inf_mp(Col &c, const Consts &K, double dt, int substeps, double mk) {
    c.lmpv_non_sat = h_open_length(c.lmpv, c.z_wf, c, substeps, mk);
    const double zw = (c.no_wf == 2 ? 0.0 : c.z_wf_t1) * mk;
    const double r = K.r_mp, r3 = r * r * r;
    const double a = c.theta_d * (r * r) * mk;
    const double k6 = 2.449489742783178 * 2;  // 6**0.5 * 2
    const double td2 = c.theta_d * c.theta_d;
    const double geo = c.ks * c.wfs;
    const double z0_di = c.z0 * (c.mp_drain_area / substeps) * mk;
    const double h = dt / substeps;
    double y = c.y_mp_m1 * mk, ym1 = c.y_mp_m1 * mk, ecs = c.inf_mp_event_csum * mk, t = 0.0, inf = 0.0;
    for (int it = 0; it < substeps; ++it) {
        const double t_in = t, ym1_in = ym1, ecs_in = ecs, inf_in = inf;
        t += h * mk;
        double cc = geo * t * mk;
        cc = (isnan(cc) ? 0.0 : cc) * mk;
        double b1 = (k6 * sqrt(cc * (6 * cc - a))) * mk;
        b1 = (isnan(b1) ? 0.0 : b1) * mk;
        double b2 = (r * td2) * (12 * cc - a + b1) * mk;
        b2 = (isnan(b2) ? 0.0 : b2) * mk;
        b2 = (b2 <= 0 ? 0.0 : b2) * mk;
        const double cb = RH_POW(b2, 1.0 / 3);
        const double y1 = (cb / c.theta_d) * 0.5 * mk;
        const double y2 = (a / cb) * 0.5 * mk;
        y = (y1 + y2 + ym1) * mk;
        y = (y < r ? r : y) * mk;
        y = (y < ym1 ? ym1 : y) * mk;
        const double pot = (K.pi * (y * y - ym1 * ym1) * c.lmpv_non_sat * c.theta_d * c.dmpv * 1e-06) * mk;
        double di = (pot > z0_di ? z0_di : pot) * mk;
        di = (c.lmpv_non_sat == 0 ? 0.0 : di) * mk;
        inf += di * mk;
        ecs += di * mk;
        y = r + sqrt((ecs / (c.dmpv * c.theta_d)) / K.pi) * mk;
        y = (y < r ? r : y) * mk;
        t = c.theta_d / (geo * r) * ((y * y * y) / 3.0 - (y * y) * r / 2.0 + r3 / 6.0) * mk;
        inf = (inf < 0 ? 0.0 : inf) * mk;
        ym1 = y * mk;
        if (h_same(t, t_in) && h_same(ym1, ym1_in) && h_same(ecs, ecs_in) && h_same(inf, inf_in)) break;
    }
    c.y_mp = y * mk;
    c.y_mp = (isnan(c.y_mp) ? 0.0 : c.y_mp) * mk;
    double inf_mp = inf * mk;
    inf_mp = (isnan(inf_mp) ? 0.0 : inf_mp) * mk;

    // split between root zone and subsoil :829-941
    double share = (c.lmpv_non_sat > 0 ? 1.0 - (c.lmpv - c.z_root) / c.lmpv_non_sat : 0.0) * mk;
    share = (c.lmpv <= c.z_root ? 1.0 : share) * mk;
    share = (zw >= c.z_root ? 0.0 : share) * mk;
    share = (share < 0 ? 0.0 : share) * mk;
    share = (share > 1 ? 1.0 : share) * mk;
    c.inf_mp_rz = inf_mp * share * mk;
    const double room = (c.S_ac_rz + c.S_ufc_rz) - (c.inf_mat_rz + c.S_lp_rz + c.S_fp_rz);  // last step's inf_mat_rz
    c.inf_mp_rz = ((c.inf_mp_rz > room) && (room >= 0) ? room : c.inf_mp_rz) * mk;
    c.inf_mp_rz = (room < 0 ? 0.0 : c.inf_mp_rz) * mk;
    c.inf_mp_ss = inf_mp * (1 - share) * mk;
    const double room_ss = (c.S_ac_ss + c.S_ufc_ss) - (c.S_lp_ss + c.S_fp_ss);
    c.inf_mp_ss = ((c.inf_mp_ss > room_ss) && (room_ss > 0) ? room_ss : c.inf_mp_ss) * mk;
    c.inf_ss = c.inf_mp_ss * mk;
    c.S_fp_ss += c.inf_ss * mk;
    bool m = c.S_fp_ss > c.S_ufc_ss;
    c.S_lp_ss += (m ? c.S_fp_ss - c.S_ufc_ss : 0.0) * mk;
    c.S_fp_ss = (m ? c.S_ufc_ss : c.S_fp_ss) * mk;
    m = c.S_lp_ss > c.S_ac_ss;
    c.inf_mp_ss += (m ? -(c.S_lp_ss - c.S_ac_ss) : 0.0) * mk;
    c.inf_mp_ss = (c.inf_mp_ss < 0 ? 0.0 : c.inf_mp_ss) * mk;
    c.S_lp_ss = (m ? c.S_ac_ss : c.S_lp_ss) * mk;
    c.inf_mp = c.inf_mp_rz + c.inf_mp_ss * mk;
    c.inf_mp_event_csum += c.inf_mp * mk;
    c.z0 += -c.inf_mp * mk;
    c.z0 = (c.z0 < 0 ? 0.0 : c.z0) * mk;
}

// shrinkage-crack infiltration :1081-1318.  In the reference the loop carry slot of inf_sc is
// never written (:1278-1283), so inf_sc stays 0 and only y_sc / z_sc_non_sat change.  Same
// fixed-point exit as above.
RH_DEV void h_inf_sc(Col &c, const Consts &K, double dt, int substeps, double mk) {
    c.z_sc_non_sat = h_open_length(c.z_sc, c.z_wf, c, substeps, mk);
    const double geo = c.ks * c.wfs;
    const double z0_di = (c.z0 / substeps) * mk;
    const double h = dt / substeps;
    double y = c.y_sc_m1 * mk, ym1 = c.y_sc_m1 * mk, ecs = c.inf_sc_event_csum * mk, t = 0.0;
    for (int it = 0; it < substeps; ++it) {
        const double t_in = t, ym1_in = ym1, ecs_in = ecs;
        t += h * mk;
        y = sqrt((geo * t * 2) / c.theta_d) * mk;
        double pot = ((c.z_sc_non_sat * c.theta_d * K.l_sc) * (y - ym1) * 1e-06) * mk;
        pot = (pot <= 0 ? 0.0 : pot) * mk;
        double di = (pot > z0_di ? z0_di : pot) * mk;
        di = (c.z_sc_non_sat <= 0 ? 0.0 : di) * mk;
        di += di * mk;  // :1248-1252
        ecs += di * mk;
        y = (ecs / K.l_sc / 2) * mk;
        t = ((ym1 * ym1 * c.theta_d) / (geo * 2)) * mk;
        ym1 = y * mk;
        if (h_same(t, t_in) && h_same(ym1, ym1_in) && h_same(ecs, ecs_in)) break;
    }
    c.y_sc = y * mk;
    c.inf_sc = 0.0 * mk;
    c.inf_sc_event_csum += c.inf_sc * mk;
    c.z0 += -c.inf_sc * mk;
    c.z0 = (c.z0 < 0 ? 0.0 : c.z0) * mk;
}

// root-zone bookkeeping, overland flow :1322-1532
// calc_inf_rz, calc_inf, calc_hof_and_sof :1322-1476
RH_DEV void h_inf_rz_hof_sof(Col &c, double mk) {
    c.inf_mat_rz = c.inf_mat * mk;
    c.inf_sc_rz = c.inf_sc * mk;
    c.inf_rz = (c.inf_mat_rz + c.inf_mp_rz + c.inf_sc_rz) * mk;
    c.S_fp_rz += c.inf_rz * mk;
    bool m = c.S_fp_rz > c.S_ufc_rz;
    c.S_lp_rz += (m ? c.S_fp_rz - c.S_ufc_rz : 0.0) * mk;
    c.S_fp_rz = (m ? c.S_ufc_rz : c.S_fp_rz) * mk;
    m = c.S_lp_rz > c.S_ac_rz;
    const double excess = c.S_lp_rz - c.S_ac_rz;
    c.inf_mp_rz += (m ? -excess : 0.0) * mk;
    c.inf_mp_rz = (c.inf_mp_rz < 0 ? 0.0 : c.inf_mp_rz) * mk;
    c.z0 += (m ? excess : 0.0) * mk;
    c.S_lp_rz = (m ? c.S_ac_rz : c.S_lp_rz) * mk;
    c.inf_mp = c.inf_mp_rz + c.inf_mp_ss * mk;
    c.inf_rz = (c.inf_mat_rz + c.inf_mp_rz + c.inf_sc_rz) * mk;
    c.inf = (c.inf_rz + c.inf_ss) * mk;

    // Hortonian and saturation overland flow :1421-1476
    c.q_hof = c.z0 * mk;
    c.q_hof = (c.q_hof < 0 ? 0.0 : c.q_hof) * mk;
    const bool full = ((c.S_lp_rz + c.S_fp_rz) > (c.S_ac_rz + c.S_ufc_rz)) && ((c.S_lp_ss + c.S_fp_ss) >= (c.S_ac_ss + c.S_ufc_ss));
    c.q_sof = (full ? (c.S_lp_rz + c.S_fp_rz) - (c.S_ac_rz + c.S_ufc_rz) : 0.0) * mk;
    m = c.q_sof > 0;
    c.S_fp_rz = (m ? c.S_ufc_rz : c.S_fp_rz) * mk;
    c.S_lp_rz = (m ? c.S_ac_rz : c.S_lp_rz) * mk;
}
// surface runoff :1480-1516 (not with settings.enable_routing_1D, :2189-2190: the ponded water stays and is routed)
RH_DEV void h_surface_runoff(Col &c, double mk) {
    c.z0 += -c.q_hof * mk;
    c.z0 = (c.z0 < 0 ? 0.0 : c.z0) * mk;
    c.q_sur = 0.0 + (c.q_hof + c.q_sof) * mk;
    c.q_sur += (c.maskRiver || c.maskLake) ? c.prec : 0.0;
}
RH_DEV void h_inf_finish(Col &c, double mk) {
    h_inf_rz_hof_sof(c, mk);
    h_surface_runoff(c, mk);
}

// calculate_infiltration :2148-2193; X.cond1..5 are the host-side `if cond.any()` branches
// calculate_infiltration (infiltration.py:2148-2193) in five parts; the fused kernel runs them as separate
// pipeline stages (each loads the planes it is the first to mention), the per-routine entry point as one call
RH_DEV int h_inf_substeps(double dt) { return (int)rint(dt / (1.0 / 5)); }  // :513, npx.round = half-to-even
RH_DEV void rt_inf_events(Col &c, const Consts &K, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    if (X.cond1) h_event_start(c, mk);
    if (X.cond2) h_pause_start(c, mk);
    if (X.cond3) h_pause_end(c, mk);
    if (X.cond5) c.t_event_csum += X.dt;
}
RH_DEV void rt_inf_matrix(Col &c, const Consts &K, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    h_green_ampt(c, X.dt, mk);
    h_inf_mat(c, X.dt, mk);
}
RH_DEV void rt_inf_macropores(Col &c, const Consts &K, const StepCtx &X) {
    h_inf_mp(c, K, X.dt, h_inf_substeps(X.dt), (double)c.maskCatch);
}
RH_DEV void rt_inf_cracks(Col &c, const Consts &K, const StepCtx &X) {
    h_inf_sc(c, K, X.dt, h_inf_substeps(X.dt), (double)c.maskCatch);
}
RH_DEV void rt_inf_finish(Col &c, const Consts &K, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    h_inf_finish(c, mk);
    if (X.cond4) h_event_end(c, mk);
}
RH_DEV void rt_infiltration(Col &c, const Consts &K, const StepCtx &X) {
    rt_inf_events(c, K, X);
    rt_inf_matrix(c, K, X);
    rt_inf_macropores(c, K, X);
    rt_inf_cracks(c, K, X);
    rt_inf_finish(c, K, X);
}
// ... with settings.enable_routing_1D: calc_surface_runoff is skipped (:2189-2190)
RH_DEV void rt_inf_finish_routed(Col &c, const Consts &K, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    h_inf_rz_hof_sof(c, mk);
    if (X.cond4) h_event_end(c, mk);
}
RH_DEV void rt_infiltration_routed(Col &c, const Consts &K, const StepCtx &X) {
    rt_inf_events(c, K, X);
    rt_inf_matrix(c, K, X);
    rt_inf_macropores(c, K, X);
    rt_inf_cracks(c, K, X);
    rt_inf_finish_routed(c, K, X);
}

// ---------------------------------------------------------------------------------------------
// subsurface_runoff.py (SVAT branch :1473-1479)
// ---------------------------------------------------------------------------------------------
// Salvucci capillary term shared by percolation and capillary rise:
// (p1 - p2) / (1 + p2 + (n - 1) p1), p1 = (z / (-ha 10.2))^-n, p2 = (-h / -ha)^-n
RH_DEV double h_salvucci(double z, double hpot, double ha, double n) {
    const double p1 = RH_POW(z / (-ha * 10.2), -n);
    const double p2 = RH_POW(-hpot / -ha, -n);
    return (p1 - p2) / (1 + p2 + (n - 1) * p1);
}

// perched water table :693-765 and its storage :7-48
RH_DEV void h_sub_water_table(Col &c, double mk) {
    const double dz_ss = c.z_soil - c.z_root;
    double lmpv_ss = c.lmpv - c.z_root * mk;
    lmpv_ss = (c.lmpv < c.z_root ? 0.0 : lmpv_ss) * mk;
    const double lp_ss_mm = c.S_lp_ss / c.theta_ac;
    const double top = (c.S_lp_ss < c.theta_ac ? lp_ss_mm : c.S_lp_rz + lp_ss_mm) * mk;
    double nomp = dz_ss - lmpv_ss - c.z_sat * mk;
    nomp = (nomp < 0 ? 0.0 : nomp);
    const double risen = ((c.S_fp_ss >= c.S_ufc_ss) && (((c.S_lp_ss + 1e-6) / c.theta_ac) < dz_ss))
                             ? lp_ss_mm
                             : (((c.S_fp_rz >= c.S_ufc_rz) && (c.S_lp_ss + 1e-6 >= c.S_ac_ss)) ? c.S_lp_rz / c.theta_ac + lp_ss_mm
                                                                                                 : lp_ss_mm);
    c.z_sat = (top > nomp ? risen : lp_ss_mm) * mk;
    c.S_zsat = (c.z_sat <= c.z_soil ? c.z_sat * c.theta_ac : c.z_soil * c.theta_ac) * mk;
    c.S_zsat_ss = (c.z_sat <= dz_ss ? c.S_zsat : dz_ss * c.theta_ac) * mk;
    c.S_zsat_rz = (c.z_sat > dz_ss ? (c.z_sat - dz_ss) * c.theta_ac : 0.0) * mk;
}

// potential + actual percolation out of the root zone :768-968
RH_DEV void h_sub_percolation_rz(Col &c, double dt, double mk) {
    const double dz_ss = c.z_soil - c.z_root;
    {
        const bool dryb = c.z_sat <= 0;
        const bool m3 = (c.z_sat > 0) && (c.z_root < c.z_soil - c.z_sat);
        double perc = ((c.z_wf < c.z_root) && dryb ? c.k_rz * dt : 0.0) * mk;
        perc = ((c.z_wf >= c.z_root) && dryb ? c.k_rz * dt : perc) * mk;
        if (m3) perc = h_salvucci(dz_ss - c.z_sat, c.h_rz, c.ha, c.n_salv) * dt * c.ks * (-1);
        perc = perc * mk;
        perc = (perc < 0 ? 0.0 : perc) * mk;
        const bool above = c.z_root_m1 < c.z_soil - c.z_sat;
        const double avail = c.S_lp_rz + c.S_fp_rz;
        double q = ((perc > 0) && (avail >= perc) && above ? perc : 0.0) * mk;
        q = ((perc > 0) && (avail < perc) && above ? c.S_fp_rz + c.S_lp_rz : q) * mk;
        const double room = (c.S_ac_ss + c.S_ufc_ss) - (c.S_lp_ss + c.S_fp_ss);
        q = ((q > 0) && (room > 0) && (q > room) && above ? room : q) * mk;
        q = ((c.S_lp_ss >= c.S_ac_ss - 1e-6) && (c.S_fp_ss >= c.S_ufc_ss - 1e-6) ? 0.0 : q) * mk;
        q = (c.z_root_m1 >= c.z_soil - c.z_sat ? 0.0 : q) * mk;
        c.q_pot_rz = q;
    }
    {
        const bool above = c.z_sat < dz_ss;
        const bool m1 = (c.S_lp_rz < c.q_pot_rz) && above, m2 = (c.S_lp_rz >= c.q_pot_rz) && above;
        c.q_rz = c.q_pot_rz * mk;
        c.q_rz = (c.z_sat >= dz_ss ? 0.0 : c.q_rz) * mk;
        c.S_fp_rz += (m1 ? -(c.q_rz - c.S_lp_rz) : 0.0) * mk;
        c.S_lp_rz = (m1 ? 0.0 : c.S_lp_rz) * mk;
        c.S_lp_rz += (m2 ? -c.q_rz : 0.0) * mk;
        c.S_fp_ss += c.q_rz * mk;
        bool m = c.S_fp_ss > c.S_ufc_ss;
        c.S_lp_ss += (m ? c.S_fp_ss - c.S_ufc_ss : 0.0) * mk;
        c.S_fp_ss = (m ? c.S_ufc_ss : c.S_fp_ss) * mk;
        m = c.S_lp_ss > c.S_ac_ss;
        const double back = c.S_lp_ss - c.S_ac_ss;
        c.q_rz += (m ? -back : 0.0) * mk;
        c.S_lp_rz += (m ? back : 0.0) * mk;
        c.S_lp_ss = (m ? c.S_ac_ss : c.S_lp_ss) * mk;
    }
}

// potential percolation out of the subsoil :971-1098 (the second of the two assignments is the live one)
RH_DEV void h_sub_pot_percolation_ss(Col &c, double dt, double mk) {
    const double dz_ss = c.z_soil - c.z_root;
    const double zgw = c.z_gw * 1000;
    const double z = (zgw - c.z_soil) + (dz_ss / 2) * mk;
    const double sal = h_salvucci(z, c.h_ss, c.ha, c.n_salv);
    const bool shallow = (c.z_gw <= 10) && (zgw > c.z_soil) && (c.z_sat > 0);
    double perc = shallow ? fmin(fmin(c.kf * dt, c.ks_ss * dt), c.k_ss * dt) : fmin(c.kf * dt, sal * dt * c.ks_ss * (-1));
    perc = perc * mk;
    const bool drain = (perc > 0) && (c.z_soil < zgw);
    const double avail = c.S_fp_ss + c.S_lp_ss;
    double q = (drain && (perc <= avail) ? perc : 0.0) * mk;
    q = (drain && (perc > avail) ? avail : q) * mk;
    double cpr = sal * dt * c.ks_ss * mk;
    cpr = (drain ? 0.0 : cpr) * mk;
    cpr = (zgw - c.z_soil > 10000 ? 0.0 : cpr) * mk;
    c.q_pot_ss = (cpr > 0 ? 0.0 : q) * mk;
}

// SVAT branch of calculate_subsurface_runoff :1473-1479
RH_DEV void rt_subsurface_runoff(Col &c, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    const double dt = X.dt;
    h_sub_water_table(c, mk);
    h_sub_percolation_rz(c, dt, mk);
    h_sub_pot_percolation_ss(c, dt, mk);
    // percolation subsoil :1101-1154
    c.q_ss = c.q_pot_ss * mk;
    c.z_sat += (c.z_sat > 0 ? -c.q_ss / c.theta_ac : 0.0) * mk;
    c.z_sat = (c.z_sat < 0 ? 0.0 : c.z_sat) * mk;
    c.S_zsat_ss = c.z_sat * c.theta_ac * mk;
    const bool m1 = c.S_lp_ss < c.q_pot_ss, m2 = c.S_lp_ss >= c.q_pot_ss;
    c.S_fp_ss += (m1 ? -(c.q_ss - c.S_lp_ss) : 0.0) * mk;
    c.S_lp_ss = (m1 ? 0.0 : c.S_lp_ss) * mk;
    c.S_lp_ss += (m2 ? -c.q_ss : 0.0) * mk;
}

// saturated thickness of one 200 mm layer, :51-245 (`top` = the deepest layer has no upper clamp)
RH_DEV double h_layer(double z_sat, double offset, bool first, bool last, double mk) {
    double v = first ? z_sat * mk : z_sat - offset * mk;
    if (!last) v = (v > 200 ? 200.0 : v) * mk;
    return (v <= 0 ? 0.0 : v) * mk;
}

// oneD model, lateral branch of calculate_subsurface_runoff :1456-1471: Darcy flow in the matrix and
// pipe flow in horizontal macropores of eight layers, leaving the column as a sink term (no
// neighbour receives it without routing, SURVEY.md a10')
RH_DEV void rt_subsurface_runoff_lateral(Col &c, const Consts &K, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    const double dt = X.dt;
    h_sub_water_table(c, mk);
    c.z_sat_layer_1 = h_layer(c.z_sat, 0.0, true, false, mk);
    c.z_sat_layer_2 = h_layer(c.z_sat, 200.0, false, false, mk);
    c.z_sat_layer_3 = h_layer(c.z_sat, 400.0, false, false, mk);
    c.z_sat_layer_4 = h_layer(c.z_sat, 600.0, false, false, mk);
    c.z_sat_layer_5 = h_layer(c.z_sat, 800.0, false, false, mk);
    c.z_sat_layer_6 = h_layer(c.z_sat, 1000.0, false, false, mk);
    c.z_sat_layer_7 = h_layer(c.z_sat, 1200.0, false, false, mk);
    c.z_sat_layer_8 = h_layer(c.z_sat, 1400.0, false, true, mk);
    h_sub_percolation_rz(c, dt, mk);

    // potential lateral runoff :248-372
    const double dx = K.dx, per_len = (1 / (dx * (c.z_soil / 1000)));
    const double r2 = K.r_mp * K.r_mp;
    c.q_sub_mat_pot = ((c.ks * c.slope * c.z_sat * dx * 1000 * dt) * 1e-6 * per_len) * mk;
    c.q_sub_mat_pot = (c.z_sat <= 0 ? 0.0 : c.q_sub_mat_pot) * mk;
    double sum = c.z_sat_layer_1 * c.v_mp_layer_1 * dt * dx * 1000 * c.dmph * 1e-6 * r2 * K.pi * 1e-6;
    sum += c.z_sat_layer_2 * c.v_mp_layer_2 * dt * dx * 1000 * c.dmph * 1e-6 * r2 * K.pi * 1e-6;
    sum += c.z_sat_layer_3 * c.v_mp_layer_3 * dt * dx * 1000 * c.dmph * 1e-6 * r2 * K.pi * 1e-6;
    sum += c.z_sat_layer_4 * c.v_mp_layer_4 * dt * dx * 1000 * c.dmph * 1e-6 * r2 * K.pi * 1e-6;
    sum += c.z_sat_layer_5 * c.v_mp_layer_5 * dt * dx * 1000 * c.dmph * 1e-6 * r2 * K.pi * 1e-6;
    sum += c.z_sat_layer_6 * c.v_mp_layer_6 * dt * dx * 1000 * c.dmph * 1e-6 * r2 * K.pi * 1e-6;
    sum += c.z_sat_layer_7 * c.v_mp_layer_7 * dt * dx * 1000 * c.dmph * 1e-6 * r2 * K.pi * 1e-6;
    sum += c.z_sat_layer_8 * c.v_mp_layer_8 * dt * dx * 1000 * c.dmph * 1e-6 * r2 * K.pi * 1e-6;
    double mp = (sum * per_len) * mk;
    mp = (mp < 0 ? 0.0 : mp) * mk;
    mp = (c.z_sat <= 0 ? 0.0 : mp) * mk;
    double pot = (mp + c.q_sub_mat_pot) * mk;
    double sh_mat = (c.q_sub_mat_pot / pot) * mk, sh_mp = (mp / pot) * mk;
    sh_mat = (pot == 0 ? 0.0 : sh_mat) * mk;
    sh_mp = (pot == 0 ? 0.0 : sh_mp) * mk;
    c.q_sub_mat_share = sh_mat;
    c.q_sub_mp_share = sh_mp;
    pot = (pot > c.S_lp_rz + c.S_lp_ss ? c.S_lp_rz + c.S_lp_ss : pot) * mk;
    c.q_sub_pot = pot;
    c.q_sub_mat_pot = pot * sh_mat * mk;
    c.q_sub_mp_pot = pot * sh_mp * mk;

    // lateral runoff from the root zone :375-457
    const double dz_ss = c.z_soil - c.z_root;
    {
        double share = (c.z_sat > 0 ? ((c.z_sat - dz_ss) / c.z_sat) : 0.0) * mk;
        share = ((c.z_sat <= dz_ss) || (c.S_lp_rz <= 0) ? 0.0 : share) * mk;
        share = (isnan(share) ? 0.0 : share) * mk;
        c.S_zsat_rz = ((c.z_sat * share) * c.theta_ac) * mk;
        c.q_sub_rz = (pot * share < c.S_zsat_rz ? pot * share : c.S_zsat_rz) * mk;
        c.q_sub_mat_rz = c.q_sub_rz * sh_mat * mk;
        c.q_sub_mp_rz = c.q_sub_rz * sh_mp * mk;
        c.q_sub_mp_pot_rz = c.q_sub_mp_pot * share * mk;
        c.z_sat += -c.q_sub_rz / c.theta_ac * mk;
        c.S_lp_rz += -c.q_sub_rz * mk;
    }
    // potential lateral runoff from the subsoil :460-515
    {
        double share = (dz_ss / c.z_sat) * mk;
        const bool nan0 = isnan(share);
        share = ((c.z_sat <= dz_ss) || (c.S_lp_rz <= 0) ? 1.0 : share) * mk;
        share = (c.z_sat <= 0 ? 0.0 : share) * mk;
        share = (nan0 ? 0.0 : share) * mk;
        c.q_sub_mat_pot_ss = c.q_sub_mat_pot * share * mk;
        c.q_sub_mp_pot_ss = c.q_sub_mp_pot * share * mk;
        c.q_sub_pot_ss = (c.q_sub_mat_pot_ss + c.q_sub_mp_pot_ss) * mk;
    }
    h_sub_pot_percolation_ss(c, dt, mk);
    // vertical and lateral drainage of the subsoil share the perched water :518-690
    {
        const double tot = c.q_pot_ss + c.q_sub_pot_ss;
        const double fv = (tot > 0 ? c.q_pot_ss / tot : 0.0) * mk, fl = (tot > 0 ? c.q_sub_pot_ss / tot : 0.0) * mk;
        const bool fits = tot <= c.S_zsat_ss;
        double q = (c.z_sat <= 0 ? c.q_pot_ss : 0.0) * mk;
        const double q_sat = (fits ? tot * fv : c.S_zsat_ss * fv) * mk;
        q = (c.z_sat > 0 ? q_sat : q);
        c.q_ss = q;
        c.q_sub_ss = (fits ? tot * fl : c.S_zsat_ss * fl) * mk;
        c.q_sub_mat_ss = c.q_sub_ss * sh_mat * mk;
        c.q_sub_mp_ss = c.q_sub_ss * sh_mp * mk;
        const bool m1 = c.S_lp_ss < q, m2 = c.S_lp_ss >= q;
        c.S_fp_ss += (m1 ? -(q - c.S_lp_ss) : 0.0) * mk;
        c.S_lp_ss = (m1 ? 0.0 : c.S_lp_ss) * mk;
        c.S_lp_ss += (m2 ? -q : 0.0) * mk;
        c.S_lp_ss += (c.z_sat > 0 ? -c.q_sub_ss : 0.0) * mk;
        c.z_sat += -((c.q_sub_ss + q) / c.theta_ac) * mk;
        c.z_sat = (c.z_sat < 0 ? 0.0 : c.z_sat) * mk;
        c.S_zsat = c.z_sat * c.theta_ac * mk;
        c.q_sub_mat = (c.q_sub_mat_rz + c.q_sub_mat_ss) * mk;
        c.q_sub_mp = (c.q_sub_mp_rz + c.q_sub_mp_ss) * mk;
        c.q_sub = (c.q_sub_rz + c.q_sub_ss) * mk;
    }
}

// ---------------------------------------------------------------------------------------------
// capillary_rise.py:7-173
// ---------------------------------------------------------------------------------------------
RH_DEV void rt_capillary_rise(Col &c, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    const double z = ((c.z_root + (c.z_soil - c.z_root) / 2) - c.z_root / 2) * mk;
    double q = h_salvucci(z, c.h_rz, c.ha, c.n_salv) * X.dt * c.ks * mk;
    q = (q < 0 ? 0.0 : q) * mk;
    q = (isnan(q) ? 0.0 : q) * mk;
    q = (c.S_lp_rz > 0 ? 0.0 : q) * mk;
    q = (c.h_rz > c.h_ss ? 0.0 : q) * mk;
    q = (q > (c.S_fp_ss + c.S_lp_ss) ? c.S_fp_ss + c.S_lp_ss : q) * mk;
    const double gap = c.S_ufc_rz - c.S_fp_rz;
    q = ((q > gap) && (gap > 0) ? gap : q) * mk;
    c.cpr_rz = q;
    const bool geo = (c.z_wf < c.z_root) || (c.z_sat < c.z_soil - c.z_root);
    const bool up = (q > 0) && geo;
    const bool m1 = up && (c.S_lp_ss <= 0);
    const bool m2 = up && (c.S_lp_ss > 0) && (q <= c.S_lp_ss);
    const bool m3 = up && (c.S_lp_ss > 0) && (q > c.S_lp_ss);
    c.S_fp_rz += (m1 ? q : 0.0) * mk;
    c.S_fp_ss += (m1 ? -q : 0.0) * mk;
    c.S_fp_rz += (m2 ? q : 0.0) * mk;
    c.S_lp_ss += (m2 ? -q : 0.0) * mk;
    c.S_fp_rz += (m3 ? q : 0.0) * mk;
    c.S_fp_ss += (m3 ? -(q - c.S_lp_ss) : 0.0) * mk;
    c.S_lp_ss = (m3 ? 0.0 : c.S_lp_ss) * mk;
    const bool m4 = c.S_fp_rz > c.S_ufc_rz;
    c.S_lp_rz += (m4 ? c.S_fp_rz - c.S_ufc_rz : 0.0) * mk;
    c.S_fp_rz = (m4 ? c.S_ufc_rz : c.S_fp_rz) * mk;
}

// ---------------------------------------------------------------------------------------------
// storages: surface.py:8-37, root_zone.py:7-166, subsoil.py:6-137, soil.py:9-140,
// numerics.py:125-214
// ---------------------------------------------------------------------------------------------
RH_DEV double h_k_bc(double ks, double theta, double theta_sat, double m_bc) { return ks / (1 + RH_POW(theta / theta_sat, -m_bc)); }
RH_DEV double h_h_bc(double ha, double theta, double theta_sat, double lambda_bc) {
    return ha / RH_POW(theta / theta_sat, 1 / lambda_bc);
}

// Conductivity and suction of the root zone and of the subsoil from their water contents (root_zone.py:113-166, subsoil.py:84-137): the
// last thing a step computes of them and the first thing the NEXT step's percolation and capillary rise read.  A lazy fused step (the
// planes were last touched by a complete step: roger_hip.hip) therefore evaluates these helpers from theta_rz / theta_ss in front of its
// subsurface stage instead of loading the five planes, and a sparse step does not store them (rl_rt_subsurface_runoff below).
RH_DEV void h_kh_rz(Col &c, double mk) {
    c.k_rz = h_k_bc(c.ks, c.theta_rz, c.theta_sat, c.m_bc) * mk;
    c.h_rz = h_h_bc(c.ha, c.theta_rz, c.theta_sat, c.lambda_bc) * mk;
}
RH_DEV void h_kh_ss(Col &c, double mk) {
    c.ks_ss = c.ks;
    c.k_ss = h_k_bc(c.ks, c.theta_ss, c.theta_sat, c.m_bc) * mk;
    c.h_ss = h_h_bc(c.ha, c.theta_ss, c.theta_sat, c.lambda_bc) * mk;
}

RH_DEV void rt_storage(Col &c, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    c.S_sur = (c.S_int_top + c.S_int_ground + c.S_dep + c.S_snow + c.z0) * mk;
    c.S_rz = (c.S_pwp_rz + c.S_fp_rz + c.S_lp_rz) * mk;
    c.dS_rz = (c.S_rz - c.S_rz_m1) * mk;
    c.theta_rz = ((c.S_fp_rz + c.S_lp_rz) / c.z_root + c.theta_pwp) * mk;
    if (X.month_tau >= 4 && X.month_tau <= 9) {  // root_zone.py:172-178
        double d = c.theta_irr - c.theta_rz;
        d = (d <= 0 ? 0.0 : d);
        c.irr_demand = d * c.z_root;
    } else {
        c.irr_demand = 0.0;
    }
    h_kh_rz(c, mk);
    c.S_ss = (c.S_pwp_ss + c.S_fp_ss + c.S_lp_ss) * mk;
    c.dS_ss = (c.S_ss - c.S_ss_m1) * mk;
    c.theta_ss = ((c.S_fp_ss + c.S_lp_ss) / (c.z_soil - c.z_root) + c.theta_pwp) * mk;
    h_kh_ss(c, mk);
    c.S_fp_s = (c.S_fp_rz + c.S_fp_ss) * mk;
    c.S_lp_s = (c.S_lp_rz + c.S_lp_ss) * mk;
    c.S_s = (c.S_pwp_s + c.S_fp_s + c.S_lp_s) * mk;
    c.dS_s = (c.S_s - c.S_s_m1) * mk;
    c.theta = ((c.S_fp_s + c.S_lp_s) / c.z_soil + c.theta_pwp) * mk;
    c.k = h_k_bc(c.ks, c.theta, c.theta_sat, c.m_bc) * mk;
    c.h = h_h_bc(c.ha, c.theta, c.theta_sat, c.lambda_bc) * mk;
    c.S = c.S_sur + c.S_s * mk;
    c.dS = c.S - c.S_m1 * mk;
}

// numerics.py: calc_dS_num_error :303-345, sanity_check :979-1011.  Returns true if the column
// violates the mass-balance / pore-space checks.
RH_DEV double h_nan0(double x) { return isnan(x) ? 0.0 : x; }
RH_DEV bool rt_num_error(Col &c, const Consts &K) {
    const double lhs = c.S - c.S_m1, rhs = c.prec - c.q_sur - c.aet - c.q_ss;
    c.dS_num_error = fabs(lhs - rhs);
    c.dS_rz_num_error =
        fabs((c.S_rz - c.S_rz_m1) - (c.inf_mat_rz + c.inf_mp_rz + c.inf_sc_rz + c.cpr_rz - c.transp - c.evap_soil - c.q_rz));
    c.dS_ss_num_error = fabs((c.S_ss - c.S_ss_m1) - (c.inf_mp_ss + c.q_rz - c.q_ss - c.cpr_rz));
    bool close = (isfinite(lhs) && isfinite(rhs)) ? (fabs(lhs - rhs) <= K.atol + K.rtol * fabs(rhs)) : (lhs == rhs);
    close = c.maskCatch ? close : true;
    const double a = h_nan0(c.S_fp_rz), b = h_nan0(c.S_lp_rz), d = h_nan0(c.S_fp_ss), e = h_nan0(c.S_lp_ss);
    const bool lower = (a > -K.atol) && (b > -K.atol) && (d > -K.atol) && (e > -K.atol);
    const bool upper = (a - K.atol <= h_nan0(c.S_ufc_rz)) && (b - K.atol <= h_nan0(c.S_ac_rz)) &&
                       (d - K.atol <= h_nan0(c.S_ufc_ss)) && (e - K.atol <= h_nan0(c.S_ac_ss));
    return !(close && lower && upper);
}

// oneD model: numerics.py:226-245 (only dS_num_error) and the sanity check with q_sub :744-759
RH_DEV bool rt_num_error_lateral(Col &c, const Consts &K) {
    const double lhs = c.S - c.S_m1, rhs = c.prec - c.q_sur - c.aet - c.q_ss - c.q_sub;
    c.dS_num_error = fabs(lhs - rhs);
    bool close = (isfinite(lhs) && isfinite(rhs)) ? (fabs(lhs - rhs) <= K.atol + K.rtol * fabs(rhs)) : (lhs == rhs);
    close = c.maskCatch ? close : true;
    const double a = h_nan0(c.S_fp_rz), b = h_nan0(c.S_lp_rz), d = h_nan0(c.S_fp_ss), e = h_nan0(c.S_lp_ss);
    const bool lower = (a > -K.atol) && (b > -K.atol) && (d > -K.atol) && (e > -K.atol);
    const bool upper = (a - K.atol <= h_nan0(c.S_ufc_rz)) && (b - K.atol <= h_nan0(c.S_ac_rz)) &&
                       (d - K.atol <= h_nan0(c.S_ufc_ss)) && (e - K.atol <= h_nan0(c.S_ac_ss));
    return !(close && lower && upper);
}

// ---------------------------------------------------------------------------------------------
// unidirectional (D8) routing, settings.enable_routing_1D: the per-column parts.  Between `_out` and `_in` the gather kernel
// (k_route_gather, roger_hip.hip) forms q_*_in of every cell from the q_*_out of its eight neighbours.
//   surface_runoff.calc_surface_runoff_routing_1D :14-227, subsurface_runoff.calc_subsurface_runoff_routing_1D :1158-1437
// ---------------------------------------------------------------------------------------------
// sum over the eight directions of where(flow_dir == code_d, q, 0) * maskCatch: at most one term
RH_DEV double h_d8_out(double q, int flow_dir, double mk) {
    const bool has = flow_dir == 64 || flow_dir == 128 || flow_dir == 1 || flow_dir == 2 || flow_dir == 4 || flow_dir == 8 ||
                     flow_dir == 16 || flow_dir == 32;
    return has ? (q * mk) * mk : 0.0 * mk;
}
RH_DEV void rt_route_surface_out(Col &c, const Consts &K, const StepCtx &X, double dt_secs) {
    const double mk = (double)c.maskCatch;
    c.z0 += c.q_sof * mk;
    const double area = (c.z0 / 1000) * 0.5 * (2 * K.dx) * mk;
    const double perimeter = 2 * (c.z0 / 1000) + K.dx * mk;
    const double radius = area / perimeter * mk;
    // Manning-Strickler, m3/s to mm per step
    c.q_sur = c.k_st * RH_POW(c.slope, 0.5) * RH_POW(radius, 2.0 / 3.0) * area * (dt_secs / (K.dx * K.dy * 1000)) * mk;
    c.q_sur = (c.q_sur > c.z0 ? c.z0 : c.q_sur) * mk;
    c.q_sur_out = h_d8_out(c.q_sur, c.flow_dir_topo, mk);
}
RH_DEV void rt_route_surface_in(Col &c) {
    const double mk = (double)c.maskCatch;
    c.q_sur_in = c.q_sur_in * mk;
    c.q_sur_in = (c.outer_boundary == 1 ? 0.0 : c.q_sur_in) * mk;
    c.z0 += -c.q_sur_out * mk;
    c.z0 += c.q_sur_in * mk;
}
RH_DEV void rt_route_subsurface_out(Col &c) { c.q_sub_out = h_d8_out(c.q_sub, c.flow_dir_topo, (double)c.maskCatch); }
RH_DEV void rt_route_subsurface_in(Col &c) {   // :1311-1437
    const double mk = (double)c.maskCatch;
    const double S1_rz = c.S_fp_rz + c.S_lp_rz, S1_ss = c.S_fp_ss + c.S_lp_ss;
    c.q_sub_in = c.q_sub_in * mk;
    c.q_sub_in = (c.outer_boundary == 1 ? 0.0 : c.q_sub_in) * mk;
    c.z_sat += (c.q_sub_in / c.theta_ac) * mk;
    c.z_sat = (c.z_sat < 0 ? 0.0 : c.z_sat) * mk;
    c.S_zsat = c.z_sat * c.theta_ac * mk;
    c.S_lp_ss += c.q_sub_in * mk;
    const bool over = c.S_lp_ss > c.S_ac_ss;
    c.S_lp_rz += (over ? c.S_lp_ss - c.S_ac_ss : 0.0) * mk;
    c.S_lp_ss = (over ? c.S_ac_ss : c.S_lp_ss) * mk;
    // saturation overland flow
    c.q_sof += (((c.S_lp_rz + c.S_fp_rz) > (c.S_ac_rz + c.S_ufc_rz)) ? (c.S_lp_rz + c.S_fp_rz) - (c.S_ac_rz + c.S_ufc_rz) : 0.0) * mk;
    c.q_sur += c.q_sof * mk;
    c.z0 += c.q_sof * mk;
    const bool sof = c.q_sof > 0;
    c.S_fp_rz = (sof ? c.S_ufc_rz : c.S_fp_rz) * mk;
    c.S_lp_rz = (sof ? c.S_ac_rz : c.S_lp_rz) * mk;
    c.q_sub_in_rz = (c.S_fp_rz + c.S_lp_rz) - S1_rz;
    c.q_sub_in_ss = (c.S_fp_ss + c.S_lp_ss) - S1_ss;
}
// numerics.py:247-270 (dS_num_error) and the sanity check with the routed fluxes :778-815
RH_DEV bool rt_num_error_routed(Col &c, const Consts &K) {
    const double lhs = c.S - c.S_m1, rhs = c.prec - c.q_sur_out + c.q_sur_in - c.aet - c.q_ss - c.q_sub_out + c.q_sub_in;
    c.dS_num_error = fabs(lhs - rhs);
    bool close = (isfinite(lhs) && isfinite(rhs)) ? (fabs(lhs - rhs) <= K.atol + K.rtol * fabs(rhs)) : (lhs == rhs);
    close = c.maskCatch ? close : true;
    const double a = h_nan0(c.S_fp_rz), b = h_nan0(c.S_lp_rz), d = h_nan0(c.S_fp_ss), e = h_nan0(c.S_lp_ss);
    const bool lower = (a > -K.atol) && (b > -K.atol) && (d > -K.atol) && (e > -K.atol);
    const bool upper = (a - K.atol <= h_nan0(c.S_ufc_rz)) && (b - K.atol <= h_nan0(c.S_ac_rz)) &&
                       (d - K.atol <= h_nan0(c.S_ufc_ss)) && (e - K.atol <= h_nan0(c.S_ac_ss));
    return !(close && lower && upper);
}

// tau -> taum1 of the prognostic variables (models/svat/svat.py:187-324, models/oneD/oneD.py)
RH_DEV void h_rotate(Col &c) {
    c.ta_m1 = c.ta;
    c.z_root_m1 = c.z_root;
    c.ground_cover_m1 = c.ground_cover;
    c.S_sur_m1 = c.S_sur;
    c.S_int_top_m1 = c.S_int_top;
    c.S_int_ground_m1 = c.S_int_ground;
    c.S_dep_m1 = c.S_dep;
    c.S_snow_m1 = c.S_snow;
    c.swe_m1 = c.swe;
    c.S_rz_m1 = c.S_rz;
    c.S_ss_m1 = c.S_ss;
    c.S_s_m1 = c.S_s;
    c.S_m1 = c.S;
    c.z_sat_m1 = c.z_sat;
    c.z_wf_m1 = c.z_wf;
    c.z_wf_t0_m1 = c.z_wf_t0;
    c.z_wf_t1_m1 = c.z_wf_t1;
    c.y_mp_m1 = c.y_mp;
    c.y_sc_m1 = c.y_sc;
    c.theta_rz_m1 = c.theta_rz;
    c.theta_ss_m1 = c.theta_ss;
    c.theta_m1 = c.theta;
    c.k_rz_m1 = c.k_rz;
    c.k_ss_m1 = c.k_ss;
    c.k_m1 = c.k;
    c.h_rz_m1 = c.h_rz;
    c.h_ss_m1 = c.h_ss;
    c.h_m1 = c.h;
    c.z0_m1 = c.z0;
    c.prec_m1 = c.prec;
}
// models/svat/svat.py:187-384: rotation + tiny negative pore storages snapped to zero (:326-345)
RH_DEV double h_snap0(double x) { return ((x > -1e-6) && (x < 0)) ? 0.0 : x; }
RH_DEV void rt_after_timestep(Col &c) {
    h_rotate(c);
    c.S_fp_rz = h_snap0(c.S_fp_rz);
    c.S_lp_rz = h_snap0(c.S_lp_rz);
    c.S_fp_ss = h_snap0(c.S_fp_ss);
    c.S_lp_ss = h_snap0(c.S_lp_ss);
}
// the oneD model's after_timestep_kernel has the rotation only
RH_DEV void rt_after_timestep_oned(Col &c) { h_rotate(c); }

// ---------------------------------------------------------------------------------------------
// setup-time / monthly parameter kernels
// ---------------------------------------------------------------------------------------------
RH_DEV int h_lut_row(const double *lut, int ncol, int key) {  // utilities._get_row_no: first match, else row 0
    for (int r = 0; r < 25; ++r)
        if (lut[r * ncol] == (double)key) return r;
    return 0;
}

// surface.py:40-71
RH_DEV void rt_topo(Col &c) {
    c.maskRiver = (c.lu_id == 20);
    c.maskLake = (c.lu_id == 14);
    c.maskCatch = (c.lu_id != 14) && (c.lu_id != 20) && (c.lu_id != 999) && c.maskCatch;
}

// surface.py:74-343: land-use x month look-ups.  Membership tests replace the reference's loops
// over land-use ids; a land use missing from a table falls back to the table's first row.
RH_DEV void rt_params_surface(Col &c, const Luts &L, const StepCtx &X) {
    const double mk = (double)c.maskCatch;
    const int lu = c.lu_id, m = (int)X.month_tau;
    const bool water = c.maskRiver || c.maskLake;
    const bool conifer = (lu == 10) || (lu == 11) || (lu == 12);
    const bool tree_top = conifer || (lu == 15);                 // {10,11,12,15,17} within 10..15
    const bool low_veg = (lu == 0) || (lu >= 5 && lu <= 9) || (lu == 13) || (lu == 31) || (lu == 32) || (lu == 33) ||
                         (lu == 40) || (lu == 41) || (lu == 50) || (lu == 60);  // 98 lies outside the loop 0..80
    const bool covered = low_veg || conifer || (lu == 15);      // cc_cond within 0..80
    const int r_ilu = h_lut_row(L.ilu, 13, lu), r_gc = h_lut_row(L.gc, 13, lu);
    double v = (tree_top ? L.ilu[r_ilu * 13 + m] : 0.0) * mk;
    c.S_int_top_tot = v * c.c_int * mk;
    v = (low_veg ? L.ilu[r_ilu * 13 + m] : 0.0) * mk;
    v = (tree_top ? 1.0 : v) * mk;                               // {10,11,12,15,16} within 10..15
    c.S_int_ground_tot = v * c.c_int * mk;
    const double gc = L.gc[r_gc * 13 + m], gcm = L.gcm[r_gc * 2 + 1];
    c.ground_cover = ((covered ? gc : 0.0) * mk) * mk;
    v = (covered ? gc / gcm : 0.0) * mk;
    c.basal_transp_coeff = (water ? 0.0 : v) * mk;
    v = (covered ? 1 - ((gc / gcm) * gcm) : 0.0) * mk;
    // `maskRiver | maskLake | lu_id == 0` binds as (maskRiver | maskLake | lu_id) == 0, surface.py:230
    c.basal_evap_coeff = ((((c.maskRiver ? 1 : 0) | (c.maskLake ? 1 : 0) | lu) == 0) ? 1.0 : v) * mk;
    c.swe_top_tot = h_swe_top_tot(c.swe_top_tot, c.ta, lu, mk);
    c.lai = log(1 / (1 - c.ground_cover)) / log(1 / 0.7) * mk;
    const double tf = (c.lai > 1 ? 0.1 : 1.0 - c.lai);
    c.throughfall_coeff_top = (conifer ? tf : 0.0) * mk;
    c.throughfall_coeff_ground = ((lu >= 500 && lu < 598) ? tf : 0.0) * mk;
}

// The parameters of calc_parameters_soil / _root_zone / _subsoil (soil.py:143-557) that follow from the primaries by a handful of IEEE
// operations, grouped by the stage of the fused step that needs them first.  rt_params_soil (the setup kernel) is built from these
// helpers, and the fused step evaluates THE SAME helpers instead of loading the 15 planes (k_step, DevState::pmask bit 63: the wave's
// planes were found to hold exactly these values) -- one definition, so the bits agree by construction (-ffp-contract=off: no
// contraction; division and comparison are IEEE on the device).  
RH_DEV void h_der_porosity(Col &c, double mk) {        // first needed by the evapotranspiration stage
    const double por = c.theta_ac + c.theta_ufc + c.theta_pwp, fc = c.theta_ufc + c.theta_pwp;
    c.theta_sat = por * mk;
    c.theta_fc = fc * mk;
}
RH_DEV void h_der_evap_depth(Col &c, const Consts &K, double rew_before, double mk) {
    double rew = (c.theta_pwp < K.theta_rew_min ? K.rew_min : rew_before) * mk;
    rew = ((c.theta_pwp >= K.theta_rew_min) && (c.theta_pwp <= K.theta_rew_max) ? c.theta_pwp / K.theta_rew_max : rew) * mk;
    c.rew = (c.theta_pwp > K.theta_rew_max ? K.rew_max : rew) * mk;
    c.z_evap = ((c.rew / K.rew_max) * K.z_evap_max) * mk;
    c.tew = ((c.theta_fc - 0.5 * c.theta_pwp) * c.z_evap) * mk;
}
RH_DEV void h_der_wfs(Col &c, double mk) {   // from lambda_bc and ha
    c.wfs = (((2 + 3 * c.lambda_bc) / (1 + 3 * c.lambda_bc) * c.ha / 2) * (-10)) * mk;
}
RH_DEV void h_der_n_salv(Col &c, const Consts &K, double mk) {
    const double nb = K.a_bc + K.b_bc * c.lambda_bc;
    c.n_salv = nb * mk;
}
RH_DEV void h_der_m_bc(Col &c, const Consts &K, double mk) {
    const double nb = K.a_bc + K.b_bc * c.lambda_bc;
    c.m_bc = (nb / c.lambda_bc) * mk;
}
RH_DEV void h_der_S_ac_rz(Col &c, double zr, double mk) { c.S_ac_rz = (c.theta_ac * zr) * mk; }
RH_DEV void h_der_S_ufc_rz(Col &c, double zr, double mk) { c.S_ufc_rz = (c.theta_ufc * zr) * mk; }
RH_DEV void h_der_S_pwp_rz(Col &c, double zr, double mk) { c.S_pwp_rz = (c.theta_pwp * zr) * mk; }
RH_DEV void h_der_S_ac_ufc_ss(Col &c, double zr, double mk) {
    const double dz = c.z_soil - zr;
    c.S_ac_ss = (c.theta_ac * dz) * mk;
    c.S_ufc_ss = (c.theta_ufc * dz) * mk;
}
RH_DEV void h_der_S_pwp_ss_s(Col &c, double zr, double mk) {
    const double dz = c.z_soil - zr;
    c.S_pwp_ss = (c.theta_pwp * dz) * mk;
    c.S_pwp_s = (c.z_soil * c.theta_pwp) * mk;
}

// soil.py:143-557 (calc_parameters_soil/root_zone/subsoil kernels)
RH_DEV void rt_params_soil(Col &c, const Consts &K, const Luts &L) {
    const double mk = (double)c.maskCatch;
    const int lu = c.lu_id;
    const double por = c.theta_ac + c.theta_ufc + c.theta_pwp, fc = c.theta_ufc + c.theta_pwp;
    c.S_ac_s = (c.z_soil * c.theta_ac) * mk;
    c.S_ufc_s = (c.z_soil * c.theta_ufc) * mk;
    c.S_fc_s = (c.z_soil * fc) * mk;
    c.S_sat_s = (c.z_soil * por) * mk;
    h_der_porosity(c, mk);
    c.lambda_bc = ((log(c.theta_fc / c.theta_sat) - log(c.theta_pwp / c.theta_sat)) / (log(15850.0) - log(63.0))) * mk;
    c.ha = (RH_POW(c.theta_pwp / c.theta_sat, 1.0 / c.lambda_bc) * (-15850)) * mk;
    h_der_m_bc(c, K, mk);
    h_der_n_salv(c, K, mk);
    h_der_wfs(c, mk);
    c.theta_27 = (RH_POW(c.ha / (-501.18723362727246), c.lambda_bc) * c.theta_sat) * mk;  // 10**2.7
    c.theta_4 = (RH_POW(c.ha / (-10000.0), c.lambda_bc) * c.theta_sat) * mk;
    c.theta_6 = (RH_POW(c.ha / (-1000000.0), c.lambda_bc) * c.theta_sat) * mk;
    double s = (1 * (c.theta_ac / 0.24)) * mk;
    s = (s < 0 ? 0.0 : s) * mk;
    c.sand = (s > 1 ? 1.0 : s) * mk;
    double cl = (K.clay_max * (c.theta_6 - K.clay_min) / 0.3) * mk;
    c.clay = (cl < K.clay_min ? K.clay_min : cl) * mk;
    c.z_sc_max = (c.clay * 700) * mk;
    c.mp_drain_area = 1 - exp((-1) * RH_POW(c.dmpv / 82, 0.887)) * mk;
    h_der_evap_depth(c, K, c.rew, mk);

    // rooting depth from land use :338-440
    const bool conifer = (lu == 10) || (lu == 11) || (lu == 12);
    const bool listed = (lu == 0) || (lu >= 5 && lu <= 13) || (lu == 15) || (lu == 31) || (lu == 32) || (lu == 33) ||
                        (lu == 40) || (lu == 41) || (lu == 50) || (lu == 60);  // cc_cond within the loop 0..60
    double zr = (listed ? L.rdlu[h_lut_row(L.rdlu, 7, lu) * 7 + 1] : c.z_root_m1) * mk;
    zr = (c.maskRiver || c.maskLake) ? 0.0 : zr;
    zr = (conifer || (lu == 15) || (lu == 16) || (lu == 17) ? 1500.0 : zr) * mk;
    zr = (lu == 100 ? 300.0 : zr) * mk;
    zr = (zr >= c.z_soil ? K.zroot_to_zsoil_max * c.z_soil : zr) * mk;
    zr = zr * c.c_root;
    zr = ((lu >= 500 && lu < 600) ? 200.0 : zr) * mk;
    zr = (zr < c.z_soil ? zr : c.z_soil * 0.9);
    c.z_root = zr;
    c.z_root_m1 = zr;
    h_der_S_ac_rz(c, zr, mk);
    h_der_S_ufc_rz(c, zr, mk);
    h_der_S_pwp_rz(c, zr, mk);
    c.S_sat_rz = (por * zr) * mk;
    c.S_fc_rz = (fc * zr) * mk;
    h_der_S_ac_ufc_ss(c, zr, mk);
    h_der_S_pwp_ss_s(c, zr, mk);
    const double dz = c.z_soil - zr;
    c.S_sat_ss = (por * dz) * mk;
    c.S_fc_ss = (fc * dz) * mk;
}

// The fused step's own evaluation of those parameters: every stage derives what it is the FIRST to mention (tools/gen_sets.py checks
// that against the sequences), so a derived value is live exactly as long as the loaded one was.  The primaries -- theta_ac, theta_ufc,
// theta_pwp, z_soil, z_root, lambda_bc, ha, maskCatch -- are in registers by then (RH_DERIVE_EARLY_* in roger_hip.hip moves theta_ac,
// lambda_bc and ha forward).  rew's "value before" only survives where theta_pwp is NaN: 0 here, and the check that sets the wave's
// bit (k_param_mask) compares every result with the planes bit for bit.
RH_DEV void rd_rt_evapotranspiration(Col &c, const Consts &K) {
    const double mk = (double)c.maskCatch;
    h_der_porosity(c, mk);
    h_der_evap_depth(c, K, 0.0, mk);
    h_der_S_ac_rz(c, c.z_root, mk);
}
RH_DEV void rd_rt_inf_matrix(Col &c, const Consts &K) {
    const double mk = (double)c.maskCatch;
    h_der_wfs(c, mk);
    h_der_S_ufc_rz(c, c.z_root, mk);
}
RH_DEV void rd_rt_inf_macropores(Col &c, const Consts &K) { h_der_S_ac_ufc_ss(c, c.z_root, (double)c.maskCatch); }
RH_DEV void rd_rt_subsurface_runoff(Col &c, const Consts &K) {
    const double mk = (double)c.maskCatch;
    h_der_n_salv(c, K, mk);
    h_der_m_bc(c, K, mk);   // (rl_rt_subsurface_runoff needs it)
}
RH_DEV void rd_rt_storage(Col &c, const Consts &K) {
    const double mk = (double)c.maskCatch;
    h_der_S_pwp_rz(c, c.z_root, mk);
    h_der_S_pwp_ss_s(c, c.z_root, mk);
}
// all of them at once (the check kernel)
RH_DEV void rd_all(Col &c, const Consts &K) {
    rd_rt_evapotranspiration(c, K);
    rd_rt_inf_matrix(c, K);
    rd_rt_inf_macropores(c, K);
    rd_rt_subsurface_runoff(c, K);
    rd_rt_storage(c, K);
}
RH_DEV void rd_rt_subsurface_runoff_lateral(Col &c, const Consts &K) { rd_rt_subsurface_runoff(c, K); }
// STATE that a lazy step derives instead of loading (rl_<stage>: every lazy kernel, whatever the wave's word says): conductivity and
// suction at the water contents the previous step left.  Valid because the previous operation on the planes was a complete step, whose
// storage stage computed exactly these from the same theta_rz / theta_ss and the same parameters.
RH_DEV void rl_rt_subsurface_runoff(Col &c, const Consts &K) {
    const double mk = (double)c.maskCatch;
    h_kh_rz(c, mk);
    h_kh_ss(c, mk);
}
RH_DEV void rl_rt_subsurface_runoff_lateral(Col &c, const Consts &K) { rl_rt_subsurface_runoff(c, K); }
RH_DEV void rl_rt_evapotranspiration(Col &c, const Consts &K) {}
RH_DEV void rl_rt_inf_matrix(Col &c, const Consts &K) {}
RH_DEV void rl_rt_inf_macropores(Col &c, const Consts &K) {}
RH_DEV void rl_rt_storage(Col &c, const Consts &K) {}

// soil.py:560-641: horizontal macropore flow velocity per layer from the slope look-up table
// lut_mlms (rows: slope in percent, then m/h of layers 8..1), converted to mm/h
RH_DEV void rt_params_lateral(Col &c, const double *mlms, int64_t nrows, int max_slope_per) {
    const double mk = (double)c.maskCatch;
    const int key = c.slope_per;
    const bool hit = (key >= 1) && (key <= max_slope_per);   // the reference loops i = 1 .. max(slope_per)
    int64_t row = 0;                                         // _get_row_no: first match, else row 0
    if (hit) {
        if (key <= nrows && mlms[(int64_t)(key - 1) * 9] == (double)key) {
            row = key - 1;
        } else {
            for (int64_t r = 0; r < nrows; ++r)
                if (mlms[r * 9] == (double)key) { row = r; break; }
        }
    }
    const double *m = mlms + row * 9;
    c.v_mp_layer_8 = ((hit ? m[1] * 1000 : 0.0) * mk) * mk;
    c.v_mp_layer_7 = ((hit ? m[2] * 1000 : 0.0) * mk) * mk;
    c.v_mp_layer_6 = ((hit ? m[3] * 1000 : 0.0) * mk) * mk;
    c.v_mp_layer_5 = ((hit ? m[4] * 1000 : 0.0) * mk) * mk;
    c.v_mp_layer_4 = ((hit ? m[5] * 1000 : 0.0) * mk) * mk;
    c.v_mp_layer_3 = ((hit ? m[6] * 1000 : 0.0) * mk) * mk;
    c.v_mp_layer_2 = ((hit ? m[7] * 1000 : 0.0) * mk) * mk;
    c.v_mp_layer_1 = ((hit ? m[8] * 1000 : 0.0) * mk) * mk;
}

// Splits an initial water content into fine / large pore fractions, soil.py:767-800 / :852-885
RH_DEV void h_split_theta(double theta, const Col &c, double &fp, double &lp, double mk) {
    fp = (theta > c.theta_pwp ? theta - c.theta_pwp : fp) * mk;
    fp = (theta <= c.theta_pwp ? 0.0 : fp) * mk;
    fp = (fp >= c.theta_ufc ? c.theta_ufc : fp) * mk;
    lp = (theta > c.theta_fc ? theta - c.theta_fc : lp) * mk;
    lp = (theta <= c.theta_fc ? 0.0 : lp) * mk;
}

// surface.py:398-414, soil.py:742-948
RH_DEV void rt_initial_conditions(Col &c) {
    const double mk = (double)c.maskCatch;
    c.S_sur = (c.S_int_top + c.S_int_ground + c.S_dep + c.S_snow) * mk;
    c.S_sur_m1 = (c.S_int_top_m1 + c.S_int_ground_m1 + c.S_dep_m1 + c.S_snow_m1) * mk;
    h_split_theta(c.theta_rz, c, c.theta_fp_rz, c.theta_lp_rz, mk);
    c.S_fp_rz = (c.theta_fp_rz * c.z_root) * mk;
    c.S_lp_rz = (c.theta_lp_rz * c.z_root) * mk;
    c.S_rz = (c.S_pwp_rz + c.S_fp_rz + c.S_lp_rz) * mk;
    c.S_rz_m1 = c.S_rz;
    c.theta_rz = ((c.S_fp_rz + c.S_lp_rz) / c.z_root + c.theta_pwp) * mk;
    c.k_rz = h_k_bc(c.ks, c.theta_rz, c.theta_sat, c.m_bc) * mk;
    c.h_rz = h_h_bc(c.ha, c.theta_rz, c.theta_sat, c.lambda_bc) * mk;
    const double dz = c.z_soil - c.z_root;
    h_split_theta(c.theta_ss, c, c.theta_fp_ss, c.theta_lp_ss, mk);
    c.S_fp_ss = (c.theta_fp_ss * dz) * mk;
    c.S_lp_ss = (c.theta_lp_ss * dz) * mk;
    c.S_ss = (c.S_pwp_ss + c.S_fp_ss + c.S_lp_ss) * mk;
    c.S_ss_m1 = c.S_ss;
    c.theta_ss = ((c.S_fp_ss + c.S_lp_ss) / dz + c.theta_pwp) * mk;
    c.k_ss = h_k_bc(c.ks, c.theta_ss, c.theta_sat, c.m_bc) * mk;
    c.h_ss = h_h_bc(c.ha, c.theta_ss, c.theta_sat, c.lambda_bc) * mk;
    c.S_fp_s = (c.S_fp_rz + c.S_fp_ss) * mk;
    c.S_lp_s = (c.S_lp_rz + c.S_lp_ss) * mk;
    c.S_s = (c.S_rz + c.S_ss) * mk;
    c.S_s_m1 = (c.S_rz_m1 + c.S_ss_m1) * mk;
    c.theta = (c.S_s / c.z_soil) * mk;
    c.theta_m1 = (c.S_s_m1 / c.z_soil) * mk;
    c.S = c.S_sur + c.S_s * mk;
    c.S_m1 = c.S_sur_m1 + c.S_s_m1 * mk;
}

// ---------------------------------------------------------------------------------------------
// adaptive time stepping, per-column parts (adaptive_time_stepping.py:128-189, 262-376)
// ---------------------------------------------------------------------------------------------
// prec/ta for the step class chosen from the day's global predicates (the reference applies the
// daily, hourly and 10-minute assignments in that order, so the last one that is enabled wins)
RH_DEV void rt_select_prec_ta(Col &c, const StepCtx &X, double prec_v, double ta_v) {
    if (X.sel_p >= 0) {
        c.prec = prec_v;
        c.ta = ta_v;
    }
}
// the same selection inside the fused kernel (summary path, shared forcing: the selected values are uniform)
// (prec_v / ta_v: X.prec_sel / X.ta_sel, or the column's own aggregates when the per-cell selection was deferred to this kernel)
RH_DEV void rt_select_prec(Col &c, const StepCtx &X, double prec_v, double ta_v) {
    if (X.apply_sel && X.sel_p >= 0) {
        c.prec = prec_v;
        c.ta = ta_v;
    }
}
// pet/ta for the final step class (cond6..cond11) and the residual PET, :262-376
RH_DEV void rt_select_pet(Col &c, const StepCtx &X, double pet_v, double ta_v) {
    if (X.sel_w >= 0) {
        c.pet = pet_v;
        c.ta = ta_v;
    }
    c.pet_res = c.pet;
}

// ---------------------------------------------------------------------------------------------
// the whole step for one column, in the order of RogerSetup.step (roger/roger.py:396-485)
// ---------------------------------------------------------------------------------------------
// interception ... numerics: everything between the `set_parameters` and `after_timestep` hooks
RH_DEV bool rt_step_core(Col &c, const Consts &K, const StepCtx &X) {
    rt_interception(c, K);
    rt_evapotranspiration(c, K);
    rt_snow(c, K, X);
    rt_infiltration(c, K, X);
    rt_subsurface_runoff(c, X);
    rt_capillary_rise(c, X);
    rt_storage(c, X);
    return rt_num_error(c, K);
}
RH_DEV bool rt_step_core_lateral(Col &c, const Consts &K, const StepCtx &X) {   // oneD model
    rt_interception(c, K);
    rt_evapotranspiration(c, K);
    rt_snow(c, K, X);
    rt_infiltration(c, K, X);
    rt_subsurface_runoff_lateral(c, K, X);
    rt_capillary_rise(c, X);
    rt_storage(c, X);
    return rt_num_error_lateral(c, K);
}
// settings.enable_routing_1D: the step core in the three passes the two gathers of the routing cut it into (roger/roger.py:410-447)
RH_DEV void rt_routed_a(Col &c, const Consts &K, const StepCtx &X, double dt_secs) {
    rt_interception(c, K);
    rt_evapotranspiration(c, K);
    rt_snow(c, K, X);
    rt_infiltration_routed(c, K, X);
    rt_route_surface_out(c, K, X, dt_secs);
}
RH_DEV void rt_routed_b(Col &c, const Consts &K, const StepCtx &X) {
    rt_route_surface_in(c);
    rt_subsurface_runoff_lateral(c, K, X);
    rt_route_subsurface_out(c);
}
RH_DEV bool rt_routed_c(Col &c, const Consts &K, const StepCtx &X) {
    rt_route_subsurface_in(c);
    rt_capillary_rise(c, X);
    rt_storage(c, X);
    return rt_num_error_routed(c, K);
}
RH_DEV bool rt_routed_c_after(Col &c, const Consts &K, const StepCtx &X) {   // ... with after_timestep (rh_step_routed)
    const bool bad = rt_routed_c(c, K, X);
    rt_after_timestep_oned(c);
    return bad;
}
RH_DEV bool rt_step(Col &c, const Consts &K, const StepCtx &X, double pet_v, double ta_v) {
    rt_select_prec(c, X, X.prec_sel, X.ta_sel);
    rt_select_pet(c, X, pet_v, ta_v);
    const bool bad = rt_step_core(c, K, X);
    rt_after_timestep(c);
    return bad;
}
// first step of a month: `set_parameters` re-derives the surface parameters (svat.py:115-120)
RH_DEV bool rt_step_monthly(Col &c, const Consts &K, const StepCtx &X, const Luts &L, double pet_v, double ta_v) {
    rt_select_prec(c, X, X.prec_sel, X.ta_sel);
    rt_select_pet(c, X, pet_v, ta_v);
    rt_params_surface(c, L, X);
    const bool bad = rt_step_core(c, K, X);
    rt_after_timestep(c);
    return bad;
}
RH_DEV bool rt_step_lateral(Col &c, const Consts &K, const StepCtx &X, double pet_v, double ta_v) {
    rt_select_prec(c, X, X.prec_sel, X.ta_sel);
    rt_select_pet(c, X, pet_v, ta_v);
    const bool bad = rt_step_core_lateral(c, K, X);
    rt_after_timestep_oned(c);
    return bad;
}
RH_DEV bool rt_step_lateral_monthly(Col &c, const Consts &K, const StepCtx &X, const Luts &L, double pet_v, double ta_v) {
    rt_select_prec(c, X, X.prec_sel, X.ta_sel);
    rt_select_pet(c, X, pet_v, ta_v);
    rt_params_surface(c, L, X);
    const bool bad = rt_step_core_lateral(c, K, X);
    rt_after_timestep_oned(c);
    return bad;
}
