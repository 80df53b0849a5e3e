/*
 * include/roger_hip_sas.h -- C ABI of the MI355X-native ("hip") backend for RoGeR's offline
 * oxygen-18 transport step with StorAge-selection (SAS) functions, deterministic solver.
 *
 * Replaces, for `enable_offline_transport and enable_oxygen18 and sas_solver == "deterministic"`,
 * the body of `svat_transport_model_deterministic` (roger/core/transport.py:949-991), i.e. the
 * `@roger_kernel`s
 *   calc_infiltration_rz_transport_iso_kernel      core/infiltration.py:2218-2346
 *   calc_evaporation_transport_iso_kernel          core/evapotranspiration.py:653-719
 *   calc_transpiration_transport_iso_kernel        core/evapotranspiration.py:831-901
 *   calc_percolation_rz_transport_iso_kernel       core/subsurface_runoff.py:1531-1626
 *   calc_infiltration_ss_transport_iso_kernel      core/infiltration.py:2441-2512
 *   calc_percolation_ss_transport_iso_kernel       core/subsurface_runoff.py:1753-1820
 *   calc_capillary_rise_rz_transport_iso_kernel    core/capillary_rise.py:404-500
 *   calc_root_zone/subsoil/soil_transport_iso_kernel  core/root_zone.py:189-217, subsoil.py:159-188, soil.py:1036-1090
 *   calculate_age_statistics_*                     core/transport.py:59-312
 *   calc_ageing_sa_msa_iso_kernel                  core/transport.py:682-739, 780-805
 * with the helpers calc_SA, calc_tt, calc_mtt, calc_conc_iso_flux, calc_conc_iso_storage,
 * conc_to_delta, update_sa (transport.py:315-619) and the SAS families of core/sas.py: uniform (code 1),
 * dirac (2), kumaraswami (3, 31-37), gamma (4), exponential (51, 52), power (6, 61, 62).  rh_sas_sync reports a column
 * whose code is none of these (RH_ERR_STATE).
 *
 * With `tracer = RH_SAS_TRACER_BROMIDE` (settings.enable_bromide) the same step runs the reference's anion kernels
 * instead, where msa_* hold solute MASS by age and C_* are concentrations in mg/l:
 *   calc_infiltration_rz/ss_transport_anion_kernel core/infiltration.py:2350-2424, 2516-2566
 *   calc_evaporation_transport_kernel              core/evapotranspiration.py:620-650 (water only)
 *   calc_transpiration_transport_anion_kernel      core/evapotranspiration.py:905-985 (alpha_transp, crop uptake switch)
 *   calc_percolation_rz/ss_transport_anion_kernel  core/subsurface_runoff.py:1630-1716, 1823-1893 (alpha_q)
 *   calc_capillary_rise_rz_transport_anion_kernel  core/capillary_rise.py:503-590
 *   calc_root_zone/subsoil_transport_anion_kernel, calculate_soil_transport_anion_kernel
 *                                                  core/root_zone.py:221-258, subsoil.py:186-223, soil.py:1094-1142
 *   calc_ageing_sa_kernel, calc_ageing_msa_kernel  core/transport.py:623-680, 743-778
 * with calc_mtt's anion branch (transport.py:583-596); RH_SAS_RESCALE then follows rescale_sa_msa_anion_soil_kernel's
 * bromide branch (core/soil.py:1399-1506) or, with RH_SAS_TRACER_CHLORIDE (settings.enable_chloride), its chloride
 * branch (:1507-1640), which RH_SAS_TRACER_VIRTUAL shares.  Nitrate is not implemented.
 *
 * Same conventions as roger_hip.h: plain pointers and sizes, 0 / negative rh_status returns,
 * rh_sas_last_error for the text, one context = one HIP device + one stream, asynchronous
 * launches fenced by rh_sas_sync / rh_sas_download.
 *
 * Data layout.  All arrays are float64 (maskCatch: int32, read as the reference's bool -- roger/variables.py:462-470 --: any non-zero
 * value is 1) over the rank's interior cells in C
 * order (x, y); age-resolved arrays are (n_cells, ages) / (n_cells, ages + 1) with the age axis
 * contiguous, exactly the reference's `vs.sa_rz[2:-2, 2:-2, vs.tau, :]` etc., so that one
 * workgroup streams one column's age vector with unit stride.  Only the prognostic state
 * (sa_rz, msa_rz, sa_ss, msa_ss) has to live in HBM; the per-flux distributions (tt, mtt, TT),
 * sa_s and msa_s are diagnostics that are written only when the context was created with
 * `keep_distributions` (they are 17 more age vectors per column).
 */
#ifndef ROGER_HIP_SAS_H
#define ROGER_HIP_SAS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RH_SAS_TRACER_OXYGEN18 0
#define RH_SAS_TRACER_BROMIDE 1
#define RH_SAS_TRACER_CHLORIDE 2 /* the anion kernels as for bromide; RH_SAS_RESCALE scales the solute with the water */
#define RH_SAS_TRACER_VIRTUAL 3  /* settings.enable_virtualtracer: as chloride, and the soil evaporation takes the tracer along
                                    (calc_evaporation_transport_virtualtracer_kernel, core/evapotranspiration.py:722-791) */
/* settings.sas_solver (roger/settings.py:119): "deterministic" (svat_transport_model_deterministic, core/transport.py:949-991) or the
 * explicit "Euler" scheme (svat_transport_model_euler :2064-2414, isotope and anion branches): every sub-step of length settings.h = 1 / substeps
 * evaluates all five fluxes on the StorAge as it stands; "RK4" (svat_transport_model_rk4 :1139-2047, both branches) evaluates them
 * four times per sub-step on trial StorAges and updates with the weighted mean of the four travel time distributions. */
#define RH_SAS_SOLVER_DETERMINISTIC 0
#define RH_SAS_SOLVER_EULER 1
#define RH_SAS_SOLVER_RK4 2
#define RH_SAS_MAX_NAGES 4096 /* ages + 1 <= this (benchmark: ages = 1000, SVATOXYGEN18_benchmark.py:28-44) */

typedef struct rh_sas_config {
    int64_t n_cells;           /* local interior cells (nx * ny of this rank) */
    int32_t ages;              /* settings.ages; nages = ages + 1 */
    int32_t substeps;          /* settings.sas_solver_substeps */
    int32_t device;            /* HIP device ordinal */
    int32_t forcing_days;      /* number of days of daily input resident on the device (>= 1) */
    int32_t age_statistics;    /* settings.enable_age_statistics */
    int32_t keep_distributions;/* also write tt_*, mtt_*, TT_*, sa_s, msa_s (diagnostics) */
    double vsmow, d18O_min, d18O_max; /* settings.VSMOW_conc18O, d18O_min, d18O_max (roger/settings.py:76-78); for
                                       * settings.enable_deuterium: VSMOW_conc2H, d2H_min, d2H_max (:79-81), same kernels */
    int32_t tracer;            /* RH_SAS_TRACER_OXYGEN18 | _BROMIDE | _CHLORIDE | _VIRTUAL (settings.enable_oxygen18 / enable_bromide / enable_chloride / enable_virtualtracer) */
    int32_t solver;            /* RH_SAS_SOLVER_DETERMINISTIC | _EULER | _RK4 (settings.sas_solver) */
} rh_sas_config;

typedef struct rh_sas_ctx rh_sas_ctx;

void rh_sas_default_config(rh_sas_config *cfg);
int rh_sas_create(const rh_sas_config *cfg, rh_sas_ctx **out);
void rh_sas_destroy(rh_sas_ctx *ctx);
const char *rh_sas_last_error(const rh_sas_ctx *ctx); /* ctx may be NULL for rh_sas_create failures */
int rh_sas_set_stream(rh_sas_ctx *ctx, void *hip_stream);
int rh_sas_sync(rh_sas_ctx *ctx);

/* ---- array registry ------------------------------------------------------------------------
 * Names follow the reference variables (roger/variables.py): state `sa_rz msa_rz sa_ss msa_ss`;
 * daily inputs (forcing_days, n) `inf_mat_rz inf_pf_rz inf_pf_ss evap_soil transp q_rz q_ss cpr_rz
 * C_in` (what `set_forcing` assigns, benchmarks/SVATOXYGEN18_benchmark.py:384-437); parameters
 * `maskCatch` (n, int32), `sas_params_<flux>` (n, 8); per-cell results `C_<flux> C_iso_<flux>`,
 * `C_inf_* C_iso_inf_*`, `C_rz C_ss C_s C_iso_rz C_iso_ss C_iso_s`, the age statistics
 * `tt{10,25,50,75,90,avg}_{transp,q_ss}`, `rt{..}_{rz,ss,s}`; diagnostics `tt_<flux> mtt_<flux>`
 * (n, ages), `TT_<flux>` (n, ages + 1), `sa_s msa_s` (n, ages). */
int rh_sas_num_arrays(void);
const char *rh_sas_array_name(int array);
int rh_sas_array_index(const char *name);                  /* -1 if unknown */
int64_t rh_sas_array_elems(const rh_sas_ctx *ctx, int array); /* elements held by this context (0: not allocated) */
int rh_sas_array_is_int(int array);
int rh_sas_upload(rh_sas_ctx *ctx, int array, const void *host, size_t bytes);
int rh_sas_download(rh_sas_ctx *ctx, int array, void *host, size_t bytes); /* synchronises */
/* The same for the rows [first_cell, first_cell + n_cells) of a per-cell array (not for the DAILY inputs):
 * lets a driver stream a state that is larger than it wants to stage on the host (32 GB at 10^6 columns
 * x 1000 ages). */
int rh_sas_upload_cells(rh_sas_ctx *ctx, int array, int64_t first_cell, int64_t n_cells, const void *host, size_t bytes);
int rh_sas_download_cells(rh_sas_ctx *ctx, int array, int64_t first_cell, int64_t n_cells, void *host, size_t bytes);
void *rh_sas_array_device_ptr(rh_sas_ctx *ctx, int array);
/* Row `day_row` of a DAILY input from n_cells float64 already on this device (device-to-device, asynchronous on the
 * context's stream; the caller orders it after the producer): the coupling point for the daily flux sums that the
 * SVAT context accumulates (rh_diag_device_ptr in roger_hip.h). */
int rh_sas_set_daily_from_device(rh_sas_ctx *ctx, int array, int64_t day_row, const double *dev_src);

/* ---- the step --------------------------------------------------------------------------------
 * Stages of one day, in the order of svat_transport_model_deterministic; `stages` is a bit mask so
 * that a driver (and the parity tests) can run the reference's kernels one at a time with the
 * state left in HBM in between.  rh_sas_step == rh_sas_stages(ctx, day, RH_SAS_ALL). */
enum {
    RH_SAS_INF_RZ = 1 << 0,   /* infiltration into the root zone (matrix, then preferential flow) */
    RH_SAS_EVAP = 1 << 1,     /* soil evaporation */
    RH_SAS_TRANSP = 1 << 2,   /* transpiration */
    RH_SAS_Q_RZ = 1 << 3,     /* root zone percolation -> subsoil */
    RH_SAS_INF_SS = 1 << 4,   /* preferential-flow infiltration into the subsoil */
    RH_SAS_Q_SS = 1 << 5,     /* subsoil percolation */
    RH_SAS_CPR = 1 << 6,      /* capillary rise subsoil -> root zone */
    RH_SAS_STORAGE = 1 << 7,  /* root zone / subsoil / soil concentrations (+ age statistics if enabled) */
    RH_SAS_AGEING = 1 << 8,   /* shift by one age class, merge the oldest */
    RH_SAS_ALL = (1 << 9) - 1,
    /* not part of a day: soil.rescale_SA after the transport warm-up (rescale_sa_msa_iso_soil_kernel,
     * core/soil.py:1250-1395): sa_rz, sa_ss scaled to S_rz_init, S_ss_init; storage concentrations recomputed */
    RH_SAS_RESCALE = 1 << 9
};
/* `day` selects the row of the daily inputs: row (day mod forcing_days). */
int rh_sas_stages(rh_sas_ctx *ctx, int64_t day, int stages);
int rh_sas_step(rh_sas_ctx *ctx, int64_t day);
/* `ndays` whole steps for days day0, day0 + 1, ... enqueued back to back. */
int rh_sas_run_days(rh_sas_ctx *ctx, int64_t day0, int64_t ndays);

/* HIP-event timing of the step kernel (same protocol as rh_enable_timing / rh_timing_summary). */
int rh_sas_enable_timing(rh_sas_ctx *ctx, int on);
int rh_sas_timing_summary(rh_sas_ctx *ctx, double *total_ms, int64_t *launches);

/* Diagnostic: the kernel's x**k routine (power-law SAS function, core/sas.py:228-231) on n host values,
 * 0 < x <= 1; lets the tests bound its error against the host's pow directly.  Uses the current device. */
int rh_sas_selftest_pow(const double *x, const double *k, double *out, int64_t n);
/* Diagnostic: the kernel's division by a loop-invariant divisor (hoisted refined reciprocal + one residual step),
 * out = a / d, to be compared bit for bit with the host's IEEE division. */
int rh_sas_selftest_div(const double *a, const double *d, double *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* ROGER_HIP_SAS_H */
