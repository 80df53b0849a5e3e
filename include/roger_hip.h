/*
 * include/roger_hip.h -- C ABI of the MI355X-native ("hip") backend for RoGeR's per-cell
 * SVAT time step.
 *
 * The reference (Hydrology-IFH/roger) has no native code; its "FFI" for this path is the
 * Python operator surface: `@roger_routine` functions that mutate `state.variables` and
 * `@roger_kernel` functions that return a `KernelOutput` (roger/routines.py:118,239).  Each
 * entry point below replaces the body of one such routine; the citation names the reference
 * routine it stands in for.  All pointers are plain host or device pointers, all sizes are
 * explicit, no C++ or torch types cross the boundary.  Every call returns 0 on success or a
 * negative rh_status; `rh_last_error` gives the text.  A context is bound to one HIP device
 * and one stream and is not thread-safe (the reference driver is single-threaded,
 * roger/roger.py:523-580).  Calls are asynchronous on the context's stream; `rh_sync` and the
 * download calls fence (the counterpart of `flush()`, roger/core/operators.py:137-145).
 */
#ifndef ROGER_HIP_H
#define ROGER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RH_ABI_VERSION 4   /* 2: rh_config gained enable_routing_1D + dy, rh_sas_config.solver (round 2); 3: rh_comm_info (round 3); 4: rh_svat_step_scalars (round 4) */
#define RH_SLOTS_PER_DAY 144 /* roger/variables.py:109 "timesteps_day": 6 * 24 */

typedef enum rh_status {
    RH_OK = 0,
    RH_ERR_ARG = -1,     /* bad argument (unknown field, size mismatch, null pointer) */
    RH_ERR_HIP = -2,     /* a HIP runtime call failed */
    RH_ERR_STATE = -3,   /* call order violated (e.g. step before forcing was set) */
    RH_ERR_NODEVICE = -4 /* no usable gfx950 device */
} rh_status;

/* Plane ids: one per (variable, time level); see rh_fields.def. */
typedef enum rh_plane {
#define RH_FIELD_1(name) RH_P_##name,
#define RH_FIELD_2(name) RH_P_##name, RH_P_##name##_m1,
#define RH_FIELD(name, type, levels) RH_FIELD_##levels(name)
#include "rh_fields.def"
#undef RH_FIELD
#undef RH_FIELD_1
#undef RH_FIELD_2
    RH_NPLANES
} rh_plane;

/* Model settings used on the path (roger/settings.py:52-122), with the reference defaults
 * filled in by rh_default_config. */
typedef struct rh_config {
    int64_t nx, ny;   /* local interior grid of this rank (reference arrays carry +4 ghosts per
                         axis, roger/variables.py:170-173; the arena stores the interior only) */
    int32_t device;   /* HIP device ordinal */
    int32_t enable_lateral_flow; /* settings.enable_lateral_flow: the oneD model (roger/models/oneD), lateral
                                    subsurface runoff (core/subsurface_runoff.py:1456-1471) and its num-error /
                                    after_timestep variants */
    double pi, r_mp, l_sc, sf, ta_fm, rmax, transp_water_stress, atol, rtol;
    double clay_min, clay_max, theta_rew_min, theta_rew_max, rew_min, rew_max;
    double z_evap_max, zroot_to_zsoil_max, a_bc, b_bc;
    int64_t end_event, hpi;
    double dx;        /* settings.dx, grid spacing in m (enters the lateral flow rates) */
    int32_t placement_probes; /* where the arena lands in HBM decides which bandwidth level the fused kernel runs at (arenas of
                                 one process at 0.30 / 0.32 / 0.33 / 0.35 ms per step at 10^6 columns, tools/arena_levels.py,
                                 DESIGN.md section 5): rh_create allocates up to this many candidate arenas, times a streaming
                                 kernel on each and keeps the fastest (default 8; 1 = take the first; the candidates held at
                                 once never exceed a quarter of the free memory; grids below 65 536 columns are not probed) */
    int32_t enable_routing_1D; /* settings.enable_routing_1D (with enable_lateral_flow): surface and subsurface runoff move to
                                  the D8 neighbour (rh_surface_routing / rh_subsurface_routing); the columns are then coupled
                                  twice per step and the fused step (rh_svat_step, rh_run_steps) is not available */
    double dy;        /* settings.dy, grid spacing in m (with dx: the routed surface runoff, surface_runoff.py:48-58) */
} rh_config;

/* Per-domain scalars of the reference (roger/variables.py:189-330).  event_id/year/month/doy
 * are [taum1, tau] as in the reference. */
typedef struct rh_scalars {
    int64_t itt, time, dt_secs, itt_day, itt_forc, time_event0, event_id_counter;
    int64_t event_id[2], year[2], month[2], doy[2];
    double dt;
    int64_t sanity_ok; /* numerics.sanity_check of the last step (roger/core/numerics.py:728) */
} rh_scalars;

typedef struct rh_ctx rh_ctx;

/* ---- life cycle -------------------------------------------------------------------------- */
void rh_default_config(rh_config *cfg);
/* What the placement probing of the last rh_create on this context found: the streaming kernel's time on every
 * candidate (ms, the chosen one first); returns the number of candidates probed (0: probing was off). */
int rh_placement_report(const rh_ctx *ctx, double *ms, int cap);
/* Replaces RogerState.initialize_variables (roger/state.py:369-374): allocates one device arena
 * with RH_NPLANES planes of nx*ny cells, zero-filled, then sets the non-zero `initial=` values of
 * the registry (maskCatch=1, ta=15, z_gw=1000, c_int=1, c_root=1; roger/variables.py). */
int rh_create(const rh_config *cfg, rh_ctx **out);
void rh_destroy(rh_ctx *ctx);
const char *rh_last_error(const rh_ctx *ctx); /* ctx may be NULL for rh_create failures */
int rh_abi_version(void);

/* Use an externally created HIP stream (e.g. torch's current stream) for all launches. */
int rh_set_stream(rh_ctx *ctx, void *hip_stream);
int rh_sync(rh_ctx *ctx);

/* ---- field registry ---------------------------------------------------------------------- */
int rh_num_planes(void);
const char *rh_plane_name(int plane); /* "<variable>" for tau, "<variable>_m1" for taum1 */
int rh_plane_is_int(int plane);       /* 1: int32 plane, 0: float64 plane */
int rh_plane_index(const char *name); /* -1 if unknown */
int64_t rh_num_cells(const rh_ctx *ctx);

/* Host <-> device copies of one plane (n_cells elements of the plane's type, C order over
 * (x, y) interior).  Counterpart of assigning / reading `vs.<name>` in a setup script. */
int rh_upload(rh_ctx *ctx, int plane, const void *host, size_t bytes);
int rh_download(rh_ctx *ctx, int plane, void *host, size_t bytes);
/* Device address of cell 0 of a plane.  The arena is tiled (64 cells per tile, a 512-byte slot per plane and tile,
 * roger_amd/csrc/rh_col.h): cell i of the plane is at
 *     ptr + (i / 64) * (rh_planes_held(ctx) * 512) + (i % 64) * sizeof(element).
 * rh_planes_held: the planes the context's arena has slots for -- all of rh_num_planes() for a routing context, otherwise all but the
 * routing's planes (flow_dir_topo ... q_sub_in_ss, the last ones of rh_fields.def), which rh_upload / rh_download then refuse. */
void *rh_plane_device_ptr(rh_ctx *ctx, int plane);
int rh_planes_held(const rh_ctx *ctx);

/* How the last fused step ran (measurement, tests): RH_STEP_MODE_LAZY = it deferred the tau -> taum1 copies of after_timestep
 * (models/svat/svat.py:187-384; materialised on demand), RH_STEP_MODE_TAIL = its last wavefront formed the control part of the next
 * step (adaptive_time_stepping.py:22-381 for the step to come), so that no control kernel runs in between. */
#define RH_STEP_MODE_LAZY 1
#define RH_STEP_MODE_TAIL 2
#define RH_STEP_MODE_SPARSE 4   /* it did not store the planes the step only produces (never the last step of a call) */
int rh_step_mode(const rh_ctx *ctx);

int rh_set_scalars(rh_ctx *ctx, const rh_scalars *s);
int rh_get_scalars(rh_ctx *ctx, rh_scalars *s); /* synchronises */

/* Look-up tables (roger/lookuptables.py; vs.lut_ilu (25,13), vs.lut_gc (25,13), vs.lut_gcm
 * (25,2), vs.lut_rdlu (25,7)), row-major float64. */
int rh_set_luts(rh_ctx *ctx, const double *ilu, const double *gc, const double *gcm, const double *rdlu);

/* vs.lut_mlms (n_slope, 9): horizontal macropore flow velocities by slope (roger/lookuptables.py ARR_MLMS),
 * needed by rh_params_lateral (oneD model).  Row-major float64; nrows <= 10000. */
int rh_set_lut_mlms(rh_ctx *ctx, const double *mlms, int64_t nrows);

/* Forcing of the current day: what `set_forcing` assigns to vs.prec_day / vs.ta_day /
 * vs.pet_day (benchmarks/SVAT_benchmark.py:151-171).  per_cell == 0: three vectors of 144
 * values shared by all cells (the broadcast the benchmark performs); per_cell != 0: three
 * (n_cells, 144) arrays. Host pointers; copied asynchronously. */
int rh_set_forcing_day(rh_ctx *ctx, const double *prec_day, const double *ta_day, const double *pet_day,
                       int per_cell);

/* The whole 10-minute forcing series plus calendar, resident on the device: what
 * `set_forcing_setup` assigns to vs.PREC / vs.TA / vs.PET / vs.YEAR / vs.MONTH / vs.DOY
 * (benchmarks/SVAT_benchmark.py:136-149).  With it the per-step user hooks `set_forcing` and
 * `set_parameters` (SVAT_benchmark.py:105-110,151-171) run on the device (rh_hooks_phase), and
 * rh_run_steps advances the model without any host synchronisation. */
int rh_set_forcing_series(rh_ctx *ctx, const double *prec, const double *ta, const double *pet, const int64_t *year,
                          const int64_t *month, const int64_t *doy, int64_t nitt_forc);

/* Per-cell weights on top of the resident series, as the distributed catchment setups apply them in `set_forcing`
 * (examples/catchment_scale/eberbaechle/svat_distributed/svat.py:169-186, 276-296):
 *   prec_day = PREC * prec_weight,  ta_day = TA + ta_offset,  pet_day = PET * pet_weight     (n_cells float64 each)
 * Needs rh_set_forcing_series first.  The day's series stays one 144-vector (staged in LDS); every column forms its
 * own values from its weights on the fly, no (n_cells, 144) array exists.  The step takes the per-cell-forcing path
 * (predicate kernels, three-phase protocol).
 * Pass three NULLs to return to the shared series. */
int rh_set_forcing_weights(rh_ctx *ctx, const double *prec_weight, const double *ta_offset, const double *pet_weight);

/* ---- setup-time kernels ------------------------------------------------------------------ */
int rh_topo(rh_ctx *ctx);               /* surface.calc_topo_kernel, roger/core/surface.py:40-71 */
int rh_params_surface(rh_ctx *ctx);     /* calc_parameters_surface_kernel, surface.py:74-343 */
int rh_params_soil(rh_ctx *ctx);        /* soil.calculate_parameters, roger/core/soil.py:143-557,727-739 */
int rh_params_lateral(rh_ctx *ctx);     /* calc_parameters_lateral_flow_kernel, roger/core/soil.py:560-641 (oneD) */
int rh_initial_conditions(rh_ctx *ctx); /* surface/soil.calculate_initial_conditions, surface.py:398-427, soil.py:742-1010 */

/* ---- one entry point per routine of RogerSetup.step (roger/roger.py:396-457,485) ---------- */
int rh_adaptive_dt(rh_ctx *ctx);        /* adaptive_time_stepping, core/adaptive_time_stepping.py:22-437 */
/* ... in three parts for several ranks: rh_step_phase1 -> exchange word 0 -> rh_step_phase2 -> exchange word 1 -> this call (dt, event
 * bookkeeping, pet / ta selection).  Replaces adaptive_time_stepping_dist_safe.py:6-380 (gather to rank 0, decide, scatter). */
int rh_adaptive_dt_finish(rh_ctx *ctx);
int rh_interception(rh_ctx *ctx);       /* calculate_interception, core/interception.py:347-356 */
int rh_evapotranspiration(rh_ctx *ctx); /* calculate_evapotranspiration, core/evapotranspiration.py:603-616 */
int rh_snow(rh_ctx *ctx);               /* calculate_snow, core/snow.py:294-304 */
int rh_infiltration(rh_ctx *ctx);       /* calculate_infiltration, core/infiltration.py:2148-2193 */
int rh_subsurface_runoff(rh_ctx *ctx);  /* calculate_subsurface_runoff (SVAT branch), core/subsurface_runoff.py:1473-1479 */
/* settings.enable_routing_1D (rh_config.enable_routing_1D): the D8 routing of surface and subsurface runoff.
 *   rh_surface_routing    = calculate_surface_runoff, core/surface_runoff.py:240-250 -> calc_surface_runoff_routing_1D :14-227
 *   rh_subsurface_routing = the routing part of calculate_subsurface_runoff, core/subsurface_runoff.py:1468-1469 ->
 *                           calc_subsurface_runoff_routing_1D :1158-1437 (call it after rh_subsurface_runoff)
 * Each is rh_route_out (per column: the outflow), the exchange of the edge columns with the x-neighbours over the context's RCCL
 * communicator when it has more than one rank (rh_comm_init / rh_set_comm; the reference itself never exchanges them: with MPI its
 * routed water is lost at the process boundaries), and rh_route_in (the gather from the eight neighbours + per column: the inflow).
 * The pieces are exported for drivers that exchange the halos themselves: rh_route_get_edges / rh_route_get_static_edges return the
 * rank's own edge columns x = 0 ("lo") and x = nx - 1 ("hi") (ny values each: q_out of the step; flow direction and mask, once),
 * rh_route_set_halo hands in the neighbour's column for side 0 (x = -1) or 1 (x = nx); q may be NULL when only the static part is
 * set, flow_dir / mask may be NULL afterwards.  `which`: 0 surface, 1 subsurface. */
int rh_surface_routing(rh_ctx *ctx);
int rh_subsurface_routing(rh_ctx *ctx);
/* One whole step with the routing in the order of RogerSetup.step (roger/roger.py:396-457), routine by routine on the device:
 * adaptive time stepping (over several ranks: the two predicate words all-reduced over the context's communicator), [monthly surface
 * parameters], interception ... infiltration, rh_surface_routing, the lateral subsurface runoff, rh_subsurface_routing, capillary rise,
 * storages, numerics, itt / time, after_timestep.  monthly: 1 / 0 = the caller's set_parameters decision, -1 = the device's (rh_run_steps
 * takes this entry for a routing context). */
int rh_step_routed(rh_ctx *ctx, int monthly);
int rh_route_out(rh_ctx *ctx, int which);
int rh_route_in(rh_ctx *ctx, int which);
int rh_route_get_edges(rh_ctx *ctx, int which, double *q_lo, double *q_hi);
int rh_route_get_static_edges(rh_ctx *ctx, int32_t *flow_dir_lo, int32_t *flow_dir_hi, int32_t *mask_lo, int32_t *mask_hi);
int rh_route_set_halo(rh_ctx *ctx, int side, const double *q, const int32_t *flow_dir, const int32_t *mask);
int rh_capillary_rise(rh_ctx *ctx);     /* calculate_capillary_rise, core/capillary_rise.py:346-358 */
int rh_storage(rh_ctx *ctx);            /* calculate_surface/root_zone/subsoil/soil + numerics.calc_storage */
int rh_num_error(rh_ctx *ctx);          /* numerics.sanity_check + calculate_num_error, core/numerics.py:716-1011 */
int rh_after_timestep(rh_ctx *ctx);     /* after_timestep_kernel, roger/models/svat/svat.py:187-384 */
/* interception ... calculate_num_error fused into one kernel plus `itt += 1; time += dt_secs`
 * (roger/roger.py:410-457): the part of step() between the user hooks set_parameters and
 * after_timestep, for drivers that keep those hooks on the host.  Needs rh_adaptive_dt first. */
int rh_step_core(rh_ctx *ctx);

/* ---- the fused step ------------------------------------------------------------------------
 * One whole time step in the order of RogerSetup.step, without the user hooks: adaptive dt ->
 * [monthly surface parameters] -> interception -> evapotranspiration -> snow -> infiltration ->
 * subsurface runoff -> capillary rise -> storages -> itt/time increment -> sanity/num error ->
 * after_timestep.  The global predicates (`.any()/.all()` in adaptive_time_stepping.py:38-81,
 * 192-201 and infiltration.py:2155-2167) are reduced on the device; the scalars never leave it.
 *
 * The step is split in three phases so that a multi-GPU host can all-reduce the predicate words
 * between them (bitwise OR over ranks; see rh_predicate_words):
 *   phase 1: reduce the start-of-step predicates          -> words[0]
 *   phase 2: select prec/ta, reduce the event predicates  -> words[1]
 *   phase 3: scalar bookkeeping + the fused per-cell kernel
 * rh_svat_step runs the three back to back (single GPU). `monthly` != 0 runs the surface
 * parameter kernel first, as `set_parameters` does on a month change (svat.py:115-120). */
int rh_step_phase1(rh_ctx *ctx);
int rh_step_phase2(rh_ctx *ctx);
int rh_step_phase3(rh_ctx *ctx, int monthly); /* monthly < 0: use the device-side month-change flag */
int rh_svat_step(rh_ctx *ctx, int monthly);
/* rh_svat_step followed by the read-back of the scalars the driver's loop looks at before the next step (`while vs.time - start_time <
 * runlen`, roger/roger.py:548-556; a script's set_forcing tests `vs.time % 86400`, benchmarks/SVAT_benchmark.py:155-156): ONE call per
 * step for a driver that keeps its hooks on the host.  The scalars (and the sanity / forcing-series flags) travel through a block of
 * pinned host memory that a one-thread kernel behind the step writes directly -- no staged copies; the call returns when that block
 * has arrived (it synchronises like rh_get_scalars, which uses the same block). */
int rh_svat_step_scalars(rh_ctx *ctx, int monthly, rh_scalars *s);
/* The parameter planes as the lazy variants of the fused kernel will read them (round 4): per wavefront of 64 columns the kernel keeps a
 * word saying which parameter planes (read by the step, assigned by none of its stages, or by the monthly surface parameters only) hold
 * ONE value over the wave -- those are read as one element per wave instead of 512 bytes -- and whether the 15 parameters that
 * calc_parameters_soil derives from the primaries (roger/core/soil.py:143-557: theta_sat, theta_fc, m_bc, n_salv, wfs, rew, z_evap,
 * tew, S_{ac,ufc,pwp}_{rz,ss}, S_pwp_s) still hold exactly those values, bit for bit -- then the stages evaluate the same expressions
 * instead of loading the planes.  The words are formed from the planes on the device whenever somebody other than the fused kernel may
 * have changed them; a plane uploaded with other values simply clears the bits of the waves it touches.
 *   derived_fraction        waves whose derived parameters are not loaded
 *   uniform_bytes_per_cell  bytes per column and step of parameter loads that are one element per wave (of what is loaded at all)
 * RH_NO_PARAM_UNIFORM / RH_NO_PARAM_DERIVE (environment, read by rh_create) switch the two off. */
int rh_param_stats(rh_ctx *ctx, double *derived_fraction, double *uniform_bytes_per_cell);
/* The same step with ONE exchange (shared forcing only).  The fused kernel leaves, per wavefront, a summary word of
 * its columns' end-of-step state from which both predicate words of the next step follow (the start-of-step snow
 * predicates directly; the event predicates together with the selected prec / ta, which are uniform when the
 * forcing is shared):
 *   rh_step_summary: [device-side hooks if a forcing series is set,] OR of the summary words -> words[3]
 *                    (after a host-side change of the planes the summary is rebuilt from the arena first)
 *   -- all-reduce words[3] over the ranks (rh_predicates_expand / _compress with word = 3) --
 *   rh_step_finish:  control kernel (dt, selection, event bookkeeping) + the fused per-cell kernel
 * rh_run_steps and rh_svat_step use this path without the exchange.  Returns RH_ERR_STATE with per-cell forcing. */
int rh_step_summary(rh_ctx *ctx);
int rh_step_finish(rh_ctx *ctx, int monthly); /* monthly < 0: use the device-side month-change flag */
/* The same two calls with the exchange format folded in: rh_step_summary_expand also writes the summary word as 64
 * int32 (0 / 1) to dev_dst64 (what rh_predicates_expand(ctx, 3, .) would produce), rh_step_finish_compress takes the
 * all-reduced 64 int32 directly (instead of rh_predicates_compress(ctx, 3, .) + rh_step_finish): two launches less
 * per step on the multi-GPU path. */
int rh_step_summary_expand(rh_ctx *ctx, int32_t *dev_dst64);
int rh_step_finish_compress(rh_ctx *ctx, int monthly, const int32_t *dev_src64);
/* Several meteorological stations (settings.enable_distributed_input: vs.PREC_DIST / TA_DIST / PET_DIST (n_stations, t_forc) and the
 * per-cell vs.station_id, roger/variables.py:882-916, 3522, 4138, 6383-6402; the distributed models' set_forcing picks, per cell, the
 * series of its station before applying the weights, roger/bmimodels/svat_dist/svat_dist.py:274-310).  prec / ta / pet: (n_stations,
 * nitt_forc) row-major; station_index: per cell the ROW of its station (0-based; < 0: no station, the cell sees zeros, like a cell whose
 * vs.station_id matches no entry of vs.station_ids).  The device-side hooks stage every station's day at midnight; each column forms its
 * own day from its station's rows and its weights (rh_set_forcing_weights; neutral if none were set).  Replaces rh_set_forcing_series. */
int rh_set_forcing_stations(rh_ctx *ctx, const double *prec, const double *ta, const double *pet, const int64_t *year, const int64_t *month,
                            const int64_t *doy, int64_t nitt_forc, int n_stations, const int32_t *station_index);
/* Device-side `set_forcing` + `set_parameters` hooks (needs rh_set_forcing_series); runs before
 * phase 1. */
int rh_hooks_phase(rh_ctx *ctx);
/* nsteps whole time steps, hooks included, enqueued back to back on the stream. */
int rh_run_steps(rh_ctx *ctx, int64_t nsteps);
/* Sparse stores inside rh_run_steps / rh_run_steps_dist (measurement, tests).  The reference overwrites its flux and diagnostic arrays
 * every step (`vs.q_ss = update(...)`), and inside ONE call nothing but the next step looks at the planes; so every step of a call
 * that another step follows leaves out the stores of the planes the step only PRODUCES -- planes no step reads before assigning
 * them, derived from rh_physics.h by the flow analysis of tools/liveness.py (72 of the SVAT step's 124 stored planes).  The last step
 * of a call stores everything: after the call every plane holds what n full steps leave.  An accumulator (rh_diag_configure) that was
 * given such a plane gets it stored after all (a third variant of the kernel); off with RH_NO_SPARSE_STORES=1.  Round 4: a sparse step
 * also leaves out conductivity and suction of root zone and subsoil (k_rz, k_ss, h_rz, h_ss, ks_ss), which the next step -- a LAZY step:
 * the planes were last touched by a complete step -- derives from the stored water contents instead of loading them; the first step
 * after the host touched planes still loads them, so these five are NOT "pure outputs" in the sense below.
 *   rh_plane_is_pure_output  1 if the step of the model only produces the plane: model 0 = SVAT, 1 = oneD (fused steps), 2 = the routed
 *                            step (settings.enable_routing_1D: there a pass also keeps what a later pass of the same step loads)
 *   rh_sparse_steps          steps of the most recent rh_run_steps / rh_run_steps_dist call that ran with sparse stores */
int rh_plane_is_pure_output(int model, int plane);
/* "Run until": the reference's driver loop is `while vs.time - start_time < runlen: step()` (roger/roger.py:548-556), and the length
 * of a step is decided on the device.  With a limit set, the control part of a step (on the device) finds the run over once the model
 * time has reached t_end: that launch and every later one do nothing (the accumulators included), so a caller may enqueue more steps
 * than the run has left -- rh_run_steps(ctx, n) then runs min(n, steps until t_end) -- and read the time afterwards.  The step that
 * reaches the limit stores every plane.  t_end < 0 clears the limit.  Observed by the summary path (forcing shared by all columns, no
 * routing); rh_run_steps / rh_run_steps_dist report RH_ERR_STATE otherwise.  Synchronises. */
int rh_set_time_limit(rh_ctx *ctx, int64_t t_end);
int64_t rh_sparse_steps(const rh_ctx *ctx);
/* Device address of the 64-bit predicate words (uint64_t[4]); combine over ranks with OR. */
void *rh_predicate_words(rh_ctx *ctx);
/* RCCL has no bitwise reduction: spread predicate word `word` into 64 int32 0/1 values at the
 * device address `dev_dst64` (then all-reduce them with MAX), and fold 64 such values back.
 * Stream-ordered, no synchronisation. */
int rh_predicates_expand(rh_ctx *ctx, int word, int32_t *dev_dst64);
int rh_predicates_compress(rh_ctx *ctx, int word, const int32_t *dev_src64);

/* ---- multi-GPU stepping without the host in the loop (SURVEY section 8b "rh_set_comm", 8e) -----------------------------------
 * One process per GPU, the grid split along x (roger/distributed.py:121-187); the only exchange of the SVAT / oneD step is the
 * OR of the ranks' summary words -- what the reference does per step by gathering 18 fields to rank 0, deciding dt there and
 * scattering them back (roger/core/adaptive_time_stepping_dist_safe.py:6-26).  rh_run_steps_dist enqueues, per step, on the
 * context's stream: ncclAllReduce(MAX, 64 x int32: the summary word the previous fused kernel's tail spread out) -> control kernel
 * (device-side hooks, dt, selection, event bookkeeping from the reduced word) -> fused kernel.  No host synchronisation and no
 * Python between the steps.  The communicator is RCCL's (librccl is loaded on first use, the library does not link against it):
 *   rh_comm_unique_id   rank 0 creates the 128-byte id (ncclGetUniqueId); the caller hands it to the other ranks
 *   rh_comm_init        every rank: ncclCommInitRank on the context's device (collective); owned by the context
 *   rh_set_comm         or: borrow a communicator the caller owns (an ncclComm_t); NULL detaches
 * With one rank the all-reduce is a copy: rh_run_steps_dist then equals rh_run_steps bit for bit (tests/test_hip_comm.py). */
int rh_comm_unique_id(void *id128);
int rh_comm_init(rh_ctx *ctx, const void *id128, int nranks, int rank);
int rh_set_comm(rh_ctx *ctx, void *nccl_comm);
/* ncclCommCount / ncclCommUserRank of the communicator the context holds, asked of RCCL itself (1 / 0 without a communicator). */
int rh_comm_info(rh_ctx *ctx, int *nranks, int *rank);
int rh_run_steps_dist(rh_ctx *ctx, int64_t nsteps);

/* ---- device-side output accumulators --------------------------------
 * The reference's "rate" diagnostic adds every registered variable after each step (`rate += var[..., tau]`,
 * roger/diagnostics/rate.py:66-84) and its "collect" diagnostic keeps the current value
 * (roger/diagnostics/collect.py); the catchment setups and the transport model's input consume them as DAILY sums
 * of the fluxes and end-of-day storages (benchmarks/SVATOXYGEN18_benchmark.py:342-377).  With adaptive time steps
 * decided on the device the host does not know where a day ends without synchronising, so the accumulators are
 * indexed by day on the device: slot = (day of the step's start time) mod n_slots; the first step of a day
 * overwrites its slot.  After rh_diag_configure every fused step (rh_svat_step, rh_run_steps, rh_step_phase3,
 * rh_step_finish) is followed by one small kernel that updates the slots.
 *   rate_planes[n_rate]:       plane ids summed per day          (24 B of traffic per plane, column and step)
 *   collect_planes[n_collect]: plane ids whose end-of-day value is kept
 * n_rate + n_collect <= 32; n_slots >= 1 days are resident: (n_slots, n_rate + n_collect, n_cells) float64. */
int rh_diag_configure(rh_ctx *ctx, const int *rate_planes, int n_rate, const int *collect_planes, int n_collect, int n_slots);
/* One (variable, day slot) array: n_cells float64.  j counts the rate planes first, then the collect planes. */
int rh_diag_download(rh_ctx *ctx, int j, int slot, double *host, size_t bytes); /* synchronises */
void *rh_diag_device_ptr(rh_ctx *ctx, int j, int slot);
/* Number of steps accumulated in a day slot: the divisor of the "average" diagnostic (roger/diagnostics/average.py:
 * `avg += var; n += 1`, output avg / n) for a variable registered as a rate plane.  Synchronises. */
/* Restart (roger/restart.py:140-174 writes the diagnostics' accumulators next to the core state): put an accumulator slot back --
 * the values of variable j, and the slot's bookkeeping (steps accumulated, start time of the interval's first step, end time of its
 * last step: what rh_diag_steps / rh_diag_slot_times report). */
int rh_diag_upload(rh_ctx *ctx, int j, int slot, const double *host, size_t bytes);
int rh_diag_set_slot_state(rh_ctx *ctx, int slot, int64_t steps, int64_t t_start, int64_t t_end);
int rh_diag_steps(rh_ctx *ctx, int slot, int64_t *steps);
/* Output intervals other than a day: 3600 or 600 seconds (the step classes; a step never straddles a boundary it does not
 * start on).  Slots are then indexed by the interval of the step's start; an interval that a longer step covers is never
 * started and its slot stays untouched.  rh_diag_slot_times tells which interval a slot holds: the start time of its first
 * step (-1: never touched) and the end time of its last one -- the reference writes a record whenever `time % frequency
 * == 0` (roger/diagnostics/api.py:47-70), i.e. at that end time.  Call rh_diag_set_interval before the first step. */
int rh_diag_set_interval(rh_ctx *ctx, int64_t seconds);
int rh_diag_slot_times(rh_ctx *ctx, int slot, int64_t *t_start, int64_t *t_end);

/* HIP-event timing of the fused per-cell kernel.  rh_enable_timing(ctx, 1) starts a new
 * measurement: every following step records an event pair around the kernel on the context's
 * stream (no synchronisation).  rh_timing_summary synchronises and returns the summed kernel
 * time and the number of timed launches since then. */
int rh_enable_timing(rh_ctx *ctx, int on);
int rh_timing_summary(rh_ctx *ctx, double *total_ms, int64_t *launches);
/* Per timed launch: the kernel's duration and the dt_secs of its step (600 / 3600 / 86400: the time-step classes of
 * adaptive_time_stepping, core/adaptive_time_stepping.py:22-381), the first min(cap, *launches) of them. */
int rh_timing_detail(rh_ctx *ctx, double *kernel_ms, int32_t *dt_secs, int64_t cap, int64_t *launches);

/* Profiling aid: one kernel that copies `nplanes` float64 planes src_plane0.. -> dst_plane0.. with
 * the access shape of the fused kernel (8 bytes per lane and plane); moves a known
 * 2 * nplanes * n_cells * 8 bytes, used to calibrate the HBM counters.  Overwrites the planes. */
int rh_calibrate_copy(rh_ctx *ctx, int src_plane0, int dst_plane0, int nplanes);
/* x[i] ** y[i] by the power function the kernels use (roger_amd/csrc/rh_pow.h), evaluated on the device: tests compare it bit for bit
 * with the host's compilation of the same header, whose accuracy is established against the C library's pow. */
int rh_selftest_pow(const double *x, const double *y, double *out, int64_t n);
/* The sum over a 144-slot series masked to the hourly window [itd, itd + 6) (adaptive_time_stepping.py:400-420) as the kernels form it, for
 * n window starts over one series: out[2 j] by the kernels' function (a rotation of numpy's pairwise tree where the window lies inside one
 * 72-block), out[2 j + 1] by its general path.  Tests compare both with numpy's own sum over the masked vector, bit for bit. */
int rh_selftest_window_sum(const double *v144, const int64_t *itd, int64_t n, double *out2n);
/* Measurement aid: two contexts of the same shape and state exchange their arenas (does a speed level belong to the allocation?). */
int rh_debug_swap_arenas(rh_ctx *a, rh_ctx *b);

#ifdef __cplusplus
}
#endif
#endif /* ROGER_HIP_H */
