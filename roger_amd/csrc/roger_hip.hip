// roger_hip.hip -- kernels and C ABI of the MI355X-native SVAT backend (see include/roger_hip.h).
//
// Kernel inventory (one thread = one soil column, 256-thread workgroups = 4 wavefronts):
//   k_pred1        start-of-step predicates over swe/swe_top, grid-stride, one word per workgroup
//   k_agg          1 workgroup: [user hooks] + OR of the workgroup words + forcing predicates,
//                  step-class flags, aggregates of the shared forcing series (numpy summation order)
//   k_select       prec/ta selection per column, event + infiltration predicates, one word per workgroup
//   k_scalars      1 workgroup: OR of the words, time-step bookkeeping (dt, event ids, itt, time),
//                  StepCtx for the step
//   k_step<M>      THE hot kernel: whole SVAT step per column, state read once / written once
//   k_reduce       only for multi-GPU runs: materialises a predicate word for the all-reduce
// plus one kernel per routine for the per-routine entry points and the setup-time kernels.
// All per-column kernels are HBM-bound streaming kernels (no data reuse, no LDS tiling, no
// MFMA); k_step moves ~2 KB per column (2.8 KB by the reference's variable read/write sets).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the entry points are resolved with dlsym on first use (rccl_api)

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#include "rh_physics.h"
#include "rh_sets.inc"
#include "roger_hip.h"

#define RH_BLOCK 256
// unused slots appended to every tile of the arena (the tile stride in units of 512 bytes decides how the tiles spread over the HBM
// channels; experiments)
#ifndef RH_STRIDE_PAD
#define RH_STRIDE_PAD 0
#endif
#define RH_PRED_BLOCKS 1024  // grid of the grid-stride predicate kernels
#define RH_DONE_GROUPS 256   // completion counters of the fused kernel (two levels: workgroup -> group -> grid), a cache line each
#define RH_DONE_STRIDE 32   // (unsigned ints: 128 bytes)
#define RH_DEVERR_FORCING 1u // a step began a day beyond the end of the resident forcing series
// flags of k_step / sources of k_ctrl
#define RH_TAIL_USE_NEXT 1   // this step runs on S_next / X_next (the previous kernel's tail formed them); its tail commits them
#define RH_TAIL_CTRL 2       // the tail forms the next step's S_next / X_next
#define RH_TAIL_HOOKS 4      // ... including the device-side set_forcing / set_parameters hooks
#define RH_TAIL_PRE 16      // (with RH_TAIL_CTRL) the launch has one workgroup more than the columns need: its first wavefront forms the half of the
                             // next step's control part that does not depend on the columns WHILE they are stepped (pre_tail); the tail does the rest
#define RH_TAIL_SKIP 8       // nobody reads this step's summary word (per-cell forcing behind k_cell_front, which looks at the planes): no summary
                             // bits posted, no completion counting, no tail -- three round trips less at the end of a launch-bound step
#define RH_SRC_WORD3 0       // the summary word sits in words[3] (a fused kernel ran last)
#ifndef RH_WSTRIDE
#define RH_WSTRIDE 16       // words between two slots of the device-wide OR words (sumw, frontw, dayw): 128 bytes -- a slot per cache line
#endif
#define RH_SRC_SUMW 1        // ... in sumw[] (k_summary rebuilt it from the arena)
#ifndef RH_STEP_PREFETCH
// 0: a stage's planes are requested right in front of it.  1: the NEXT stage's planes are requested before the current stage computes
// -- in the source; the compiler sinks the loads back to their first use (no difference measured).  2 (default since round 3): the same
// with a scheduling barrier behind the requests (__builtin_amdgcn_sched_barrier), so that a stage's arithmetic runs under the next
// stage's loads: 197 VGPRs as before, nothing spilled, 2 - 3 % per step (10^6 columns 0.2294 -> 0.2228 ms, 10^7 2.047 -> 2.000 ms, oneD
// 0.2413 -> 0.2354 ms; library variants alternating inside one call, tools/ab_variants.sh).
#define RH_STEP_PREFETCH 2
#endif
#ifndef RH_STEP_WAVES
#define RH_STEP_WAVES 2     // waves per SIMD the fused kernel is compiled for (register budget 512 / waves)
#endif

// ---------------------------------------------------------------------------------------------
// device-resident control block
// ---------------------------------------------------------------------------------------------
struct DevState {
    Consts K;
    rh_scalars S;
    StepCtx X;
    unsigned long long words[4];  // predicate words 0,1; word 2 = "sanity violated"; word 3 scratch
    // per-workgroup partial predicate words of k_pred1 / k_select: plain stores, OR-reduced by the
    // single-workgroup kernel that follows (a single word hammered by atomics from every wave
    // costs ~100 us per pass: one address sustains ~90 atomics/us)
    unsigned long long bflags[2][RH_PRED_BLOCKS];
    unsigned long long day_bflags[RH_PRED_BLOCKS];   // k_pred1, weighted station forcing: the forcing bits of the day per workgroup
    int pred_blocks;               // workgroups launched for k_pred1 / k_select
    // summary path: the QB_* bits of every column at the end of a step, OR-ed by the fused kernel's wavefronts into 64 words
    // (device-scope atomics, word = workgroup mod 64: ~250 atomics per address and step at 10^6 columns); the last wavefront
    // to finish folds them into words[3] and runs the control part of the NEXT step on S_next / X_next (step_tail)
    unsigned long long sumw[64 * RH_WSTRIDE];   // 64 slots, one per cache line (RH_WSTRIDE)
    unsigned int done_grp[RH_DONE_GROUPS * RH_DONE_STRIDE];   // workgroups finished per completion group (workgroup b belongs to group b mod n_groups)
    unsigned int done_top;                   // groups finished
    unsigned long long sanity_last;          // words[2] of the last fused step (the tail clears words[2] for the next one)
    // what the control part keeps of the DAY's shared series between two midnights (ctrl_wave): the OR of the slots' forcing bits and the
    // three daily aggregates -- a step inside the day then needs the six slots of its hourly window only.  Everything that writes forc
    // clears day_cache_ok.
    unsigned long long pre_words[64];        // pre_tail -> tail of one fused launch (TailPre, a word per lane)
    int day_cache_ok, day_cache_pad;
    unsigned long long day_fb;
    double day_agg[3];
    unsigned int err_flags;                  // RH_DEVERR_*
    rh_scalars S_next;                       // scalars / step context of the next step, formed by the tail of the last fused kernel;
    StepCtx X_next;                          // committed to S / X by the tail of the kernel that runs that step
    // device-side output accumulators (rh_diag_configure): (diag_slots, diag_rate + diag_collect, n) float64
    double *diag;
    long long *diag_steps;         // per slot: {steps accumulated (the divisor of the "average" diagnostic), start time of the
                                   // interval's first step, end time of its last step}
    long long diag_interval;       // output interval in seconds (86400, 3600 or 600)
    int diag_rate, diag_collect, diag_slots;
    int diag_planes[32];
    // sparse stores with accumulators: the pure-output planes an accumulator was given are stored by the sparse kernel after all
    // (bit p of keep[p / 64]; keep_any = any bit set) -- the other ~70 stay unwritten
    unsigned long long keep[(RH_NPLANES + 63) / 64];
    int keep_any;
    // rh_enable_timing: dt_secs of every step since then (the time-step class of each timed launch)
    int *dt_log;
    int dt_log_cap, dt_log_n;
    double forc[3][RH_SLOTS_PER_DAY];  // shared forcing of the day: prec, ta, pet
    const double *forc_cell[3];        // per-cell forcing, TRANSPOSED on upload to (144, n): slot s of column i at [s * n + i], unit stride over
                                       // the columns (a wave reads 512 contiguous bytes per slot instead of 64 values 1152 bytes apart); or null
    double *agg_cell;                  // per-cell aggregates, 9 planes of n (written by k_cell_agg)
    // per-cell forcing, one launch in front of the fused kernel (k_cell_front): frontw = the waves' column bits of the step (word 0's
    // snow bits, word 1's terms for each candidate selection), dayw = the forcing bits of the DAY over all columns and slots (formed
    // once a day, folded into day_word by the front kernel's last wavefront)
    unsigned long long frontw[64 * RH_WSTRIDE], dayw[64 * RH_WSTRIDE], day_word;
    int per_cell;
    // whole forcing series resident on the device (rh_set_forcing_series): 10-minute PREC/TA/PET
    // and the calendar vectors, as the benchmark's set_forcing_setup holds them in vs.PREC, ...
    const double *series[3];
    const int64_t *calendar[3];
    int64_t nitt_forc;
    long long t_end;                   // rh_set_time_limit: no step begins at or beyond this model time (< 0: no limit)
    int skipped;                       // the last fused launch found its step halted and did nothing (read by k_diag)
    int monthly;                       // set_parameters' month-change test, evaluated on the device
    const double *weights[3];          // per-cell prec_weight, ta_offset, pet_weight (rh_set_forcing_weights) or null
    // several meteorological stations (settings.enable_distributed_input, roger/variables.py:6383-6402): the resident series are
    // (n_stations, nitt_forc) each, a column takes the series of station station_idx[column] (< 0: none, all zeros); the day of
    // every station is staged in forc_multi (3, n_stations, 144) at midnight
    int n_stations;
    const int *station_idx;
    double *forc_multi;

    // Parameter planes of the fused step (RH_PARAM_BITS): one 64-bit word per wavefront's 64 columns.  Bit b: the wave's columns hold
    // ONE value of parameter plane b, so the wave reads one element instead of 512 bytes; bit 63: the planes of RH_DERIVED_FIELDS hold
    // exactly what the stages' rd_* functions compute from the primaries, so they are derived instead of loaded.  Written by
    // k_param_mask whenever somebody other than the fused kernel may have changed planes; all zeros = the plain loads.
    const unsigned long long *pmask;

    const double *mlms;                // lut_mlms rows (oneD model), device copy
    int64_t mlms_rows;
    int max_slope_per;
    Luts L;
};

// What the host reads after a step, in pinned host memory that the device writes directly (hipHostMallocMapped): k_export copies the
// committed scalars and flags and stores `seq` last (system scope), the host waits for its sequence number.  rh_get_scalars used
// four staged copies into pageable memory (>= 40 us); this is one one-thread kernel.
struct HostExport {
    rh_scalars S;
    unsigned long long bad, bad_last;
    unsigned int err, pad;
    unsigned long long seq;
};

struct rh_ctx {
    unsigned long long *pmask_buf = nullptr;   // DevState::pmask
    bool pmask_valid = false;                   // ... describes the planes as they are now
    int pmask_flags = 7;                        // bit 0: uniform loads, bit 1: derived parameters, bit 2: the catchment mask as a constant (RH_NO_PARAM_UNIFORM / RH_NO_PARAM_DERIVE / RH_NO_MASK_CONSTANT clear them)
    HostExport *hexp = nullptr;      // pinned + mapped
    unsigned long long hexp_seq = 0;
    rh_config cfg;
    int64_t n;
    Arena arena;
    DevState *dev;
    hipStream_t stream;
    bool own_stream;
    double *forc_cell_buf[3];
    double *weight_buf[3];
    int *station_buf;
    double *forc_multi_buf;
    double *transpose_buf;   // staging of one (n, 144) per-cell forcing array before its transposition
    double *agg_cell_buf;
    void *series_buf;
    double *mlms_buf;
    bool per_cell;
    bool summary_valid;   // the summary word (words[3]) describes the columns as they are in the arena now
    bool routed_summary = false;     // routing: sumw holds the summary bits of the arena's state (posted by k_routed_a2)
    bool routed_device_ok = true;    // RH_ROUTED_BY_ROUTINE: rh_run_steps takes rh_step_routed per step (A/B, tests)
    bool defer_select_ok = true;     // RH_NO_DEFERRED_SELECT: k_select stores the per-cell prec / ta itself (A/B, tests)
    int64_t cell_agg_split_min = 65536;   // columns from which the per-cell aggregates run as two kernels (RH_CELL_AGG_SPLIT_MIN: tests)
    bool pending_valid;   // S_next / X_next hold the control part of the next step (formed by the last fused kernel's tail)
    bool pre_valid = false;   // ... or, multi-GPU step: pre_words hold its columns-independent half (pre_tail of the last fused launch), for k_ctrl behind the exchange
    int pending_hooks;    // ... formed with / without the device-side hooks
    bool tail_ok;         // RH_NO_TAIL_CTRL unset
    int n_groups;         // fused kernel: completion groups (about 64 workgroups each, at most RH_DONE_GROUPS)
    // lazy tau -> taum1 rotation (k_step<.,.,LAZY>): rot_consistent = the last thing that touched the planes was a complete
    // fused step, i.e. X_m1 == X logically for every rotation pair; m1_stale = the X_m1 PLANES do not hold that yet
    bool rot_consistent, m1_stale, lazy_ok, diag_reads_m1;
    // sparse stores (k_step<.,.,LAZY,SPARSE>): sparse_next = the step being enqueued is followed by another step of the same
    // rh_run_steps call; outputs_stale = the last fused step did not store the pure-output planes (only ever true INSIDE a call,
    // or after a call that failed half-way); diag_reads_sparse = an accumulator was given one of those planes
    bool sparse_ok = true, sparse_next = false, outputs_stale = false, diag_reads_sparse = false, last_sparse = false;
    int64_t t_end = -1;              // rh_set_time_limit (host copy of DevState::t_end)
    int64_t call_sparse_steps = 0;   // steps of the most recent rh_run_steps / rh_run_steps_dist call that ran with sparse stores
    bool agg_daily_stale;   // per-cell daily forcing sums must be re-formed (new weights; first use)
    bool pred_daily_stale;  // the same for the day's forcing bits kept by k_pred1
    bool front_daily_stale = true;   // ... and for the one-launch front (k_cell_front: daily sums + DevState::day_word)
    bool cell_front_ok = true;       // RH_PER_CELL_OLD_FRONT unset: per-cell forcing takes k_cell_front instead of the five predicate-generation launches
    int64_t cell_front_max = 2097152; // ... on grids up to this many columns (RH_CELL_FRONT_MAX).  Measured, round 4 (profiles/r04_per_cell_front.txt), ms
                                     // per step with the predicate kernels / with the front: 80 x 53 columns 0.055 / 0.039 (launch-bound: one
                                     // launch in front of the fused kernel instead of six -- the set_forcing hook rides along), 10^6 columns
                                     // 0.261 / 0.252, 10^7 columns 2.15 / 2.23 (one thread doing a column's aggregates, plane reads and bits in
                                     // sequence is latency-bound; two of the five predicate kernels are grid-stride).  Before the slots of the
                                     // device-wide words and the completion counters had a cache line each, the front took 0.320 ms at 10^6.
    int last_front = 0;              // which of the two formed the day's cached parts last (1 old, 2 new): the other re-forms them when it takes over
    double *diag_buf;
    long long *diag_steps_buf;
    long long diag_interval;
    int diag_n, diag_slots;
    int pred_blocks;
    bool forcing_set;
    bool timing;
    std::vector<hipEvent_t> events;  // pairs (start, stop) around the fused kernel, one per timed step
    size_t ev_used;
    int *dt_log_buf;
    std::vector<double> probe_ms;   // placement probing: streaming-kernel time per candidate arena, the chosen one first
    char *arena_alloc;    // what hipMalloc returned for the arena (arena.base = arena_alloc + arena_offset)
    size_t arena_offset, arena_pad = 0;
    void *stage_buf;      // one contiguous plane (n * 8 bytes): uploads and downloads pass through it
    // multi-GPU: RCCL communicator and the exchange buffers of the summary word (64 int32 sent, 64 received)
    ncclComm_t comm;
    bool own_comm;
    int *exch_buf;
    bool exch_valid;      // exch_buf[0..63] holds the summary word of the columns as they are now (written by the last fused kernel's tail)
    int comm_nranks, comm_rank;
    int planes_held;      // planes the arena has slots for: all of them for a routing context, otherwise all but the routing's (the last
                          // ones of rh_fields.def) -- the tile stride of the non-routing contexts stays what it was before the routing was
                          // added (at 10^6 columns the fused step ran 13 % slower with nine more slots per tile: 2.21 instead of 2.14 GB,
                          // A/B on one box)
    // routing (settings.enable_routing_1D): edge columns of this rank and halo columns of its x-neighbours, ny values each
    double *route_q;      // [0, 2 ny): own edges lo / hi of q_out; [2 ny, 4 ny): halo lo / hi
    int *route_i;         // flow direction and mask: [0, 4 ny) own edges (fd lo, fd hi, mk lo, mk hi), [4 ny, 8 ny) halos (same order)
    bool route_halo[2];   // a halo column is present on that side (rh_route_set_halo or the RCCL exchange)
    bool route_static_done;   // the neighbours' flow direction and mask have been exchanged over RCCL
    std::string err;
};
#define RH_DT_LOG_CAP 65536

static std::string g_create_err;

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
// OR a wavefront's predicate bits into a global word.  The word only ever gains bits during a
// kernel, so a wave whose bits are already present skips the atomic: after the first few waves
// nobody touches the word any more (one address sustains only ~90 atomics/us chip-wide).  The
// pre-check may read a stale (smaller) value, which costs an extra atomic, never a lost bit.
RH_DEV void wave_or_to(unsigned long long *word, unsigned long long bits) {
    for (int off = 32; off; off >>= 1) bits |= __shfl_xor(bits, off);
    if ((threadIdx.x & 63) == 0 && bits) {
        const unsigned long long seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bits & ~seen) atomicOr(word, bits);
    }
}
// OR over the workgroup, then one plain store per workgroup.
RH_DEV void block_or_store(unsigned long long *slot, unsigned long long bits) {
    __shared__ unsigned long long wv[RH_BLOCK / 64];
    for (int off = 32; off; off >>= 1) bits |= __shfl_xor(bits, off);
    if ((threadIdx.x & 63) == 0) wv[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = 0;
        for (int k = 0; k < RH_BLOCK / 64; ++k) b |= wv[k];
        *slot = b;
    }
}
// OR over the wavefront, one plain store per wave: no barrier, so a wave that is done retires at once (the fused
// kernel's waves finish at different times; a closing barrier would hold their registers until the slowest is done)
RH_DEV void wave_or_store(unsigned long long *wave_slots, unsigned long long bits) {
    for (int off = 32; off; off >>= 1) bits |= __shfl_xor(bits, off);
    if ((threadIdx.x & 63) == 0) wave_slots[threadIdx.x >> 6] = bits;
}
// OR-reduce the per-workgroup words (one workgroup of RH_BLOCK threads); result valid in thread 0.
RH_DEV unsigned long long reduce_bflags(const unsigned long long *bf, int nblk) {
    __shared__ unsigned long long wv[RH_BLOCK / 64];
    unsigned long long b = 0;
    // 32 independent loads in flight per thread: the words were written by other CUs (other XCDs' L2s), a single
    // workgroup reading 15 000 of them a few dependent loads at a time is latency-bound (2 us per round trip)
    for (int k = threadIdx.x; k < nblk; k += 32 * RH_BLOCK) {
        unsigned long long v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int idx = k + j * RH_BLOCK;
            v[j] = idx < nblk ? bf[idx] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) b |= v[j];
    }
    for (int off = 32; off; off >>= 1) b |= __shfl_xor(b, off);
    if ((threadIdx.x & 63) == 0) wv[threadIdx.x >> 6] = b;
    __syncthreads();
    b = 0;
    for (int k = 0; k < RH_BLOCK / 64; ++k) b |= wv[k];
    __syncthreads();
    return b;
}
#define BIT(b) (1ull << (b))
RH_DEV bool bit(unsigned long long w, int b) { return (w >> b) & 1ull; }

// numpy's pairwise add.reduce over 144 contiguous float64 (two blocks of 72, eight partial sums
// each) -- the reference aggregates the day's forcing with np.sum / np.nanmean
// (adaptive_time_stepping.py:384-437), so the grouping is part of the result.
RH_DEV double np_sum72(const double *a) {
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    for (int i = 8; i < 72; i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    return ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
}
// ... of values given by an accessor: the eight partial sums run in registers (a staging buffer of 72 doubles per thread lived in scratch
// memory: the daily sums of 10^6 columns with weighted forcing took 1.4 ms, all of it scratch traffic)
template <class Get>
RH_DEV double np_sum72_of(Get get, int base) {
    double r0 = get(base), r1 = get(base + 1), r2 = get(base + 2), r3 = get(base + 3);
    double r4 = get(base + 4), r5 = get(base + 5), r6 = get(base + 6), r7 = get(base + 7);
#pragma unroll 1   // (fully unrolled the 144 loads of a sum are hoisted together: 512 registers and spills)
    for (int i = 8; i < 72; i += 8) {
        r0 += get(base + i); r1 += get(base + i + 1); r2 += get(base + i + 2); r3 += get(base + i + 3);
        r4 += get(base + i + 4); r5 += get(base + i + 5); r6 += get(base + i + 6); r7 += get(base + i + 7);
    }
    return ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
}
template <class Get>
RH_DEV double np_sum144(Get get) {
    const double h0 = np_sum72_of(get, 0);
    return 0.0 + (h0 + np_sum72_of(get, 72));
}

// aggregates {prec, ta, pet} x {daily, hourly, 10 min} of one forcing series (stride between
// consecutive slots given, so the same code serves the shared vector and per-cell rows)
// np.sum over the 144 slots of a series that is 0 outside the hourly window [itd, itd + 6) (the masked sums of
// adaptive_time_stepping.py:400-420), in numpy's pairwise order without walking the 138 zeros: the window's six
// consecutive slots fall into six different lanes of the two 72-blocks (lane = slot mod 8), every lane also receives
// zeros (v + 0.0: a negative zero becomes positive, as in the full sum), and the lanes are combined as np_sum72 does.
// itd is uniform over the grid, so the lane selection is scalar work.
// A window that lies inside one 72-block (every hourly step's: itd a multiple of 6) with a start that is the same over the wavefront
// takes the short way: the six values sit in six of the eight lanes of ONE block in rotated order, r = itd mod 8 says where, and each of
// the eight rotations is the tree ((l0 + l1) + (l2 + l3)) + ((l4 + l5) + (l6 + l7)) with its two zero lanes written out of it (x + 0.0 = x
// for everything but a negative zero, which `+ 0.0` on the way in has removed; the other block's 0.0 and the leading 0.0 + likewise):
// five additions behind a scalar branch instead of 96 selects per sum (k_cell_agg<1> at 10^6 columns: 39 -> 17 us).
template <class Get>
RH_DEV double np_sum144_window_general(Get get, int64_t itd) {
    double lane[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) lane[h][j] = 0.0;
#pragma unroll
    for (int w = 0; w < 6; ++w) {
        const int64_t k = itd + w;
        if (k < 0 || k >= RH_SLOTS_PER_DAY) continue;
        const double v = get((int)k) + 0.0;
        const int h = k >= 72 ? 1 : 0, j = (int)((k - 72 * h) & 7);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                if (hh == h && jj == j) lane[hh][jj] = v;
    }
    const double h0 = ((lane[0][0] + lane[0][1]) + (lane[0][2] + lane[0][3])) + ((lane[0][4] + lane[0][5]) + (lane[0][6] + lane[0][7]));
    const double h1 = ((lane[1][0] + lane[1][1]) + (lane[1][2] + lane[1][3])) + ((lane[1][4] + lane[1][5]) + (lane[1][6] + lane[1][7]));
    return 0.0 + (h0 + h1);
}
template <class Get>
RH_DEV double np_sum144_window(Get get, int64_t itd) {
    const int iu = __builtin_amdgcn_readfirstlane((int)itd);
    if (__all((int64_t)iu == itd) && iu >= 0 && iu + 6 <= RH_SLOTS_PER_DAY && !(iu < 72 && iu + 6 > 72)) {
        const double v0 = get(iu) + 0.0, v1 = get(iu + 1) + 0.0, v2 = get(iu + 2) + 0.0, v3 = get(iu + 3) + 0.0, v4 = get(iu + 4) + 0.0,
                     v5 = get(iu + 5) + 0.0;
        switch (iu & 7) {
        case 0: return ((v0 + v1) + (v2 + v3)) + (v4 + v5);
        case 1: return (v0 + (v1 + v2)) + ((v3 + v4) + v5);
        case 2: return (v0 + v1) + ((v2 + v3) + (v4 + v5));
        case 3: return (v5 + v0) + ((v1 + v2) + (v3 + v4));
        case 4: return (v4 + v5) + ((v0 + v1) + (v2 + v3));
        case 5: return ((v3 + v4) + v5) + (v0 + (v1 + v2));
        case 6: return ((v2 + v3) + (v4 + v5)) + (v0 + v1);
        default: return ((v1 + v2) + (v3 + v4)) + (v5 + v0);
        }
    }
    return np_sum144_window_general(get, itd);
}

// aggregates {prec, ta, pet} x {daily, hourly, 10 min} of one forcing series given by accessors (per-cell rows, or the
// weighted station forcing: PREC[k] * w, TA[k] + offset, PET[k] * w).  daily = false leaves a[0..2] alone: the daily
// sums only change with the day.
template <class P, class T, class E>
RH_DEV void forcing_aggregates_of(P p, T t, E e, int64_t itd, double *a, bool daily = true, bool hourly = true);
RH_DEV void forcing_aggregates(const double *p, const double *t, const double *e, int64_t itd, double *a) {
    forcing_aggregates_of([&](int k) { return p[k]; }, [&](int k) { return t[k]; }, [&](int k) { return e[k]; }, itd, a);
}
template <class P, class T, class E>
RH_DEV void forcing_aggregates_of(P p, T t, E e, int64_t itd, double *a, bool daily, bool hourly) {
    if (daily) {
        a[0] = np_sum144([&](int k) { return p(k); });
        int cnt = 0;
        for (int k = 0; k < 144; ++k) cnt += !isnan(t(k));
        a[1] = np_sum144([&](int k) { const double v = t(k); return isnan(v) ? 0.0 : v; }) / (double)cnt;
        a[2] = np_sum144([&](int k) { return e(k); });
    }
    if (!hourly) return;
    a[3] = np_sum144_window([&](int k) { return p(k); }, itd);
    {
        int cnt = 0;
        for (int w = 0; w < 6; ++w) {
            const int64_t k = itd + w;
            cnt += (k >= 0 && k < 144) && !isnan(t((int)k));
        }
        a[4] = np_sum144_window([&](int k) { const double v = t(k); return isnan(v) ? 0.0 : v; }, itd) / (double)cnt;
    }
    a[5] = np_sum144_window([&](int k) { return e(k); }, itd);
    int64_t k = itd < 0 ? itd + 144 : itd;
    k = k > 143 ? 143 : k;
    a[6] = p((int)k);
    a[7] = t((int)k);
    a[8] = e((int)k);
}

RH_DEV unsigned long long forcing_bits(double p, double t, const Consts &K) {
    unsigned long long b = 0;
    const double hpi = (double)K.hpi;
    b |= !(p <= 0) ? BIT(PB_P_NOT_LE0) : 0;
    b |= (p > 0) ? BIT(PB_P_GT0) : 0;
    b |= (p > hpi) ? BIT(PB_P_GT_HPI) : 0;
    b |= !(p <= hpi) ? BIT(PB_P_NOT_LE_HPI) : 0;
    b |= !(t > K.ta_fm) ? BIT(PB_TA_NOT_GT) : 0;
    b |= (t > K.ta_fm) ? BIT(PB_TA_GT) : 0;
    b |= ((p > 0) && (t <= K.ta_fm)) ? BIT(PB_PGT0_TALE) : 0;
    b |= !((p <= 0) && (t <= K.ta_fm)) ? BIT(PB_NOT_PLE0_TALE) : 0;
    return b;
}

// ---------------------------------------------------------------------------------------------
// adaptive time stepping (adaptive_time_stepping.py:22-381)
// ---------------------------------------------------------------------------------------------
// The benchmark's `set_forcing` and `set_parameters` hooks (benchmarks/SVAT_benchmark.py:105-110,
// 151-171) on the device: at midnight take the next 144 forcing slots and the calendar entry;
// flag a month change.  Called by one whole workgroup of RH_BLOCK threads.
RH_DEV void hooks_set_forcing(DevState *D) {
    rh_scalars &S = D->S;
    const bool midnight = (S.time % 86400 == 0);
    const int64_t i0 = S.itt_forc;
    const bool have = midnight && (i0 + RH_SLOTS_PER_DAY <= D->nitt_forc);
    __syncthreads();  // everybody has read S before thread 0 changes it
    if (have && threadIdx.x < RH_SLOTS_PER_DAY)
        for (int k = 0; k < 3; ++k) D->forc[k][threadIdx.x] = D->series[k][i0 + threadIdx.x];
    if (have && threadIdx.x == 0) D->day_cache_ok = 0;
    if (have && D->n_stations > 0) {   // the series are (n_stations, nitt_forc): the day of every station
        const int S = D->n_stations;
        for (int q = threadIdx.x; q < 3 * S * RH_SLOTS_PER_DAY; q += RH_BLOCK) {
            const int v = q / (S * RH_SLOTS_PER_DAY), r = q % (S * RH_SLOTS_PER_DAY), st = r / RH_SLOTS_PER_DAY, j = r % RH_SLOTS_PER_DAY;
            D->forc_multi[q] = D->series[v][(size_t)st * D->nitt_forc + i0 + j];
        }
    }
    if (threadIdx.x == 0) {
        if (have) {
            S.itt_day = 0;
            S.year[1] = D->calendar[0][i0];
            S.month[1] = D->calendar[1][i0];
            S.doy[1] = D->calendar[2][i0];
            S.itt_forc = i0 + RH_SLOTS_PER_DAY;
            D->per_cell = D->weights[0] ? 1 : 0;
        }
        D->monthly = (S.month[1] != S.month[0]) && (S.itt > 1);
        if (midnight && !have) D->err_flags |= RH_DEVERR_FORCING;   // the step would run on yesterday's forcing: reported by rh_sync / rh_get_scalars
    }
    __threadfence();
    __syncthreads();
}
__global__ __launch_bounds__(RH_BLOCK) void k_set_forcing(DevState *D) { hooks_set_forcing(D); }
// Weighted station forcing (eberbaechle/svat_distributed/svat.py:276-296: prec_day = PREC * prec_weight, ta_day = TA +
// ta_offset, pet_day = PET * pet_weight): the day's series stays one 144-vector, staged in LDS, and every column forms
// its own values from its three weights on the fly -- no (n, 144) arrays, 24 bytes of weights per column and step.
struct DaySeries {
    double f[3][RH_SLOTS_PER_DAY];
};
// The day a kernel in front of the fused step works on when the set_forcing hook rides along (k_cell_front): at midnight, with a day left
// in the resident series, that is the NEXT day -- read straight from the series; the kernel's last wavefront then does what the hook does to
// the state (front_ctrl), and nothing has to run in front of the kernel.
struct FreshDay {
    bool fresh;     // the hook finds midnight and a day left: the new day
    bool missing;   // midnight, but the series is exhausted (the step runs on yesterday's forcing and the error is reported)
    int64_t i0;     // first slot of the new day in the series
    int64_t itd;    // itt_day as the hook leaves it
};
RH_DEV FreshDay fresh_day(const DevState *D, int hooks) {
    FreshDay f;
    const bool midnight = hooks && (D->S.time % 86400 == 0);
    f.i0 = D->S.itt_forc;
    f.fresh = midnight && (f.i0 + RH_SLOTS_PER_DAY <= D->nitt_forc);
    f.missing = midnight && !f.fresh;
    f.itd = f.fresh ? 0 : D->S.itt_day;
    return f;
}
RH_DEV void stage_day(const DevState *D, DaySeries &s, const FreshDay *fd = nullptr) {
    const bool fresh = fd && fd->fresh;
    for (int k = threadIdx.x; k < 3 * RH_SLOTS_PER_DAY; k += RH_BLOCK) {
        const int v = k / RH_SLOTS_PER_DAY, j = k % RH_SLOTS_PER_DAY;
        s.f[v][j] = fresh ? D->series[v][fd->i0 + j] : D->forc[v][j];
    }
    __syncthreads();
}
// slot k (uniform) of a variable from a wavefront's registers (ctrl_inputs, k_cell_front) (x0, x1, x2: the lane's slots lane, lane + 64, lane + 128)
RH_DEV double lane_double(double x, int l) {   // lane l's x (l uniform): v_readlane, no trip through the LDS crossbar
    const long long bits = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)bits, l), hi = __builtin_amdgcn_readlane((int)(bits >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
RH_DEV double ctrl_slot(double x0, double x1, double x2, int k) {
    // The register is chosen by a UNIFORM branch, not by a select in front of the read: under a divergent mask a select leaves the
    // switched-off lanes' copy unwritten, and v_readlane reads whatever lane it is told to (the registers themselves were loaded with
    // every lane on).
    const int r = __builtin_amdgcn_readfirstlane(k >> 6), l = k & 63;
    if (r == 0) return lane_double(x0, l);
    if (r == 1) return lane_double(x1, l);
    return lane_double(x2, l);
}
// The day's series as a column sees it: the one shared series staged in LDS, or -- with several stations -- its station's rows of
// forc_multi (a table of 3 x n_stations x 144 values: cache-resident; on a fresh day the rows of the series themselves); a column without
// a station reads zeros.
struct DayView {
    const DaySeries *lds;  // null: the one series is read where it lies (DevState::forc) -- a step that needs six slots of it does not stage the day
    const double *sv[3];   // several stations: slot 0 of station 0, per variable; one series without staging: DevState::forc[v]
    size_t ststride;       // ... and the distance between two stations
    bool multi;
    int st;
    RH_DEV double operator()(int v, int k) const {
        if (!multi) return lds ? lds->f[v][k] : sv[v][k];
        return st < 0 ? 0.0 : sv[v][(size_t)st * ststride + k];
    }
};
RH_DEV DayView day_view(const DevState *D, const DaySeries &lds, int64_t i, const FreshDay *fd = nullptr, bool staged = true) {
    DayView d;
    d.lds = staged ? &lds : nullptr;
    d.multi = D->n_stations > 0;
    d.st = d.multi ? D->station_idx[i] : 0;
    const bool fresh = fd && fd->fresh;

    for (int v = 0; v < 3; ++v)
        d.sv[v] = !d.multi ? (staged ? nullptr : (fresh ? D->series[v] + fd->i0 : &D->forc[v][0]))   // (fresh: the day the hook is about to bring)
                           : (fresh ? D->series[v] + fd->i0 : D->forc_multi + (size_t)v * D->n_stations * RH_SLOTS_PER_DAY);
    d.ststride = fresh ? (size_t)D->nitt_forc : (size_t)RH_SLOTS_PER_DAY;
    return d;
}
// start-of-step predicates over the columns (adaptive_time_stepping.py:38-81), grid-stride
__global__ __launch_bounds__(RH_BLOCK) void k_pred1(Arena a, DevState *D, int force_daily) {
    const Consts K = D->K;
    unsigned long long b = 0;
    const bool per_cell = D->per_cell != 0, weighted = D->weights[0] != nullptr;
    // weighted station forcing: the day's series changes at midnight only, so the forcing bits of a workgroup's columns
    // (the same columns every step: the grid-stride mapping is fixed) are formed once a day and kept
    const bool daily = force_daily || D->S.itt_day == 0;
    __shared__ DaySeries day;
    if (per_cell && weighted && daily) stage_day(D, day);
    for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * RH_BLOCK) {
        double swe, swe_top;
        rh_ld(a, RH_P_swe, i, swe);
        rh_ld(a, RH_P_swe_top, i, swe_top);
        b |= !(swe <= 0) ? BIT(PB_SWE_NOT_LE0) : 0;
        b |= (swe > 0) ? BIT(PB_SWE_GT0) : 0;
        b |= !(swe_top <= 0) ? BIT(PB_SWETOP_NOT_LE0) : 0;
        b |= (swe_top > 0) ? BIT(PB_SWETOP_GT0) : 0;
        if (per_cell && weighted) {
            if (daily) {
                const double pw = D->weights[0][i], toff = D->weights[1][i];
                const DayView F = day_view(D, day, i);
                for (int k = 0; k < RH_SLOTS_PER_DAY; ++k) b |= forcing_bits(F(0, k) * pw, F(1, k) + toff, K);
            }
        } else if (per_cell) {
            const double *p = D->forc_cell[0] + i, *t = D->forc_cell[1] + i;   // (144, n): stride n between the slots
            for (int k = 0; k < RH_SLOTS_PER_DAY; ++k) b |= forcing_bits(p[(size_t)k * a.n], t[(size_t)k * a.n], K);
        }
    }
    if (per_cell && weighted) {   // keep / reuse the day's forcing bits of this workgroup
        const unsigned long long cols = BIT(PB_SWE_NOT_LE0) | BIT(PB_SWE_GT0) | BIT(PB_SWETOP_NOT_LE0) | BIT(PB_SWETOP_GT0);
        __shared__ unsigned long long s_day;
        if (daily) {
            block_or_store(&D->day_bflags[blockIdx.x], b & ~cols);
            __syncthreads();   // block_or_store's scratch is used again below
        } else {
            if (threadIdx.x == 0) s_day = D->day_bflags[blockIdx.x];
            __syncthreads();
            b |= s_day;
        }
    }
    block_or_store(&D->bflags[0][blockIdx.x], b);
}

// word 0 = OR of the workgroup words of k_pred1 (or, summary path, of the fused kernel's summary words: their
// bits 0..3 are word 0's column bits) and the predicates of the shared forcing series.  Returns the OR of the
// workgroup words in thread 0.
RH_DEV unsigned long long finish_word0(DevState *D, const unsigned long long *flags, int nflags, unsigned long long cell_mask) {
    unsigned long long fb = 0;
    if (!D->per_cell && threadIdx.x < RH_SLOTS_PER_DAY) fb = forcing_bits(D->forc[0][threadIdx.x], D->forc[1][threadIdx.x], D->K);
    __shared__ unsigned long long fw[RH_BLOCK / 64];
    for (int off = 32; off; off >>= 1) fb |= __shfl_xor(fb, off);
    if ((threadIdx.x & 63) == 0) fw[threadIdx.x >> 6] = fb;
    const unsigned long long cells = reduce_bflags(flags, nflags);  // contains __syncthreads
    if (threadIdx.x == 0) {
        unsigned long long w = cells & cell_mask;
        for (int k = 0; k < RH_BLOCK / 64; ++k) w |= fw[k];
        D->words[0] = w;
    }
    __threadfence();
    __syncthreads();
    return cells;
}
__global__ __launch_bounds__(RH_BLOCK) void k_reduce(DevState *D, int which) {
    if (which == 0) {
        finish_word0(D, D->bflags[0], D->pred_blocks, ~0ull);
    } else {
        const unsigned long long w = reduce_bflags(D->bflags[1], D->pred_blocks);
        if (threadIdx.x == 0) D->words[1] = w;
    }
}

// Shared forcing: the nine aggregates of the day's 144-slot series in numpy's summation order,
// computed by one workgroup from LDS.  np.sum over 144 contiguous float64 is
// 0 + (half(0..71) + half(72..143)), each half = ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) with
// r_j = a[j] + a[j+8] + ... + a[j+64] accumulated in that order (numpy pairwise_sum, n <= 128).
// Lane j of a 16-lane group owns one r_j; six groups = six sums.
RH_DEV double np_tree8(const double *r) { return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7])); }

RH_DEV void agg_body(DevState *D) {
    __shared__ double f[3][RH_SLOTS_PER_DAY];   // prec, ta, pet of the day
    __shared__ double part[6][16];
    const int tid = threadIdx.x;
    const bool shared_forcing = !D->per_cell;
    const int64_t itd = D->S.itt_day;
    if (shared_forcing && tid < RH_SLOTS_PER_DAY)
        for (int k = 0; k < 3; ++k) f[k][tid] = D->forc[k][tid];
    __syncthreads();
    if (shared_forcing && tid < 96) {
        const int sum_id = tid >> 4, lane = tid & 15, half = lane >> 3, j = lane & 7;
        const int var = sum_id % 3;          // 0 prec, 1 ta, 2 pet
        const bool hourly = sum_id >= 3;     // sums 0..2 daily, 3..5 hourly window
        double r = 0.0;
        for (int q = 0; q < 9; ++q) {
            const int k = half * 72 + j + 8 * q;
            double v = f[var][k];
            const bool in = !hourly || ((k >= itd) && (k < itd + 6));
            if (var == 1) v = (in && !isnan(v)) ? v : 0.0;  // nanmean: NaN (and masked) slots count as 0
            else v = in ? v : 0.0;
            r = (q == 0) ? v : r + v;
        }
        part[sum_id][lane] = r;
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long w = D->words[0];
        const bool all_p_le0 = !bit(w, PB_P_NOT_LE0), any_p_gt0 = bit(w, PB_P_GT0), any_p_gthpi = bit(w, PB_P_GT_HPI);
        const bool all_p_lehpi = !bit(w, PB_P_NOT_LE_HPI), all_ta_gt = !bit(w, PB_TA_NOT_GT), any_ta_gt = bit(w, PB_TA_GT);
        const bool any_pgt0_tale = bit(w, PB_PGT0_TALE), all_ple0_tale = !bit(w, PB_NOT_PLE0_TALE);
        const bool all_swe_le0 = !bit(w, PB_SWE_NOT_LE0), all_swetop_le0 = !bit(w, PB_SWETOP_NOT_LE0);
        const bool snow_any = (bit(w, PB_SWE_GT0) || bit(w, PB_SWETOP_GT0)) && any_ta_gt;
        const bool cond0 = all_p_le0 && all_swe_le0 && all_swetop_le0 && all_ta_gt;
        const bool cond00 = any_pgt0_tale || all_ple0_tale;
        const bool cond1 = any_p_gthpi && any_p_gt0 && any_ta_gt;
        const bool cond2 = all_p_lehpi && any_p_gt0 && any_ta_gt;
        const bool cond3 = any_p_gthpi && any_p_gt0 && snow_any;
        const bool cond4 = all_p_lehpi && any_p_gt0 && snow_any;
        const bool cond5 = all_p_le0 && snow_any;
        StepCtx X = D->X;
        X.cond_time = (D->S.time % 86400 == 0);
        X.sel_daily = cond0 || cond00;
        X.sel_hourly = (cond2 || cond4 || cond5) && !cond1 && !cond3;
        X.sel_10min = (cond1 || cond3) && !cond2 && !cond4 && !cond5;
        // :143-144 (the second assignment overwrites the first), :166, :190
        int64_t dts = X.cond_time ? 86400 : 3600;
        if (X.sel_hourly) dts = 3600;
        if (X.sel_10min) dts = 600;
        X.dt_secs_prelim = dts;
        X.itt_day = itd;
        X.sel_p = X.sel_10min ? 2 : (X.sel_hourly ? 1 : (X.sel_daily ? 0 : -1));
        if (shared_forcing) {
            double s[6];
            for (int q = 0; q < 6; ++q) s[q] = 0.0 + (np_tree8(&part[q][0]) + np_tree8(&part[q][8]));
            int cnt_d = 0, cnt_h = 0;
            for (int k = 0; k < RH_SLOTS_PER_DAY; ++k) {
                const bool ok = !isnan(f[1][k]);
                cnt_d += ok;
                cnt_h += ok && (k >= itd) && (k < itd + 6);
            }
            X.agg[0] = s[0];
            X.agg[1] = s[1] / (double)cnt_d;
            X.agg[2] = s[2];
            X.agg[3] = s[3];
            X.agg[4] = s[4] / (double)cnt_h;
            X.agg[5] = s[5];
            int64_t k = itd < 0 ? itd + RH_SLOTS_PER_DAY : itd;
            k = k > RH_SLOTS_PER_DAY - 1 ? RH_SLOTS_PER_DAY - 1 : k;
            X.agg[6] = f[0][k];
            X.agg[7] = f[1][k];
            X.agg[8] = f[2][k];
            if (X.sel_p >= 0) {
                X.prec_sel = X.agg[3 * X.sel_p];
                X.ta_sel = X.agg[3 * X.sel_p + 1];
            }
        }
        D->X = X;
    }
}
__global__ __launch_bounds__(RH_BLOCK) void k_agg(DevState *D, int do_hooks, int do_reduce) {
    if (do_hooks) hooks_set_forcing(D);   // rh_run_steps: the user hooks ride along
    if (do_reduce) finish_word0(D, D->bflags[0], D->pred_blocks, ~0ull);  // single GPU: no exchange between k_pred1 and here
    agg_body(D);
}

// Per-cell forcing only: aggregates of every column's own 144-slot series, once per step, into
// nine SoA planes (so the per-column kernels stay free of the 144-element loops).
// PART: 0 = everything in one launch; 1 = the hourly window and the current slot only; 2 = the daily sums only (returns at once unless they
// are due).  From 65 536 columns on the host launches 1 and 2: the daily sums' code needs 214 registers, which leaves the hourly part --
// every step's part -- two waves per SIMD for a chain of dependent loads (58 us at 10^6 columns; on its own 25 us).
template <int PART>
__global__ __launch_bounds__(RH_BLOCK) void k_cell_agg(Arena a, DevState *D, int force_daily) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    const bool weighted = D->weights[0] != nullptr;
    // the station series of the day changes at midnight only (device-side hooks): its daily sums are formed once a day,
    // the first step of the day has itt_day == 0; rows uploaded by the host may change at any time
    const bool daily = (PART != 1) && (!weighted || force_daily || D->S.itt_day == 0);
    if (PART == 2 && !daily) return;
    __shared__ DaySeries day;
    if (weighted) stage_day(D, day);
    if (i >= a.n) return;
    double agg[9];
    if (weighted) {
        const double pw = D->weights[0][i], toff = D->weights[1][i], ew = D->weights[2][i];
        const DayView F = day_view(D, day, i);
        forcing_aggregates_of([&](int k) { return F(0, k) * pw; }, [&](int k) { return F(1, k) + toff; },
                              [&](int k) { return F(2, k) * ew; }, D->S.itt_day, agg, daily, PART != 2);
    } else {
        const double *p = D->forc_cell[0] + i, *t = D->forc_cell[1] + i, *e = D->forc_cell[2] + i;
        const size_t n = (size_t)a.n;
        forcing_aggregates_of([&](int k) { return p[k * n]; }, [&](int k) { return t[k * n]; }, [&](int k) { return e[k * n]; },
                              D->S.itt_day, agg, daily, PART != 2);
    }
    if (daily)
        for (int k = 0; k < 3; ++k) D->agg_cell[(size_t)k * a.n + i] = agg[k];
    if (PART != 2)
        for (int k = 3; k < 9; ++k) D->agg_cell[(size_t)k * a.n + i] = agg[k];
}
RH_DEV double cell_agg(const DevState *D, int64_t n, int64_t i, int k) { return D->agg_cell[(size_t)k * n + i]; }


// mode: RH_SELECT_M1_PENDING = the tau -> taum1 copies of the last fused step are still pending (lazy rotation): prec_m1 / swe_m1 are
// the tau planes as they stand; RH_SELECT_DEFER = the selected prec / ta are not stored, the fused kernel applies the selection itself
// (StepCtx.apply_sel = 2) -- the planes stay untouched between two fused steps, so the rotation can stay pending
#define RH_SELECT_M1_PENDING 1
#define RH_SELECT_DEFER 2
__global__ __launch_bounds__(RH_BLOCK) void k_select(Arena a, DevState *D, int mode) {
    const Consts K = D->K;
    const StepCtx X = D->X;
    const bool per_cell = D->per_cell != 0;
    unsigned long long b = 0;
    for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * RH_BLOCK) {
        Col c;
        rh_ld(a, RH_P_prec, i, c.prec);
        rh_ld(a, RH_P_ta, i, c.ta);
        double prec_m1, swe, swe_top, swe_m1;
        rh_ld(a, RH_P_swe, i, swe);
        rh_ld(a, RH_P_swe_top, i, swe_top);
        if (mode & RH_SELECT_M1_PENDING) {
            prec_m1 = c.prec;
            swe_m1 = swe;
        } else {
            rh_ld(a, RH_P_prec_m1, i, prec_m1);
            rh_ld(a, RH_P_swe_m1, i, swe_m1);
        }
        if (X.sel_p >= 0) {
            if (per_cell)
                rt_select_prec_ta(c, X, cell_agg(D, a.n, i, 3 * X.sel_p), cell_agg(D, a.n, i, 3 * X.sel_p + 1));
            else
                rt_select_prec_ta(c, X, X.prec_sel, X.ta_sel);
            if (!(mode & RH_SELECT_DEFER)) {
                rh_st(a, RH_P_prec, i, c.prec);
                rh_st(a, RH_P_ta, i, c.ta);
            }
        }
        const bool warm = c.ta > K.ta_fm;
        b |= ((c.prec > 0) && warm) ? BIT(PC_RAIN) : 0;
        b |= (((swe > 0) || (swe_top > 0)) && warm) ? BIT(PC_SNOWMELT) : 0;
        b |= !(c.prec <= 0) ? BIT(PC_PREC_NOT_LE0) : 0;
        b |= !((c.prec > 0) && (c.ta <= K.ta_fm)) ? BIT(PC_NOT_PGT0_TALE) : 0;
        b |= (swe_m1 > 0) ? BIT(PC_SWEM1_GT0) : 0;
        b |= !(swe <= 0) ? BIT(PC_SWE_NOT_LE0) : 0;
        b |= (c.prec == 0) ? BIT(PC_P_EQ0) : 0;
        b |= (prec_m1 != 0) ? BIT(PC_PM1_NE0) : 0;
        b |= (c.prec != 0) ? BIT(PC_P_NE0) : 0;
        b |= (prec_m1 == 0) ? BIT(PC_PM1_EQ0) : 0;
    }
    block_or_store(&D->bflags[1][blockIdx.x], b);
}

// infiltration.py:2155-2167 from the predicate word and the event ids
RH_DEV void infiltration_conds(const rh_scalars &S, StepCtx &X, unsigned long long w) {
    X.cond1 = (S.event_id[0] == 0) && (S.event_id[1] >= 1);
    X.cond2 = bit(w, PC_P_EQ0) && bit(w, PC_PM1_NE0) && (S.event_id[0] >= 1);
    X.cond3 = bit(w, PC_P_NE0) && bit(w, PC_PM1_EQ0) && (S.event_id[0] == S.event_id[1]);
    X.cond4 = (S.event_id[0] >= 1) && (S.event_id[1] == 0);
    X.cond5 = S.event_id[1] >= 1;
}

// adaptive_time_stepping.py:192-373, scalar part
RH_DEV void scalars_body(DevState *D, unsigned long long w, int do_finish, int apply_sel);
__global__ __launch_bounds__(RH_BLOCK) void k_scalars(DevState *D, int do_reduce, int do_finish, int apply_sel = 0) {
    unsigned long long w = 0;
    if (do_reduce) w = reduce_bflags(D->bflags[1], D->pred_blocks);
    if (threadIdx.x != 0) return;
    if (!do_reduce) w = D->words[1];
    scalars_body(D, w, do_finish, apply_sel);
}
// agg[3 * sel + off] without a dynamic index (which would put the whole step context into scratch memory)
RH_DEV double agg_pick(const StepCtx &X, int sel, int off) {
    const double a0 = off == 0 ? X.agg[0] : (off == 1 ? X.agg[1] : X.agg[2]);
    const double a1 = off == 0 ? X.agg[3] : (off == 1 ? X.agg[4] : X.agg[5]);
    const double a2 = off == 0 ? X.agg[6] : (off == 1 ? X.agg[7] : X.agg[8]);
    return sel == 0 ? a0 : (sel == 1 ? a1 : a2);
}
// the bookkeeping itself, on copies of the scalars and of the step context
RH_DEV int64_t scalars_update(rh_scalars &S, StepCtx &X, unsigned long long w, int do_finish, int apply_sel, bool per_cell, int64_t ee) {
    X.apply_sel = apply_sel;
    X.halt = 0;   // (the time limit is the summary path's, ctrl_wave: it sets `last` after this bookkeeping)
    X.last = 0;
    const bool ev_start = bit(w, PC_RAIN) || bit(w, PC_SNOWMELT);
    const bool ev_end = !bit(w, PC_PREC_NOT_LE0) || !bit(w, PC_NOT_PGT0_TALE) || (bit(w, PC_SWEM1_GT0) && !bit(w, PC_SWE_NOT_LE0));
    int64_t dts = X.dt_secs_prelim;
    if (ev_start) S.time_event0 = 0;
    if (ev_end) S.time_event0 = S.time_event0 + dts;
    const int64_t te0 = S.time_event0, tm = S.time;
    const bool c6 = (te0 <= ee) && (dts == 600), c7 = (te0 <= ee) && (dts == 3600), c8 = (te0 <= ee) && (dts == 86400);
    const bool c9 = (te0 > ee) && (tm % 3600 != 0) && (dts == 600);
    const bool c10 = (te0 > ee) && (tm % 3600 == 0) && ((dts == 600) || (dts == 3600));
    const bool c11 = (te0 > ee) && (tm % 86400 == 0) && (dts == 86400);
    int w_sel = -1;
    double dt = S.dt;
    int64_t itd = S.itt_day;
    if (c6) { w_sel = 2; S.event_id[1] = S.event_id_counter; dt = 1.0 / 6; itd += 1; }
    if (c7) { w_sel = 1; S.event_id[1] = S.event_id_counter; dt = 1; itd += 6; }
    if (c8) { w_sel = 0; dt = 24; itd = 0; }
    if (c9) { w_sel = 2; S.event_id[1] = 0; dt = 1.0 / 6; dts = 600; itd += 1; }
    if (c10) { w_sel = 1; S.event_id[1] = 0; dt = 1; dts = 3600; itd += 6; }
    if (c11) { w_sel = 0; S.event_id[1] = 0; dt = 24; dts = 86400; itd = 0; }
    S.dt = dt;
    S.dt_secs = dts;
    S.itt_day = itd;
    if ((S.event_id[0] > 0) && (S.event_id[1] == 0)) S.event_id_counter += 1;
    X.sel_w = w_sel;
    if (w_sel >= 0 && !per_cell) {
        X.pet_sel_w = agg_pick(X, w_sel, 2);
        X.ta_sel_w = agg_pick(X, w_sel, 1);
    }
    X.dt = dt;
    X.month_tau = S.month[1];
    infiltration_conds(S, X, w);
    if (do_finish) {
        // roger.py:449-450 and the scalar half of after_timestep (svat.py:352-366).  Nothing below
        // this kernel reads these scalars during the step (k_step works from StepCtx), so they are
        // advanced here instead of in a kernel of their own.
        S.itt += 1;
        S.time += S.dt_secs;
        S.event_id[0] = S.event_id[1];
        S.year[0] = S.year[1];
        S.month[0] = S.month[1];
        S.doy[0] = S.doy[1];
    }
    return dts;
}
RH_DEV void log_dt(DevState *D, int64_t dts) {
    if (D->dt_log) {
        const int k = D->dt_log_n;
        if (k < D->dt_log_cap) D->dt_log[k] = (int)dts;
        D->dt_log_n = k + 1;
    }
}
// one thread
RH_DEV void scalars_body(DevState *D, unsigned long long w, int do_finish, int apply_sel) {
    // one burst of loads, the bookkeeping in registers, one burst of stores: working on D->S / D->X in place costs a
    // global-memory round trip per field for this single thread (the control kernel took 20 us that way)
    rh_scalars S = D->S;
    StepCtx X = D->X;
    const int64_t dts = scalars_update(S, X, w, do_finish, apply_sel, D->per_cell != 0, D->K.end_event);
    D->words[0] = 0;
    D->words[1] = 0;
    D->words[2] = 0;
    D->S = S;
    D->X = X;
    log_dt(D, dts);
}

// ---- summary path (shared forcing): the whole control part of a step in ONE single-workgroup kernel ----------
// Word 1 of this step from the summary bits the fused kernel left at the end of the previous step and the
// (uniform) selected prec / ta; same terms as k_select evaluates per column.
RH_DEV unsigned long long derive_word1(unsigned long long s, const StepCtx &X, const Consts &K) {
    unsigned long long w = 0;
    if (X.sel_p >= 0) {
        const double P = X.prec_sel, T = X.ta_sel;
        const bool warm = T > K.ta_fm;
        w |= ((P > 0) && warm) ? BIT(PC_RAIN) : 0;
        w |= ((bit(s, QB_SWE_GT0) || bit(s, QB_SWETOP_GT0)) && warm) ? BIT(PC_SNOWMELT) : 0;
        w |= !(P <= 0) ? BIT(PC_PREC_NOT_LE0) : 0;
        w |= !((P > 0) && (T <= K.ta_fm)) ? BIT(PC_NOT_PGT0_TALE) : 0;
        w |= (P == 0) ? BIT(PC_P_EQ0) : 0;
        w |= (P != 0) ? BIT(PC_P_NE0) : 0;
    } else {
        w |= bit(s, QB_RAIN_KEEP) ? BIT(PC_RAIN) : 0;
        w |= bit(s, QB_SNOWMELT_KEEP) ? BIT(PC_SNOWMELT) : 0;
        w |= bit(s, QB_P_NOT_LE0) ? BIT(PC_PREC_NOT_LE0) : 0;
        w |= bit(s, QB_NOT_PGT0_TALE) ? BIT(PC_NOT_PGT0_TALE) : 0;
        w |= bit(s, QB_P_EQ0) ? BIT(PC_P_EQ0) : 0;
        w |= bit(s, QB_P_NE0) ? BIT(PC_P_NE0) : 0;
    }
    w |= bit(s, QB_SWE_GT0) ? BIT(PC_SWEM1_GT0) : 0;       // swe[taum1] of this step = swe the last step left
    w |= bit(s, QB_SWE_NOT_LE0) ? BIT(PC_SWE_NOT_LE0) : 0;
    w |= bit(s, QB_P_NE0) ? BIT(PC_PM1_NE0) : 0;           // prec[taum1] likewise
    w |= bit(s, QB_P_EQ0) ? BIT(PC_PM1_EQ0) : 0;
    return w;
}
// Summary bits of one column (values as they stand in the arena at the start of the next step), in two halves
// so that the fused kernel can sample prec / ta and swe / swe_top where each pair is final: the prec/ta half
// (bit 63 carries `warm` to the second half), then the snow half.
#define QB_WARM_TMP 63
RH_DEV unsigned long long summary_bits_pt(double prec, double ta, const Consts &K) {
    unsigned long long b = 0;
    const bool warm = ta > K.ta_fm;
    b |= warm ? BIT(QB_WARM_TMP) : 0;
    b |= ((prec > 0) && warm) ? BIT(QB_RAIN_KEEP) : 0;
    b |= !(prec <= 0) ? BIT(QB_P_NOT_LE0) : 0;
    b |= !((prec > 0) && (ta <= K.ta_fm)) ? BIT(QB_NOT_PGT0_TALE) : 0;
    b |= (prec == 0) ? BIT(QB_P_EQ0) : 0;
    b |= (prec != 0) ? BIT(QB_P_NE0) : 0;
    return b;
}
RH_DEV unsigned long long summary_bits_sw(unsigned long long b, double swe, double swe_top) {
    const bool warm = bit(b, QB_WARM_TMP);
    b &= ~BIT(QB_WARM_TMP);
    b |= !(swe <= 0) ? BIT(QB_SWE_NOT_LE0) : 0;
    b |= (swe > 0) ? BIT(QB_SWE_GT0) : 0;
    b |= !(swe_top <= 0) ? BIT(QB_SWETOP_NOT_LE0) : 0;
    b |= (swe_top > 0) ? BIT(QB_SWETOP_GT0) : 0;
    b |= (((swe > 0) || (swe_top > 0)) && warm) ? BIT(QB_SNOWMELT_KEEP) : 0;
    return b;
}
RH_DEV unsigned long long summary_bits(double swe, double swe_top, double prec, double ta, const Consts &K) {
    return summary_bits_sw(summary_bits_pt(prec, ta, K), swe, swe_top);
}
// ---- the control part of a step by ONE wavefront (no workgroup barrier): k_ctrl, and the tail of the fused kernel ----------
RH_DEV void wave_sync() {   // LDS written by some lanes of a wavefront, read by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
RH_DEV unsigned long long wave_or(unsigned long long b) {
    for (int off = 32; off; off >>= 1) b |= __shfl_xor(b, off);
    return b;
}
struct CtrlLds {
    double f[3][RH_SLOTS_PER_DAY];   // prec, ta, pet of the day
    double part[6][16];              // numpy's eight partial sums of both halves, six sums
};
// [Device-side hooks,] predicate word 0, forcing aggregates in numpy's order, dt and event bookkeeping -- what k_agg + k_select +
// k_scalars do for the predicate-kernel generation -- on the copies S / X every lane holds (uniform); `cells` = OR of the
// summary bits of all columns (of all ranks).  The caller stores S / X.  Shared forcing only (the summary path).
#ifdef RH_STEP_PHASES   // measurement builds: cycles of the tail's parts (one wavefront per launch)
__device__ unsigned long long g_tail_phases[8];
#define RH_TPH(k)                                                  \
    if ((threadIdx.x & 63) == 0) {                                 \
        const unsigned long long t_ = clock64();                   \
        atomicAdd(&g_tail_phases[k], t_ - tph);                    \
        tph = t_;                                                  \
    }
#else
#define RH_TPH(k)
#endif
// The day's shared series spread over the wavefront's registers (lane l holds slots l, l + 64, l + 128 of each variable) and the scalars
// the control part needs besides S / X: requested by the caller BEFORE it waits for anything else, so that the tail of the fused kernel
// makes ONE round trip to memory for all its inputs (it made three in sequence: the summary words and S / X, then the constants, then the
// series into LDS -- 19 600 cycles per tail, a tenth of a 10^6-column step and a third of an 80 x 53 one; profiles/r04_tail_phases.txt).
struct CtrlIn {
    double f[3][3];            // f[v][r]: slot lane + 64 r of variable v (slots >= 144: 0)
    double ta_fm;
    int64_t hpi, end_event, nitt_forc;
    long long t_end;
    int cache_ok;
    unsigned long long day_fb;
    double day_agg[3];
    int *dt_log;               // log_dt's three loads, with the others
    int dt_log_n, dt_log_cap;
};
RH_DEV CtrlIn ctrl_inputs(const DevState *D) {
    CtrlIn in;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int k = lane + 64 * r;
            in.f[v][r] = k < RH_SLOTS_PER_DAY ? D->forc[v][k] : 0.0;
        }
    in.ta_fm = D->K.ta_fm;
    in.hpi = D->K.hpi;
    in.end_event = D->K.end_event;
    in.nitt_forc = D->nitt_forc;
    in.t_end = D->t_end;
    in.cache_ok = D->day_cache_ok;
    in.day_fb = D->day_fb;
    in.day_agg[0] = D->day_agg[0]; in.day_agg[1] = D->day_agg[1]; in.day_agg[2] = D->day_agg[2];
    in.dt_log = D->dt_log;
    in.dt_log_n = D->dt_log_n;
    in.dt_log_cap = D->dt_log_cap;
    return in;
}
// The control part in two halves.  ctrl_pre: everything that does not depend on the columns -- the time limit, the set_forcing hook, the
// day's forcing bits (fb) and the aggregates of the day and of the hourly window; it may run while the columns are still being stepped
// (the extra workgroup of a fused launch, pre_tail), so what running wavefronts read (D->monthly, D->per_cell) is NOT written here but
// handed on in `side` (RH_SIDE_*).  ctrl_post: the decisions on word 0 and word 1 (`cells` = OR of the columns' summary bits), dt and the
// event bookkeeping.
#define RH_SIDE_MONTHLY_SET 1   // D->monthly = bit RH_SIDE_MONTHLY
#define RH_SIDE_MONTHLY 2
#define RH_SIDE_PER_CELL_SET 4  // D->per_cell = bit RH_SIDE_PER_CELL
#define RH_SIDE_PER_CELL 8
RH_DEV void ctrl_side_effects(DevState *D, int side) {   // (one lane)
    if (side & RH_SIDE_PER_CELL_SET) D->per_cell = (side & RH_SIDE_PER_CELL) ? 1 : 0;
    if (side & RH_SIDE_MONTHLY_SET) D->monthly = (side & RH_SIDE_MONTHLY) ? 1 : 0;
}
RH_DEV void ctrl_pre(DevState *D, CtrlLds &L, rh_scalars &S, StepCtx &X, int do_hooks, const CtrlIn &in, unsigned long long &fb_out, int &side) {
#ifdef RH_STEP_PHASES
    unsigned long long tph = clock64();
#endif
    const int lane = threadIdx.x & 63;
    Consts Kf;   // forcing_bits reads hpi and ta_fm only
    Kf.hpi = in.hpi;
    Kf.ta_fm = in.ta_fm;
    bool fresh_day = false;
    side = 0;
    fb_out = 0;
    const long long t_end = in.t_end;
    X.halt = (t_end >= 0 && S.time >= t_end) ? 1 : 0;   // the run is over (roger/roger.py:548): nothing is formed, S stays as it is
    X.last = 0;
    if (X.halt) return;
    X.forc_exhausted = 0;
    if (do_hooks) {   // hooks_set_forcing: benchmarks/SVAT_benchmark.py:105-110, 151-171
        const bool midnight = (S.time % 86400 == 0);
        const int64_t i0 = S.itt_forc;
        const bool have = midnight && (i0 + RH_SLOTS_PER_DAY <= in.nitt_forc);
        X.forc_exhausted = (midnight && !have) ? 1 : 0;
        if (have) {
            for (int k = lane; k < 3 * RH_SLOTS_PER_DAY; k += 64) {
                const int v = k / RH_SLOTS_PER_DAY, j = k % RH_SLOTS_PER_DAY;
                const double x = D->series[v][i0 + j];
                L.f[v][j] = x;
                D->forc[v][j] = x;
            }
            S.itt_day = 0;
            S.year[1] = D->calendar[0][i0];
            S.month[1] = D->calendar[1][i0];
            S.doy[1] = D->calendar[2][i0];
            S.itt_forc = i0 + RH_SLOTS_PER_DAY;
            side |= RH_SIDE_PER_CELL_SET | (D->weights[0] ? RH_SIDE_PER_CELL : 0);
            fresh_day = true;
        }
        side |= RH_SIDE_MONTHLY_SET | (((S.month[1] != S.month[0]) && (S.itt > 1)) ? RH_SIDE_MONTHLY : 0);
    }
    const int64_t itd = S.itt_day;
    if (!fresh_day && in.cache_ok) {
        // inside a day whose bits and daily aggregates an earlier control part has formed: the six slots of the hourly window, out of the
        // registers (forcing_aggregates_of: the sums in numpy's order, as the full path forms them)
        RH_TPH(1)
        fb_out = in.day_fb;
        double agg[9];
        // (copies by value: a closure holding a reference to `in` keeps the whole struct in scratch memory)
        const double p0 = in.f[0][0], p1 = in.f[0][1], p2 = in.f[0][2], t0 = in.f[1][0], t1 = in.f[1][1], t2 = in.f[1][2];
        const double e0 = in.f[2][0], e1 = in.f[2][1], e2 = in.f[2][2];
        forcing_aggregates_of([=](int k) { return ctrl_slot(p0, p1, p2, k); }, [=](int k) { return ctrl_slot(t0, t1, t2, k); },
                              [=](int k) { return ctrl_slot(e0, e1, e2, k); }, itd, agg, false, true);
        X.agg[0] = in.day_agg[0]; X.agg[1] = in.day_agg[1]; X.agg[2] = in.day_agg[2];
        X.agg[3] = agg[3]; X.agg[4] = agg[4]; X.agg[5] = agg[5];
        X.agg[6] = agg[6]; X.agg[7] = agg[7]; X.agg[8] = agg[8];
        RH_TPH(2)
    } else {
        if (!fresh_day) {
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (lane + 64 * r < RH_SLOTS_PER_DAY) L.f[v][lane + 64 * r] = in.f[v][r];
        }
        wave_sync();
        RH_TPH(1)
        // word 0: the columns' bits 0..3 and the predicates of the day's series (adaptive_time_stepping.py:38-81)
        unsigned long long fb = 0;
        int cnt_d = 0, cnt_h = 0;
        for (int k = lane; k < RH_SLOTS_PER_DAY; k += 64) fb |= forcing_bits(L.f[0][k], L.f[1][k], Kf);
        for (int k0 = 0; k0 < RH_SLOTS_PER_DAY; k0 += 64) {   // nanmean's divisors
            const int k = k0 + lane;
            const bool ok = k < RH_SLOTS_PER_DAY && !isnan(L.f[1][k < RH_SLOTS_PER_DAY ? k : 0]);
            cnt_d += __popcll(__ballot(ok));
            cnt_h += __popcll(__ballot(ok && (k >= itd) && (k < itd + 6)));
        }
        fb = wave_or(fb);
        fb_out = fb;
        // numpy's partial sums (agg_body): six sums x 16 (half, lane-of-eight) pairs
        for (int item = lane; item < 96; item += 64) {
            const int sum_id = item >> 4, l16 = item & 15, half = l16 >> 3, j = l16 & 7;
            const int var = sum_id % 3;          // 0 prec, 1 ta, 2 pet
            const bool hourly = sum_id >= 3;     // sums 0..2 daily, 3..5 hourly window
            double r = 0.0;
            for (int q = 0; q < 9; ++q) {
                const int k = half * 72 + j + 8 * q;
                double v = L.f[var][k];
                const bool inw = !hourly || ((k >= itd) && (k < itd + 6));
                if (var == 1) v = (inw && !isnan(v)) ? v : 0.0;  // nanmean: NaN (and masked) slots count as 0
                else v = inw ? v : 0.0;
                r = (q == 0) ? v : r + v;
            }
            L.part[sum_id][l16] = r;
        }
        wave_sync();
        RH_TPH(2)
#define RH_SUM6(q) (0.0 + (np_tree8(&L.part[q][0]) + np_tree8(&L.part[q][8])))
        X.agg[0] = RH_SUM6(0);
        X.agg[1] = RH_SUM6(1) / (double)cnt_d;
        X.agg[2] = RH_SUM6(2);
        X.agg[3] = RH_SUM6(3);
        X.agg[4] = RH_SUM6(4) / (double)cnt_h;
        X.agg[5] = RH_SUM6(5);
#undef RH_SUM6
        int64_t k = itd < 0 ? itd + RH_SLOTS_PER_DAY : itd;
        k = k > RH_SLOTS_PER_DAY - 1 ? RH_SLOTS_PER_DAY - 1 : k;
        X.agg[6] = L.f[0][k];
        X.agg[7] = L.f[1][k];
        X.agg[8] = L.f[2][k];
        if (lane == 0) {   // the day's part, for the steps until somebody writes forc again
            D->day_fb = fb;
            D->day_agg[0] = X.agg[0]; D->day_agg[1] = X.agg[1]; D->day_agg[2] = X.agg[2];
            D->day_cache_ok = 1;
        }
    }
    X.cond_time = (S.time % 86400 == 0);
    X.itt_day = itd;
}
RH_DEV void ctrl_post(DevState *D, rh_scalars &S, StepCtx &X, unsigned long long cells, unsigned long long fb, double ta_fm, int64_t hpi_i, int64_t end_event,
                      long long t_end, int *dt_log, int dt_log_n, int dt_log_cap) {
#ifdef RH_STEP_PHASES
    unsigned long long tph = clock64();
#endif
    const int lane = threadIdx.x & 63;
    if (X.halt) return;
    Consts Kf;   // derive_word1 reads hpi and ta_fm only
    Kf.hpi = hpi_i;
    Kf.ta_fm = ta_fm;
    const unsigned long long w = (cells & 0xFull) | fb;
    {   // uniform from here on (every lane computes the same)
        const bool all_p_le0 = !bit(w, PB_P_NOT_LE0), any_p_gt0 = bit(w, PB_P_GT0), any_p_gthpi = bit(w, PB_P_GT_HPI);
        const bool all_p_lehpi = !bit(w, PB_P_NOT_LE_HPI), all_ta_gt = !bit(w, PB_TA_NOT_GT), any_ta_gt = bit(w, PB_TA_GT);
        const bool any_pgt0_tale = bit(w, PB_PGT0_TALE), all_ple0_tale = !bit(w, PB_NOT_PLE0_TALE);
        const bool all_swe_le0 = !bit(w, PB_SWE_NOT_LE0), all_swetop_le0 = !bit(w, PB_SWETOP_NOT_LE0);
        const bool snow_any = (bit(w, PB_SWE_GT0) || bit(w, PB_SWETOP_GT0)) && any_ta_gt;
        const bool cond0 = all_p_le0 && all_swe_le0 && all_swetop_le0 && all_ta_gt;
        const bool cond00 = any_pgt0_tale || all_ple0_tale;
        const bool cond1 = any_p_gthpi && any_p_gt0 && any_ta_gt;
        const bool cond2 = all_p_lehpi && any_p_gt0 && any_ta_gt;
        const bool cond3 = any_p_gthpi && any_p_gt0 && snow_any;
        const bool cond4 = all_p_lehpi && any_p_gt0 && snow_any;
        const bool cond5 = all_p_le0 && snow_any;
        X.sel_daily = cond0 || cond00;
        X.sel_hourly = (cond2 || cond4 || cond5) && !cond1 && !cond3;
        X.sel_10min = (cond1 || cond3) && !cond2 && !cond4 && !cond5;
        int64_t dts = X.cond_time ? 86400 : 3600;   // :143-144 (the second assignment overwrites the first), :166, :190
        if (X.sel_hourly) dts = 3600;
        if (X.sel_10min) dts = 600;
        X.dt_secs_prelim = dts;
        X.sel_p = X.sel_10min ? 2 : (X.sel_hourly ? 1 : (X.sel_daily ? 0 : -1));
        if (X.sel_p >= 0) {
            X.prec_sel = agg_pick(X, X.sel_p, 0);
            X.ta_sel = agg_pick(X, X.sel_p, 1);
        }
    }
    RH_TPH(3)
    const int64_t dts = scalars_update(S, X, derive_word1(cells, X, Kf), 1, 1, false, end_event);
    X.last = (t_end >= 0 && S.time >= t_end) ? 1 : 0;   // (S.time is the END of the step that is being formed)
    RH_TPH(4)
    if (lane == 0) {
        D->words[0] = 0;
        D->words[1] = 0;
        D->words[2] = 0;
        if (dt_log) {   // log_dt on values requested with the tail's other loads
            if (dt_log_n < dt_log_cap) dt_log[dt_log_n] = (int)dts;
            D->dt_log_n = dt_log_n + 1;
        }
    }
    RH_TPH(5)
}
// both halves in one wavefront (the control kernel; a fused launch without the extra workgroup)
RH_DEV void ctrl_wave(DevState *D, CtrlLds &L, rh_scalars &S, StepCtx &X, unsigned long long cells, int do_hooks, const CtrlIn &in) {
    unsigned long long fb;
    int side;
    ctrl_pre(D, L, S, X, do_hooks, in, fb, side);
    if ((threadIdx.x & 63) == 0) ctrl_side_effects(D, side);
    ctrl_post(D, S, X, cells, fb, in.ta_fm, in.hpi, in.end_event, in.t_end, in.dt_log, in.dt_log_n, in.dt_log_cap);
}
RH_DEV unsigned long long dev_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RH_DEV void dev_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// What pre_tail hands to the tail: S after the hook, X with the aggregates, the day's forcing bits, the deferred writes (RH_SIDE_*) -- one
// 8-byte word per lane (WI: integer fields, WF: doubles), packed by a select chain and taken apart by v_readlane with constant lanes:
// the structs never exist in memory on either side (as aggregates through LDS they cost both kernels a scratch frame).
#define RH_SX_FIELDS(WI, WF)                                                                                                         \
    WI(S.itt) WI(S.time) WI(S.dt_secs) WI(S.itt_day) WI(S.itt_forc) WI(S.time_event0) WI(S.event_id_counter)                         \
    WI(S.event_id[0]) WI(S.event_id[1]) WI(S.year[0]) WI(S.year[1]) WI(S.month[0]) WI(S.month[1]) WI(S.doy[0]) WI(S.doy[1])          \
    WF(S.dt) WI(S.sanity_ok)                                                                                                         \
    WF(X.dt) WF(X.agg[0]) WF(X.agg[1]) WF(X.agg[2]) WF(X.agg[3]) WF(X.agg[4]) WF(X.agg[5]) WF(X.agg[6]) WF(X.agg[7]) WF(X.agg[8])    \
    WI(X.month_tau) WI(X.sel_daily) WI(X.sel_hourly) WI(X.sel_10min) WI(X.sel_p) WF(X.prec_sel) WF(X.ta_sel) WI(X.sel_w)             \
    WF(X.pet_sel_w) WF(X.ta_sel_w) WI(X.cond1) WI(X.cond2) WI(X.cond3) WI(X.cond4) WI(X.cond5) WI(X.cond_time)                       \
    WI(X.dt_secs_prelim) WI(X.itt_day) WI(X.apply_sel) WI(X.forc_exhausted) WI(X.halt) WI(X.last)
#define RH_PRE_FIELDS(WI, WF) RH_SX_FIELDS(WI, WF) WI(fb) WI(side)
static_assert(sizeof(rh_scalars) == 17 * 8, "RH_PRE_FIELDS lists every field of rh_scalars");
static_assert(sizeof(StepCtx) == 200, "RH_PRE_FIELDS lists every field of StepCtx");
RH_DEV unsigned long long lane_word(unsigned long long w, int l) {   // (l: a constant)
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)w, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(w >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
// The control kernel of a step that does not find S_next / X_next ready (first step, after the host touched planes or
// scalars, multi-GPU after the exchange): one wavefront.  src: where the columns' summary word is (RH_SRC_*); src64 != null:
// it arrives as 64 int32 (0 / 1) from the exchange between the ranks and is folded here.
__global__ __launch_bounds__(64) void k_ctrl(DevState *D, int do_hooks, int src, const int *src64, int use_pre = 0) {
    __shared__ CtrlLds L;
    const int lane = threadIdx.x & 63;
    const CtrlIn in = ctrl_inputs(D);
    const unsigned long long pw = use_pre ? D->pre_words[lane] : 0ull;   // (written by the launch in front: a plain load)
    unsigned long long cells;
    if (src64) {
        cells = wave_or(src64[lane] ? (1ull << lane) : 0ull);
    } else if (src == RH_SRC_SUMW) {
        cells = wave_or(dev_load(&D->sumw[lane * RH_WSTRIDE]));
        dev_store(&D->sumw[lane * RH_WSTRIDE], 0ull);
    } else {
        cells = D->words[3];
    }
    rh_scalars S;
    StepCtx X;
    if (use_pre) {
        // the columns-independent half was formed by pre_tail of the fused launch in front of the exchange (from the S / X this kernel
        // would read): the decisions on the exchanged word are left
        unsigned long long fb = 0;
        int side = 0;
        int k = 0;
#define RH_WI(f) f = (std::remove_reference_t<decltype((f))>)(long long)lane_word(pw, k); ++k;
#define RH_WF(f) f = __longlong_as_double((long long)lane_word(pw, k)); ++k;
        RH_PRE_FIELDS(RH_WI, RH_WF)
#undef RH_WI
#undef RH_WF
        if (lane == 0) ctrl_side_effects(D, side);
        ctrl_post(D, S, X, cells, fb, in.ta_fm, in.hpi, in.end_event, in.t_end, in.dt_log, in.dt_log_n, in.dt_log_cap);
    } else {
        S = D->S;
        X = D->X;
        ctrl_wave(D, L, S, X, cells, do_hooks, in);
    }
    if (lane == 0) {
        D->words[3] = cells;
        if (!X.halt) D->sanity_last = 0;
        D->S = S;
        D->X = X;
        if (X.forc_exhausted && !X.halt) D->err_flags |= RH_DEVERR_FORCING;
    }
}
// The tail of the fused kernel, run by the wavefront that finishes last: folds the summary words into words[3] (and, for the
// exchange between ranks, spreads them over 64 int32), latches the sanity word, commits S_next / X_next if the step ran on them,
// and forms the next step's S_next / X_next (the control part of the next step: no control kernel between two fused kernels).
// A workgroup reports itself done; true for the one that is the last of the grid.  Two levels of counters -- workgroup b counts into
// group b mod n_groups, a full group into the top counter --, every counter in a cache line of its own and neighbouring workgroups in
// different groups: device-scope atomics on ONE line are served one after the other (~ 25 ns each; 3 907 workgroups counting into the two
// lines of 62 packed counters kept a 50 us kernel waiting for them, and the fused kernel's waves for their slot).  The counters reset
// themselves.  Called by one lane, after everything the workgroup posted through device-scope atomics has returned.
RH_DEV bool grid_completion(DevState *D, int n_groups) {
    const unsigned nblk = gridDim.x, ng = (unsigned)n_groups < nblk ? (unsigned)n_groups : nblk;
    const unsigned g = blockIdx.x % ng, cnt = nblk / ng + (g < nblk % ng ? 1u : 0u);
    unsigned int *c = &D->done_grp[g * RH_DONE_STRIDE];
    if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != cnt - 1) return false;
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ng == 1) return true;   // (small grids: one level -- a round trip less in a launch-bound step)
    if (__hip_atomic_fetch_add(&D->done_top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ng - 1) return false;
    __hip_atomic_store(&D->done_top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
// The first wavefront of a fused launch's extra workgroup (RH_TAIL_PRE), at the START of the launch: commits the running step's S / X
// (nothing of the launch reads D->S / D->X: the columns work from X_next), forms the columns-independent half of the next control part
// and publishes it through returning device-scope atomics (the tail reads it with device-scope loads: sumw's scheme).  `dep`: the wave's
// completion count, made to depend on the atomics' return.
RH_DEV void pre_tail(DevState *D, CtrlLds &L, int flags, unsigned &dep) {
    const int lane = threadIdx.x & 63;
    const CtrlIn in = ctrl_inputs(D);
    const bool use_next = (flags & RH_TAIL_USE_NEXT) != 0;
    rh_scalars S = *(use_next ? &D->S_next : &D->S);   // (one load site: two would leave a copy of the structs in scratch memory)
    StepCtx X = *(use_next ? &D->X_next : &D->X);
    if (use_next && lane == 0) {
        D->S = S;
        D->X = X;
        if (X.forc_exhausted) D->err_flags |= RH_DEVERR_FORCING;
    }
    unsigned long long fb;
    int side;
    ctrl_pre(D, L, S, X, (flags & RH_TAIL_HOOKS) != 0, in, fb, side);
    unsigned long long mine = 0;
    {
        int k = 0;
#define RH_WI(f) if (lane == k) mine = (unsigned long long)(long long)(f); ++k;
#define RH_WF(f) if (lane == k) mine = (unsigned long long)__double_as_longlong(f); ++k;
        RH_PRE_FIELDS(RH_WI, RH_WF)
#undef RH_WI
#undef RH_WF
    }
    const unsigned long long old = __hip_atomic_exchange(&D->pre_words[lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" : "+v"(dep) : "v"((unsigned)old));   // (dep is used only after the exchange has returned -- for every lane: one instruction)
}
RH_DEV void step_tail(DevState *D, CtrlLds &L, int flags, int *dst64) {
    const int lane = threadIdx.x & 63;
    const bool pre = (flags & RH_TAIL_PRE) && (flags & RH_TAIL_CTRL);   // (RH_TAIL_PRE alone: the control kernel behind the exchange takes the hand-over)
    CtrlIn in;
    if (!pre) in = ctrl_inputs(D);   // (requested first: one round trip for everything the tail reads)
#ifdef RH_STEP_PHASES
    const unsigned long long tph0 = clock64();
#endif
    // every load of the tail is requested before the first result is used (one round trip)
    const unsigned long long sw = dev_load(&D->sumw[lane * RH_WSTRIDE]);
    const unsigned long long bad = dev_load(&D->words[2]);
    const unsigned long long pw = pre ? dev_load(&D->pre_words[lane]) : 0ull;
    const double ta_fm = D->K.ta_fm;
    const int64_t hpi = D->K.hpi, end_event = D->K.end_event;
    const long long t_end = D->t_end;
    int *const dt_log = D->dt_log;
    const int dt_log_n = D->dt_log_n, dt_log_cap = D->dt_log_cap;
    rh_scalars S;
    StepCtx X;
    unsigned long long fb = 0;
    int side = 0;
    if (!pre) {
        if (flags & RH_TAIL_USE_NEXT) {
            S = D->S_next;
            X = D->X_next;
        } else {
            S = D->S;
            X = D->X;
        }
    }
    const unsigned long long cells = wave_or(sw);
    dev_store(&D->sumw[lane * RH_WSTRIDE], 0ull);
    if (pre) {   // pre_tail has done the commit and its half of the control part
        int k = 0;
#define RH_WI(f) f = (std::remove_reference_t<decltype((f))>)(long long)lane_word(pw, k); ++k;
#define RH_WF(f) f = __longlong_as_double((long long)lane_word(pw, k)); ++k;
        RH_PRE_FIELDS(RH_WI, RH_WF)
#undef RH_WI
#undef RH_WF
    }
    if (lane == 0) {
        D->words[3] = cells;
        D->sanity_last = bad;
        if (!pre && (flags & RH_TAIL_USE_NEXT)) {
            D->S = S;
            D->X = X;
            if (X.forc_exhausted) D->err_flags |= RH_DEVERR_FORCING;
        }
    }
    if (dst64) dst64[lane] = (int)((cells >> lane) & 1ull);
    if (!(flags & RH_TAIL_CTRL)) return;
#ifdef RH_STEP_PHASES
    if (lane == 0) {
        atomicAdd(&g_tail_phases[0], clock64() - tph0 + (S.time & 0) + (unsigned long long)(X.halt & 0));   // (loads of S / X used)
        atomicAdd(&g_tail_phases[7], 1ull);
    }
#endif
    if (pre) {
        if (lane == 0) ctrl_side_effects(D, side);
        ctrl_post(D, S, X, cells, fb, ta_fm, hpi, end_event, t_end, dt_log, dt_log_n, dt_log_cap);
    } else {
        ctrl_wave(D, L, S, X, cells, (flags & RH_TAIL_HOOKS) != 0, in);
    }
    if (lane == 0) {
        D->S_next = S;
        D->X_next = X;
    }
#ifdef RH_STEP_PHASES
    __threadfence();
    if (lane == 0) atomicAdd(&g_tail_phases[6], clock64() - tph0);
#endif
}
// ---- per-cell forcing: ONE per-column launch in front of the fused kernel (round 4) --------------------------------
// What k_pred1 -> k_agg -> k_cell_agg<1> -> k_select -> k_scalars did in five launches (four of them passes over the columns or
// single-workgroup reductions waiting for each other).  The two global decisions of a step depend on each other -- word 0 (snow state +
// the day's forcing over all columns) decides which aggregate a column takes as its prec / ta, word 1 is formed from THOSE --, so a
// column evaluates word 1's terms for every candidate selection (keep, daily, hourly, ten minutes: 6 bits each) next to its
// aggregates; the wavefront that finishes last folds the words, takes the candidate word 0 selects and does the bookkeeping
// (agg_body's decisions + scalars_update, the same device functions).  The day's forcing bits are formed once a day by the part that
// forms the daily sums and kept in day_word.
enum { FC_RAIN = 0, FC_SNOWMELT, FC_PREC_NOT_LE0, FC_NOT_PGT0_TALE, FC_P_EQ0, FC_P_NE0, FC_PER_CANDIDATE };
#define FC_COMMON 24   // bits 24..: PC_SWEM1_GT0, PC_SWE_NOT_LE0, PC_PM1_NE0, PC_PM1_EQ0, then word 0's four snow bits
RH_DEV unsigned long long front_candidate_bits(double prec, double ta, bool snow, double ta_fm) {
    unsigned long long b = 0;
    const bool warm = ta > ta_fm;
    b |= ((prec > 0) && warm) ? BIT(FC_RAIN) : 0;
    b |= (snow && warm) ? BIT(FC_SNOWMELT) : 0;
    b |= !(prec <= 0) ? BIT(FC_PREC_NOT_LE0) : 0;
    b |= !((prec > 0) && (ta <= ta_fm)) ? BIT(FC_NOT_PGT0_TALE) : 0;
    b |= (prec == 0) ? BIT(FC_P_EQ0) : 0;
    b |= (prec != 0) ? BIT(FC_P_NE0) : 0;
    return b;
}
// the daily part of a column: its daily sums (aggregate planes 0..2) and the forcing bits of its 144 slots
template <class P, class T, class E>
RH_DEV unsigned long long front_daily(P p, T t, E e, double *agg, const Consts &K) {
    forcing_aggregates_of(p, t, e, 0, agg, true, false);
    unsigned long long b = 0;
    for (int k = 0; k < RH_SLOTS_PER_DAY; ++k) b |= forcing_bits(p(k), t(k), K);
    return b;
}
// the control part by the last wavefront: S / X as k_agg's thread 0 and k_scalars form them
// fd: the set_forcing hook rode along with the kernel (fresh_day) -- what it does to the state happens here, once
RH_DEV void front_ctrl(DevState *D, bool daily_due, const FreshDay &fd, int hooks) {
    const int lane = threadIdx.x & 63;
    // everything this wavefront reads is requested before the first result is used (one round trip, as in the fused kernel's tail)
    const unsigned long long fw = dev_load(&D->frontw[lane * RH_WSTRIDE]);
    const unsigned long long dw = daily_due ? dev_load(&D->dayw[lane * RH_WSTRIDE]) : 0ull;
    unsigned long long day = D->day_word;
    const int64_t end_event = D->K.end_event;
    int *const dt_log = D->dt_log;
    const int dt_log_n = D->dt_log_n, dt_log_cap = D->dt_log_cap;
    const unsigned long long cells = wave_or(fw);
    dev_store(&D->frontw[lane * RH_WSTRIDE], 0ull);
    if (daily_due) {
        day = wave_or(dw);
        dev_store(&D->dayw[lane * RH_WSTRIDE], 0ull);
    }
    if (fd.fresh) {   // hooks_set_forcing: the day of the resident series becomes the current one for everybody behind this kernel
        for (int k = lane; k < 3 * RH_SLOTS_PER_DAY; k += 64) D->forc[k / RH_SLOTS_PER_DAY][k % RH_SLOTS_PER_DAY] = D->series[k / RH_SLOTS_PER_DAY][fd.i0 + k % RH_SLOTS_PER_DAY];
        if (lane == 0) D->day_cache_ok = 0;
        const int ns = D->n_stations;
        for (int q = lane; q < 3 * ns * RH_SLOTS_PER_DAY; q += 64) {
            const int v = q / (ns * RH_SLOTS_PER_DAY), r = q % (ns * RH_SLOTS_PER_DAY), st = r / RH_SLOTS_PER_DAY, j = r % RH_SLOTS_PER_DAY;
            D->forc_multi[q] = D->series[v][(size_t)st * D->nitt_forc + fd.i0 + j];
        }
    }
    if (lane != 0) return;
    D->day_word = day;
    rh_scalars S = D->S;   // (hoisted above the folds these two structs end up in scratch memory)
    StepCtx X = D->X;
    if (hooks) {
        if (fd.fresh) {
            S.itt_day = 0;
            S.year[1] = D->calendar[0][fd.i0];
            S.month[1] = D->calendar[1][fd.i0];
            S.doy[1] = D->calendar[2][fd.i0];
            S.itt_forc = fd.i0 + RH_SLOTS_PER_DAY;
            D->per_cell = D->weights[0] ? 1 : 0;
        }
        D->monthly = (S.month[1] != S.month[0]) && (S.itt > 1);
        if (fd.missing) D->err_flags |= RH_DEVERR_FORCING;
    }
    const unsigned long long w = ((cells >> (FC_COMMON + 4)) & 0xFull) | day;   // word 0: bits 0..3 are the columns' snow bits
    {
        const bool all_p_le0 = !bit(w, PB_P_NOT_LE0), any_p_gt0 = bit(w, PB_P_GT0), any_p_gthpi = bit(w, PB_P_GT_HPI);
        const bool all_p_lehpi = !bit(w, PB_P_NOT_LE_HPI), all_ta_gt = !bit(w, PB_TA_NOT_GT), any_ta_gt = bit(w, PB_TA_GT);
        const bool any_pgt0_tale = bit(w, PB_PGT0_TALE), all_ple0_tale = !bit(w, PB_NOT_PLE0_TALE);
        const bool all_swe_le0 = !bit(w, PB_SWE_NOT_LE0), all_swetop_le0 = !bit(w, PB_SWETOP_NOT_LE0);
        const bool snow_any = (bit(w, PB_SWE_GT0) || bit(w, PB_SWETOP_GT0)) && any_ta_gt;
        const bool cond0 = all_p_le0 && all_swe_le0 && all_swetop_le0 && all_ta_gt;
        const bool cond00 = any_pgt0_tale || all_ple0_tale;
        const bool cond1 = any_p_gthpi && any_p_gt0 && any_ta_gt;
        const bool cond2 = all_p_lehpi && any_p_gt0 && any_ta_gt;
        const bool cond3 = any_p_gthpi && any_p_gt0 && snow_any;
        const bool cond4 = all_p_lehpi && any_p_gt0 && snow_any;
        const bool cond5 = all_p_le0 && snow_any;
        X.cond_time = (S.time % 86400 == 0);
        X.sel_daily = cond0 || cond00;
        X.sel_hourly = (cond2 || cond4 || cond5) && !cond1 && !cond3;
        X.sel_10min = (cond1 || cond3) && !cond2 && !cond4 && !cond5;
        int64_t dts = X.cond_time ? 86400 : 3600;   // adaptive_time_stepping.py:143-144, :166, :190 (agg_body)
        if (X.sel_hourly) dts = 3600;
        if (X.sel_10min) dts = 600;
        X.dt_secs_prelim = dts;
        X.itt_day = S.itt_day;
        X.sel_p = X.sel_10min ? 2 : (X.sel_hourly ? 1 : (X.sel_daily ? 0 : -1));
    }
    // word 1 from the candidate the selection takes (candidate 0: keep; 1 + sel_p otherwise) and the common terms
    const unsigned long long cand = (cells >> (FC_PER_CANDIDATE * (X.sel_p + 1))) & ((1ull << FC_PER_CANDIDATE) - 1);
    unsigned long long w1 = 0;
    w1 |= bit(cand, FC_RAIN) ? BIT(PC_RAIN) : 0;
    w1 |= bit(cand, FC_SNOWMELT) ? BIT(PC_SNOWMELT) : 0;
    w1 |= bit(cand, FC_PREC_NOT_LE0) ? BIT(PC_PREC_NOT_LE0) : 0;
    w1 |= bit(cand, FC_NOT_PGT0_TALE) ? BIT(PC_NOT_PGT0_TALE) : 0;
    w1 |= bit(cand, FC_P_EQ0) ? BIT(PC_P_EQ0) : 0;
    w1 |= bit(cand, FC_P_NE0) ? BIT(PC_P_NE0) : 0;
    w1 |= bit(cells, FC_COMMON + 0) ? BIT(PC_SWEM1_GT0) : 0;
    w1 |= bit(cells, FC_COMMON + 1) ? BIT(PC_SWE_NOT_LE0) : 0;
    w1 |= bit(cells, FC_COMMON + 2) ? BIT(PC_PM1_NE0) : 0;
    w1 |= bit(cells, FC_COMMON + 3) ? BIT(PC_PM1_EQ0) : 0;
    const int64_t dts = scalars_update(S, X, w1, 1, 2, true, end_event);
    D->words[0] = 0;
    D->words[1] = 0;
    D->words[2] = 0;
    D->sanity_last = 0;   // (the fused kernel behind this front has no tail: words[2] is the whole record of its step)
    D->S = S;
    D->X = X;
    if (dt_log) {   // log_dt
        if (dt_log_n < dt_log_cap) dt_log[dt_log_n] = (int)dts;
        D->dt_log_n = dt_log_n + 1;
    }
}
// PART 0: every step's part; 1: with the daily part inline (small grids: one launch); 2: the daily part alone (in front of PART 0 on large
// grids, returning at once unless it is due: the daily sums' code needs > 200 registers, which would leave every step's part two waves per
// SIMD).  m1_pending as k_select's RH_SELECT_M1_PENDING.  The column's planes, weights and daily sums are requested FIRST, so that their
// round trips to HBM run under the staging of the day and the window sums.
template <int PART>
__global__ __launch_bounds__(RH_BLOCK) void k_cell_front(Arena a, DevState *D, int force_daily, int m1_pending, int n_groups, int hooks) {
    constexpr bool daily_only = PART == 2, with_daily = PART != 0;
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    const bool weighted = D->weights[0] != nullptr;
    // hooks: the device-side set_forcing hook rides along -- at midnight the kernel works on the day the hook is about to bring, the last
    // wavefront does the hook's bookkeeping (the state is read, not written, until then: PART 2 in front of PART 0 decides the same)
    const FreshDay fd = fresh_day(D, hooks);
    const bool daily_due = !weighted || force_daily || fd.itd == 0;   // (uniform over the grid; k_cell_agg's rule)
    __shared__ unsigned wg_done;
    __shared__ DaySeries day;
    if (daily_only && !daily_due) return;
    const bool in = i < a.n;
    const int64_t ii = in ? i : a.n - 1;
    double prec = 0, ta = 0, swe = 0, swe_top = 0, prec_m1 = 0, swe_m1 = 0, pw = 0, toff = 0, ew = 0, day0 = 0, day1 = 0;
    if (!daily_only) {
        rh_ld(a, RH_P_prec, ii, prec);
        rh_ld(a, RH_P_ta, ii, ta);
        rh_ld(a, RH_P_swe, ii, swe);
        rh_ld(a, RH_P_swe_top, ii, swe_top);
        if (!m1_pending) {
            rh_ld(a, RH_P_prec_m1, ii, prec_m1);
            rh_ld(a, RH_P_swe_m1, ii, swe_m1);
        }
        if (!(with_daily && daily_due)) {   // the daily sums of the day: formed earlier
            day0 = D->agg_cell[ii];
            day1 = D->agg_cell[(size_t)a.n + ii];
        }
    }
    if (weighted) {
        pw = D->weights[0][ii];
        toff = D->weights[1][ii];
        ew = D->weights[2][ii];
    }
    if (threadIdx.x == 0) wg_done = 0;
    // the whole day in LDS only where all of it is walked (the daily part); every step's part reads its window's six slots where they lie
    const bool staged = weighted && with_daily && daily_due;
    if (staged) stage_day(D, day, &fd);
    else __syncthreads();
    const Consts K = D->K;
    const int64_t itd = fd.itd;
    unsigned long long b = 0, db = 0;
    if (in) {
        double agg[9];
        if (weighted) {
            const DayView F = day_view(D, day, i, &fd, staged);
            auto p = [&](int k) { return F(0, k) * pw; };
            auto t = [&](int k) { return F(1, k) + toff; };
            auto e = [&](int k) { return F(2, k) * ew; };
            if (with_daily && daily_due) db = front_daily(p, t, e, agg, K);
            if (!daily_only) forcing_aggregates_of(p, t, e, itd, agg, false, true);
        } else {
            const double *pp = D->forc_cell[0] + i, *tp = D->forc_cell[1] + i, *ep = D->forc_cell[2] + i;
            const size_t n = (size_t)a.n;
            auto p = [&](int k) { return pp[k * n]; };
            auto t = [&](int k) { return tp[k * n]; };
            auto e = [&](int k) { return ep[k * n]; };
            if (with_daily && daily_due) db = front_daily(p, t, e, agg, K);
            if (!daily_only) forcing_aggregates_of(p, t, e, itd, agg, false, true);
        }
        if (with_daily && daily_due)
            for (int k = 0; k < 3; ++k) D->agg_cell[(size_t)k * a.n + i] = agg[k];
        if (!daily_only) {
            for (int k = 3; k < 9; ++k) D->agg_cell[(size_t)k * a.n + i] = agg[k];
            if (!(with_daily && daily_due)) {
                agg[0] = day0;
                agg[1] = day1;
            }
            if (m1_pending) {
                prec_m1 = prec;
                swe_m1 = swe;
            }
            const bool snow = (swe > 0) || (swe_top > 0);
            b |= front_candidate_bits(prec, ta, snow, K.ta_fm);                                   // keep (sel_p < 0)
            b |= front_candidate_bits(agg[0], agg[1], snow, K.ta_fm) << FC_PER_CANDIDATE;        // daily
            b |= front_candidate_bits(agg[3], agg[4], snow, K.ta_fm) << (2 * FC_PER_CANDIDATE);  // hourly
            b |= front_candidate_bits(agg[6], agg[7], snow, K.ta_fm) << (3 * FC_PER_CANDIDATE);  // ten minutes
            b |= (swe_m1 > 0) ? BIT(FC_COMMON + 0) : 0;
            b |= !(swe <= 0) ? BIT(FC_COMMON + 1) : 0;
            b |= (prec_m1 != 0) ? BIT(FC_COMMON + 2) : 0;
            b |= (prec_m1 == 0) ? BIT(FC_COMMON + 3) : 0;
            b |= !(swe <= 0) ? BIT(FC_COMMON + 4 + PB_SWE_NOT_LE0) : 0;
            b |= (swe > 0) ? BIT(FC_COMMON + 4 + PB_SWE_GT0) : 0;
            b |= !(swe_top <= 0) ? BIT(FC_COMMON + 4 + PB_SWETOP_NOT_LE0) : 0;
            b |= (swe_top > 0) ? BIT(FC_COMMON + 4 + PB_SWETOP_GT0) : 0;
        }
    }
    // the wave's bits into the device-wide words (atomics that have RETURNED before the wave counts itself done: k_step's completion scheme)
    unsigned dep = 1;
    b = wave_or(b);
    db = wave_or(db);
    if ((threadIdx.x & 63) == 0) {
        if (db) dep |= (unsigned)(__hip_atomic_fetch_or(&D->dayw[(blockIdx.x & 63) * RH_WSTRIDE], db, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 63);
        if (b) dep |= (unsigned)(__hip_atomic_fetch_or(&D->frontw[(blockIdx.x & 63) * RH_WSTRIDE], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 63);
    }
    if (daily_only) return;   // (the front kernel behind this launch folds dayw)
    bool last = false;
    if ((threadIdx.x & 63) == 0) {
        const unsigned o = atomicAdd(&wg_done, dep);   // LDS; dep == 1 (bit 63 of the words is never set)
        if (o == (RH_BLOCK / 64) - 1) {
            last = grid_completion(D, n_groups);
        }
    }
    if (__shfl((int)last, 0)) {
        front_ctrl(D, daily_due, fd, hooks);
    }
}

// summary bits straight from the arena (first step, or after the host changed planes), OR-ed into sumw (zeroed by the host)
__global__ __launch_bounds__(RH_BLOCK) void k_summary(Arena a, DevState *D) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    unsigned long long b = 0;
    if (i < a.n) {
        double swe, swe_top, prec, ta;
        rh_ld(a, RH_P_swe, i, swe);
        rh_ld(a, RH_P_swe_top, i, swe_top);
        rh_ld(a, RH_P_prec, i, prec);
        rh_ld(a, RH_P_ta, i, ta);
        b = summary_bits(swe, swe_top, prec, ta, D->K);
    }
    b = wave_or(b);
    if ((threadIdx.x & 63) == 0 && b) __hip_atomic_fetch_or(&D->sumw[(blockIdx.x & 63) * RH_WSTRIDE], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Output accumulators: after a step that covered (t0, t1], day = t0 / 86400, slot = day mod diag_slots; the first
// step of a day (t0 on midnight) overwrites.  Rate planes add this step's value (Rate.diagnose, roger/diagnostics/
// rate.py:66-84: `rate += var[..., tau]`), collect planes keep the current one.  S.time / S.dt_secs were advanced by
// the control kernel before the fused kernel ran.
// after_fused: the launch follows a fused k_step, whose prologue has just written D->skipped (the step found the run over,
// rh_set_time_limit: nothing to add).  Behind the routine-by-routine step (rh_step_core) and the routed passes there is no such
// launch in front and the flag may be left over from an earlier device run under a limit: they pass 0.
__global__ __launch_bounds__(RH_BLOCK) void k_diag(Arena a, DevState *D, int after_fused) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n || (after_fused && D->skipped)) return;
    const int64_t t0 = D->S.time - D->S.dt_secs, iv = D->diag_interval;
    const int64_t slot = (t0 / iv) % D->diag_slots;
    const bool first = (t0 % iv) == 0;   // steps never straddle an interval boundary they do not start on (adaptive_time_stepping)
    const int nr = D->diag_rate, nv = D->diag_rate + D->diag_collect;
    double *base = D->diag + (size_t)slot * nv * a.n;
    if (i == 0) {
        long long *m = D->diag_steps + 3 * slot;
        m[0] = first ? 1 : m[0] + 1;
        if (first) m[1] = t0;
        m[2] = D->S.time;
    }
    for (int j = 0; j < nv; ++j) {
        double v;
        rh_ld(a, D->diag_planes[j], i, v);
        double *p = base + (size_t)j * a.n + i;
        *p = (j < nr && !first) ? *p + v : v;
    }
}
// multi-GPU: OR of the summary words into words[3] for the exchange
// dst64 != null: also spread over 64 int32 (0 / 1) for the MAX all-reduce (k_words_expand folded in)
__global__ __launch_bounds__(RH_BLOCK) void k_summary_reduce(DevState *D, int do_hooks, int *dst64, int src) {
    if (do_hooks) hooks_set_forcing(D);
    if (threadIdx.x < 64) {
        unsigned long long w;
        if (src == RH_SRC_SUMW) {
            w = wave_or(dev_load(&D->sumw[threadIdx.x * RH_WSTRIDE]));
            dev_store(&D->sumw[threadIdx.x * RH_WSTRIDE], 0ull);
        } else {
            w = D->words[3];
        }
        if (threadIdx.x == 0) D->words[3] = w;
        if (dst64) dst64[threadIdx.x] = (int)((w >> threadIdx.x) & 1ull);
    }
}

__global__ void k_export(const DevState *D, HostExport *H, unsigned long long seq) {
    H->S = D->S;
    H->bad = D->words[2];
    H->bad_last = D->sanity_last;
    H->err = D->err_flags;
    __threadfence_system();
    __hip_atomic_store(&H->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_advance(DevState *D) {  // roger.py:449-450
    D->S.itt += 1;
    D->S.time += D->S.dt_secs;
}
__global__ void k_rotate_scalars(DevState *D) {
    rh_scalars &S = D->S;
    S.event_id[0] = S.event_id[1];
    S.year[0] = S.year[1];
    S.month[0] = S.month[1];
    S.doy[0] = S.doy[1];
}
__global__ void k_sync_ctx(DevState *D) {
    D->X.dt = D->S.dt;
    D->X.month_tau = D->S.month[1];
    D->X.itt_day = D->S.itt_day;
}
__global__ void k_sanity_to_scalars(DevState *D) { D->S.sanity_ok = D->words[2] ? 0 : 1; }
// Multi-GPU: NCCL/RCCL has no bitwise-OR reduction, so a predicate word is spread over 64 int32
// (0/1) for a MAX all-reduce and folded back afterwards.
__global__ void k_words_expand(DevState *D, int w, int *dst) { dst[threadIdx.x] = (int)((D->words[w] >> threadIdx.x) & 1ull); }
__global__ void k_words_compress(DevState *D, int w, const int *src) {
    unsigned long long b = src[threadIdx.x] ? (1ull << threadIdx.x) : 0ull;
    for (int off = 32; off; off >>= 1) b |= __shfl_xor(b, off);
    if (threadIdx.x == 0) D->words[w] = b;
}
__global__ void k_zero_words(DevState *D) { D->words[0] = D->words[1] = D->words[2] = D->words[3] = 0; }

// ---------------------------------------------------------------------------------------------
// per-column kernels
// ---------------------------------------------------------------------------------------------
#define LD(name) rh_ld(a, RH_P_##name, i, c.name);
#define ST(name) rh_st(a, RH_P_##name, i, c.name);

// ---- parameter planes: uniform over a wave -> one element; derivable -> not loaded at all (DevState::pmask) ----
constexpr int rh_param_bit(int plane) {
    switch (plane) {
#define RH_PB(name, bit) case RH_P_##name: return bit;
        RH_PARAM_BITS(RH_PB)
#undef RH_PB
        default: return -1;
    }
}
constexpr bool rh_param_derived(int plane) {
    switch (plane) {
#define RH_PD(name) case RH_P_##name:
        RH_DERIVED_FIELDS(RH_PD)
#undef RH_PD
        return true;
        default: return false;
    }
}
#if defined(RH_CENSUS)   // tools/isa_census.py: the wave's word as a compile-time constant (RH_CENSUS_PMASK), so that the count is of ONE path
#ifndef RH_CENSUS_PMASK
#define RH_CENSUS_PMASK 0ull
#endif
#endif
// Load of plane PLANE for the LAZY kernels: a parameter plane whose bit is set in the wave's word is read at the wave's FIRST column by
// every lane (one 64-byte sector from HBM instead of 512 bytes: the values are equal, k_param_mask compared them bit for bit); a
// derived parameter is not loaded when bit 63 is set (the stage's rd_* function assigns it).
// MK1: the wave's columns all lie in the catchment (bit 62: maskCatch == 1 on every one): the mask is the constant 1, not loaded, and
// the ~ 350 multiplications by it per column and step (the reference masks every assignment) fold away -- x * 1.0 is x, bit for bit.
template <int PLANE, bool MK1, typename T>
RH_DEV void rh_ld_p(const Arena &a, int64_t i, T &dst, unsigned long long um) {
    constexpr int bit = rh_param_bit(PLANE);
    if constexpr (MK1 && PLANE == RH_P_maskCatch) {
        dst = 1;
    } else if constexpr (bit < 0) {
        rh_ld(a, PLANE, i, dst);
    } else {
        if constexpr (rh_param_derived(PLANE)) {
            if (um >> 63) return;
        }
        const bool uni = (um >> bit) & 1ull;
#if RH_TILED
        const int tile = __builtin_amdgcn_readfirstlane((int)(i >> RH_TILE_SHIFT));
        const int piece = __builtin_amdgcn_readfirstlane((int)(i & (RH_TILE_CELLS - 1)) & ~63);
        const T *p = reinterpret_cast<const T *>(a.base + (size_t)tile * a.stride + (size_t)PLANE * RH_SLOT_BYTES) + piece + (uni ? 0 : (int)(i & 63));
#else
        const T *p = reinterpret_cast<const T *>(a.base + (size_t)PLANE * a.stride) + (uni ? (i & ~(int64_t)63) : i);
#endif
        dst = (RH_NT & 1) ? __builtin_nontemporal_load(p) : *p;
    }
}
#define LDP(name) rh_ld_p<RH_P_##name, MK1>(a, i, c.name, um);
RH_DEV unsigned long long bits_of(double v) { return (unsigned long long)__double_as_longlong(v); }
RH_DEV unsigned long long bits_of(int v) { return (unsigned long long)(unsigned)v; }
// The wave's word of DevState::pmask.  flags bit 0: uniformity bits, bit 1: the derive bit.  One thread per column; a wave whose upper
// lanes lie beyond the grid compares its active lanes only (the uniform load reads the wave's first column, which exists).
__global__ __launch_bounds__(RH_BLOCK) void k_param_mask(Arena a, DevState *D, unsigned long long *out, int flags) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    const bool active = i < a.n;
    const int64_t ii = active ? i : a.n - 1;   // (a lane beyond the grid looks at the last column: it does not vote)
    if ((int64_t)blockIdx.x * RH_BLOCK + (threadIdx.x & ~63) >= a.n) return;   // the whole wave lies beyond the grid
    Col c;
    unsigned long long um = 0;
#define RH_PU(name, bit)                                                                                       \
    {                                                                                                          \
        rh_ld(a, RH_P_##name, ii, c.name);                                                                     \
        const unsigned long long mine = bits_of(c.name);                                                       \
        const unsigned long long first = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mine >> 32)) << 32) | \
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)mine);                  \
        if (__ballot(active && mine != first) == 0) um |= 1ull << bit;                                         \
    }
    RH_PARAM_BITS(RH_PU)
#undef RH_PU
    // a plane the monthly surface parameters assign stays uniform over the wave only if what they are computed from is
    unsigned long long in_bits = 0, mon_bits = 0;
#define RH_PB(name, bit) const unsigned long long pbit_##name = 1ull << bit;
    RH_PARAM_BITS(RH_PB)
#undef RH_PB
#define RH_PI(name) in_bits |= pbit_##name;
    RH_PARAM_MONTHLY_INPUTS(RH_PI)
#undef RH_PI
#define RH_PM(name) mon_bits |= pbit_##name;
    RH_PARAM_MONTHLY(RH_PM)
#undef RH_PM
    if ((um & in_bits) != in_bits) um &= ~mon_bits;
    const bool all_in = __ballot(active && c.maskCatch != 1) == 0;   // every column of the wave lies in the catchment
    if (!(flags & 1)) um = 0;
    if (all_in && (flags & 4)) um |= 1ull << 62;
    if (flags & 2) {
        // the derived parameters: what the stages would compute from the primaries (already in c) against what the planes hold
        Col d = c;
        rd_all(d, D->K);
        bool same = true;
#define RH_PD(name) same = same && bits_of(d.name) == bits_of(c.name);
        RH_DERIVED_FIELDS(RH_PD)
#undef RH_PD
        if (__ballot(active && !same) == 0) um |= 1ull << 63;
    }
    if ((threadIdx.x & 63) == 0) out[i >> 6] = um;
}
#define ROT(name) rh_st(a, RH_P_##name##_m1, i, c.name);  // tau -> taum1 copy of after_timestep, done early

// THE hot kernel.  Loads every plane the step reads once, runs the whole step in registers,
// stores every plane the step assigns once.
// The fused step runs as a pipeline of stages (sets generated per sequence by tools/gen_sets.py):
// every routine stores the planes it is the last to assign right away, and the planes the NEXT
// routine is the first to mention are requested before the current routine computes, so their
// latency hides behind its arithmetic.
// LAZY (template parameter of step_column): the previous operation was a complete step, so X_m1 == X for every rotation
// pair of after_timestep.  Then the X_m1 planes are neither loaded (the register is filled from X's, AL) nor stored;
// they are materialised from the X planes when somebody else needs them (materialise_m1).  Per column and step this
// saves the 30 rotation stores and the 11 X_m1 loads of the step: 328 of 1 986 bytes.
// SPARSE (with LAZY; every step of an rh_run_steps call that another step of the same call follows): the planes the step only
// PRODUCES -- fluxes and diagnostics that no step reads back, RH_SPARSE_FIELDS_* from the flow analysis of tools/liveness.py -- are
// not stored.  Nothing but the next step looks at the planes between two steps of one call, that step overwrites them as the
// reference's arrays are overwritten, and the call's last step stores everything: what the caller can observe is unchanged.
#define AL(xm1, x) c.xm1 = c.x;
#define RH_LOADS(seq, rt)                                          \
    if constexpr (LAZY) {                                          \
        RH_SEQ_##seq##_LLOAD_##rt(LDP) RH_SEQ_##seq##_ALIAS_##rt(AL) \
    } else {                                                       \
        RH_SEQ_##seq##_LOAD_##rt(LD)                               \
    }
// the stage's derived parameters (rh_physics.h rd_<stage>), where the wave's planes were found to hold exactly these values
// ... and the state every lazy step derives itself (rl_<stage>: k / h of root zone and subsoil from the previous step's water contents)
#define RH_DERIVE(rt)                        \
    if constexpr (LAZY) {                    \
        if (um >> 63) rd_##rt(c, K);         \
        rl_##rt(c, K);                       \
    }
#ifdef RH_CENSUS   // (tools/isa_census.py counts what the sparse kernel stores when no accumulator asks for more)
#define STK(name)
#else
#define STK(name) \
    if ((keepw[RH_P_##name >> 6] >> (RH_P_##name & 63)) & 1ull) rh_st(a, RH_P_##name, i, c.name);
#endif
#define RH_STORES(seq, rt)                                               \
    if constexpr (LAZY && SPARSE) {                                      \
        RH_SEQ_##seq##_SSTORE_##rt(ST)                                   \
        if constexpr (KEEP) { RH_SEQ_##seq##_KSTORE_##rt(STK) }          \
    } else if constexpr (LAZY) {                                         \
        RH_SEQ_##seq##_LSTORE_##rt(ST)                                   \
    } else {                                                             \
        RH_SEQ_##seq##_STORE_##rt(ST) RH_SEQ_##seq##_ROT_##rt(ROT)       \
    }
#if RH_STEP_PREFETCH
#if RH_STEP_PREFETCH >= 2
#define RH_PIN __builtin_amdgcn_sched_barrier(0);
#else
#define RH_PIN
#endif
#ifdef RH_STEP_PHASES   // measurement builds (tools/step_phases.sh): wave cycles per stage of the step, by class of step length
__device__ unsigned long long g_step_phases[256 * 64];   // 256 copies (one address would serialise the chip's atomics)
#define RH_PH(k)                                                                          \
    if ((threadIdx.x & 63) == 0) {                                                        \
        const unsigned long long t_ = clock64();                                          \
        atomicAdd(&g_step_phases[(blockIdx.x & 255) * 64 + ph_cls * 20 + (k)], t_ - ph_t);                          \
        ph_t = t_;                                                                        \
    }
#else
#define RH_PH(k)
#endif
#define RH_STEP_BODY(seq, mon_rt, MON_LOADS, MON_RUN, sub_rt, sub_call, ne_rt, ne_call, at_rt, at_call) \
    RH_LOADS(seq, rt_select_prec) RH_LOADS(seq, rt_select_pet) MON_LOADS RH_LOADS(seq, rt_interception) RH_PIN  \
    rt_select_prec(c, X, prec_s, ta_s); RH_STORES(seq, rt_select_prec)                                                \
    rt_select_pet(c, X, pet_v, ta_v); RH_STORES(seq, rt_select_pet) RH_PH(1)                                      \
    q = summary_bits_pt(c.prec, c.ta, K);                                                                \
    MON_RUN                                                                                              \
    RH_LOADS(seq, rt_evapotranspiration) RH_PIN                                                                 \
    rt_interception(c, K); RH_STORES(seq, rt_interception) RH_PH(2)                                               \
    RH_LOADS(seq, rt_snow) RH_PIN                                                                               \
    RH_DERIVE(rt_evapotranspiration) rt_evapotranspiration(c, K); RH_STORES(seq, rt_evapotranspiration) RH_PH(3)  \
    RH_LOADS(seq, rt_inf_events) RH_PIN                                                                         \
    rt_snow(c, K, X); RH_STORES(seq, rt_snow) RH_PH(4)                                                            \
    q = summary_bits_sw(q, c.swe, c.swe_top); if (post) post_summary(D, q, dep);                         \
    RH_LOADS(seq, rt_inf_matrix) RH_PIN                                                                         \
    rt_inf_events(c, K, X); RH_STORES(seq, rt_inf_events) RH_PH(5)                                                \
    RH_LOADS(seq, rt_inf_macropores) RH_PIN                                                                     \
    RH_DERIVE(rt_inf_matrix) rt_inf_matrix(c, K, X); RH_STORES(seq, rt_inf_matrix) RH_PH(6)                       \
    RH_LOADS(seq, rt_inf_cracks) RH_PIN                                                                         \
    RH_DERIVE(rt_inf_macropores) rt_inf_macropores(c, K, X); RH_STORES(seq, rt_inf_macropores) RH_PH(7)           \
    RH_LOADS(seq, rt_inf_finish) RH_PIN                                                                         \
    rt_inf_cracks(c, K, X); RH_STORES(seq, rt_inf_cracks) RH_PH(8)                                                \
    RH_LOADS(seq, sub_rt) RH_PIN                                                                                \
    rt_inf_finish(c, K, X); RH_STORES(seq, rt_inf_finish) RH_PH(9)                                                \
    RH_LOADS(seq, rt_capillary_rise) RH_PIN                                                                     \
    RH_DERIVE(sub_rt) sub_call; RH_STORES(seq, sub_rt) RH_PH(10)                                                   \
    RH_LOADS(seq, rt_storage) RH_PIN                                                                            \
    rt_capillary_rise(c, X); RH_STORES(seq, rt_capillary_rise) RH_PH(11)                                           \
    RH_LOADS(seq, ne_rt) RH_PIN                                                                                 \
    RH_DERIVE(rt_storage) rt_storage(c, X); RH_STORES(seq, rt_storage) RH_PH(12)                                   \
    RH_LOADS(seq, at_rt) RH_PIN                                                                                 \
    bad = ne_call; RH_STORES(seq, ne_rt) RH_PH(13)                                                                 \
    at_call; RH_STORES(seq, at_rt) RH_PH(14)
#else
#define RH_STAGE(seq, rt, call) RH_LOADS(seq, rt) call; RH_STORES(seq, rt)
#ifdef RH_NO_SELSTAGE  // timing experiments only
#define RH_DBG_SEL(x)
#else
#define RH_DBG_SEL(x) x
#endif
#ifdef RH_NO_SUMMARY
#define RH_DBG_SUM(x)
#else
#define RH_DBG_SUM(x) x
#endif
// q_pt / q_sw: the column's summary values for the next step's predicates, sampled where they are final
// (tools/gen_sets.py asserts that no later stage assigns them)
#define RH_STEP_BODY(seq, mon_rt, MON_LOADS, MON_RUN, sub_rt, sub_call, ne_rt, ne_call, at_rt, at_call) \
    RH_DBG_SEL(RH_STAGE(seq, rt_select_prec, rt_select_prec(c, X, prec_s, ta_s)))                                     \
    RH_STAGE(seq, rt_select_pet, rt_select_pet(c, X, pet_v, ta_v))                                       \
    RH_DBG_SUM(q = summary_bits_pt(c.prec, c.ta, K);)                                                    \
    MON_LOADS MON_RUN                                                                                    \
    RH_STAGE(seq, rt_interception, rt_interception(c, K))                                                \
    RH_STAGE(seq, rt_evapotranspiration, RH_DERIVE(rt_evapotranspiration) rt_evapotranspiration(c, K))   \
    RH_STAGE(seq, rt_snow, rt_snow(c, K, X))                                                             \
    RH_DBG_SUM(q = summary_bits_sw(q, c.swe, c.swe_top); if (post) post_summary(D, q, dep);)             \
    RH_STAGE(seq, rt_inf_events, rt_inf_events(c, K, X))                                                 \
    RH_STAGE(seq, rt_inf_matrix, RH_DERIVE(rt_inf_matrix) rt_inf_matrix(c, K, X))                        \
    RH_STAGE(seq, rt_inf_macropores, RH_DERIVE(rt_inf_macropores) rt_inf_macropores(c, K, X))            \
    RH_STAGE(seq, rt_inf_cracks, rt_inf_cracks(c, K, X))                                                 \
    RH_STAGE(seq, rt_inf_finish, rt_inf_finish(c, K, X))                                                 \
    RH_STAGE(seq, sub_rt, RH_DERIVE(sub_rt) sub_call)                                                    \
    RH_STAGE(seq, rt_capillary_rise, rt_capillary_rise(c, X))                                            \
    RH_STAGE(seq, rt_storage, RH_DERIVE(rt_storage) rt_storage(c, X))                                    \
    RH_STAGE(seq, ne_rt, bad = ne_call)                                                                  \
    RH_STAGE(seq, at_rt, at_call)
#endif

// A wavefront's summary bits into the device-wide words, as soon as they are final (right after the snow stage: the latency of
// the returning atomic hides behind the infiltration stages).  `dep` carries the returned value to the completion count at the end
// of the kernel, so that the wave is counted as done only after its bits have arrived.
RH_DEV void post_summary(DevState *D, unsigned long long q, unsigned &dep) {
    // (by ballots, not lane exchanges: the last wavefront of the grid may run with its upper lanes switched off)
    unsigned long long qq = 0;
#pragma unroll
    for (int b = 0; b <= QB_P_NE0; ++b) qq |= __ballot((q >> b) & 1ull) ? (1ull << b) : 0ull;
    if ((threadIdx.x & 63) == 0 && qq) {
        const unsigned long long old = __hip_atomic_fetch_or(&D->sumw[(blockIdx.x & 63) * RH_WSTRIDE], qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dep |= (unsigned)(old >> 63);   // (bit 63 is never set: dep stays as it is, but depends on the atomic's return)
    }
}

// the step of one column: loads, the staged pipeline, stores; q = summary bits of the column for the next step's predicates
// KEEP (with SPARSE): an accumulator was given planes the sparse kernel does not store -- those are stored after all (DevState::keep)
template <bool MONTHLY, bool LATERAL, bool LAZY, bool SPARSE, bool KEEP = false, bool MK1 = false>
RH_DEV void step_column(const Arena &a, DevState *D, const StepCtx *Xp, int64_t i, unsigned long long &q, bool &bad, unsigned &dep, unsigned long long um,
                        bool post) {
    {
    const Consts K = D->K;
    const StepCtx X = *Xp;
    Col c;
    double pet_v = X.pet_sel_w, ta_v = X.ta_sel_w;
    double prec_s = X.prec_sel, ta_s = X.ta_sel;   // the column's own when the per-cell selection was deferred to this kernel
    // KEEP: the words that say which pure-output planes an accumulator reads, read ONCE -- tested at the store sites out of D they were
    // loaded again behind every store (the stores may alias them, for all the compiler knows): 72 dependent trips per wavefront,
    // + 65 us per step at 10^6 columns (k_step<..., KEEP> 244 against 179 us; tools/experiments/diag_prof.sh)
    unsigned long long keepw[(RH_NPLANES + 63) / 64];
#pragma unroll
    for (int k = 0; k < (RH_NPLANES + 63) / 64; ++k) keepw[k] = KEEP ? D->keep[k] : 0ull;
#ifdef RH_STEP_PHASES
    const int ph_cls = X.dt < 0.5 ? 0 : (X.dt < 12 ? 1 : 2);
    unsigned long long ph_t = clock64();
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_step_phases[(blockIdx.x & 255) * 64 + ph_cls * 20 + 19], 1ull);
#endif
#ifndef RH_CENSUS   // (tools/isa_census.py counts the step with shared forcing: these four loads belong to the per-cell path only)
    if (D->per_cell && X.sel_w >= 0) {
        pet_v = cell_agg(D, a.n, i, 3 * X.sel_w + 2);
        ta_v = cell_agg(D, a.n, i, 3 * X.sel_w + 1);
    }
    if (X.apply_sel == 2 && D->per_cell && X.sel_p >= 0) {
        prec_s = cell_agg(D, a.n, i, 3 * X.sel_p);
        ta_s = cell_agg(D, a.n, i, 3 * X.sel_p + 1);
    }
#endif
    if (MONTHLY && LATERAL) {
        RH_STEP_BODY(step_lateral_monthly, rt_params_surface, RH_LOADS(step_lateral_monthly, rt_params_surface),
                     rt_params_surface(c, D->L, X); RH_STORES(step_lateral_monthly, rt_params_surface),
                     rt_subsurface_runoff_lateral, rt_subsurface_runoff_lateral(c, K, X), rt_num_error_lateral,
                     rt_num_error_lateral(c, K), rt_after_timestep_oned, rt_after_timestep_oned(c))
    } else if (LATERAL) {
        RH_STEP_BODY(step_lateral, , , , rt_subsurface_runoff_lateral, rt_subsurface_runoff_lateral(c, K, X),
                     rt_num_error_lateral, rt_num_error_lateral(c, K), rt_after_timestep_oned, rt_after_timestep_oned(c))
    } else if (MONTHLY) {
        RH_STEP_BODY(step_monthly, rt_params_surface, RH_LOADS(step_monthly, rt_params_surface),
                     rt_params_surface(c, D->L, X); RH_STORES(step_monthly, rt_params_surface), rt_subsurface_runoff,
                     rt_subsurface_runoff(c, X), rt_num_error, rt_num_error(c, K), rt_after_timestep, rt_after_timestep(c))
    } else {
        RH_STEP_BODY(step, , , , rt_subsurface_runoff, rt_subsurface_runoff(c, X), rt_num_error, rt_num_error(c, K),
                     rt_after_timestep, rt_after_timestep(c))
    }
    }
}

// MODE: 0 = the plain step, 1 = with the monthly surface parameters (calc_parameters_surface_kernel first), 2 = decided
// by the device-side month-change flag (rh_run_steps, rh_step_finish: one launch whatever the month does)
// flags: RH_TAIL_*; n_groups: completion groups (grid_completion); dst64: the summary word for the exchange between
// ranks, written by the tail (or null)
template <int MODE, bool LATERAL, bool LAZY, bool SPARSE = false, bool KEEP = false>
__global__ __launch_bounds__(RH_BLOCK, RH_STEP_WAVES) void k_step(Arena a, DevState *D, int flags, int n_groups, int *dst64) {
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2 and address-translation cache).  Mapping
    // workgroup b to the column block  (b mod 8) * blocks_per_xcd + b / 8  lets every XCD walk ONE contiguous eighth of
    // the arena instead of every XCD touching every page.  Never slower; on one box 6 - 11 % faster at 10^7 columns (21 GB
    // arena: 3.87 -> 3.44 .. 3.64 ms per step; oneD 4.38 -> 4.12 ms), on another box and up to 4 x 10^6 columns the same
    // (DESIGN.md section 5 on the speed levels of this kernel).  -DRH_XCD_ROUND_ROBIN: the plain mapping.
    // RH_TAIL_PRE: the last workgroup of the grid has no columns -- its first wavefront is pre_tail
    const unsigned nb = gridDim.x - ((flags & RH_TAIL_PRE) ? 1u : 0u);
    const bool extra = blockIdx.x >= nb;
#ifndef RH_XCD_ROUND_ROBIN
    const unsigned x = blockIdx.x & 7u, base_cnt = nb >> 3, rem = nb & 7u;
    const unsigned blk = x * base_cnt + (x < rem ? x : rem) + (blockIdx.x >> 3);
    const int64_t i = extra ? a.n : (int64_t)blk * RH_BLOCK + threadIdx.x;
#else
    const int64_t i = extra ? a.n : (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
#endif
    __shared__ unsigned wg_done;      // wavefronts of this workgroup that are through
    __shared__ CtrlLds tail_lds;      // scratch of the tail and of pre_tail (one wavefront each of the whole grid)
    const StepCtx *Xp = (flags & RH_TAIL_USE_NEXT) ? &D->X_next : &D->X;
    // rh_set_time_limit: the control part found the run over before this step (uniform over the grid).  Nothing runs, the tail
    // included: S_next / X_next keep saying so to every launch that follows.
#ifndef RH_CENSUS   // (tools/isa_census.py counts the per-column memory instructions of ONE pipeline: no halt prologue, no tail)
    const bool halted = Xp->halt != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        D->skipped = halted ? 1 : 0;
        if (halted) log_dt(D, 0);     // (timing: a launch that did nothing is logged with dt = 0)
    }
    if (halted) return;
#endif
    if (threadIdx.x == 0) wg_done = 0;
    __syncthreads();                  // (at the start, where all waves are in step; the kernel has no closing barrier)
    unsigned long long q = 0;
    bool bad = false;
    unsigned dep = 1;
    const bool post = !(flags & RH_TAIL_SKIP);
    if (i < a.n) {
        const bool monthly = MODE == 1 || (MODE == 2 && D->monthly != 0);
        // the wave's word of the parameter planes (uniform / derivable / all in the catchment; zero: plain loads) -- wave-uniform, in
        // scalar registers
        unsigned long long um = 0;
        if constexpr (LAZY) {
#ifdef RH_CENSUS
            um = RH_CENSUS_PMASK;
#else
            const unsigned long long *pm = D->pmask;
            const unsigned long long w = pm ? pm[__builtin_amdgcn_readfirstlane((int)(i >> 6))] : 0ull;
            um = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)w);
#endif
        }
        const bool mk1 = LAZY && ((um >> 62) & 1ull);   // (the monthly pipeline, once a month, keeps the generic code)
        // the SPARSE kernel holds the full-store pipeline too: the step that reaches the time limit is the run's last one
        // (X.last, decided by the control part on the device) and stores every plane -- a wave-uniform branch
#ifdef RH_CENSUS
        if (SPARSE) {
#else
        if (SPARSE && !Xp->last) {
#endif
            if (monthly) step_column<true, LATERAL, LAZY, SPARSE, KEEP>(a, D, Xp, i, q, bad, dep, um, post);
            else if (mk1) step_column<false, LATERAL, LAZY, SPARSE, KEEP, LAZY>(a, D, Xp, i, q, bad, dep, um, post);
            else step_column<false, LATERAL, LAZY, SPARSE, KEEP>(a, D, Xp, i, q, bad, dep, um, post);
        } else {
            // (the full-store pipeline keeps the generic code: with a third copy the full-store kernels spill registers)
            if (monthly) step_column<true, LATERAL, LAZY, false>(a, D, Xp, i, q, bad, dep, um, post);
            else step_column<false, LATERAL, LAZY, false>(a, D, Xp, i, q, bad, dep, um, post);
        }
    } else if (post) {
        post_summary(D, 0ull, dep);
    }
    // (behind the column pipeline in the kernel's text: in front of it, the allocator spilled 36 - 63 of the pipeline's registers.
    //  RH_X_NOPRE / RH_X_NOCOMPLETION / RH_X_NOTAIL: timing-only builds whose results are NOT valid -- profiles/r04_tail_phases.txt)
#if !defined(RH_CENSUS) && !defined(RH_X_NOPRE)
    if (__builtin_expect(extra && threadIdx.x < 64, 0)) pre_tail(D, tail_lds, flags, dep);
#endif
    const bool any_bad = __any(bad);
    if (!post) {   // RH_TAIL_SKIP: the sanity word is all anybody reads of this launch
        if (any_bad && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(&D->words[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // Completion, without a barrier and without fences (a release fence at device scope writes the XCD's L2 back): everything the
    // tail reads from other waves went through device-scope atomics that have RETURNED before the wave counts itself done.
    bool last = false;
    if ((threadIdx.x & 63) == 0) {
        if (any_bad) dep |= (unsigned)(__hip_atomic_fetch_or(&D->words[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 63);
        const unsigned o = atomicAdd(&wg_done, dep);   // LDS; dep == 1
        if (o == (RH_BLOCK / 64) - 1) {                // the last wave of the workgroup reports the workgroup
#ifndef RH_X_NOCOMPLETION
            last = grid_completion(D, n_groups);
#endif
        }
    }
#ifdef RH_X_NOTAIL
    last = false;
#endif
#ifndef RH_CENSUS   // tools/isa_census.py counts the per-column memory instructions of the kernel without its tail
    if (__shfl((int)last, 0)) step_tail(D, tail_lds, flags, dst64);
#endif
}

#ifdef RH_CENSUS   // tools/isa_census.py: the sparse variant of the non-monthly pipeline on its own (the product launches MODE 2 only)
template __global__ void k_step<0, false, true, true>(Arena, DevState *, int, int, int *);
template __global__ void k_step<0, true, true, true>(Arena, DevState *, int, int, int *);
#endif

#define RH_CELL_KERNEL(kname, rt, call)                                       \
    __global__ __launch_bounds__(RH_BLOCK) void kname(Arena a, DevState *D) { \
        const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;       \
        if (i >= a.n) return;                                                 \
        const Consts K = D->K;                                                \
        const StepCtx X = D->X;                                               \
        (void)K;                                                              \
        (void)X;                                                              \
        Col c;                                                                \
        RH_SET_LOAD_##rt(LD) call;                                            \
        RH_SET_STORE_##rt(ST)                                                 \
    }

RH_CELL_KERNEL(k_interception, rt_interception, rt_interception(c, K))
RH_CELL_KERNEL(k_evapotranspiration, rt_evapotranspiration, rt_evapotranspiration(c, K))
RH_CELL_KERNEL(k_snow, rt_snow, rt_snow(c, K, X))
RH_CELL_KERNEL(k_infiltration, rt_infiltration, rt_infiltration(c, K, X))
RH_CELL_KERNEL(k_subsurface_runoff, rt_subsurface_runoff, rt_subsurface_runoff(c, X))
RH_CELL_KERNEL(k_capillary_rise, rt_capillary_rise, rt_capillary_rise(c, X))
RH_CELL_KERNEL(k_storage, rt_storage, rt_storage(c, X))
RH_CELL_KERNEL(k_num_error, rt_num_error, if (rt_num_error(c, K)) atomicOr(&D->words[2], 1ull))
RH_CELL_KERNEL(k_after_timestep, rt_after_timestep, rt_after_timestep(c))
RH_CELL_KERNEL(k_step_core, rt_step_core, if (rt_step_core(c, K, X)) atomicOr(&D->words[2], 1ull))
// oneD model variants
RH_CELL_KERNEL(k_subsurface_runoff_lateral, rt_subsurface_runoff_lateral, rt_subsurface_runoff_lateral(c, K, X))
RH_CELL_KERNEL(k_num_error_lateral, rt_num_error_lateral, if (rt_num_error_lateral(c, K)) atomicOr(&D->words[2], 1ull))
RH_CELL_KERNEL(k_after_timestep_oned, rt_after_timestep_oned, rt_after_timestep_oned(c))
RH_CELL_KERNEL(k_step_core_lateral, rt_step_core_lateral, if (rt_step_core_lateral(c, K, X)) atomicOr(&D->words[2], 1ull))
RH_CELL_KERNEL(k_params_lateral, rt_params_lateral, rt_params_lateral(c, D->mlms, D->mlms_rows, D->max_slope_per))
// settings.enable_routing_1D: the per-column parts of the D8 routing (rh_physics.h) ...
RH_CELL_KERNEL(k_infiltration_routed, rt_infiltration_routed, rt_infiltration_routed(c, K, X))
RH_CELL_KERNEL(k_route_surface_out, rt_route_surface_out, rt_route_surface_out(c, K, X, (double)D->S.dt_secs))
RH_CELL_KERNEL(k_route_surface_in, rt_route_surface_in, rt_route_surface_in(c))
RH_CELL_KERNEL(k_route_subsurface_out, rt_route_subsurface_out, rt_route_subsurface_out(c))
RH_CELL_KERNEL(k_route_subsurface_in, rt_route_subsurface_in, rt_route_subsurface_in(c))
RH_CELL_KERNEL(k_num_error_routed, rt_num_error_routed, if (rt_num_error_routed(c, K)) atomicOr(&D->words[2], 1ull))
// the step core in three passes, one kernel each (the infiltration's branch conditions come from the adaptive time stepping's
// predicate word, as in k_step_core: global over the ranks)
// ... staged like the fused step (tools/gen_sets.py PLAIN_SEQUENCES): every plane is loaded right before the first stage that mentions it
// and stored right after the last one that assigns it (short live ranges instead of all loads up front)
#define RH_PSTAGE(seq, rt, call) RH_SEQ_##seq##_LOAD_##rt(LD) call; RH_SEQ_##seq##_STORE_##rt(ST)
// ... in a kernel with a template parameter SPARSE (the device-driven routed step inside rh_run_steps: every step of a call but the last
// leaves out the stores of the planes the routed step only produces and no later pass of the step loads, RH_SEQ_*_SSTORE_*, tools/gen_sets.py)
#define RH_PSTAGE_S(seq, rt, call)                            \
    RH_SEQ_##seq##_LOAD_##rt(LD) call;                        \
    if constexpr (SPARSE) { RH_SEQ_##seq##_SSTORE_##rt(ST) }  \
    else { RH_SEQ_##seq##_STORE_##rt(ST) }
#define RH_PASS_KERNEL(kname, body)                                                                        \
    __global__ __launch_bounds__(RH_BLOCK, RH_STEP_WAVES) void kname(Arena a, DevState *D) {               \
        const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;                                    \
        if (i >= a.n) return;                                                                              \
        const Consts K = D->K;                                                                             \
        const StepCtx X = D->X;                                                                            \
        Col c;                                                                                             \
        bool bad = false;                                                                                  \
        body                                                                                               \
        if (bad) atomicOr(&D->words[2], 1ull);                                                             \
    }
RH_PASS_KERNEL(k_routed_a,
               RH_PSTAGE(routed_a, rt_interception, rt_interception(c, K))
               RH_PSTAGE(routed_a, rt_evapotranspiration, rt_evapotranspiration(c, K))
               RH_PSTAGE(routed_a, rt_snow, rt_snow(c, K, X))
               RH_PSTAGE(routed_a, rt_inf_events, rt_inf_events(c, K, X))
               RH_PSTAGE(routed_a, rt_inf_matrix, rt_inf_matrix(c, K, X))
               RH_PSTAGE(routed_a, rt_inf_macropores, rt_inf_macropores(c, K, X))
               RH_PSTAGE(routed_a, rt_inf_cracks, rt_inf_cracks(c, K, X))
               RH_PSTAGE(routed_a, rt_inf_finish_routed, rt_inf_finish_routed(c, K, X))
               RH_PSTAGE(routed_a, rt_route_surface_out, rt_route_surface_out(c, K, X, (double)D->S.dt_secs)))
RH_PASS_KERNEL(k_routed_b,
               RH_PSTAGE(routed_b, rt_route_surface_in, rt_route_surface_in(c))
               RH_PSTAGE(routed_b, rt_subsurface_runoff_lateral, rt_subsurface_runoff_lateral(c, K, X))
               RH_PSTAGE(routed_b, rt_route_subsurface_out, rt_route_subsurface_out(c)))
RH_PASS_KERNEL(k_routed_c,
               RH_PSTAGE(routed_c, rt_route_subsurface_in, rt_route_subsurface_in(c))
               RH_PSTAGE(routed_c, rt_capillary_rise, rt_capillary_rise(c, X))
               RH_PSTAGE(routed_c, rt_storage, rt_storage(c, X))
               RH_PSTAGE(routed_c, rt_num_error_routed, bad = rt_num_error_routed(c, K)))
RH_PASS_KERNEL(k_routed_c_after,
               RH_PSTAGE(routed_c_after, rt_route_subsurface_in, rt_route_subsurface_in(c))
               RH_PSTAGE(routed_c_after, rt_capillary_rise, rt_capillary_rise(c, X))
               RH_PSTAGE(routed_c_after, rt_storage, rt_storage(c, X))
               RH_PSTAGE(routed_c_after, rt_num_error_routed, bad = rt_num_error_routed(c, K))
               RH_PSTAGE(routed_c_after, rt_after_timestep_oned, rt_after_timestep_oned(c)))
// The step core of the hook-preserving flow (rh_step_core: RogerSetup.step() with the user hooks on the host) as ONE staged pass -- the
// fused kernel's pipeline without its selection, rotation and control parts.  k_step_core / k_step_core_lateral, which load every
// plane up front, need 256 VGPRs + 92 / 118 AGPRs and run at one wave per SIMD; kept for A/B (RH_STEP_CORE_UNSTAGED).
#define RH_CORE_HEAD(seq)                                                                          \
    RH_PSTAGE(seq, rt_interception, rt_interception(c, K))                                         \
    RH_PSTAGE(seq, rt_evapotranspiration, rt_evapotranspiration(c, K))                             \
    RH_PSTAGE(seq, rt_snow, rt_snow(c, K, X))                                                      \
    RH_PSTAGE(seq, rt_inf_events, rt_inf_events(c, K, X))                                          \
    RH_PSTAGE(seq, rt_inf_matrix, rt_inf_matrix(c, K, X))                                          \
    RH_PSTAGE(seq, rt_inf_macropores, rt_inf_macropores(c, K, X))                                  \
    RH_PSTAGE(seq, rt_inf_cracks, rt_inf_cracks(c, K, X))                                          \
    RH_PSTAGE(seq, rt_inf_finish, rt_inf_finish(c, K, X))
RH_PASS_KERNEL(k_core_staged,
               RH_CORE_HEAD(core)
               RH_PSTAGE(core, rt_subsurface_runoff, rt_subsurface_runoff(c, X))
               RH_PSTAGE(core, rt_capillary_rise, rt_capillary_rise(c, X))
               RH_PSTAGE(core, rt_storage, rt_storage(c, X))
               RH_PSTAGE(core, rt_num_error, bad = rt_num_error(c, K)))
RH_PASS_KERNEL(k_core_staged_lateral,
               RH_CORE_HEAD(core_lateral)
               RH_PSTAGE(core_lateral, rt_subsurface_runoff_lateral, rt_subsurface_runoff_lateral(c, K, X))
               RH_PSTAGE(core_lateral, rt_capillary_rise, rt_capillary_rise(c, X))
               RH_PSTAGE(core_lateral, rt_storage, rt_storage(c, X))
               RH_PSTAGE(core_lateral, rt_num_error_lateral, bad = rt_num_error_lateral(c, K)))
// Device-driven stepping (rh_run_steps / rh_run_steps_dist on a routing context): the first pass with the step's forcing selection [and
// the monthly surface parameters, D->monthly] in front, as the fused kernel has them, and the columns' summary bits for the NEXT step's
// control kernel posted as soon as they are final (k_ctrl reads them from sumw: no predicate passes over the arena between two steps).
#define RH_ROUTED_A2_TAIL(seq)                                                                                  \
    RH_PSTAGE_S(seq, rt_interception, rt_interception(c, K))                                                      \
    RH_PSTAGE_S(seq, rt_evapotranspiration, rt_evapotranspiration(c, K))                                          \
    RH_PSTAGE_S(seq, rt_snow, rt_snow(c, K, X))                                                                   \
    q = summary_bits_sw(q, c.swe, c.swe_top);                                                                   \
    post_summary(D, q, dep);                                                                                    \
    RH_PSTAGE_S(seq, rt_inf_events, rt_inf_events(c, K, X))                                                       \
    RH_PSTAGE_S(seq, rt_inf_matrix, rt_inf_matrix(c, K, X))                                                       \
    RH_PSTAGE_S(seq, rt_inf_macropores, rt_inf_macropores(c, K, X))                                               \
    RH_PSTAGE_S(seq, rt_inf_cracks, rt_inf_cracks(c, K, X))                                                       \
    RH_PSTAGE_S(seq, rt_inf_finish_routed, rt_inf_finish_routed(c, K, X))                                         \
    RH_PSTAGE_S(seq, rt_route_surface_out, rt_route_surface_out(c, K, X, (double)D->S.dt_secs))
template <bool SPARSE>
__global__ __launch_bounds__(RH_BLOCK, RH_STEP_WAVES) void k_routed_a2(Arena a, DevState *D) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const Consts K = D->K;
    const StepCtx X = D->X;
    Col c;
    unsigned long long q = 0;
    unsigned dep = 1;
    double pet_v = X.pet_sel_w, ta_v = X.ta_sel_w;
    if (D->per_cell && X.sel_w >= 0) {
        pet_v = cell_agg(D, a.n, i, 3 * X.sel_w + 2);
        ta_v = cell_agg(D, a.n, i, 3 * X.sel_w + 1);
    }
#ifdef RH_CENSUS   // tools/isa_census.py counts the pipeline a step runs unless the month changes
    if (false) {
#else
    if (D->monthly != 0) {
#endif
        RH_PSTAGE_S(routed_a2_monthly, rt_select_prec, rt_select_prec(c, X, X.prec_sel, X.ta_sel))
        RH_PSTAGE_S(routed_a2_monthly, rt_select_pet, rt_select_pet(c, X, pet_v, ta_v))
        q = summary_bits_pt(c.prec, c.ta, K);
        RH_PSTAGE_S(routed_a2_monthly, rt_params_surface, rt_params_surface(c, D->L, X))
        RH_ROUTED_A2_TAIL(routed_a2_monthly)
    } else {
        RH_PSTAGE_S(routed_a2, rt_select_prec, rt_select_prec(c, X, X.prec_sel, X.ta_sel))
        RH_PSTAGE_S(routed_a2, rt_select_pet, rt_select_pet(c, X, pet_v, ta_v))
        q = summary_bits_pt(c.prec, c.ta, K);
        RH_ROUTED_A2_TAIL(routed_a2)
    }
}
// set_parameters' month-change test was evaluated on the device by the set_forcing hook (D->monthly)
__global__ __launch_bounds__(RH_BLOCK) void k_params_surface_if_monthly(Arena a, DevState *D) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n || !D->monthly) return;
    const StepCtx X = D->X;
    Col c;
    RH_SET_LOAD_rt_params_surface(LD) rt_params_surface(c, D->L, X);
    RH_SET_STORE_rt_params_surface(ST)
}

// ... and the gather between them: q_in of cell (ix, iy) = np.sum over the eight *_in_d8 entries, in_d8[c, d] = where(flow_dir[s] ==
// code_d, q_out[s], 0) * maskCatch[s] with s = c - (dx_d, dy_d) an interior cell (surface_runoff.py:137-204; the reference scatters
// into shifted slices, a cell next to the edge of the grid receives nothing from outside).  The reference's direction order
// N, NE, E, SE, S, SW, W, NW and numpy's sum of 8 contiguous values, ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7)).
// Several ranks (decomposition along x, the slow index): the neighbour ranks' edge columns -- q_out per step, flow direction and mask
// once -- arrive in halo[0] (the column x = -1) and halo[1] (x = nx); null where the rank has no neighbour.
struct RouteHalo {
    const double *q[2];
    const int *flow_dir[2];
    const int *mask[2];
};
RH_DEV double route_gather_value(const Arena &a, int nx, int ny, int src_plane, int64_t i, const RouteHalo &H) {
    const int ix = (int)(i / ny), iy = (int)(i % ny);
    const int CODE[8] = {64, 128, 1, 2, 4, 8, 16, 32};
    const int DX[8] = {0, -1, 1, 1, 0, -1, -1, -1};
    const int DY[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
    double v[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int sx = ix - DX[d], sy = iy - DY[d];
        double q = 0.0;
        int fd = 0, mk = 0;
        if (sy >= 0 && sy < ny) {
            if (sx >= 0 && sx < nx) {
                const int64_t s = (int64_t)sx * ny + sy;
                q = *rh_cell_any<const double>(a, src_plane, s);
                fd = *rh_cell_any<const int>(a, RH_P_flow_dir_topo, s);
                mk = *rh_cell_any<const int>(a, RH_P_maskCatch, s);
            } else {
                const int side = sx < 0 ? 0 : 1;
                if (H.q[side]) {
                    q = H.q[side][sy];
                    fd = H.flow_dir[side][sy];
                    mk = H.mask[side][sy];
                }
            }
        }
        v[d] = (fd == CODE[d] ? q : 0.0) * (double)mk;
    }
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}
__global__ __launch_bounds__(RH_BLOCK) void k_route_gather(Arena a, int nx, int ny, int src_plane, int dst_plane, RouteHalo H) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    *rh_cell_any<double>(a, dst_plane, i) = route_gather_value(a, nx, ny, src_plane, i, H);
}
// Device-driven routed stepping: the second and third pass with the gather in front of them folded in -- a column reads its eight
// neighbours' q_out (own columns from the arena, the x-neighbour ranks' edge columns from the halo buffers) instead of a q_in plane
// that a kernel of its own wrote: 4 launches per step instead of 6 (k_ctrl, k_routed_a2, k_routed_bg, k_routed_cg[_after]).
template <int P, int WHICH, typename T>
RH_DEV void ld_or_gather(const Arena &a, int64_t i, T &dst, int nx, int ny, const RouteHalo &H) {
    if constexpr (P == (WHICH == 0 ? (int)RH_P_q_sur_in : (int)RH_P_q_sub_in))
        dst = route_gather_value(a, nx, ny, WHICH == 0 ? (int)RH_P_q_sur_out : (int)RH_P_q_sub_out, i, H);
    else
        rh_ld(a, P, i, dst);
}
#define RH_PSTAGE_G(which, seq, rt, call)                     \
    RH_SEQ_##seq##_LOAD_##rt(LDG##which) call;                \
    if constexpr (SPARSE) { RH_SEQ_##seq##_SSTORE_##rt(ST) }  \
    else { RH_SEQ_##seq##_STORE_##rt(ST) }
#define LDG0(name) ld_or_gather<RH_P_##name, 0>(a, i, c.name, nx, ny, H);
#define LDG1(name) ld_or_gather<RH_P_##name, 1>(a, i, c.name, nx, ny, H);
template <bool SPARSE>
__global__ __launch_bounds__(RH_BLOCK, RH_STEP_WAVES) void k_routed_bg(Arena a, DevState *D, int nx, int ny, RouteHalo H) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const Consts K = D->K;
    const StepCtx X = D->X;
    Col c;
    RH_PSTAGE_G(0, routed_b, rt_route_surface_in, rt_route_surface_in(c))
    RH_PSTAGE_G(0, routed_b, rt_subsurface_runoff_lateral, rt_subsurface_runoff_lateral(c, K, X))
    RH_PSTAGE_G(0, routed_b, rt_route_subsurface_out, rt_route_subsurface_out(c))
}
template <bool AFTER, bool SPARSE>
__global__ __launch_bounds__(RH_BLOCK, RH_STEP_WAVES) void k_routed_cg(Arena a, DevState *D, int nx, int ny, RouteHalo H) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const Consts K = D->K;
    const StepCtx X = D->X;
    Col c;
    bool bad = false;
    if constexpr (AFTER) {
        RH_PSTAGE_G(1, routed_c_after, rt_route_subsurface_in, rt_route_subsurface_in(c))
        RH_PSTAGE_G(1, routed_c_after, rt_capillary_rise, rt_capillary_rise(c, X))
        RH_PSTAGE_G(1, routed_c_after, rt_storage, rt_storage(c, X))
        RH_PSTAGE_G(1, routed_c_after, rt_num_error_routed, bad = rt_num_error_routed(c, K))
        RH_PSTAGE_G(1, routed_c_after, rt_after_timestep_oned, rt_after_timestep_oned(c))
    } else {   // (with the output accumulators between the numerics and the rotation: never sparse)
#define RH_PSTAGE_GF(seq, rt, call) RH_SEQ_##seq##_LOAD_##rt(LDG1) call; RH_SEQ_##seq##_STORE_##rt(ST)
        RH_PSTAGE_GF(routed_c, rt_route_subsurface_in, rt_route_subsurface_in(c))
        RH_PSTAGE_GF(routed_c, rt_capillary_rise, rt_capillary_rise(c, X))
        RH_PSTAGE_GF(routed_c, rt_storage, rt_storage(c, X))
        RH_PSTAGE_GF(routed_c, rt_num_error_routed, bad = rt_num_error_routed(c, K))
#undef RH_PSTAGE_GF
    }
    if (bad) atomicOr(&D->words[2], 1ull);
}
// the rank's own edge columns (x = 0 and x = nx - 1) of a plane into two contiguous rows of ny (what the neighbours' halos take)
template <typename T>
__global__ void k_route_edges(Arena a, int nx, int ny, int plane, T *lo, T *hi) {
    const int iy = blockIdx.x * blockDim.x + threadIdx.x;
    if (iy >= ny) return;
    lo[iy] = *rh_cell_any<const T>(a, plane, iy);
    hi[iy] = *rh_cell_any<const T>(a, plane, (int64_t)(nx - 1) * ny + iy);
}

// max over the columns of slope_per (the trip count of the reference's look-up loop, soil.py:621)
__global__ __launch_bounds__(RH_BLOCK) void k_max_slope(Arena a, DevState *D) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    int v = 0;
    if (i < a.n) rh_ld(a, RH_P_slope_per, i, v);
    for (int off = 32; off; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(&D->max_slope_per, v);
}
RH_CELL_KERNEL(k_topo, rt_topo, rt_topo(c))
RH_CELL_KERNEL(k_params_surface, rt_params_surface, rt_params_surface(c, D->L, X))
RH_CELL_KERNEL(k_params_soil, rt_params_soil, rt_params_soil(c, K, D->L))
RH_CELL_KERNEL(k_initial_conditions, rt_initial_conditions, rt_initial_conditions(c))

__global__ __launch_bounds__(RH_BLOCK) void k_select_pet(Arena a, DevState *D) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const StepCtx X = D->X;
    Col c;
    RH_SET_LOAD_rt_select_pet(LD)
    if (D->per_cell && X.sel_w >= 0)
        rt_select_pet(c, X, cell_agg(D, a.n, i, 3 * X.sel_w + 2), cell_agg(D, a.n, i, 3 * X.sel_w + 1));
    else
        rt_select_pet(c, X, X.pet_sel_w, X.ta_sel_w);
    RH_SET_STORE_rt_select_pet(ST)
}

// predicates of calculate_infiltration for the stand-alone entry point
__global__ __launch_bounds__(RH_BLOCK) void k_inf_pred(Arena a, DevState *D) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    unsigned long long b = 0;
    if (i < a.n) {
        double prec, prec_m1;
        rh_ld(a, RH_P_prec, i, prec);
        rh_ld(a, RH_P_prec_m1, i, prec_m1);
        b |= (prec == 0) ? BIT(PC_P_EQ0) : 0;
        b |= (prec_m1 != 0) ? BIT(PC_PM1_NE0) : 0;
        b |= (prec != 0) ? BIT(PC_P_NE0) : 0;
        b |= (prec_m1 == 0) ? BIT(PC_PM1_EQ0) : 0;
    }
    wave_or_to(&D->words[3], b);
}
__global__ void k_inf_conds(DevState *D) {
    infiltration_conds(D->S, D->X, D->words[3]);
    D->words[3] = 0;
}

// Counter calibration (profiles/): copies `nplanes` float64 planes with the access shape of k_step
// (one 8-byte element per lane and plane), so FETCH_SIZE / WRITE_SIZE can be scaled on a known
// byte count as MI355X_MICROARCH.md prescribes for access widths other than 16 B per lane.
__global__ __launch_bounds__(RH_BLOCK) void k_calib_copy(Arena a, int src0, int dst0, int nplanes) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    double v[32];
    for (int p0 = 0; p0 < nplanes; p0 += 32) {
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (p0 + k < nplanes) rh_ld(a, src0 + p0 + k, i, v[k]);
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (p0 + k < nplanes) rh_st(a, dst0 + p0 + k, i, v[k]);
    }
}

// one plane between the arena and a contiguous buffer of n elements (rh_upload / rh_download / rh_plane_device_ptr)
template <typename T>
__global__ __launch_bounds__(RH_BLOCK) void k_plane_gather(Arena a, int plane, T *dst) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i < a.n) dst[i] = *rh_cell<const T>(a, plane, i);
}
template <typename T>
__global__ __launch_bounds__(RH_BLOCK) void k_plane_scatter(Arena a, int plane, const T *src) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i < a.n) *rh_cell<T>(a, plane, i) = src[i];
}

// (n, 144) -> (144, n): a per-cell forcing array as the host hands it over (the reference's vs.prec_day[x, y, :]) into the layout the
// kernels read with unit stride over the columns.  One 64 x 64 tile per workgroup through LDS, both sides coalesced.
__global__ __launch_bounds__(RH_BLOCK) void k_transpose_forcing(const double *src, double *dst, int64_t n) {
    __shared__ double tile[64][65];
    const int64_t c0 = (int64_t)blockIdx.x * 64;   // first column of the tile
    const int s0 = blockIdx.y * 64;                // first slot
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += RH_BLOCK / 64) {   // rows = columns of the grid, contiguous slots
        const int64_t c = c0 + r;
        const int sl = s0 + tx;
        tile[r][tx] = (c < n && sl < RH_SLOTS_PER_DAY) ? src[c * RH_SLOTS_PER_DAY + sl] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += RH_BLOCK / 64) {   // rows = slots, contiguous columns
        const int sl = s0 + r;
        const int64_t c = c0 + tx;
        if (c < n && sl < RH_SLOTS_PER_DAY) dst[(size_t)sl * n + c] = tile[tx][r];
    }
}

// X_m1 = X for every rotation pair of after_timestep: what the lazy steps left undone (materialise_m1)
__global__ __launch_bounds__(RH_BLOCK) void k_rotate_all(Arena a) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    Col c;
    RH_ROTATION_FIELDS(LD)
    RH_ROTATION_FIELDS(ROT)
}

// initial values of the variable registry that are not zero (roger/variables.py `initial=`)
__global__ __launch_bounds__(RH_BLOCK) void k_init_registry(Arena a) {
    const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    rh_st(a, RH_P_maskCatch, i, 1);
    rh_st(a, RH_P_ta, i, 15.0);
    rh_st(a, RH_P_ta_m1, i, 15.0);
    rh_st(a, RH_P_z_gw, i, 1000.0);
    rh_st(a, RH_P_z_gw_m1, i, 1000.0);
    rh_st(a, RH_P_c_int, i, 1.0);
    rh_st(a, RH_P_c_root, i, 1.0);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static const char *const PLANE_NAMES[] = {
#define RH_N1(name) #name,
#define RH_N2(name) #name, #name "_m1",
#define RH_FIELD(name, type, levels) RH_N##levels(name)
#include "rh_fields.def"
#undef RH_FIELD
#undef RH_N1
#undef RH_N2
};
static const unsigned char PLANE_IS_INT[] = {
#define RH_T_F64 0
#define RH_T_I32 1
#define RH_I1(type) RH_T_##type,
#define RH_I2(type) RH_T_##type, RH_T_##type,
#define RH_FIELD(name, type, levels) RH_I##levels(type)
#include "rh_fields.def"
#undef RH_FIELD
#undef RH_I1
#undef RH_I2
};

// planes the fused step only produces (tools/liveness.py -> RH_SPARSE_FIELDS_* in rh_sets.inc), per model: [0] SVAT, [1] oneD
static const std::vector<unsigned char> *pure_output_planes() {   // [0] SVAT, [1] oneD (fused steps), [2] the routed step
    static const std::vector<unsigned char> tab[3] = {
        [] { std::vector<unsigned char> t(RH_NPLANES, 0);
#define RH_MARK(name) t[RH_P_##name] = 1;
             RH_SPARSE_FIELDS_SVAT(RH_MARK) return t; }(),
        [] { std::vector<unsigned char> t(RH_NPLANES, 0);
             RH_SPARSE_FIELDS_ONED(RH_MARK) return t; }(),
        [] { std::vector<unsigned char> t(RH_NPLANES, 0);
             RH_SPARSE_FIELDS_ROUTED(RH_MARK)
#undef RH_MARK
             return t; }()};
    return tab;
}

static int fail(rh_ctx *ctx, int code, const std::string &msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_create_err = msg;
    return code;
}
#define HIPCHK(ctx, call)                                                                                      \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return fail(ctx, RH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + RH_BLOCK - 1) / RH_BLOCK); }
// any per-column kernel other than the fused step may change what the summary words describe
// The X_m1 planes from the X planes, if lazy steps left them behind (anything but the fused kernel that looks at the
// planes calls this first).
static void materialise_m1(rh_ctx *ctx) {
    if (!ctx->m1_stale) return;
    hipLaunchKernelGGL(k_rotate_all, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena);
    ctx->m1_stale = false;
}
// somebody other than the fused kernel is about to change planes: X_m1 == X cannot be taken for granted afterwards
static void planes_touched(rh_ctx *ctx) {
    materialise_m1(ctx);
    ctx->pmask_valid = false;   // (a parameter plane may be about to change: the wave words are formed again before the next lazy step)
    ctx->rot_consistent = false;
    ctx->summary_valid = false;
    ctx->routed_summary = false;
    ctx->pending_valid = ctx->pre_valid = false;
    ctx->exch_valid = false;
}
#define LAUNCH_CELLS(ctx, kern)                                                                                          \
    do {                                                                                                                 \
        planes_touched(ctx);                                                                                             \
        hipLaunchKernelGGL(kern, dim3(grid_for((ctx)->n)), dim3(RH_BLOCK), 0, (ctx)->stream, (ctx)->arena, (ctx)->dev); \
    } while (0)
#define LAUNCH_ONE(ctx, kern, ...) hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, (ctx)->stream, __VA_ARGS__)
#define LAUNCH_WG(ctx, kern, ...) hipLaunchKernelGGL(kern, dim3(1), dim3(RH_BLOCK), 0, (ctx)->stream, __VA_ARGS__)
#define CHECK_LAUNCH(ctx) HIPCHK(ctx, hipGetLastError())

// RCCL, resolved at run time: a single-GPU user needs no librccl, and a process that already holds one (PyTorch ships its own
// copy under the same soname) keeps using that one.
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    const char *(*GetErrorString)(ncclResult_t);
    bool ok;
    std::string why;
};
static RcclApi *rccl_api() {
    static RcclApi api = [] {
        RcclApi a{};
        void *h = nullptr;
        // RH_RCCL_LIB: this RCCL build and no other (a site's own build; tests/loopback_nccl.cpp, whose "ranks" are threads on one GPU)
        if (const char *own = std::getenv("RH_RCCL_LIB")) {
            h = dlopen(own, RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                a.why = std::string("RH_RCCL_LIB: ") + (dlerror() ? dlerror() : "cannot be loaded");
                return a;
            }
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!h) {
            a.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
            return a;
        }
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
        a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
        a.CommUserRank = (decltype(a.CommUserRank))dlsym(h, "ncclCommUserRank");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString && a.Send && a.Recv && a.GroupStart &&
               a.GroupEnd && a.CommCount && a.CommUserRank;
        if (!a.ok) a.why = "librccl lacks an expected entry point";
        return a;
    }();
    return &api;
}
static void release_comm(rh_ctx *ctx) {
    if (ctx->comm && ctx->own_comm && rccl_api()->ok) (void)rccl_api()->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->own_comm = false;
    ctx->comm_nranks = 1;
    ctx->comm_rank = 0;
    ctx->route_static_done = false;
}
#define NCCLCHK(ctx, call)                                                                                                   \
    do {                                                                                                                     \
        ncclResult_t r_ = (call);                                                                                            \
        if (r_ != ncclSuccess) return fail(ctx, RH_ERR_HIP, std::string(#call) + ": " + rccl_api()->GetErrorString(r_));     \
    } while (0)

extern "C" {

int rh_abi_version(void) { return RH_ABI_VERSION; }

void rh_default_config(rh_config *cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->nx = cfg->ny = 1;
    // roger/settings.py:52-122
    cfg->pi = 3.14159265358979323846264338327950588;
    cfg->r_mp = 2.5;
    cfg->l_sc = 10000;
    cfg->sf = 3;
    cfg->ta_fm = 0;
    cfg->rmax = 30;
    cfg->transp_water_stress = 0.75;
    cfg->atol = 1e-2;
    cfg->rtol = 1e-2;
    cfg->clay_min = 0.01;
    cfg->clay_max = 0.71;
    cfg->theta_rew_min = 0.02;
    cfg->theta_rew_max = 0.24;
    cfg->rew_min = 2;
    cfg->rew_max = 12;
    cfg->z_evap_max = 150;
    cfg->zroot_to_zsoil_max = 0.7;
    cfg->a_bc = 2;
    cfg->b_bc = 2;
    cfg->end_event = 21600;
    cfg->hpi = 5;
    cfg->dx = 1;
    cfg->enable_lateral_flow = 0;
    cfg->enable_routing_1D = 0;
    cfg->dy = 1.0;
    cfg->placement_probes = 8;   // up to eight candidate arenas, never more than a quarter of the free memory held at once (1 or RH_PLACEMENT_PROBES=1: none)
}

const char *rh_last_error(const rh_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
int rh_num_planes(void) { return RH_NPLANES; }
int rh_planes_held(const rh_ctx *ctx) { return ctx ? ctx->planes_held : 0; }
const char *rh_plane_name(int p) { return (p >= 0 && p < RH_NPLANES) ? PLANE_NAMES[p] : nullptr; }
int rh_plane_is_int(int p) { return (p >= 0 && p < RH_NPLANES) ? PLANE_IS_INT[p] : -1; }
int rh_plane_index(const char *name) {
    if (!name) return -1;
    for (int p = 0; p < RH_NPLANES; ++p)
        if (!std::strcmp(PLANE_NAMES[p], name)) return p;
    return -1;
}
int64_t rh_num_cells(const rh_ctx *ctx) { return ctx ? ctx->n : 0; }

int rh_create(const rh_config *cfg, rh_ctx **out) {
    if (!cfg || !out) return fail(nullptr, RH_ERR_ARG, "rh_create: null argument");
    if (cfg->nx <= 0 || cfg->ny <= 0) return fail(nullptr, RH_ERR_ARG, "rh_create: nx and ny must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RH_ERR_NODEVICE, "rh_create: no HIP device visible (this backend has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, RH_ERR_ARG, "rh_create: device ordinal out of range");
    HIPCHK(nullptr, hipSetDevice(cfg->device));
    rh_ctx *ctx = new (std::nothrow) rh_ctx();
    if (!ctx) return fail(nullptr, RH_ERR_ARG, "rh_create: out of host memory");
    ctx->cfg = *cfg;
    ctx->n = cfg->nx * cfg->ny;
    ctx->stream = nullptr;
    ctx->own_stream = false;
    ctx->forcing_set = false;
    ctx->timing = false;
    ctx->ev_used = 0;
    ctx->dt_log_buf = nullptr;
    ctx->dev = nullptr;
    ctx->arena.base = nullptr;
    ctx->arena_alloc = nullptr;
    ctx->arena_offset = 0;
    ctx->rot_consistent = false;
    ctx->m1_stale = false;
    ctx->agg_daily_stale = true;
    ctx->pred_daily_stale = true;
    ctx->front_daily_stale = true;
    ctx->diag_reads_m1 = false;
    ctx->lazy_ok = std::getenv("RH_NO_LAZY_ROTATION") == nullptr;
    ctx->sparse_ok = std::getenv("RH_NO_SPARSE_STORES") == nullptr;
    for (auto &b : ctx->forc_cell_buf) b = nullptr;
    for (auto &b : ctx->weight_buf) b = nullptr;
    ctx->station_buf = nullptr;
    ctx->forc_multi_buf = nullptr;
    ctx->transpose_buf = nullptr;
    ctx->agg_cell_buf = nullptr;
    ctx->series_buf = nullptr;
    ctx->mlms_buf = nullptr;
    ctx->diag_buf = nullptr;
    ctx->diag_steps_buf = nullptr;
    ctx->diag_interval = 86400;
    ctx->diag_n = 0;
    ctx->diag_slots = 0;
    ctx->per_cell = false;
#if RH_TILED
    const size_t n_tiles = ((size_t)ctx->n + RH_TILE_CELLS - 1) / RH_TILE_CELLS;
    ctx->planes_held = cfg->enable_routing_1D ? (int)RH_NPLANES : (int)RH_P_flow_dir_topo;
    const size_t stride = (size_t)(ctx->planes_held + RH_STRIDE_PAD) * RH_SLOT_BYTES, arena_bytes = n_tiles * stride;
#else
    ctx->planes_held = RH_NPLANES;
    const size_t stride = (((size_t)ctx->n * sizeof(double)) + 255) / 256 * 256, arena_bytes = stride * RH_NPLANES;
#endif
    ctx->arena.stride = stride;
    ctx->arena.n = ctx->n;
    ctx->stage_buf = nullptr;
    ctx->comm = nullptr;
    ctx->own_comm = false;
    ctx->exch_buf = nullptr;
    ctx->exch_valid = false;
    ctx->comm_nranks = 1;
    ctx->comm_rank = 0;
    ctx->route_q = nullptr;
    ctx->route_i = nullptr;
    ctx->route_halo[0] = ctx->route_halo[1] = false;
    ctx->route_static_done = false;
    if (cfg->enable_routing_1D && !cfg->enable_lateral_flow) {
        delete ctx;
        return fail(nullptr, RH_ERR_ARG, "rh_create: enable_routing_1D needs enable_lateral_flow (the routed subsurface runoff is the lateral flow)");
    }
    auto bail = [&](hipError_t e, const char *what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        rh_destroy(ctx);
        return fail(nullptr, RH_ERR_HIP, msg);
    };
    hipError_t e;
    if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) return bail(e, "hipStreamCreate");
    ctx->own_stream = true;
    {   // experiments (tools/placement_diag*.py): RH_ARENA_OFFSET_KB shifts the arena inside a larger allocation
        const char *off = std::getenv("RH_ARENA_OFFSET_KB");
        ctx->arena_offset = off ? (size_t)std::atoll(off) * 1024 : 0;
        const char *pad = std::getenv("RH_ARENA_PAD_KB");   // ... and RH_ARENA_PAD_KB pads the allocation behind the arena
        ctx->arena_pad = pad ? (size_t)std::atoll(pad) * 1024 : 0;
    }
    {
        // Placement probing (rh_config.placement_probes): candidates are allocated one after the other and held until
        // the choice is made, so that each lands somewhere else; a copy of 96 planes with the fused kernel's access
        // shape (k_calib_copy) is timed on each (its time tracks the fused kernel's level, tools/placement_diag3.py).
        int probes = cfg->placement_probes;
        if (const char *env = std::getenv("RH_PLACEMENT_PROBES")) probes = std::atoi(env);
        if (probes < 1 || ctx->n < 65536) probes = 1;   // small grids are latency-bound
        std::vector<char *> cand;
        std::vector<double> cand_ms;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (probes > 1 && (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess)) probes = 1;
        for (int k = 0; k < probes; ++k) {
            if (k > 0) {   // all candidates are held until the choice is made: never more than a quarter of the free memory in total
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess ||
                    (free_b + k * (arena_bytes + ctx->arena_offset + ctx->arena_pad)) / 4 < (k + 1) * (arena_bytes + ctx->arena_offset + ctx->arena_pad)) break;
            }
            char *p = nullptr;
            if ((e = hipMalloc((void **)&p, arena_bytes + ctx->arena_offset + ctx->arena_pad)) != hipSuccess) {
                if (k == 0) return bail(e, "hipMalloc(arena)");
                (void)hipGetLastError();
                break;
            }
            cand.push_back(p);
            if ((e = hipMemsetAsync(p + ctx->arena_offset, 0, arena_bytes, ctx->stream)) != hipSuccess) {
                for (char *q : cand) (void)hipFree(q);
                return bail(e, "hipMemset");
            }
            double ms = 0;
            if (probes > 1) {
                Arena probe = ctx->arena;
                probe.base = p + ctx->arena_offset;
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {   // the first repetition warms up
                    (void)hipEventRecord(ev0, ctx->stream);
                    hipLaunchKernelGGL(k_calib_copy, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, probe, 0, 100, 96);
                    (void)hipEventRecord(ev1, ctx->stream);
                    (void)hipEventSynchronize(ev1);
                    float t = 0;
                    if (rep > 0 && hipEventElapsedTime(&t, ev0, ev1) == hipSuccess && t < best) best = t;
                }
                ms = best;
            }
            cand_ms.push_back(ms);
        }
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        size_t pick = 0;
        for (size_t k = 1; k < cand.size(); ++k)
            if (cand_ms[k] < cand_ms[pick]) pick = k;
        ctx->arena_alloc = cand[pick];
        ctx->arena.base = cand[pick] + ctx->arena_offset;
        if (probes > 1) {
            ctx->probe_ms.push_back(cand_ms[pick]);
            for (size_t k = 0; k < cand.size(); ++k)
                if (k != pick) ctx->probe_ms.push_back(cand_ms[k]);
        }
        for (size_t k = 0; k < cand.size(); ++k)
            if (k != pick) (void)hipFree(cand[k]);
    }
    if ((e = hipMalloc((void **)&ctx->stage_buf, (size_t)ctx->n * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(staging plane)");
    if ((e = hipMalloc((void **)&ctx->dev, sizeof(DevState))) != hipSuccess) return bail(e, "hipMalloc(DevState)");
    if ((e = hipHostMalloc((void **)&ctx->hexp, sizeof(HostExport), hipHostMallocMapped)) != hipSuccess) return bail(e, "hipHostMalloc(scalar export block)");
    std::memset(ctx->hexp, 0, sizeof(HostExport));
    if ((e = hipMemsetAsync(ctx->dev, 0, sizeof(DevState), ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
    {   // the parameter words of the fused step's wavefronts (all zero: plain loads; AFTER the control block was cleared) and their address in the control block
        const size_t words = ((size_t)ctx->n + 63) / 64;
        if ((e = hipMalloc((void **)&ctx->pmask_buf, words * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(parameter words)");
        if ((e = hipMemsetAsync(ctx->pmask_buf, 0, words * sizeof(unsigned long long), ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
        if ((e = hipMemcpyAsync(&ctx->dev->pmask, &ctx->pmask_buf, sizeof(ctx->pmask_buf), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return bail(e, "hipMemcpy(pmask)");
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
        ctx->pmask_flags = (std::getenv("RH_NO_PARAM_UNIFORM") ? 0 : 1) | (std::getenv("RH_NO_PARAM_DERIVE") ? 0 : 2) | (std::getenv("RH_NO_MASK_CONSTANT") ? 0 : 4);
    }
    {
        static const long long no_limit = -1;
        if ((e = hipMemcpyAsync(&ctx->dev->t_end, &no_limit, sizeof(no_limit), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return bail(e, "hipMemcpy(t_end)");
    }
    Consts K;
    K.pi = cfg->pi; K.r_mp = cfg->r_mp; K.l_sc = cfg->l_sc; K.sf = cfg->sf; K.ta_fm = cfg->ta_fm; K.rmax = cfg->rmax;
    K.transp_water_stress = cfg->transp_water_stress; K.atol = cfg->atol; K.rtol = cfg->rtol;
    K.clay_min = cfg->clay_min; K.clay_max = cfg->clay_max; K.theta_rew_min = cfg->theta_rew_min;
    K.theta_rew_max = cfg->theta_rew_max; K.rew_min = cfg->rew_min; K.rew_max = cfg->rew_max;
    K.z_evap_max = cfg->z_evap_max; K.zroot_to_zsoil_max = cfg->zroot_to_zsoil_max; K.a_bc = cfg->a_bc; K.b_bc = cfg->b_bc;
    K.end_event = cfg->end_event; K.hpi = cfg->hpi;
    K.dx = cfg->dx; K.lateral = cfg->enable_lateral_flow ? 1 : 0;
    K.dy = cfg->dy; K.routing = cfg->enable_routing_1D ? 1 : 0;
    if ((e = hipMemcpyAsync(&ctx->dev->K, &K, sizeof(K), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        return bail(e, "hipMemcpy(Consts)");
    // scalars: roger/variables.py initial values (dt=1, dt_secs=3600, event_id_counter=1, year=1900, month=doy=1)
    rh_scalars S;
    std::memset(&S, 0, sizeof(S));
    S.dt = 1;
    S.dt_secs = 3600;
    S.event_id_counter = 1;
    S.year[0] = S.year[1] = 1900;
    S.month[0] = S.month[1] = 1;
    S.doy[0] = S.doy[1] = 1;
    S.sanity_ok = 1;
    if ((e = hipMemcpyAsync(&ctx->dev->S, &S, sizeof(S), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        return bail(e, "hipMemcpy(scalars)");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");  // K, S are stack locals
    ctx->pending_valid = ctx->pre_valid = false;
    ctx->pending_hooks = 0;
    ctx->tail_ok = std::getenv("RH_NO_TAIL_CTRL") == nullptr;
    ctx->routed_device_ok = std::getenv("RH_ROUTED_BY_ROUTINE") == nullptr;
    ctx->defer_select_ok = std::getenv("RH_NO_DEFERRED_SELECT") == nullptr;
    ctx->cell_front_ok = std::getenv("RH_PER_CELL_OLD_FRONT") == nullptr && ctx->defer_select_ok;
    if (const char *v = std::getenv("RH_CELL_FRONT_MAX")) ctx->cell_front_max = std::atoll(v);
    if (const char *v = std::getenv("RH_CELL_AGG_SPLIT_MIN")) ctx->cell_agg_split_min = std::atoll(v);
    ctx->n_groups = (int)((grid_for(ctx->n) + 63) / 64);
    if (ctx->n_groups > RH_DONE_GROUPS) ctx->n_groups = RH_DONE_GROUPS;
    if (ctx->n_groups < 1) ctx->n_groups = 1;
    ctx->summary_valid = false;
    ctx->pred_blocks = (int)(grid_for(ctx->n) < RH_PRED_BLOCKS ? grid_for(ctx->n) : RH_PRED_BLOCKS);
    if ((e = hipMemcpyAsync(&ctx->dev->pred_blocks, &ctx->pred_blocks, sizeof(int), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        return bail(e, "hipMemcpy(pred_blocks)");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    hipLaunchKernelGGL(k_init_registry, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena);
    hipLaunchKernelGGL(k_sync_ctx, dim3(1), dim3(1), 0, ctx->stream, ctx->dev);
    if ((e = hipGetLastError()) != hipSuccess) return bail(e, "kernel launch (is this a gfx950 device?)");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    *out = ctx;
    return RH_OK;
}

void rh_destroy(rh_ctx *ctx) {
    if (!ctx) return;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
#ifdef RH_STEP_PHASES
    {
        unsigned long long tp[8];
        if (hipMemcpyFromSymbol(tp, HIP_SYMBOL(g_tail_phases), sizeof(tp)) == hipSuccess && tp[7])
            std::fprintf(stderr, "tail phases: %llu tails, cycles per tail: loads %.0f, hooks+staging %.0f, bits+sums %.0f, decisions %.0f, scalars %.0f, log %.0f; whole tail %.0f\n",
                         tp[7], (double)tp[0] / tp[7], (double)tp[1] / tp[7], (double)tp[2] / tp[7], (double)tp[3] / tp[7], (double)tp[4] / tp[7],
                         (double)tp[5] / tp[7], (double)tp[6] / tp[7]);
        static unsigned long long all[256 * 64];
        unsigned long long h[60] = {0};
        if (hipMemcpyFromSymbol(all, HIP_SYMBOL(g_step_phases), sizeof(all)) == hipSuccess) {
            for (int b = 0; b < 256; ++b)
                for (int k = 0; k < 60; ++k) h[k] += all[b * 64 + k];
            static const char *cls[3] = {"10min", "hourly", "daily"};
            for (int c = 0; c < 3; ++c) {
                if (!h[c * 20 + 19]) continue;
                double tot = 0;
                for (int k = 0; k < 19; ++k) tot += (double)h[c * 20 + k];
                std::fprintf(stderr, "step phases %s: %llu waves, cycles per wave %.0f:", cls[c], h[c * 20 + 19], tot / (double)h[c * 20 + 19]);
                for (int k = 1; k <= 14; ++k) std::fprintf(stderr, " %d:%.1f%%", k, 100.0 * (double)h[c * 20 + k] / tot);
                std::fprintf(stderr, "\n");
            }
        }
    }
#endif
    for (auto &ev : ctx->events) (void)hipEventDestroy(ev);
    if (ctx->dt_log_buf) (void)hipFree(ctx->dt_log_buf);
    for (auto &b : ctx->forc_cell_buf)
        if (b) (void)hipFree(b);
    for (auto &b : ctx->weight_buf)
        if (b) (void)hipFree(b);
    if (ctx->agg_cell_buf) (void)hipFree(ctx->agg_cell_buf);
    if (ctx->station_buf) (void)hipFree(ctx->station_buf);
    if (ctx->forc_multi_buf) (void)hipFree(ctx->forc_multi_buf);
    if (ctx->transpose_buf) (void)hipFree(ctx->transpose_buf);
    if (ctx->series_buf) (void)hipFree(ctx->series_buf);
    if (ctx->mlms_buf) (void)hipFree(ctx->mlms_buf);
    if (ctx->diag_buf) (void)hipFree(ctx->diag_buf);
    if (ctx->diag_steps_buf) (void)hipFree(ctx->diag_steps_buf);
    if (ctx->arena_alloc) (void)hipFree(ctx->arena_alloc);
    if (ctx->stage_buf) (void)hipFree(ctx->stage_buf);
    if (ctx->exch_buf) (void)hipFree(ctx->exch_buf);
    if (ctx->route_q) (void)hipFree(ctx->route_q);
    if (ctx->route_i) (void)hipFree(ctx->route_i);
    release_comm(ctx);
    if (ctx->dev) (void)hipFree(ctx->dev);
    if (ctx->hexp) (void)hipHostFree(ctx->hexp);
    if (ctx->pmask_buf) (void)hipFree(ctx->pmask_buf);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rh_set_stream(rh_ctx *ctx, void *hip_stream) {
    if (!ctx) return RH_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) HIPCHK(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return RH_OK;
}

static int device_error(rh_ctx *ctx, unsigned err) {
    if (err & RH_DEVERR_FORCING)
        return fail(ctx, RH_ERR_STATE, "a step began a day beyond the end of the resident forcing series (rh_set_forcing_series): it ran on the "
                                       "previous day's forcing; hand over a longer series (the reference fails on the short slice, "
                                       "benchmarks/SVAT_benchmark.py:151-171)");
    return RH_OK;
}
int rh_sync(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    unsigned err = 0;
    HIPCHK(ctx, hipMemcpyAsync(&err, &ctx->dev->err_flags, sizeof(err), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return device_error(ctx, err);
}

static int plane_bytes(rh_ctx *ctx, int plane, size_t bytes, size_t *elem) {
    if (!ctx) return RH_ERR_ARG;
    if (plane < 0 || plane >= RH_NPLANES) return fail(ctx, RH_ERR_ARG, "unknown plane id");
    if (plane >= ctx->planes_held)
        return fail(ctx, RH_ERR_STATE, std::string("plane ") + PLANE_NAMES[plane] + " belongs to the routing: the context was created without enable_routing_1D");
    *elem = PLANE_IS_INT[plane] ? sizeof(int32_t) : sizeof(double);
    if (bytes != *elem * (size_t)ctx->n)
        return fail(ctx, RH_ERR_ARG, std::string("size mismatch for plane ") + PLANE_NAMES[plane]);
    return RH_OK;
}

int rh_upload(rh_ctx *ctx, int plane, const void *host, size_t bytes) {
    size_t elem;
    int rc = plane_bytes(ctx, plane, bytes, &elem);
    if (rc) return rc;
    if (!host) return fail(ctx, RH_ERR_ARG, "rh_upload: null host pointer");
    planes_touched(ctx);
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_buf, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (elem == sizeof(double))
        hipLaunchKernelGGL(k_plane_scatter<double>, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, plane, (const double *)ctx->stage_buf);
    else
        hipLaunchKernelGGL(k_plane_scatter<int>, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, plane, (const int *)ctx->stage_buf);
    CHECK_LAUNCH(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the host buffer may be a temporary
    ctx->summary_valid = false;
    return RH_OK;
}

int rh_download(rh_ctx *ctx, int plane, void *host, size_t bytes) {
    size_t elem;
    int rc = plane_bytes(ctx, plane, bytes, &elem);
    if (rc) return rc;
    if (!host) return fail(ctx, RH_ERR_ARG, "rh_download: null host pointer");
    if (ctx->outputs_stale && pure_output_planes()[ctx->cfg.enable_routing_1D ? 2 : (ctx->cfg.enable_lateral_flow ? 1 : 0)][plane])   // only after an rh_run_steps call that failed half-way
        return fail(ctx, RH_ERR_STATE, "rh_download: the last rh_run_steps call ended before its final step; this flux / diagnostic plane holds an "
                                       "earlier step's values (run one more step)");
    materialise_m1(ctx);
    if (elem == sizeof(double))
        hipLaunchKernelGGL(k_plane_gather<double>, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, plane, (double *)ctx->stage_buf);
    else
        hipLaunchKernelGGL(k_plane_gather<int>, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, plane, (int *)ctx->stage_buf);
    CHECK_LAUNCH(ctx);
    HIPCHK(ctx, hipMemcpyAsync(host, ctx->stage_buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}

void *rh_plane_device_ptr(rh_ctx *ctx, int plane) {
    if (!ctx || plane < 0 || plane >= ctx->planes_held) return nullptr;
    if (ctx->outputs_stale && pure_output_planes()[ctx->cfg.enable_routing_1D ? 2 : (ctx->cfg.enable_lateral_flow ? 1 : 0)][plane]) {
        // (as rh_download: only after a stepping call that ended before its final, full-store step -- ADVICE r3)
        fail(ctx, RH_ERR_STATE, "rh_plane_device_ptr: the last rh_run_steps call ended before its final step; this plane holds an earlier step's values");
        return nullptr;
    }
    planes_touched(ctx);  // the caller may write through the pointer
#if RH_TILED
    return ctx->arena.base + (size_t)plane * RH_SLOT_BYTES;   // cell i: + (i / 64) * tile_bytes + (i % 64) * element size
#else
    return ctx->arena.base + (size_t)plane * ctx->arena.stride;
#endif
}

int rh_set_scalars(rh_ctx *ctx, const rh_scalars *s) {
    if (!ctx || !s) return RH_ERR_ARG;
    ctx->pending_valid = ctx->pre_valid = false;
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->err_flags, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->S, s, sizeof(*s), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    LAUNCH_ONE(ctx, k_sync_ctx, ctx->dev);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

// Enqueue the export of the scalars behind whatever is on the stream and wait for it: the host spins on the block's sequence number
// (the write arrives a few microseconds after the kernel in front has finished; hipStreamSynchronize wakes up later) and looks at the
// stream from time to time, so that a launch that failed ends the wait with its error instead of hanging.
static int export_scalars(rh_ctx *ctx, rh_scalars *s) {
    const unsigned long long seq = ++ctx->hexp_seq;
    hipLaunchKernelGGL(k_export, dim3(1), dim3(1), 0, ctx->stream, (const DevState *)ctx->dev, ctx->hexp, seq);
    CHECK_LAUNCH(ctx);
    const unsigned long long *p = &ctx->hexp->seq;
    for (unsigned long spins = 1;; ++spins) {
        if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0xfffful) == 0) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == seq) break;
                return fail(ctx, RH_ERR_HIP, "the scalar export kernel finished without its block arriving in host memory");
            }
            if (q != hipErrorNotReady) HIPCHK(ctx, q);
        }
    }
    const HostExport &H = *ctx->hexp;
    *s = H.S;
    // word 2 collects the sanity violations of the last step; the fused kernel's tail moves it to sanity_last
    s->sanity_ok = (H.bad | H.bad_last) ? 0 : 1;
    return device_error(ctx, H.err);
}

int rh_get_scalars(rh_ctx *ctx, rh_scalars *s) {
    if (!ctx || !s) return RH_ERR_ARG;
    return export_scalars(ctx, s);
}

int rh_set_luts(rh_ctx *ctx, const double *ilu, const double *gc, const double *gcm, const double *rdlu) {
    if (!ctx || !ilu || !gc || !gcm || !rdlu) return RH_ERR_ARG;
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->L.ilu, ilu, sizeof(double) * 25 * 13, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->L.gc, gc, sizeof(double) * 25 * 13, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->L.gcm, gcm, sizeof(double) * 25 * 2, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->L.rdlu, rdlu, sizeof(double) * 25 * 7, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}

int rh_set_lut_mlms(rh_ctx *ctx, const double *mlms, int64_t nrows) {
    if (!ctx || !mlms || nrows <= 0) return RH_ERR_ARG;
    if (ctx->mlms_buf) HIPCHK(ctx, hipFree(ctx->mlms_buf));
    ctx->mlms_buf = nullptr;
    const size_t nb = sizeof(double) * 9 * (size_t)nrows;
    HIPCHK(ctx, hipMalloc((void **)&ctx->mlms_buf, nb));
    HIPCHK(ctx, hipMemcpyAsync(ctx->mlms_buf, mlms, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->mlms, &ctx->mlms_buf, sizeof(double *), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->mlms_rows, &nrows, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}

int rh_params_lateral(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    if (!ctx->mlms_buf) return fail(ctx, RH_ERR_STATE, "rh_set_lut_mlms must be called first");
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->max_slope_per, 0, sizeof(int), ctx->stream));
    LAUNCH_CELLS(ctx, k_max_slope);
    LAUNCH_CELLS(ctx, k_params_lateral);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

int rh_set_forcing_day(rh_ctx *ctx, const double *prec_day, const double *ta_day, const double *pet_day, int per_cell) {
    if (!ctx || !prec_day || !ta_day || !pet_day) return RH_ERR_ARG;
    const double *src[3] = {prec_day, ta_day, pet_day};
    int pc = per_cell ? 1 : 0;
    if (!pc) {
        for (int k = 0; k < 3; ++k)
            HIPCHK(ctx, hipMemcpyAsync(ctx->dev->forc[k], src[k], sizeof(double) * RH_SLOTS_PER_DAY, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(&ctx->dev->day_cache_ok, 0, sizeof(int), ctx->stream));   // (ctrl_wave's cache of the day)
    } else {
        const size_t bytes = sizeof(double) * RH_SLOTS_PER_DAY * (size_t)ctx->n;
        if (!ctx->transpose_buf) HIPCHK(ctx, hipMalloc((void **)&ctx->transpose_buf, bytes));
        for (int k = 0; k < 3; ++k) {   // (n, 144) from the host -> (144, n) on the device
            if (!ctx->forc_cell_buf[k]) HIPCHK(ctx, hipMalloc((void **)&ctx->forc_cell_buf[k], bytes));
            HIPCHK(ctx, hipMemcpyAsync(ctx->transpose_buf, src[k], bytes, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_transpose_forcing, dim3((unsigned)((ctx->n + 63) / 64), (RH_SLOTS_PER_DAY + 63) / 64), dim3(RH_BLOCK), 0, ctx->stream,
                               (const double *)ctx->transpose_buf, ctx->forc_cell_buf[k], ctx->n);
            CHECK_LAUNCH(ctx);
        }
        HIPCHK(ctx, hipMemcpyAsync(ctx->dev->forc_cell, ctx->forc_cell_buf, sizeof(double *) * 3, hipMemcpyHostToDevice, ctx->stream));
        if (!ctx->agg_cell_buf) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->agg_cell_buf, sizeof(double) * 9 * (size_t)ctx->n));
            HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->agg_cell, &ctx->agg_cell_buf, sizeof(double *), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    ctx->per_cell = pc != 0;
    ctx->pending_valid = ctx->pre_valid = false;
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->per_cell, &pc, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->forcing_set = true;
    return RH_OK;
}

#define SIMPLE_ENTRY(fname, kern)           \
    int fname(rh_ctx *ctx) {                \
        if (!ctx) return RH_ERR_ARG;        \
        LAUNCH_CELLS(ctx, kern);            \
        CHECK_LAUNCH(ctx);                  \
        return RH_OK;                       \
    }

SIMPLE_ENTRY(rh_topo, k_topo)
SIMPLE_ENTRY(rh_params_surface, k_params_surface)
SIMPLE_ENTRY(rh_params_soil, k_params_soil)
SIMPLE_ENTRY(rh_initial_conditions, k_initial_conditions)
SIMPLE_ENTRY(rh_interception, k_interception)
SIMPLE_ENTRY(rh_evapotranspiration, k_evapotranspiration)
SIMPLE_ENTRY(rh_snow, k_snow)
SIMPLE_ENTRY(rh_capillary_rise, k_capillary_rise)
SIMPLE_ENTRY(rh_storage, k_storage)

int rh_subsurface_runoff(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    if (ctx->cfg.enable_lateral_flow)
        LAUNCH_CELLS(ctx, k_subsurface_runoff_lateral);
    else
        LAUNCH_CELLS(ctx, k_subsurface_runoff);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

int rh_infiltration(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    LAUNCH_CELLS(ctx, k_inf_pred);
    LAUNCH_ONE(ctx, k_inf_conds, ctx->dev);
    if (ctx->cfg.enable_routing_1D) LAUNCH_CELLS(ctx, k_infiltration_routed);
    else LAUNCH_CELLS(ctx, k_infiltration);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

int rh_num_error(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->words[2], 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->sanity_last, 0, sizeof(unsigned long long), ctx->stream));
    if (ctx->cfg.enable_routing_1D)
        LAUNCH_CELLS(ctx, k_num_error_routed);
    else if (ctx->cfg.enable_lateral_flow)
        LAUNCH_CELLS(ctx, k_num_error_lateral);
    else
        LAUNCH_CELLS(ctx, k_num_error);
    LAUNCH_ONE(ctx, k_sanity_to_scalars, ctx->dev);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

// interception ... numerics in one kernel, then itt/time (roger/roger.py:410-457); for drivers that
// keep the user hooks `set_parameters` and `after_timestep` on the host
static int routed_core(rh_ctx *ctx, bool with_after);
int rh_step_core(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    if (ctx->cfg.enable_routing_1D) return routed_core(ctx, false);   // the columns are coupled: routine by routine with the two gathers
    static const bool unstaged = std::getenv("RH_STEP_CORE_UNSTAGED") != nullptr;   // A/B, tests: the single-function kernels
    if (ctx->cfg.enable_lateral_flow) {
        if (unstaged) LAUNCH_CELLS(ctx, k_step_core_lateral);
        else LAUNCH_CELLS(ctx, k_core_staged_lateral);
    } else {
        if (unstaged) LAUNCH_CELLS(ctx, k_step_core);
        else LAUNCH_CELLS(ctx, k_core_staged);
    }
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, ctx->stream, ctx->dev);
    // the output accumulators follow every step, also in the hook-preserving flow (itt / time were just advanced)
    if (ctx->diag_n) hipLaunchKernelGGL(k_diag, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, ctx->dev, 0);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

int rh_after_timestep(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    if (ctx->cfg.enable_lateral_flow)
        LAUNCH_CELLS(ctx, k_after_timestep_oned);
    else
        LAUNCH_CELLS(ctx, k_after_timestep);
    LAUNCH_ONE(ctx, k_rotate_scalars, ctx->dev);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

// ---- settings.enable_routing_1D -------------------------------------------------------------------------------------------------
static int route_buffers(rh_ctx *ctx) {
    if (ctx->route_q) return RH_OK;
    const size_t ny = (size_t)ctx->cfg.ny;
    HIPCHK(ctx, hipMalloc((void **)&ctx->route_q, 4 * ny * sizeof(double)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->route_i, 8 * ny * sizeof(int)));
    HIPCHK(ctx, hipMemsetAsync(ctx->route_q, 0, 4 * ny * sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->route_i, 0, 8 * ny * sizeof(int), ctx->stream));
    return RH_OK;
}
static int route_check(rh_ctx *ctx, int which, const char *who) {
    if (!ctx) return RH_ERR_ARG;
    if (!ctx->cfg.enable_routing_1D) return fail(ctx, RH_ERR_STATE, std::string(who) + ": the context was created without enable_routing_1D");
    if (which != 0 && which != 1) return fail(ctx, RH_ERR_ARG, std::string(who) + ": which must be 0 (surface) or 1 (subsurface)");
    return route_buffers(ctx);
}
static void route_pack_edges(rh_ctx *ctx, int plane, bool ints, int slot) {
    const int ny = (int)ctx->cfg.ny, nx = (int)ctx->cfg.nx;
    const dim3 grid((ny + 255) / 256), block(256);
    if (ints)
        hipLaunchKernelGGL(k_route_edges<int>, grid, block, 0, ctx->stream, ctx->arena, nx, ny, plane, ctx->route_i + (size_t)slot * ny,
                           ctx->route_i + (size_t)(slot + 1) * ny);
    else
        hipLaunchKernelGGL(k_route_edges<double>, grid, block, 0, ctx->stream, ctx->arena, nx, ny, plane, ctx->route_q, ctx->route_q + ny);
}
int rh_route_out(rh_ctx *ctx, int which) {
    int rc = route_check(ctx, which, "rh_route_out");
    if (rc) return rc;
    if (which == 0) LAUNCH_CELLS(ctx, k_route_surface_out);
    else LAUNCH_CELLS(ctx, k_route_subsurface_out);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
int rh_route_get_edges(rh_ctx *ctx, int which, double *q_lo, double *q_hi) {
    int rc = route_check(ctx, which, "rh_route_get_edges");
    if (rc) return rc;
    if (!q_lo || !q_hi) return fail(ctx, RH_ERR_ARG, "rh_route_get_edges: null pointer");
    const size_t ny = (size_t)ctx->cfg.ny;
    route_pack_edges(ctx, which == 0 ? RH_P_q_sur_out : RH_P_q_sub_out, false, 0);
    CHECK_LAUNCH(ctx);
    HIPCHK(ctx, hipMemcpyAsync(q_lo, ctx->route_q, ny * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q_hi, ctx->route_q + ny, ny * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}
int rh_route_get_static_edges(rh_ctx *ctx, int32_t *fd_lo, int32_t *fd_hi, int32_t *mk_lo, int32_t *mk_hi) {
    int rc = route_check(ctx, 0, "rh_route_get_static_edges");
    if (rc) return rc;
    if (!fd_lo || !fd_hi || !mk_lo || !mk_hi) return fail(ctx, RH_ERR_ARG, "rh_route_get_static_edges: null pointer");
    const size_t ny = (size_t)ctx->cfg.ny;
    route_pack_edges(ctx, RH_P_flow_dir_topo, true, 0);
    route_pack_edges(ctx, RH_P_maskCatch, true, 2);
    CHECK_LAUNCH(ctx);
    int32_t *dst[4] = {fd_lo, fd_hi, mk_lo, mk_hi};
    for (int k = 0; k < 4; ++k) HIPCHK(ctx, hipMemcpyAsync(dst[k], ctx->route_i + k * ny, ny * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}
int rh_route_set_halo(rh_ctx *ctx, int side, const double *q, const int32_t *flow_dir, const int32_t *mask) {
    int rc = route_check(ctx, 0, "rh_route_set_halo");
    if (rc) return rc;
    if (side != 0 && side != 1) return fail(ctx, RH_ERR_ARG, "rh_route_set_halo: side must be 0 (x = -1) or 1 (x = nx)");
    const size_t ny = (size_t)ctx->cfg.ny;
    if (flow_dir && mask) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->route_i + (4 + side) * ny, flow_dir, ny * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->route_i + (6 + side) * ny, mask, ny * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        ctx->route_halo[side] = true;
    }
    if (q) {
        if (!ctx->route_halo[side]) return fail(ctx, RH_ERR_STATE, "rh_route_set_halo: the side's flow direction and mask must be set first");
        HIPCHK(ctx, hipMemcpyAsync(ctx->route_q + (2 + side) * ny, q, ny * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}
static RouteHalo route_halo_of(rh_ctx *ctx) {
    const size_t ny = (size_t)ctx->cfg.ny;
    RouteHalo H;
    for (int side = 0; side < 2; ++side) {
        const bool have = ctx->route_halo[side];
        H.q[side] = have ? ctx->route_q + (2 + side) * ny : nullptr;
        H.flow_dir[side] = have ? ctx->route_i + (4 + side) * ny : nullptr;
        H.mask[side] = have ? ctx->route_i + (6 + side) * ny : nullptr;
    }
    return H;
}
static int rh_route_gather_only(rh_ctx *ctx, int which) {
    int rc = route_check(ctx, which, "rh_route_in");
    if (rc) return rc;
    const RouteHalo H = route_halo_of(ctx);
    planes_touched(ctx);
    hipLaunchKernelGGL(k_route_gather, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, (int)ctx->cfg.nx, (int)ctx->cfg.ny,
                       which == 0 ? (int)RH_P_q_sur_out : (int)RH_P_q_sub_out, which == 0 ? (int)RH_P_q_sur_in : (int)RH_P_q_sub_in, H);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
int rh_route_in(rh_ctx *ctx, int which) {
    int rc = rh_route_gather_only(ctx, which);
    if (rc) return rc;
    if (which == 0) LAUNCH_CELLS(ctx, k_route_surface_in);
    else LAUNCH_CELLS(ctx, k_route_subsurface_in);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
// the neighbours' edge columns over RCCL (decomposition along x: rank r - 1 holds x < 0, rank r + 1 holds x >= nx)
static int route_exchange(rh_ctx *ctx, int which) {
    RcclApi *api = rccl_api();
    if (!api->ok) return fail(ctx, RH_ERR_STATE, "routing: " + api->why);
    const size_t ny = (size_t)ctx->cfg.ny;
    const int r = ctx->comm_rank, N = ctx->comm_nranks;
    if (!ctx->route_static_done) {
        route_pack_edges(ctx, RH_P_flow_dir_topo, true, 0);
        route_pack_edges(ctx, RH_P_maskCatch, true, 2);
        CHECK_LAUNCH(ctx);
        NCCLCHK(ctx, api->GroupStart());
        for (int k = 0; k < 2; ++k) {   // k = 0: flow direction, 1: mask
            int *own = ctx->route_i + (size_t)(2 * k) * ny, *halo = ctx->route_i + (size_t)(4 + 2 * k) * ny;
            if (r > 0) {
                NCCLCHK(ctx, api->Send(own, ny, ncclInt32, r - 1, ctx->comm, ctx->stream));
                NCCLCHK(ctx, api->Recv(halo, ny, ncclInt32, r - 1, ctx->comm, ctx->stream));
            }
            if (r < N - 1) {
                NCCLCHK(ctx, api->Send(own + ny, ny, ncclInt32, r + 1, ctx->comm, ctx->stream));
                NCCLCHK(ctx, api->Recv(halo + ny, ny, ncclInt32, r + 1, ctx->comm, ctx->stream));
            }
        }
        NCCLCHK(ctx, api->GroupEnd());
        ctx->route_halo[0] = r > 0;
        ctx->route_halo[1] = r < N - 1;
        ctx->route_static_done = true;
    }
    route_pack_edges(ctx, which == 0 ? RH_P_q_sur_out : RH_P_q_sub_out, false, 0);
    CHECK_LAUNCH(ctx);
    NCCLCHK(ctx, api->GroupStart());
    if (r > 0) {
        NCCLCHK(ctx, api->Send(ctx->route_q, ny, ncclDouble, r - 1, ctx->comm, ctx->stream));
        NCCLCHK(ctx, api->Recv(ctx->route_q + 2 * ny, ny, ncclDouble, r - 1, ctx->comm, ctx->stream));
    }
    if (r < N - 1) {
        NCCLCHK(ctx, api->Send(ctx->route_q + ny, ny, ncclDouble, r + 1, ctx->comm, ctx->stream));
        NCCLCHK(ctx, api->Recv(ctx->route_q + 3 * ny, ny, ncclDouble, r + 1, ctx->comm, ctx->stream));
    }
    NCCLCHK(ctx, api->GroupEnd());
    return RH_OK;
}
static int route_all(rh_ctx *ctx, int which) {
    int rc = rh_route_out(ctx, which);
    if (rc) return rc;
    if (ctx->comm && ctx->comm_nranks > 1) {
        rc = route_exchange(ctx, which);
        if (rc) return rc;
    }
    return rh_route_in(ctx, which);
}
int rh_surface_routing(rh_ctx *ctx) { return route_all(ctx, 0); }
int rh_subsurface_routing(rh_ctx *ctx) { return route_all(ctx, 1); }

#define LAUNCH_PRED(ctx, kern)                                                                                           \
    do {                                                                                                                 \
        planes_touched(ctx);                                                                                             \
        hipLaunchKernelGGL(kern, dim3((ctx)->pred_blocks), dim3(RH_BLOCK), 0, (ctx)->stream, (ctx)->arena, (ctx)->dev, 0); \
    } while (0)

// the next pair of timing events (rh_enable_timing): they ride on a kernel's own dispatch (hipExtLaunchKernelGGL)
static int timing_pair(rh_ctx *ctx, hipEvent_t *ev0, hipEvent_t *ev1) {
    if (ctx->ev_used + 2 > 2 * (size_t)RH_DT_LOG_CAP)
        return fail(ctx, RH_ERR_STATE, "timing: more than 65536 timed steps since rh_enable_timing(1); read the timings and enable again");
    while (ctx->events.size() < ctx->ev_used + 2) {
        hipEvent_t ev;
        HIPCHK(ctx, hipEventCreate(&ev));
        ctx->events.push_back(ev);
    }
    *ev0 = ctx->events[ctx->ev_used];
    *ev1 = ctx->events[ctx->ev_used + 1];
    return RH_OK;
}

static int launch_fused_kernel(rh_ctx *ctx, int monthly, int flags = 0, int *dst64 = nullptr) {
    // Timing: the event pair rides on the kernel's own dispatch (hipExtLaunchKernelGGL: start / stop are taken from the
    // dispatch's completion signal) instead of two hipEventRecord packets around it, which cost 5 us per step at 10^6
    // columns.  -DRH_EVENT_RECORD: the hipEventRecord pair.
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->timing) {
        if (ctx->ev_used + 2 > 2 * (size_t)RH_DT_LOG_CAP)
            return fail(ctx, RH_ERR_STATE, "timing: more than 65536 timed steps since rh_enable_timing(1); read the timings and enable again");
        while (ctx->events.size() < ctx->ev_used + 2) {   // the pool is reused by the next rh_enable_timing(1), never beyond the cap
            hipEvent_t ev;
            HIPCHK(ctx, hipEventCreate(&ev));
            ctx->events.push_back(ev);
        }
        ev0 = ctx->events[ctx->ev_used];
        ev1 = ctx->events[ctx->ev_used + 1];
#ifdef RH_EVENT_RECORD
        HIPCHK(ctx, hipEventRecord(ev0, ctx->stream));
#endif
    }
    // a launch whose tail forms the next step's control part gets one workgroup more: pre_tail (RH_NO_PRE_TAIL=1: the tail does it all)
    static const bool pre_ok = std::getenv("RH_NO_PRE_TAIL") == nullptr;
    if ((flags & RH_TAIL_CTRL) && pre_ok) flags |= RH_TAIL_PRE;
    const dim3 grid(grid_for(ctx->n) + ((flags & RH_TAIL_PRE) ? 1u : 0u)), block(RH_BLOCK);
    const bool lat = ctx->cfg.enable_lateral_flow != 0;
#ifdef RH_EVENT_RECORD
#define RH_LAUNCH_K(K) hipLaunchKernelGGL(K, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, flags, ctx->n_groups, dst64)
#else
#define RH_LAUNCH_K(K) hipExtLaunchKernelGGL(K, grid, block, 0, ctx->stream, ev0, ev1, 0, ctx->arena, ctx->dev, flags, ctx->n_groups, dst64)
#endif
    // lazy rotation: the planes were last touched by a complete fused step (X_m1 == X) and nobody who reads X_m1 planes
    // follows inside this call (the accumulator kernel may, if it was given an X_m1 plane)
    const bool lazy = ctx->lazy_ok && ctx->rot_consistent && !ctx->diag_reads_m1;
    // sparse stores: another step of the same rh_run_steps call follows and nothing in between reads what this one only produces
    const bool sparse = lazy && ctx->sparse_next && monthly < 0;   // (planes an accumulator reads are kept: DevState::keep, the KEEP variant)
    const bool keep = sparse && ctx->diag_reads_sparse;
    ctx->sparse_next = false;
    if (lazy && !ctx->pmask_valid) {   // the lazy kernels read the parameter planes through the wave words: formed from the planes as they are
        hipLaunchKernelGGL(k_param_mask, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, ctx->pmask_buf, ctx->pmask_flags);
        CHECK_LAUNCH(ctx);
        ctx->pmask_valid = true;
    }
#define RH_LAUNCH_STEP(MODE)                                          \
    do {                                                              \
        if (MODE == 2 && lat && keep)                                 \
            RH_LAUNCH_K((k_step<2, true, true, true, true>));         \
        else if (MODE == 2 && keep)                                   \
            RH_LAUNCH_K((k_step<2, false, true, true, true>));        \
        else if (MODE == 2 && lat && sparse)                          \
            RH_LAUNCH_K((k_step<2, true, true, true>));               \
        else if (MODE == 2 && sparse)                                 \
            RH_LAUNCH_K((k_step<2, false, true, true>));              \
        else if (lat && lazy)                                         \
            RH_LAUNCH_K((k_step<MODE, true, true>));                  \
        else if (lat)                                                 \
            RH_LAUNCH_K((k_step<MODE, true, false>));                 \
        else if (lazy)                                                \
            RH_LAUNCH_K((k_step<MODE, false, true>));                 \
        else                                                          \
            RH_LAUNCH_K((k_step<MODE, false, false>));                \
    } while (0)
    if (monthly < 0) RH_LAUNCH_STEP(2);  // decided on the device
    else if (monthly) RH_LAUNCH_STEP(1);
    else RH_LAUNCH_STEP(0);
    ctx->rot_consistent = true;   // a complete step: after_timestep's X_m1 = X holds, physically (eager) or logically (lazy)
    ctx->m1_stale = lazy;
    ctx->outputs_stale = ctx->last_sparse = sparse;
    ctx->call_sparse_steps += sparse ? 1 : 0;
#undef RH_LAUNCH_STEP
#undef RH_LAUNCH_K
    CHECK_LAUNCH(ctx);
    if (ctx->timing) {
#ifdef RH_EVENT_RECORD
        HIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
#endif
        ctx->ev_used += 2;
    }
    ctx->summary_valid = !(flags & RH_TAIL_SKIP);  // the fused kernel's tail leaves the summary word of the state it wrote (words[3])
    ctx->pending_valid = (flags & RH_TAIL_CTRL) != 0;
    ctx->pre_valid = (flags & RH_TAIL_PRE) && !(flags & RH_TAIL_CTRL);
    ctx->pending_hooks = (flags & RH_TAIL_HOOKS) != 0;
    ctx->exch_valid = dst64 != nullptr;
    if (ctx->diag_n) {
        hipLaunchKernelGGL(k_diag, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, 1);
        CHECK_LAUNCH(ctx);
    }
    return RH_OK;
}

int rh_step_phase1(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    if (!ctx->forcing_set) return fail(ctx, RH_ERR_STATE, "rh_set_forcing_day / rh_set_forcing_series must be called before the first step");
    planes_touched(ctx);
    if (ctx->last_front != 1) ctx->agg_daily_stale = ctx->pred_daily_stale = true;   // (the one-launch front formed the day's parts last)
    ctx->last_front = 1;
    hipLaunchKernelGGL(k_pred1, dim3(ctx->pred_blocks), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, ctx->dev, ctx->pred_daily_stale ? 1 : 0);
    ctx->pred_daily_stale = false;
    LAUNCH_WG(ctx, k_reduce, ctx->dev, 0);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
// per-cell forcing: the columns' aggregates of the step (k_cell_agg); does not touch the planes
static void launch_cell_agg(rh_ctx *ctx) {
    const dim3 grid(grid_for(ctx->n)), block(RH_BLOCK);
    const int force = ctx->agg_daily_stale ? 1 : 0;
    if (ctx->n >= ctx->cell_agg_split_min) {
        hipLaunchKernelGGL(k_cell_agg<2>, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, force);
        hipLaunchKernelGGL(k_cell_agg<1>, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, force);
    } else
        hipLaunchKernelGGL(k_cell_agg<0>, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, force);
}
int rh_step_phase2(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    LAUNCH_WG(ctx, k_agg, ctx->dev, 0, 0);
    if (ctx->per_cell) {   // (does not touch the planes: no LAUNCH_CELLS)
        launch_cell_agg(ctx);
        ctx->agg_daily_stale = false;
    }
    LAUNCH_PRED(ctx, k_select);
    LAUNCH_WG(ctx, k_reduce, ctx->dev, 1);
    CHECK_LAUNCH(ctx);
    ctx->summary_valid = false;  // k_select rewrote prec / ta
    return RH_OK;
}
int rh_step_phase3(rh_ctx *ctx, int monthly) {
    if (!ctx) return RH_ERR_ARG;
    LAUNCH_WG(ctx, k_scalars, ctx->dev, 0, 1);
    int rc = launch_fused_kernel(ctx, monthly);
    if (rc) return rc;
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
static void launch_hooks(rh_ctx *ctx);
// single GPU: the same step with the reductions folded into the single-workgroup kernels
static int step_fused_launches(rh_ctx *ctx, int monthly, int hooks) {
    if (ctx->cfg.enable_routing_1D)
        return fail(ctx, RH_ERR_STATE, "enable_routing_1D couples the columns twice per step: the fused step is not available, run the step "
                                       "routine by routine (rh_adaptive_dt ... rh_infiltration, rh_surface_routing, rh_subsurface_runoff, "
                                       "rh_subsurface_routing, ... rh_after_timestep)");
    if (!ctx->forcing_set) return fail(ctx, RH_ERR_STATE, "rh_set_forcing_day / rh_set_forcing_series must be called before the first step");
    if (!ctx->per_cell) {
        // summary path: the previous fused kernel left what the predicates need; one control kernel, one fused kernel
        // ... unless the previous fused kernel's tail has formed this step's control part already (S_next / X_next)
        const bool use_next = ctx->pending_valid && ctx->pending_hooks == (hooks != 0);
        if (!use_next) {
            int src = RH_SRC_WORD3;
            if (!ctx->summary_valid) {
                HIPCHK(ctx, hipMemsetAsync(ctx->dev->sumw, 0, sizeof(ctx->dev->sumw), ctx->stream));
                LAUNCH_CELLS(ctx, k_summary);
                src = RH_SRC_SUMW;
            }
            LAUNCH_ONE(ctx, k_ctrl, ctx->dev, hooks, src, (const int *)nullptr);
            CHECK_LAUNCH(ctx);
        }
        const int flags = (use_next ? RH_TAIL_USE_NEXT : 0) | (ctx->tail_ok ? RH_TAIL_CTRL | (hooks ? RH_TAIL_HOOKS : 0) : 0);
        return launch_fused_kernel(ctx, monthly, flags);
    }
    const bool front = ctx->cell_front_ok && ctx->n <= ctx->cell_front_max;
    if (hooks && !front) {  // per-cell forcing from the resident series: the hooks must have formed it before k_pred1 reads it
        launch_hooks(ctx);
        hooks = 0;
    }
    if (front) {
        // (the set_forcing hook rides along with the front kernel: fresh_day / front_ctrl)
        // ONE per-column launch in front of the fused kernel (k_cell_front; a second one, returning at once unless a new day began, for the
        // daily sums of large grids).  Nothing in front of the fused kernel writes a plane: its lazy rotation stays.
        ctx->summary_valid = false;
        ctx->pending_valid = ctx->pre_valid = false;
        ctx->exch_valid = false;
        const dim3 grid(grid_for(ctx->n)), block(RH_BLOCK);
        const int force = (ctx->front_daily_stale || ctx->last_front != 2) ? 1 : 0, m1 = ctx->m1_stale ? 1 : 0;
        if (ctx->n >= ctx->cell_agg_split_min) {
            hipLaunchKernelGGL(k_cell_front<2>, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, force, m1, ctx->n_groups, hooks);
            hipLaunchKernelGGL(k_cell_front<0>, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, force, m1, ctx->n_groups, hooks);
        } else {
            hipLaunchKernelGGL(k_cell_front<1>, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, force, m1, ctx->n_groups, hooks);
        }
        CHECK_LAUNCH(ctx);
        ctx->front_daily_stale = false;
        ctx->last_front = 2;
        int rc = launch_fused_kernel(ctx, monthly, RH_TAIL_SKIP);   // (the next front reads the planes, not the summary word)
        if (rc) return rc;
        CHECK_LAUNCH(ctx);
        return RH_OK;
    }
    if (ctx->last_front != 1) ctx->agg_daily_stale = ctx->pred_daily_stale = true;
    ctx->last_front = 1;
    // None of the kernels in front of the fused one writes a plane: the selected prec / ta are applied inside the fused kernel
    // (apply_sel = 2, from the per-cell aggregates), the predicate kernels read tau planes only -- with the rotation pending, prec_m1 /
    // swe_m1 are the tau planes themselves.  So the fused kernel keeps its lazy rotation in the per-cell path too.
    ctx->summary_valid = false;
    ctx->pending_valid = ctx->pre_valid = false;
    ctx->exch_valid = false;
    hipLaunchKernelGGL(k_pred1, dim3(ctx->pred_blocks), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, ctx->dev, ctx->pred_daily_stale ? 1 : 0);
    ctx->pred_daily_stale = false;
    LAUNCH_WG(ctx, k_agg, ctx->dev, hooks, 1);
    if (ctx->per_cell) {
        launch_cell_agg(ctx);
        ctx->agg_daily_stale = false;
    }
    const bool defer = ctx->per_cell && ctx->defer_select_ok;   // (shared forcing on this path, e.g. before the series is resident: k_select stores)
    if (!defer) planes_touched(ctx);
    hipLaunchKernelGGL(k_select, dim3(ctx->pred_blocks), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, ctx->dev,
                       (ctx->m1_stale ? RH_SELECT_M1_PENDING : 0) | (defer ? RH_SELECT_DEFER : 0));
    LAUNCH_WG(ctx, k_scalars, ctx->dev, 1, 1, defer ? 2 : 0);
    int rc = launch_fused_kernel(ctx, monthly);
    if (rc) return rc;
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
static int step_summary(rh_ctx *ctx, int32_t *dev_dst64) {
    if (!ctx) return RH_ERR_ARG;
    if (!ctx->forcing_set) return fail(ctx, RH_ERR_STATE, "rh_set_forcing_day / rh_set_forcing_series must be called before the first step");
    if (ctx->per_cell) return fail(ctx, RH_ERR_STATE, "rh_step_summary: the summary path needs forcing shared by all columns; use rh_step_phase1/2/3");
    int src = RH_SRC_WORD3;
    if (!ctx->summary_valid) {
        HIPCHK(ctx, hipMemsetAsync(ctx->dev->sumw, 0, sizeof(ctx->dev->sumw), ctx->stream));
        LAUNCH_CELLS(ctx, k_summary);
        ctx->summary_valid = true;
        src = RH_SRC_SUMW;
    }
    ctx->pending_valid = ctx->pre_valid = false;   // the ranks decide together: the control kernel follows the exchange
    LAUNCH_WG(ctx, k_summary_reduce, ctx->dev, ctx->series_buf ? 1 : 0, (int *)dev_dst64, src);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
static int step_finish(rh_ctx *ctx, int monthly, const int32_t *dev_src64) {
    if (!ctx) return RH_ERR_ARG;
    if (ctx->per_cell) return fail(ctx, RH_ERR_STATE, "rh_step_finish: the summary path needs forcing shared by all columns");
    LAUNCH_ONE(ctx, k_ctrl, ctx->dev, 0, RH_SRC_WORD3, (const int *)dev_src64);
    int rc = launch_fused_kernel(ctx, monthly);
    if (rc) return rc;
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
int rh_step_summary(rh_ctx *ctx) { return step_summary(ctx, nullptr); }
int rh_step_finish(rh_ctx *ctx, int monthly) { return step_finish(ctx, monthly, nullptr); }
int rh_step_summary_expand(rh_ctx *ctx, int32_t *dev_dst64) {
    if (!dev_dst64) return RH_ERR_ARG;
    return step_summary(ctx, dev_dst64);
}
int rh_step_finish_compress(rh_ctx *ctx, int monthly, const int32_t *dev_src64) {
    if (!dev_src64) return RH_ERR_ARG;
    return step_finish(ctx, monthly, dev_src64);
}
int rh_svat_step(rh_ctx *ctx, int monthly) {
    if (!ctx) return RH_ERR_ARG;
    return step_fused_launches(ctx, monthly, 0);
}
int rh_svat_step_scalars(rh_ctx *ctx, int monthly, rh_scalars *s) {
    if (!ctx || !s) return RH_ERR_ARG;
    if (int rc = step_fused_launches(ctx, monthly, 0)) return rc;
    return export_scalars(ctx, s);
}

// stand-alone adaptive time stepping: phases 1-2, the scalar kernel and the pet/ta selection
int rh_adaptive_dt(rh_ctx *ctx) {
    int rc = rh_step_phase1(ctx);
    if (rc) return rc;
    rc = rh_step_phase2(ctx);
    if (rc) return rc;
    LAUNCH_WG(ctx, k_scalars, ctx->dev, 0, 0);
    LAUNCH_CELLS(ctx, k_select_pet);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

// the last part of rh_adaptive_dt on its own: with several ranks the two predicate words are exchanged between rh_step_phase1 /
// rh_step_phase2 and this call (adaptive_time_stepping_dist_safe.py:6-26 gathers 18 fields to rank 0 for the same decision)
int rh_adaptive_dt_finish(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    LAUNCH_WG(ctx, k_scalars, ctx->dev, 0, 0);
    LAUNCH_CELLS(ctx, k_select_pet);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

int rh_set_forcing_series(rh_ctx *ctx, const double *prec, const double *ta, const double *pet, const int64_t *year,
                          const int64_t *month, const int64_t *doy, int64_t nitt_forc) {
    if (!ctx || !prec || !ta || !pet || !year || !month || !doy || nitt_forc <= 0) return RH_ERR_ARG;
    const size_t nb = sizeof(double) * (size_t)nitt_forc;
    if (ctx->series_buf) HIPCHK(ctx, hipFree(ctx->series_buf));
    ctx->series_buf = nullptr;
    HIPCHK(ctx, hipMalloc(&ctx->series_buf, 6 * nb));
    char *base = (char *)ctx->series_buf;
    const void *src[6] = {prec, ta, pet, year, month, doy};
    for (int k = 0; k < 6; ++k) HIPCHK(ctx, hipMemcpyAsync(base + k * nb, src[k], nb, hipMemcpyHostToDevice, ctx->stream));
    const double *sp[3] = {(double *)base, (double *)(base + nb), (double *)(base + 2 * nb)};
    const int64_t *cp[3] = {(int64_t *)(base + 3 * nb), (int64_t *)(base + 4 * nb), (int64_t *)(base + 5 * nb)};
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->series, sp, sizeof(sp), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->calendar, cp, sizeof(cp), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->nitt_forc, &nitt_forc, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->err_flags, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->n_stations, 0, sizeof(int), ctx->stream));   // one series for all columns
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->forcing_set = true;
    ctx->pending_valid = ctx->pre_valid = false;
    ctx->per_cell = false;
    return RH_OK;
}

int rh_set_forcing_stations(rh_ctx *ctx, const double *prec, const double *ta, const double *pet, const int64_t *year, const int64_t *month,
                            const int64_t *doy, int64_t nitt_forc, int n_stations, const int32_t *station_index) {
    if (!ctx || !prec || !ta || !pet || !year || !month || !doy || !station_index || nitt_forc <= 0 || n_stations < 1)
        return ctx ? fail(ctx, RH_ERR_ARG, "rh_set_forcing_stations: bad arguments") : RH_ERR_ARG;
    if (n_stations > 4096) return fail(ctx, RH_ERR_ARG, "rh_set_forcing_stations: at most 4096 stations");
    // the series: (n_stations, nitt_forc) per variable, then the calendar
    const size_t nb1 = sizeof(double) * (size_t)nitt_forc, nbS = nb1 * (size_t)n_stations;
    if (ctx->series_buf) HIPCHK(ctx, hipFree(ctx->series_buf));
    ctx->series_buf = nullptr;
    HIPCHK(ctx, hipMalloc(&ctx->series_buf, 3 * nbS + 3 * nb1));
    char *base = (char *)ctx->series_buf;
    const void *fsrc[3] = {prec, ta, pet}, *csrc[3] = {year, month, doy};
    for (int k = 0; k < 3; ++k) HIPCHK(ctx, hipMemcpyAsync(base + k * nbS, fsrc[k], nbS, hipMemcpyHostToDevice, ctx->stream));
    for (int k = 0; k < 3; ++k) HIPCHK(ctx, hipMemcpyAsync(base + 3 * nbS + k * nb1, csrc[k], nb1, hipMemcpyHostToDevice, ctx->stream));
    const double *sp[3] = {(double *)base, (double *)(base + nbS), (double *)(base + 2 * nbS)};
    const int64_t *cp[3] = {(int64_t *)(base + 3 * nbS), (int64_t *)(base + 3 * nbS + nb1), (int64_t *)(base + 3 * nbS + 2 * nb1)};
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->series, sp, sizeof(sp), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->calendar, cp, sizeof(cp), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->nitt_forc, &nitt_forc, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    // the station of every column, the staging table of a day
    if (!ctx->station_buf) HIPCHK(ctx, hipMalloc((void **)&ctx->station_buf, sizeof(int) * (size_t)ctx->n));
    HIPCHK(ctx, hipMemcpyAsync(ctx->station_buf, station_index, sizeof(int) * (size_t)ctx->n, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->forc_multi_buf) HIPCHK(ctx, hipFree(ctx->forc_multi_buf));
    ctx->forc_multi_buf = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&ctx->forc_multi_buf, sizeof(double) * 3 * (size_t)n_stations * RH_SLOTS_PER_DAY));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->station_idx, &ctx->station_buf, sizeof(int *), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->forc_multi, &ctx->forc_multi_buf, sizeof(double *), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->n_stations, &n_stations, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->err_flags, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->forcing_set = true;
    ctx->pending_valid = ctx->pre_valid = false;
    // the station series reach the columns through the per-cell (weighted) path: neutral weights unless the caller sets some
    if (!ctx->weight_buf[0]) {
        std::vector<double> one((size_t)ctx->n, 1.0), zero((size_t)ctx->n, 0.0);
        const int rc = rh_set_forcing_weights(ctx, one.data(), zero.data(), one.data());
        if (rc) return rc;
    }
    ctx->per_cell = true;
    ctx->agg_daily_stale = true;
    ctx->pred_daily_stale = true;
    ctx->front_daily_stale = true;
    return RH_OK;
}

static void launch_hooks(rh_ctx *ctx) {
    // the hook rewrites D->S (itt_forc, itt_day, the calendar), D->forc and D->monthly: a control part the previous fused kernel's
    // tail formed for the next step (S_next / X_next) was formed BEFORE this hook ran and must not be used
    ctx->pending_valid = ctx->pre_valid = false;
    ctx->exch_valid = false;
    hipLaunchKernelGGL(k_set_forcing, dim3(1), dim3(RH_BLOCK), 0, ctx->stream, ctx->dev);
}
int rh_hooks_phase(rh_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    if (!ctx->series_buf) return fail(ctx, RH_ERR_STATE, "rh_set_forcing_series must be called first");
    launch_hooks(ctx);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

// One whole step with the routing, routine by routine in the order of RogerSetup.step (roger/roger.py:396-457): the columns are
// coupled twice (after the infiltration and after the lateral flow), so the step is eleven per-column passes with two gathers in
// between instead of the fused kernel.  monthly: 1 / 0 = the caller's set_parameters decision, -1 = the device's (after the
// device-side set_forcing hook, rh_run_steps).  Several ranks: the two predicate words of the adaptive time stepping are all-reduced
// over the context's communicator (64 int32 each, as rh_run_steps_dist's summary word), the edge columns go to the x-neighbours.
static int allreduce_word(rh_ctx *ctx, int word) {
    RcclApi *api = rccl_api();
    if (!api->ok) return fail(ctx, RH_ERR_STATE, "rh_step_routed: " + api->why);
    if (!ctx->exch_buf) HIPCHK(ctx, hipMalloc((void **)&ctx->exch_buf, 128 * sizeof(int)));
    int rc = rh_predicates_expand(ctx, word, ctx->exch_buf);
    if (rc) return rc;
    NCCLCHK(ctx, api->AllReduce(ctx->exch_buf, ctx->exch_buf + 64, 64, ncclInt32, ncclMax, ctx->comm, ctx->stream));
    ctx->exch_valid = false;
    return rh_predicates_compress(ctx, word, ctx->exch_buf + 64);
}
// interception ... numerics, itt / time (what rh_step_core is for the uncoupled columns)
static int routed_core(rh_ctx *ctx, bool with_after) {
    int rc;
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->words[2], 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(&ctx->dev->sanity_last, 0, sizeof(unsigned long long), ctx->stream));
    {   // interception ... infiltration, the surface outflow: the longest of the three passes, the one rh_enable_timing times
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (ctx->timing && (rc = timing_pair(ctx, &ev0, &ev1))) return rc;
        planes_touched(ctx);
        hipExtLaunchKernelGGL(k_routed_a, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ev0, ev1, 0, ctx->arena, ctx->dev);
        if (ctx->timing) ctx->ev_used += 2;
    }
    if (ctx->comm && ctx->comm_nranks > 1 && (rc = route_exchange(ctx, 0))) return rc;
    if ((rc = rh_route_gather_only(ctx, 0))) return rc;
    LAUNCH_CELLS(ctx, k_routed_b);                    // the surface inflow, the lateral subsurface runoff, its outflow
    if (ctx->comm && ctx->comm_nranks > 1 && (rc = route_exchange(ctx, 1))) return rc;
    if ((rc = rh_route_gather_only(ctx, 1))) return rc;
    // the subsurface inflow, capillary rise, storages, numerics [, after_timestep: the output accumulators then read the taum1 planes
    // only for variables the rotation has just made equal to tau -- they accumulate tau values, kept by a separate pass otherwise]
    const bool fuse_after = with_after && !ctx->diag_n;
    if (fuse_after) LAUNCH_CELLS(ctx, k_routed_c_after);
    else LAUNCH_CELLS(ctx, k_routed_c);
    LAUNCH_ONE(ctx, k_sanity_to_scalars, ctx->dev);
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, ctx->stream, ctx->dev);
    if (ctx->diag_n) hipLaunchKernelGGL(k_diag, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, ctx->dev, 0);
    if (with_after) {
        if (fuse_after) LAUNCH_ONE(ctx, k_rotate_scalars, ctx->dev);
        else if ((rc = rh_after_timestep(ctx))) return rc;
    }
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
// One routed step of rh_run_steps / rh_run_steps_dist (forcing shared by all columns): control kernel on the summary word (all-reduced
// between the ranks), three passes around the two gathers -- 6 launches instead of 17.
static int routed_step_device(rh_ctx *ctx, bool sparse_wanted = false) {
    int rc;
    // sparse stores: another step of the same rh_run_steps call follows and no accumulator reads the planes in between
    const bool sparse = sparse_wanted && ctx->sparse_ok && !ctx->diag_n && std::getenv("RH_ROUTED_SEPARATE_GATHERS") == nullptr;
    RcclApi *api = nullptr;
    const bool ranks = ctx->comm && ctx->comm_nranks > 1;
    if (ctx->comm) {
        api = rccl_api();
        if (!api->ok) return fail(ctx, RH_ERR_STATE, "routed step: " + api->why);
    }
    if (!ctx->routed_summary) {   // first step, or somebody else touched the planes: the summary bits from the arena
        HIPCHK(ctx, hipMemsetAsync(ctx->dev->sumw, 0, sizeof(ctx->dev->sumw), ctx->stream));
        LAUNCH_CELLS(ctx, k_summary);
    }
    if (ctx->comm) {   // (a one-rank communicator takes the same path: tests)
        if (!ctx->exch_buf) HIPCHK(ctx, hipMalloc((void **)&ctx->exch_buf, 128 * sizeof(int)));
        int *send = ctx->exch_buf, *recv = ctx->exch_buf + 64;
        LAUNCH_WG(ctx, k_summary_reduce, ctx->dev, 0, send, RH_SRC_SUMW);
        NCCLCHK(ctx, api->AllReduce(send, recv, 64, ncclInt32, ncclMax, ctx->comm, ctx->stream));
        LAUNCH_ONE(ctx, k_ctrl, ctx->dev, 1, RH_SRC_WORD3, (const int *)recv);
    } else
        LAUNCH_ONE(ctx, k_ctrl, ctx->dev, 1, RH_SRC_SUMW, (const int *)nullptr);
    planes_touched(ctx);
    {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (ctx->timing && (rc = timing_pair(ctx, &ev0, &ev1))) return rc;
        if (sparse) hipExtLaunchKernelGGL(k_routed_a2<true>, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ev0, ev1, 0, ctx->arena, ctx->dev);
        else hipExtLaunchKernelGGL(k_routed_a2<false>, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ev0, ev1, 0, ctx->arena, ctx->dev);
        if (ctx->timing) ctx->ev_used += 2;
    }
    static const bool separate_gathers = std::getenv("RH_ROUTED_SEPARATE_GATHERS") != nullptr;   // A/B, tests: the 6-launch step
    const dim3 grid(grid_for(ctx->n)), block(RH_BLOCK);
    const int nx = (int)ctx->cfg.nx, ny = (int)ctx->cfg.ny;
    if ((rc = route_check(ctx, 0, "routed step"))) return rc;
    if (ranks && (rc = route_exchange(ctx, 0))) return rc;
    if (separate_gathers) {
        if ((rc = rh_route_gather_only(ctx, 0))) return rc;
        LAUNCH_CELLS(ctx, k_routed_b);
    } else if (sparse)
        hipLaunchKernelGGL(k_routed_bg<true>, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, nx, ny, route_halo_of(ctx));
    else
        hipLaunchKernelGGL(k_routed_bg<false>, grid, block, 0, ctx->stream, ctx->arena, ctx->dev, nx, ny, route_halo_of(ctx));
    if (ranks && (rc = route_exchange(ctx, 1))) return rc;
    if (separate_gathers && (rc = rh_route_gather_only(ctx, 1))) return rc;
    // (the control kernel has advanced itt / time and rotated the scalars, scalars_update; the sanity word stays in words[2], where
    // rh_get_scalars reads it)
    if (ctx->diag_n) {   // the accumulators read the planes between the numerics and the rotation
        if (separate_gathers) LAUNCH_CELLS(ctx, k_routed_c);
        else hipLaunchKernelGGL((k_routed_cg<false, false>), grid, block, 0, ctx->stream, ctx->arena, ctx->dev, nx, ny, route_halo_of(ctx));
        hipLaunchKernelGGL(k_diag, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, ctx->dev, 0);
        LAUNCH_CELLS(ctx, k_after_timestep_oned);
    } else if (separate_gathers)
        LAUNCH_CELLS(ctx, k_routed_c_after);
    else if (sparse)
        hipLaunchKernelGGL((k_routed_cg<true, true>), grid, block, 0, ctx->stream, ctx->arena, ctx->dev, nx, ny, route_halo_of(ctx));
    else
        hipLaunchKernelGGL((k_routed_cg<true, false>), grid, block, 0, ctx->stream, ctx->arena, ctx->dev, nx, ny, route_halo_of(ctx));
    CHECK_LAUNCH(ctx);
    ctx->routed_summary = true;   // k_routed_a2 left the summary bits of the state the step ends in
    ctx->outputs_stale = ctx->last_sparse = sparse;
    ctx->call_sparse_steps += sparse ? 1 : 0;
    return RH_OK;
}
int rh_step_routed(rh_ctx *ctx, int monthly) {
    if (!ctx) return RH_ERR_ARG;
    if (!ctx->cfg.enable_routing_1D) return fail(ctx, RH_ERR_STATE, "rh_step_routed: the context was created without enable_routing_1D");
    int rc;
    if (ctx->comm && ctx->comm_nranks > 1) {
        if ((rc = rh_step_phase1(ctx)) || (rc = allreduce_word(ctx, 0)) || (rc = rh_step_phase2(ctx)) || (rc = allreduce_word(ctx, 1)) ||
            (rc = rh_adaptive_dt_finish(ctx)))
            return rc;
    } else if ((rc = rh_adaptive_dt(ctx)))
        return rc;
    if (monthly < 0) LAUNCH_CELLS(ctx, k_params_surface_if_monthly);
    else if (monthly) LAUNCH_CELLS(ctx, k_params_surface);
    return routed_core(ctx, true);
}

int rh_set_forcing_weights(rh_ctx *ctx, const double *prec_weight, const double *ta_offset, const double *pet_weight) {
    if (!ctx) return RH_ERR_ARG;
    if (!ctx->series_buf) return fail(ctx, RH_ERR_STATE, "rh_set_forcing_series must be called first");
    const double *src[3] = {prec_weight, ta_offset, pet_weight};
    const bool clear = !prec_weight && !ta_offset && !pet_weight;
    if (!clear && (!prec_weight || !ta_offset || !pet_weight)) return fail(ctx, RH_ERR_ARG, "rh_set_forcing_weights: give all three arrays or none");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t nb = sizeof(double) * (size_t)ctx->n;
    const double *dptr[3] = {nullptr, nullptr, nullptr};
    if (clear) {
        for (auto &b : ctx->weight_buf) {
            if (b) HIPCHK(ctx, hipFree(b));
            b = nullptr;
        }
    } else {
        for (int k = 0; k < 3; ++k) {
            if (!ctx->weight_buf[k]) HIPCHK(ctx, hipMalloc((void **)&ctx->weight_buf[k], nb));
            HIPCHK(ctx, hipMemcpyAsync(ctx->weight_buf[k], src[k], nb, hipMemcpyHostToDevice, ctx->stream));
            dptr[k] = ctx->weight_buf[k];
        }
        if (!ctx->agg_cell_buf) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->agg_cell_buf, sizeof(double) * 9 * (size_t)ctx->n));
            HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->agg_cell, &ctx->agg_cell_buf, sizeof(double *), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->weights, dptr, sizeof(dptr), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pending_valid = ctx->pre_valid = false;
    ctx->per_cell = !clear;   // from the next midnight on; rh_set_forcing_weights is a setup-time call
    ctx->agg_daily_stale = true;
    ctx->pred_daily_stale = true;
    ctx->front_daily_stale = true;
    return RH_OK;
}

int rh_set_time_limit(rh_ctx *ctx, int64_t t_end) {
    if (!ctx) return RH_ERR_ARG;
    const long long v = t_end < 0 ? -1 : (long long)t_end;
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->t_end, &v, sizeof(v), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // (v is a stack local)
    ctx->t_end = v;
    ctx->pending_valid = ctx->pre_valid = false;   // a control part formed under the old limit (possibly "halt") is not the next step's
    return RH_OK;
}
// with a time limit: 1 if the limit is reached already (nothing to enqueue), 0 if the first launch of the call will run a step --
// the host-side flags that a fused launch leaves behind (rotation, summary, pending control part) are those of a launch that RAN,
// which holds for every later launch of the call too once the first one did (a halted launch changes nothing on the device)
static int limit_reached(rh_ctx *ctx, bool *reached) {
    *reached = false;
    if (ctx->t_end < 0) return RH_OK;
    if (ctx->cfg.enable_routing_1D || ctx->per_cell)
        return fail(ctx, RH_ERR_STATE, "rh_set_time_limit: the limit is observed by the summary path's control part (forcing shared by all columns, "
                                       "no routing); clear it (rh_set_time_limit(ctx, -1)) and bound the steps from the host");
    int64_t now = 0;
    HIPCHK(ctx, hipMemcpyAsync(&now, &ctx->dev->S.time, sizeof(now), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *reached = now >= ctx->t_end;
    return RH_OK;
}

// sparse_next is the request of ONE enqueued step; whatever way a stepping call ends (a failing launch, the timing cap, forcing that was
// never set), it must not survive the call: the next single-step entry point would run the sparse kernel as a call's last step
struct SparseRequestScope {
    rh_ctx *ctx;
    explicit SparseRequestScope(rh_ctx *c) : ctx(c) { ctx->sparse_next = false; }
    ~SparseRequestScope() { ctx->sparse_next = false; }
};

int rh_run_steps(rh_ctx *ctx, int64_t nsteps) {
    if (!ctx || nsteps < 0) return RH_ERR_ARG;
    if (!ctx->series_buf) return fail(ctx, RH_ERR_STATE, "rh_set_forcing_series must be called first");
    SparseRequestScope sparse_scope(ctx);
    ctx->call_sparse_steps = 0;
    bool over = false;
    if (int rc = limit_reached(ctx, &over)) return rc;
    if (over) return RH_OK;
    for (int64_t k = 0; k < nsteps; ++k) {
        int rc;
        if (ctx->cfg.enable_routing_1D) {
            if (!ctx->per_cell && ctx->routed_device_ok)
                rc = routed_step_device(ctx, k + 1 < nsteps);
            else {   // the hooks, then the step routine by routine (rh_step_routed)
                launch_hooks(ctx);
                rc = rh_step_routed(ctx, -1);
            }
        } else {
            ctx->sparse_next = ctx->sparse_ok && k + 1 < nsteps;   // the call's last step stores every plane
            rc = step_fused_launches(ctx, -1, 1);
        }
        if (rc) return rc;
    }
    return RH_OK;
}

int rh_comm_unique_id(void *id128) {
    if (!id128) return fail(nullptr, RH_ERR_ARG, "rh_comm_unique_id: null pointer");
    RcclApi *api = rccl_api();
    if (!api->ok) return fail(nullptr, RH_ERR_STATE, "rh_comm_unique_id: " + api->why);
    ncclUniqueId id;
    NCCLCHK(nullptr, api->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return RH_OK;
}
int rh_comm_init(rh_ctx *ctx, const void *id128, int nranks, int rank) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return ctx ? fail(ctx, RH_ERR_ARG, "rh_comm_init: bad arguments") : RH_ERR_ARG;
    RcclApi *api = rccl_api();
    if (!api->ok) return fail(ctx, RH_ERR_STATE, "rh_comm_init: " + api->why);
    release_comm(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    NCCLCHK(ctx, api->CommInitRank(&ctx->comm, nranks, id, rank));
    ctx->own_comm = true;
    ctx->comm_nranks = nranks;
    ctx->comm_rank = rank;
    return RH_OK;
}
int rh_set_comm(rh_ctx *ctx, void *nccl_comm) {
    if (!ctx) return RH_ERR_ARG;
    release_comm(ctx);
    ctx->comm = (ncclComm_t)nccl_comm;
    if (ctx->comm) {
        RcclApi *api = rccl_api();
        if (!api->ok) return fail(ctx, RH_ERR_STATE, "rh_set_comm: " + api->why);
        NCCLCHK(ctx, api->CommCount(ctx->comm, &ctx->comm_nranks));
        NCCLCHK(ctx, api->CommUserRank(ctx->comm, &ctx->comm_rank));
    }
    return RH_OK;
}
int rh_comm_info(rh_ctx *ctx, int *nranks, int *rank) {
    if (!ctx || !nranks || !rank) return ctx ? fail(ctx, RH_ERR_ARG, "rh_comm_info: null pointer") : RH_ERR_ARG;
    *nranks = 1;
    *rank = 0;
    if (!ctx->comm) return RH_OK;
    RcclApi *api = rccl_api();
    if (!api->ok) return fail(ctx, RH_ERR_STATE, "rh_comm_info: " + api->why);
    NCCLCHK(ctx, api->CommCount(ctx->comm, nranks));
    NCCLCHK(ctx, api->CommUserRank(ctx->comm, rank));
    return RH_OK;
}
int rh_run_steps_dist(rh_ctx *ctx, int64_t nsteps) {
    if (!ctx || nsteps < 0) return RH_ERR_ARG;
    if (!ctx->series_buf) return fail(ctx, RH_ERR_STATE, "rh_set_forcing_series must be called first");
    if (!ctx->comm) return fail(ctx, RH_ERR_STATE, "rh_run_steps_dist: no communicator (rh_comm_init / rh_set_comm)");
    SparseRequestScope sparse_scope(ctx);
    ctx->call_sparse_steps = 0;
    bool over = false;
    if (int rc = limit_reached(ctx, &over)) return rc;
    if (over) return RH_OK;
    if (ctx->cfg.enable_routing_1D) {   // the routed step exchanges its predicate words and edge columns itself
        for (int64_t k = 0; k < nsteps; ++k) {
            int rc;
            if (!ctx->per_cell && ctx->routed_device_ok)
                rc = routed_step_device(ctx, k + 1 < nsteps);
            else {
                launch_hooks(ctx);
                rc = rh_step_routed(ctx, -1);
            }
            if (rc) return rc;
        }
        return RH_OK;
    }
    if (ctx->per_cell) return fail(ctx, RH_ERR_STATE, "rh_run_steps_dist: the one-exchange step needs forcing shared by all columns (rh_step_phase1/2/3 otherwise)");
    RcclApi *api = rccl_api();
    if (!api->ok) return fail(ctx, RH_ERR_STATE, "rh_run_steps_dist: " + api->why);
    if (!ctx->exch_buf) HIPCHK(ctx, hipMalloc((void **)&ctx->exch_buf, 128 * sizeof(int)));
    int *send = ctx->exch_buf, *recv = ctx->exch_buf + 64;
    for (int64_t k = 0; k < nsteps; ++k) {
        if (!ctx->exch_valid) {   // first step, or the host touched the planes: the summary word from words[3] or from the arena
            int src = RH_SRC_WORD3;
            if (!ctx->summary_valid) {
                HIPCHK(ctx, hipMemsetAsync(ctx->dev->sumw, 0, sizeof(ctx->dev->sumw), ctx->stream));
                LAUNCH_CELLS(ctx, k_summary);
                ctx->summary_valid = true;
                src = RH_SRC_SUMW;
            }
            LAUNCH_WG(ctx, k_summary_reduce, ctx->dev, 0, send, src);
            CHECK_LAUNCH(ctx);
        }
        static const bool copy_only = std::getenv("RH_DIST_COPY_ONLY") != nullptr;   // timing experiment (one rank): the exchange as a plain copy
        if (copy_only && ctx->comm_nranks == 1)
            HIPCHK(ctx, hipMemcpyAsync(recv, send, 64 * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        else
            NCCLCHK(ctx, api->AllReduce(send, recv, 64, ncclInt32, ncclMax, ctx->comm, ctx->stream));
        // the fused launch in front of the exchange formed the columns-independent half of this control part (pre_tail): the kernel behind
        // the exchange keeps the decisions
        LAUNCH_ONE(ctx, k_ctrl, ctx->dev, 1, RH_SRC_WORD3, (const int *)recv, ctx->pre_valid ? 1 : 0);
        CHECK_LAUNCH(ctx);
        ctx->sparse_next = ctx->sparse_ok && k + 1 < nsteps;
        static const bool pre_ok = std::getenv("RH_NO_PRE_TAIL") == nullptr;
        int rc = launch_fused_kernel(ctx, -1, pre_ok ? (RH_TAIL_PRE | RH_TAIL_HOOKS) : 0, send);   // the tail spreads the next step's summary word into `send`
        if (rc) return rc;
    }
    return RH_OK;
}

int rh_diag_configure(rh_ctx *ctx, const int *rate_planes, int n_rate, const int *collect_planes, int n_collect, int n_slots) {
    if (!ctx) return RH_ERR_ARG;
    if (n_rate < 0 || n_collect < 0 || n_rate + n_collect > 32 || n_slots < 1 || (n_rate && !rate_planes) || (n_collect && !collect_planes))
        return fail(ctx, RH_ERR_ARG, "rh_diag_configure: bad counts (n_rate + n_collect <= 32, n_slots >= 1)");
    int planes[32];
    for (int j = 0; j < n_rate + n_collect; ++j) {
        planes[j] = j < n_rate ? rate_planes[j] : collect_planes[j - n_rate];
        if (planes[j] < 0 || planes[j] >= ctx->planes_held || PLANE_IS_INT[planes[j]])
            return fail(ctx, RH_ERR_ARG, "rh_diag_configure: plane ids must name float64 planes");
    }
    // an accumulated pure-output plane must be in memory after every step: the sparse kernel keeps storing THOSE planes (DevState::keep)
    ctx->diag_reads_sparse = false;
    {
        unsigned long long keep[(RH_NPLANES + 63) / 64] = {};
        for (int j = 0; j < n_rate + n_collect; ++j)
            if (pure_output_planes()[ctx->cfg.enable_lateral_flow ? 1 : 0][planes[j]]) {
                ctx->diag_reads_sparse = true;
                keep[planes[j] >> 6] |= 1ull << (planes[j] & 63);
            }
        const int any = ctx->diag_reads_sparse ? 1 : 0;
        HIPCHK(ctx, hipMemcpyAsync(ctx->dev->keep, keep, sizeof(keep), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->keep_any, &any, sizeof(any), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->diag_reads_m1 = false;   // an accumulated X_m1 plane keeps the fused kernel from skipping its stores
    for (int j = 0; j < n_rate + n_collect; ++j) {
        const size_t len = std::strlen(PLANE_NAMES[planes[j]]);
        if (len > 3 && !std::strcmp(PLANE_NAMES[planes[j]] + len - 3, "_m1")) ctx->diag_reads_m1 = true;
    }
    materialise_m1(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->diag_buf) {
        HIPCHK(ctx, hipFree(ctx->diag_buf));
        ctx->diag_buf = nullptr;
    }
    if (ctx->diag_steps_buf) {
        HIPCHK(ctx, hipFree(ctx->diag_steps_buf));
        ctx->diag_steps_buf = nullptr;
    }
    const int nv = n_rate + n_collect;
    ctx->diag_n = nv;
    ctx->diag_slots = n_slots;
    if (nv) {
        const size_t bytes = (size_t)n_slots * nv * ctx->n * sizeof(double);
        HIPCHK(ctx, hipMalloc((void **)&ctx->diag_buf, bytes));
        HIPCHK(ctx, hipMemsetAsync(ctx->diag_buf, 0, bytes, ctx->stream));
        HIPCHK(ctx, hipMalloc((void **)&ctx->diag_steps_buf, (size_t)n_slots * 3 * sizeof(long long)));
        HIPCHK(ctx, hipMemsetAsync(ctx->diag_steps_buf, 0xff, (size_t)n_slots * 3 * sizeof(long long), ctx->stream));   // -1: never touched
    }
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->diag_steps, &ctx->diag_steps_buf, sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    {
        const long long day = 86400;
        if (ctx->diag_interval <= 0) ctx->diag_interval = day;
        HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->diag_interval, &ctx->diag_interval, sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->diag, &ctx->diag_buf, sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->diag_rate, &n_rate, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->diag_collect, &n_collect, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->diag_slots, &n_slots, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dev->diag_planes, planes, sizeof(int) * 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the sources are stack locals
    return RH_OK;
}
static int diag_check(rh_ctx *ctx, int j, int slot) {
    if (!ctx) return RH_ERR_ARG;
    if (!ctx->diag_n) return fail(ctx, RH_ERR_STATE, "rh_diag_configure has not been called");
    if (j < 0 || j >= ctx->diag_n || slot < 0 || slot >= ctx->diag_slots) return fail(ctx, RH_ERR_ARG, "rh_diag: variable or slot out of range");
    return RH_OK;
}
int rh_diag_download(rh_ctx *ctx, int j, int slot, double *host, size_t bytes) {
    const int rc = diag_check(ctx, j, slot);
    if (rc) return rc;
    if (!host || bytes != (size_t)ctx->n * sizeof(double)) return fail(ctx, RH_ERR_ARG, "rh_diag_download: size mismatch");
    HIPCHK(ctx, hipMemcpyAsync(host, ctx->diag_buf + ((size_t)slot * ctx->diag_n + j) * ctx->n, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}
int rh_diag_upload(rh_ctx *ctx, int j, int slot, const double *host, size_t bytes) {
    const int rc = diag_check(ctx, j, slot);
    if (rc) return rc;
    if (!host || bytes != (size_t)ctx->n * sizeof(double)) return fail(ctx, RH_ERR_ARG, "rh_diag_upload: size mismatch");
    HIPCHK(ctx, hipMemcpyAsync(ctx->diag_buf + ((size_t)slot * ctx->diag_n + j) * ctx->n, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}
int rh_diag_set_slot_state(rh_ctx *ctx, int slot, int64_t steps, int64_t t_start, int64_t t_end) {
    const int rc = diag_check(ctx, 0, slot);
    if (rc) return rc;
    const long long v[3] = {(long long)steps, (long long)t_start, (long long)t_end};
    HIPCHK(ctx, hipMemcpyAsync(ctx->diag_steps_buf + 3 * slot, v, sizeof(v), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}
int rh_diag_steps(rh_ctx *ctx, int slot, int64_t *steps) {
    const int rc = diag_check(ctx, 0, slot);
    if (rc) return rc;
    if (!steps) return fail(ctx, RH_ERR_ARG, "rh_diag_steps: null pointer");
    long long v = 0;
    HIPCHK(ctx, hipMemcpyAsync(&v, ctx->diag_steps_buf + 3 * slot, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *steps = (int64_t)(v < 0 ? 0 : v);
    return RH_OK;
}
int rh_diag_set_interval(rh_ctx *ctx, int64_t seconds) {
    if (!ctx) return RH_ERR_ARG;
    if (seconds != 86400 && seconds != 3600 && seconds != 600)
        return fail(ctx, RH_ERR_ARG, "rh_diag_set_interval: the output interval is a day, an hour or ten minutes (the step classes)");
    ctx->diag_interval = seconds;
    if (ctx->diag_n) {
        HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->diag_interval, &ctx->diag_interval, sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->diag_steps_buf, 0xff, (size_t)ctx->diag_slots * 3 * sizeof(long long), ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RH_OK;
}
int rh_diag_slot_times(rh_ctx *ctx, int slot, int64_t *t_start, int64_t *t_end) {
    const int rc = diag_check(ctx, 0, slot);
    if (rc) return rc;
    if (!t_start || !t_end) return fail(ctx, RH_ERR_ARG, "rh_diag_slot_times: null pointer");
    long long v[3] = {0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(v, ctx->diag_steps_buf + 3 * slot, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *t_start = (int64_t)v[1];
    *t_end = (int64_t)v[2];
    return RH_OK;
}
void *rh_diag_device_ptr(rh_ctx *ctx, int j, int slot) {
    if (diag_check(ctx, j, slot)) return nullptr;
    return ctx->diag_buf + ((size_t)slot * ctx->diag_n + j) * ctx->n;
}

int rh_predicates_expand(rh_ctx *ctx, int word, int32_t *dev_dst64) {
    if (!ctx || word < 0 || word > 3 || !dev_dst64) return RH_ERR_ARG;
    hipLaunchKernelGGL(k_words_expand, dim3(1), dim3(64), 0, ctx->stream, ctx->dev, word, dev_dst64);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}
int rh_predicates_compress(rh_ctx *ctx, int word, const int32_t *dev_src64) {
    if (!ctx || word < 0 || word > 3 || !dev_src64) return RH_ERR_ARG;
    hipLaunchKernelGGL(k_words_compress, dim3(1), dim3(64), 0, ctx->stream, ctx->dev, word, dev_src64);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

int rh_calibrate_copy(rh_ctx *ctx, int src_plane0, int dst_plane0, int nplanes) {
    if (!ctx || nplanes <= 0 || src_plane0 < 0 || dst_plane0 < 0 || src_plane0 + nplanes > ctx->planes_held ||
        dst_plane0 + nplanes > ctx->planes_held)
        return RH_ERR_ARG;
    for (int p = 0; p < nplanes; ++p)
        if (PLANE_IS_INT[src_plane0 + p] || PLANE_IS_INT[dst_plane0 + p]) return fail(ctx, RH_ERR_ARG, "calibration planes must be float64");
    planes_touched(ctx);
    Arena probe = ctx->arena;
    if (const char *sh = std::getenv("RH_CALIB_SHIFT_KB")) {   // experiment (tools/arena_phase.py): the same copy shifted inside the allocation's padding
        const size_t shift = (size_t)std::atoll(sh) * 1024;
        if (shift > ctx->arena_pad) return fail(ctx, RH_ERR_ARG, "RH_CALIB_SHIFT_KB exceeds RH_ARENA_PAD_KB");
        probe.base = ctx->arena.base + shift;
    }
    hipLaunchKernelGGL(k_calib_copy, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, probe, src_plane0, dst_plane0, nplanes);
    CHECK_LAUNCH(ctx);
    return RH_OK;
}

// Experiment (tools/swap_levels.py): two contexts of the same shape exchange their arenas -- does the fused kernel's speed level follow the
// arena or the rest of the context?  The caller has brought both to the same state (same steps from the same start).
int rh_debug_swap_arenas(rh_ctx *a, rh_ctx *b) {
    if (!a || !b || a->n != b->n || a->arena.stride != b->arena.stride || a->planes_held != b->planes_held) return RH_ERR_ARG;
    HIPCHK(a, hipStreamSynchronize(a->stream));
    HIPCHK(b, hipStreamSynchronize(b->stream));
    std::swap(a->arena.base, b->arena.base);
    std::swap(a->arena_alloc, b->arena_alloc);
    std::swap(a->arena_offset, b->arena_offset);
    std::swap(a->arena_pad, b->arena_pad);
    a->pmask_valid = b->pmask_valid = false;   // (the wave words describe the planes of the arena a context steps on)
    return RH_OK;
}

// rh_pow on the device for n argument pairs (tests: the same bits as the host's compilation of rh_pow.h)
__global__ void k_selftest_rh_pow(const double *x, const double *y, double *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rh_pow(x[i], y[i]);
}
int rh_selftest_pow(const double *x, const double *y, double *out, int64_t n) {
    if (!x || !y || !out || n <= 0) return RH_ERR_ARG;
    double *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * 3 * sizeof(double)) != hipSuccess) return RH_ERR_HIP;
    int rc = RH_OK;
    if (hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        rc = RH_ERR_HIP;
    if (rc == RH_OK) {
        hipLaunchKernelGGL(k_selftest_rh_pow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, d + n, d + 2 * n, n);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(out, d + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = RH_ERR_HIP;
    }
    (void)hipFree(d);
    return rc;
}

// np_sum144_window for n window starts over one 144-vector, one wavefront per start: out[2 j] by the kernels' function (the rotation path
// where it applies), out[2 j + 1] by the general path (tests: both are numpy's sum over the masked vector, bit for bit)
__global__ void k_selftest_window(const double *v, const int64_t *itd, double *out) {
    const int64_t t = itd[blockIdx.x];
    auto get = [&](int k) { return v[k]; };
    const double a = np_sum144_window(get, t), b = np_sum144_window_general(get, t);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = a;
        out[2 * blockIdx.x + 1] = b;
    }
}
int rh_selftest_window_sum(const double *v144, const int64_t *itd, int64_t n, double *out2n) {
    if (!v144 || !itd || !out2n || n <= 0) return RH_ERR_ARG;
    char *d = nullptr;
    const size_t bv = RH_SLOTS_PER_DAY * sizeof(double), bi = (size_t)n * sizeof(int64_t), bo = (size_t)n * 2 * sizeof(double);
    if (hipMalloc((void **)&d, bv + bi + bo) != hipSuccess) return RH_ERR_HIP;
    int rc = RH_OK;
    if (hipMemcpy(d, v144, bv, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d + bv, itd, bi, hipMemcpyHostToDevice) != hipSuccess) rc = RH_ERR_HIP;
    if (rc == RH_OK) {
        hipLaunchKernelGGL(k_selftest_window, dim3((unsigned)n), dim3(64), 0, 0, (const double *)d, (const int64_t *)(d + bv), (double *)(d + bv + bi));
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(out2n, d + bv + bi, bo, hipMemcpyDeviceToHost) != hipSuccess)
            rc = RH_ERR_HIP;
    }
    (void)hipFree(d);
    return rc;
}

int rh_placement_report(const rh_ctx *ctx, double *ms, int cap) {
    if (!ctx) return 0;
    const int n = (int)ctx->probe_ms.size();
    for (int k = 0; k < n && k < cap && ms; ++k) ms[k] = ctx->probe_ms[k];
    return n;
}

// planes NO variant of the fused step reads (the sparse kernels additionally leave out the state the next lazy step derives itself:
// RH_LAZY_DERIVED_FIELDS, which the eager kernel still loads)
int rh_plane_is_pure_output(int model, int plane) {
    static const std::vector<unsigned char> tab[2] = {
        [] { std::vector<unsigned char> t(RH_NPLANES, 0);
#define RH_MARK(name) t[RH_P_##name] = 1;
             RH_NEVER_READ_FIELDS_SVAT(RH_MARK) return t; }(),
        [] { std::vector<unsigned char> t(RH_NPLANES, 0);
             RH_NEVER_READ_FIELDS_ONED(RH_MARK)
#undef RH_MARK
             return t; }()};
    if (plane < 0 || plane >= RH_NPLANES || model < 0 || model > 2) return -1;
    return model == 2 ? pure_output_planes()[2][plane] : tab[model][plane];
}
int64_t rh_sparse_steps(const rh_ctx *ctx) { return ctx ? ctx->call_sparse_steps : 0; }
int rh_param_stats(rh_ctx *ctx, double *derived_fraction, double *uniform_bytes_per_cell) {
    if (!ctx || !derived_fraction || !uniform_bytes_per_cell) return ctx ? fail(ctx, RH_ERR_ARG, "rh_param_stats: null pointer") : RH_ERR_ARG;
    if (!ctx->pmask_valid) {
        hipLaunchKernelGGL(k_param_mask, dim3(grid_for(ctx->n)), dim3(RH_BLOCK), 0, ctx->stream, ctx->arena, ctx->dev, ctx->pmask_buf, ctx->pmask_flags);
        CHECK_LAUNCH(ctx);
        ctx->pmask_valid = true;
    }
    const size_t words = ((size_t)ctx->n + 63) / 64;
    std::vector<unsigned long long> w(words);
    HIPCHK(ctx, hipMemcpyAsync(w.data(), ctx->pmask_buf, words * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // the parameter planes the step loads unless the month changes, by element size; the derived ones apart
    unsigned long long f64_bits = 0, i32_bits = 0, derived_bits = 0;
#define RH_PB(name, bit) const unsigned long long pbit_##name = 1ull << bit;
    RH_PARAM_BITS(RH_PB)
#undef RH_PB
#define RH_PL(name) (PLANE_IS_INT[RH_P_##name] ? i32_bits : f64_bits) |= pbit_##name;
    if (ctx->cfg.enable_lateral_flow) { RH_PARAM_LOADED_ONED(RH_PL) } else { RH_PARAM_LOADED_SVAT(RH_PL) }
#undef RH_PL
#define RH_PD(name) derived_bits |= pbit_##name;
    RH_DERIVED_FIELDS(RH_PD)
#undef RH_PD
    double derived = 0, bytes = 0;
    for (size_t k = 0; k < words; ++k) {
        const bool der = (w[k] >> 63) & 1ull;
        derived += der ? 1 : 0;
        const unsigned long long u = w[k] & (der ? ~derived_bits : ~0ull);   // (a derived plane is not loaded at all)
        bytes += 8.0 * __builtin_popcountll(u & f64_bits) + 4.0 * __builtin_popcountll(u & i32_bits);
    }
    *derived_fraction = derived / (double)words;
    *uniform_bytes_per_cell = bytes / (double)words;
    return RH_OK;
}
int rh_step_mode(const rh_ctx *ctx) {
    if (!ctx) return 0;
    return (ctx->m1_stale ? RH_STEP_MODE_LAZY : 0) | (ctx->pending_valid ? RH_STEP_MODE_TAIL : 0) | (ctx->last_sparse ? RH_STEP_MODE_SPARSE : 0);
}

void *rh_predicate_words(rh_ctx *ctx) { return ctx ? (void *)ctx->dev->words : nullptr; }

int rh_enable_timing(rh_ctx *ctx, int on) {
    if (!ctx) return RH_ERR_ARG;
    ctx->timing = on != 0;
    ctx->ev_used = 0;
    ctx->pending_valid = ctx->pre_valid = false;   // the step log restarts: the next step's entry must be written after this call
    if (on && !ctx->dt_log_buf) HIPCHK(ctx, hipMalloc((void **)&ctx->dt_log_buf, sizeof(int) * RH_DT_LOG_CAP));
    int *log = on ? ctx->dt_log_buf : nullptr;
    const int cap = RH_DT_LOG_CAP, zero = 0;
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->dt_log, &log, sizeof(int *), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->dt_log_cap, &cap, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ctx->dev->dt_log_n, &zero, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}
int rh_timing_detail(rh_ctx *ctx, double *kernel_ms, int32_t *dt_secs, int64_t cap, int64_t *launches) {
    if (!ctx || !kernel_ms || !dt_secs || !launches || cap < 0) return RH_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    int logged = 0;
    if (ctx->dt_log_buf) HIPCHK(ctx, hipMemcpy(&logged, &ctx->dev->dt_log_n, sizeof(int), hipMemcpyDeviceToHost));
    const int64_t n = (int64_t)(ctx->ev_used / 2);
    // (the tail of the last timed kernel has logged the step after it already: one entry more than launches)
    if ((logged != n && logged != n + 1) || n > RH_DT_LOG_CAP)
        return fail(ctx, RH_ERR_STATE, "rh_timing_detail: the step log does not match the timed launches (timing enabled mid-step, "
                                       "or more than 65536 steps)");
    *launches = n;
    const int64_t m = n < cap ? n : cap;
    for (int64_t k = 0; k < m; ++k) {
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->events[2 * k], ctx->events[2 * k + 1]));
        kernel_ms[k] = ms;
    }
    if (m) HIPCHK(ctx, hipMemcpy(dt_secs, ctx->dt_log_buf, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
    return RH_OK;
}
int rh_timing_summary(rh_ctx *ctx, double *total_ms, int64_t *launches) {
    if (!ctx || !total_ms || !launches) return RH_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0;
    for (size_t k = 0; k + 1 < ctx->ev_used; k += 2) {
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->events[k], ctx->events[k + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)(ctx->ev_used / 2);
    return RH_OK;
}

}  // extern "C"
