// rh_col.h -- one soil column ("cell") in registers, and the SoA device arena it is loaded from.
//
// Data layout in HBM: one arena per GPU holding RH_NPLANES planes (float64 or int32) over the rank's interior (x, y)
// grid in C order, laid out in tiles of 64 cells (see Arena below).  Lane l of a wavefront owns cell blockIdx*256 +
// 64*wave + l, so every plane access is one fully coalesced 512-byte (float64) request per wave-instruction; nothing is
// re-read within a kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "roger_hip.h"

#define RH_DEV __device__ __forceinline__

struct Col {
#define RH_DECL_F64_1(name) double name;
#define RH_DECL_F64_2(name) double name, name##_m1;
#define RH_DECL_I32_1(name) int name;
#define RH_DECL_I32_2(name) int name, name##_m1;
#define RH_FIELD(name, type, levels) RH_DECL_##type##_##levels(name)
#include "rh_fields.def"
#undef RH_FIELD
#undef RH_DECL_F64_1
#undef RH_DECL_F64_2
#undef RH_DECL_I32_1
#undef RH_DECL_I32_2
};

// RH_TILED = 1 (default): the arena is a sequence of tiles of 64 cells (one wavefront's columns); inside a tile every
// plane has a 512-byte slot (64 float64, or 64 int32 in its first half), slots in plane order.  Everything one
// wavefront loads and stores during a step lies in ONE contiguous span of RH_NPLANES * 512 bytes (77 KB) instead of
// RH_NPLANES addresses 8 MB apart: one or two address translations per wave instead of one per plane, and consecutive
// plane accesses fall into the same DRAM pages.  A plane access is still one fully coalesced 512-byte request.
// RH_TILED = 0: plane-major (plane p at base + p * stride), kept for experiments.
#ifndef RH_TILED
#define RH_TILED 1
#endif
// RH_TILE_CELLS: columns per tile, 64 (a wavefront's) or a multiple of it up to the workgroup's 256 -- then a plane's slot holds the
// columns of 2 or 4 consecutive wavefronts of a workgroup (1 / 2 KiB contiguous per plane and workgroup instead of 512-byte pieces).
// 256 since the end of round 3: with the kernel's arithmetic cut by rh_pow the longer pieces show a little -- alternating with the 64-column
// library inside one call, SVAT 10^6 columns 0.2143 -> 0.2065 and 0.2192 -> 0.2118 ms (medians of 3 and 4 pairs), 10^7 columns 1.959 -> 1.942
// and 1.991 -> 1.962 ms, oneD alike; the driver's 20-step command and the 80 x 53 grid within +- 1 %, the routed step 3 % slower (0.396 ->
// 0.408 ms).  With the library's pow the three sizes were within the noise.  The whole GPU suite runs on either.
#ifndef RH_TILE_CELLS
#define RH_TILE_CELLS 256
#endif
#define RH_TILE_SHIFT (RH_TILE_CELLS == 64 ? 6 : (RH_TILE_CELLS == 128 ? 7 : 8))
#define RH_SLOT_BYTES (RH_TILE_CELLS * 8)
static_assert(RH_TILE_CELLS == 64 || RH_TILE_CELLS == 128 || RH_TILE_CELLS == 256, "RH_TILE_CELLS: 64, 128 or 256");

struct Arena {
    char *base;
    size_t stride;  // tiled: bytes per tile (RH_NPLANES * RH_SLOT_BYTES); plane-major: bytes between planes
    int64_t n;      // cells
};

// Address of cell i of a plane.  Tiled: the tile index is uniform over the wavefront (every kernel maps lane l of a
// wave to cell 64 * k + l), so it is taken from the first active lane and the whole tile/plane part of the address is
// scalar arithmetic; the per-lane part is lane * element size.
template <typename T>
RH_DEV T *rh_cell(const Arena &a, int plane, int64_t i) {
#if RH_TILED
    // (uniform over the wavefront: the tile index and, inside a tile of several wavefronts, the wavefront's 64-column piece)
    const int tile = __builtin_amdgcn_readfirstlane((int)(i >> RH_TILE_SHIFT));
    const int piece = __builtin_amdgcn_readfirstlane((int)(i & (RH_TILE_CELLS - 1)) & ~63);
    return reinterpret_cast<T *>(a.base + (size_t)tile * a.stride + (size_t)plane * RH_SLOT_BYTES) + piece + (int)(i & 63);
#else
    return reinterpret_cast<T *>(a.base + (size_t)plane * a.stride) + i;
#endif
}
// Address of ANY cell of a plane (the routing's gather reads the eight neighbours: the tile is not uniform over the wavefront).
template <typename T>
RH_DEV T *rh_cell_any(const Arena &a, int plane, int64_t i) {
#if RH_TILED
    return reinterpret_cast<T *>(a.base + (size_t)(i >> RH_TILE_SHIFT) * a.stride + (size_t)plane * RH_SLOT_BYTES) + (int)(i & (RH_TILE_CELLS - 1));
#else
    return reinterpret_cast<T *>(a.base + (size_t)plane * a.stride) + i;
#endif
}
// RH_NT: plane accesses as non-temporal (streaming) loads / stores -- every plane is touched once per kernel, nothing is worth
// keeping in the caches.  bit 0: loads, bit 1: stores.  Measured on the fused step at 10^6 columns, alternating in one call
// (tools/ab_variants.sh): 0.3158 ms plain, 0.3136 loads only, 0.3108 stores only, 0.3016 both (- 4.5 %); a plain copy with the
// same access shape gains 3 - 5 % (tools/experiments/bw_probe.hip).
#ifndef RH_NT
#define RH_NT 3
#endif
RH_DEV void rh_ld(const Arena &a, int plane, int64_t i, double &dst) {
    dst = (RH_NT & 1) ? __builtin_nontemporal_load(rh_cell<const double>(a, plane, i)) : *rh_cell<const double>(a, plane, i);
}
RH_DEV void rh_ld(const Arena &a, int plane, int64_t i, int &dst) {
    dst = (RH_NT & 1) ? __builtin_nontemporal_load(rh_cell<const int>(a, plane, i)) : *rh_cell<const int>(a, plane, i);
}
RH_DEV void rh_st(const Arena &a, int plane, int64_t i, double v) {
    if (RH_NT & 2) __builtin_nontemporal_store(v, rh_cell<double>(a, plane, i));
    else *rh_cell<double>(a, plane, i) = v;
}
RH_DEV void rh_st(const Arena &a, int plane, int64_t i, int v) {
    if (RH_NT & 2) __builtin_nontemporal_store(v, rh_cell<int>(a, plane, i));
    else *rh_cell<int>(a, plane, i) = v;
}

// Settings that the kernels read (subset of rh_config, device copy).
struct Consts {
    double pi, r_mp, l_sc, sf, ta_fm, rmax, transp_water_stress, atol, rtol;
    double clay_min, clay_max, theta_rew_min, theta_rew_max, rew_min, rew_max;
    double z_evap_max, zroot_to_zsoil_max, a_bc, b_bc;
    int64_t end_event, hpi;
    double dx;      // grid spacing in m (settings.dx), enters the lateral flow rates
    int lateral;    // settings.enable_lateral_flow (oneD model)
    double dy;      // settings.dy (routing)
    int routing;    // settings.enable_routing_1D
};

// Look-up tables, row-major (roger/lookuptables.py).
struct Luts {
    double ilu[25 * 13];
    double gc[25 * 13];
    double gcm[25 * 2];
    double rdlu[25 * 7];
};

// Per-step uniform values produced on the device by the scalar kernels and consumed by the
// per-cell kernels (never round-trips to the host).
struct StepCtx {
    double dt;
    double agg[9];        // shared-forcing aggregates: {prec, ta, pet} x {daily, hourly, 10 min}
    int64_t month_tau;
    int sel_daily, sel_hourly, sel_10min;  // prec/ta selection, adaptive_time_stepping.py:128-189
    int sel_p;                             // the one that wins (applied last): -1 none, 0 daily, 1 hourly, 2 10 min
    double prec_sel, ta_sel;               // its values when the forcing is shared by all columns
    int sel_w;                             // pet/ta selection of cond6..11: -1 none, 0 daily, 1 hourly, 2 10 min
    double pet_sel_w, ta_sel_w;            // its values when the forcing is shared
    int cond1, cond2, cond3, cond4, cond5; // calculate_infiltration, infiltration.py:2155-2167
    int cond_time;
    int64_t dt_secs_prelim;
    int64_t itt_day;
    int apply_sel;        // 1: the fused kernel applies the prec/ta selection itself (summary path); 2: from the per-cell aggregates; 0: k_select did
    int forc_exhausted;   // the device-side set_forcing hook found midnight beyond the end of the resident series
    // rh_set_time_limit (RogerSetup.run(): `while vs.time - start_time < runlen`, roger/roger.py:548-556, decided on the device):
    int halt;             // the time limit was reached before this step: the launch does nothing
    int last;             // this step reaches the limit: it stores every plane (never the sparse variant)
};

// predicate bit positions, word 0 (start of step) and word 1 (after prec/ta selection)
enum {
    PB_SWE_NOT_LE0 = 0, PB_SWE_GT0, PB_SWETOP_NOT_LE0, PB_SWETOP_GT0, PB_P_NOT_LE0, PB_P_GT0, PB_P_GT_HPI,
    PB_P_NOT_LE_HPI, PB_TA_NOT_GT, PB_TA_GT, PB_PGT0_TALE, PB_NOT_PLE0_TALE
};
// Summary bits a column contributes at the END of a step (fused kernel epilogue) from which the NEXT step's
// predicate words 0 and 1 are derived without another pass over the columns (shared forcing): bits 0..3 are
// word 0's column bits; the *_KEEP / P_* bits are word 1's column terms for a step that keeps prec/ta
// (sel_p < 0); swe and prec double as next step's swe[taum1] and prec[taum1].
enum {
    QB_SWE_NOT_LE0 = 0, QB_SWE_GT0, QB_SWETOP_NOT_LE0, QB_SWETOP_GT0, QB_RAIN_KEEP, QB_SNOWMELT_KEEP, QB_P_NOT_LE0,
    QB_NOT_PGT0_TALE, QB_P_EQ0, QB_P_NE0
};
enum {
    PC_RAIN = 0, PC_SNOWMELT, PC_PREC_NOT_LE0, PC_NOT_PGT0_TALE, PC_SWEM1_GT0, PC_SWE_NOT_LE0, PC_P_EQ0, PC_PM1_NE0,
    PC_P_NE0, PC_PM1_EQ0
};
