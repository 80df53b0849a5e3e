// rh_sas.hip -- the C ABI of the SAS / oxygen-18 transport step for gfx950 (kernels: rh_sas_kernels.h, rh_sas_solvers_impl.h)
// (include/roger_hip_sas.h).
//
// One workgroup per soil column.  The age axis is laid out blocked over the workgroup: thread t
// owns the E consecutive age classes [t * E, (t + 1) * E) of every age vector in registers, so the
// whole day -- 2 inflows, 5 outgoing fluxes with `substeps` sub-steps each, storage concentrations,
// age statistics, ageing -- runs on one read and one write of the four state vectors
// (sa_rz, msa_rz, sa_ss, msa_ss): 8 * ages * 8 bytes per column and day.  The arithmetic is
// dominated by the power-law SAS function, 5 * substeps * (ages + 1) `pow` per column and day
// (3 * 10^4 at ages = 1000), which makes this kernel fp64-ALU bound, not HBM bound.
//
// Every formula below is a per-element restatement of the reference's array expressions (file:line
// in the comments; roger/core/transport.py unless said otherwise) with the reference's operation
// order; no FMA contraction (-ffp-contract=off).  Two deliberate differences, both at rounding
// level: prefix sums are block scans instead of sequential `cumsum`s, sums over ages are tree
// reductions instead of numpy's pairwise sums.  The scan is built so that what the algorithm is
// sensitive to still holds exactly: cumulative values are non-decreasing for non-negative input and
// an empty age class contributes an exact 0 to every difference of cumulative values.
#include <hip/hip_runtime.h>
#include <cstdio>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "roger_hip.h"
#include "roger_hip_sas.h"
#include "rh_sas_dev.h"

static const char *const SAS_NAMES[] = {
#define RH_SAS_ARRAY(name, kind, when) #name,
#include "rh_sas_arrays.def"
#undef RH_SAS_ARRAY
};
static const unsigned char SAS_KIND[] = {
#define RH_SAS_ARRAY(name, kind, when) K_##kind,
#include "rh_sas_arrays.def"
#undef RH_SAS_ARRAY
};
static const unsigned char SAS_WHEN[] = {
#define RH_SAS_ARRAY(name, kind, when) W_##when,
#include "rh_sas_arrays.def"
#undef RH_SAS_ARRAY
};

__global__ void k_selftest_div(const double *a, const double *d, double *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = udiv(a[i], udiv_prepare(d[i]));
}
__global__ void k_selftest_pow(const double *x, const double *k, double *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const PowConsts C = load_pow_consts(SAS_LOG_T);
    if (i >= n) return;
    // the same dispatch as the power-law branch of calc_tt_family: exponents 0.5 / 1.5 / 1 through the square root
    if (k[i] == 0.5) out[i] = sqrt_unit(x[i]);
    else if (k[i] == 1.5) out[i] = x[i] * sqrt_unit(x[i]);
    else if (k[i] == 1.0) out[i] = x[i];
    else if (k[i] == 0.2) {   // the fifth-root path with its range check
        int e2;
        const double y = pow_fifth(x[i], &e2);
        out[i] = (e2 < -126) ? sas_pow_ratio(C, x[i], 1.0, 0.0, k[i]) : y;
    } else out[i] = sas_pow_ratio(C, x[i], 1.0, 0.0, k[i]);
}

// ---------------------------------------------------------------------------------------------
// host side: context and C ABI
// ---------------------------------------------------------------------------------------------
struct rh_sas_ctx {
    rh_sas_config cfg;
    hipStream_t stream;
    bool own_stream;
    void *arr[SA_COUNT];
    int64_t elems[SA_COUNT];
    int *unsupported;
    bool timing;
    std::vector<hipEvent_t> events;
    size_t ev_used;
    std::string err;
};
static std::string g_sas_create_err;

static int sfail(rh_sas_ctx *ctx, int code, const std::string &msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_sas_create_err = msg;
    return code;
}
#define SHIPCHK(ctx, call)                                                                                      \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) return sfail(ctx, RH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

static int64_t sas_elems(const rh_sas_config &c, int a) {
    const int when = SAS_WHEN[a];
    if (when == W_STATS && !c.age_statistics) return 0;
    if (when == W_DIAG && !c.keep_distributions) return 0;
    if (when == W_ANION && c.tracer == RH_SAS_TRACER_OXYGEN18) return 0;
    switch (SAS_KIND[a]) {
    case K_AGE: return c.n_cells * c.ages;
    case K_NAGE: return c.n_cells * (c.ages + 1);
    case K_CELL: return c.n_cells;
    case K_DAILY: return c.n_cells * c.forcing_days;
    case K_PARAM: return c.n_cells * 8;
    case K_MASK: return c.n_cells;
    }
    return 0;
}

extern "C" {

void rh_sas_default_config(rh_sas_config *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->n_cells = 1;
    cfg->ages = 1000;         // benchmarks/SVATOXYGEN18_benchmark.py:28-44
    cfg->substeps = 1;        // settings.sas_solver_substeps default, roger/settings.py:120
    cfg->forcing_days = 1;
    cfg->vsmow = 2005.2e-6;   // roger/settings.py:76-78
    cfg->d18O_min = -20;
    cfg->d18O_max = 0;
}

const char *rh_sas_last_error(const rh_sas_ctx *ctx) { return ctx ? ctx->err.c_str() : g_sas_create_err.c_str(); }
int rh_sas_num_arrays(void) { return SA_COUNT; }
const char *rh_sas_array_name(int a) { return (a >= 0 && a < SA_COUNT) ? SAS_NAMES[a] : nullptr; }
int rh_sas_array_is_int(int a) { return (a >= 0 && a < SA_COUNT) ? (SAS_KIND[a] == K_MASK) : -1; }
int rh_sas_array_index(const char *name) {
    if (!name) return -1;
    for (int a = 0; a < SA_COUNT; ++a)
        if (std::strcmp(SAS_NAMES[a], name) == 0) return a;
    return -1;
}
int64_t rh_sas_array_elems(const rh_sas_ctx *ctx, int a) { return (ctx && a >= 0 && a < SA_COUNT) ? ctx->elems[a] : 0; }

int rh_sas_create(const rh_sas_config *cfg, rh_sas_ctx **out) {
    if (!cfg || !out) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: null argument");
    if (cfg->n_cells <= 0 || cfg->n_cells > 0x7fffffffLL) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: n_cells out of range");
    if (cfg->ages < 2 || cfg->ages + 1 > RH_SAS_MAX_NAGES)
        return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: ages must be in [2, RH_SAS_MAX_NAGES - 1]");
    if (cfg->substeps < 1 || cfg->forcing_days < 1) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: substeps and forcing_days must be >= 1");
    if (cfg->tracer < RH_SAS_TRACER_OXYGEN18 || cfg->tracer > RH_SAS_TRACER_VIRTUAL)
        return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: tracer must be RH_SAS_TRACER_OXYGEN18, _BROMIDE, _CHLORIDE or _VIRTUAL");
    if (cfg->solver < RH_SAS_SOLVER_DETERMINISTIC || cfg->solver > RH_SAS_SOLVER_RK4)
        return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: solver must be RH_SAS_SOLVER_DETERMINISTIC, _EULER or _RK4");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return sfail(nullptr, RH_ERR_NODEVICE, "rh_sas_create: no HIP device visible (this backend has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: device ordinal out of range");
    SHIPCHK(nullptr, hipSetDevice(cfg->device));
    rh_sas_ctx *ctx = new (std::nothrow) rh_sas_ctx();
    if (!ctx) return sfail(nullptr, RH_ERR_ARG, "rh_sas_create: out of host memory");
    ctx->cfg = *cfg;
    ctx->stream = nullptr;
    ctx->own_stream = false;
    ctx->unsupported = nullptr;
    ctx->timing = false;
    ctx->ev_used = 0;
    for (int a = 0; a < SA_COUNT; ++a) {
        ctx->arr[a] = nullptr;
        ctx->elems[a] = 0;
    }
    auto bail = [&](hipError_t e, const char *what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        rh_sas_destroy(ctx);
        return sfail(nullptr, RH_ERR_HIP, msg);
    };
    hipError_t e;
    if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) return bail(e, "hipStreamCreate");
    ctx->own_stream = true;
    for (int a = 0; a < SA_COUNT; ++a) {
        const int64_t ne = sas_elems(*cfg, a);
        if (!ne) continue;
        const size_t bytes = (size_t)ne * (SAS_KIND[a] == K_MASK ? sizeof(int32_t) : sizeof(double));
        if ((e = hipMalloc(&ctx->arr[a], bytes)) != hipSuccess) return bail(e, "hipMalloc(SAS array)");
        if ((e = hipMemsetAsync(ctx->arr[a], 0, bytes, ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
        ctx->elems[a] = ne;
    }
    if ((e = hipMalloc((void **)&ctx->unsupported, sizeof(int))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemsetAsync(ctx->unsupported, 0, sizeof(int), ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
    {   // maskCatch defaults to 1 (roger/variables.py)
        std::vector<int32_t> ones((size_t)cfg->n_cells, 1);
        if ((e = hipMemcpyAsync(ctx->arr[SA_maskCatch], ones.data(), ones.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return bail(e, "hipMemcpy(maskCatch)");
        if (cfg->tracer != RH_SAS_TRACER_OXYGEN18) {   // alpha_transp, alpha_q: initial=1 (roger/variables.py:5377-5405)
            std::vector<double> one((size_t)cfg->n_cells, 1.0);
            for (int a : {SA_alpha_transp, SA_alpha_q})
                if ((e = hipMemcpyAsync(ctx->arr[a], one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
                    return bail(e, "hipMemcpy(alpha)");
        }
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    }
    *out = ctx;
    return RH_OK;
}

void rh_sas_destroy(rh_sas_ctx *ctx) {
    if (!ctx) return;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->events) (void)hipEventDestroy(ev);
    for (int a = 0; a < SA_COUNT; ++a)
        if (ctx->arr[a]) (void)hipFree(ctx->arr[a]);
    if (ctx->unsupported) (void)hipFree(ctx->unsupported);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rh_sas_set_stream(rh_sas_ctx *ctx, void *hip_stream) {
    if (!ctx) return RH_ERR_ARG;
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) SHIPCHK(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return RH_OK;
}

int rh_sas_sync(rh_sas_ctx *ctx) {
    if (!ctx) return RH_ERR_ARG;
    int bad = 0;
    SHIPCHK(ctx, hipMemcpyAsync(&bad, ctx->unsupported, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (bad & 2)   // (the invariant blk_cumsum's top value rests on, checked by every day's kernel before it stores: ADVICE r3)
        return sfail(ctx, RH_ERR_STATE, "internal: an age class above ages - 1 (register padding of the age axis) holds water after a day's fluxes");
    if (bad)
        return sfail(ctx, RH_ERR_STATE,
                     "a column's sas_params select a code that is none of the reference's SAS families (1, 2, 3, 31-37, 4, 51, 52, 6, 61, 62)");
    return RH_OK;
}

static int sas_check(rh_sas_ctx *ctx, int a, size_t bytes, const void *host, const char *who) {
    if (!ctx) return RH_ERR_ARG;
    if (a < 0 || a >= SA_COUNT) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": unknown array id");
    if (!ctx->arr[a]) return sfail(ctx, RH_ERR_STATE, std::string(who) + ": array " + SAS_NAMES[a] + " is not held by this context (age_statistics / keep_distributions)");
    const size_t want = (size_t)ctx->elems[a] * (SAS_KIND[a] == K_MASK ? sizeof(int32_t) : sizeof(double));
    if (bytes != want) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": size mismatch for array " + SAS_NAMES[a]);
    if (!host) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": null host pointer");
    return RH_OK;
}

int rh_sas_upload(rh_sas_ctx *ctx, int a, const void *host, size_t bytes) {
    const int rc = sas_check(ctx, a, bytes, host, "rh_sas_upload");
    if (rc) return rc;
    SHIPCHK(ctx, hipMemcpyAsync(ctx->arr[a], host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}

int rh_sas_download(rh_sas_ctx *ctx, int a, void *host, size_t bytes) {
    const int rc = sas_check(ctx, a, bytes, host, "rh_sas_download");
    if (rc) return rc;
    SHIPCHK(ctx, hipMemcpyAsync(host, ctx->arr[a], bytes, hipMemcpyDeviceToHost, ctx->stream));
    return rh_sas_sync(ctx);
}

static int sas_check_cells(rh_sas_ctx *ctx, int a, int64_t first, int64_t cnt, size_t bytes, const void *host, const char *who,
                           size_t *offset) {
    if (!ctx) return RH_ERR_ARG;
    if (a < 0 || a >= SA_COUNT) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": unknown array id");
    if (!ctx->arr[a]) return sfail(ctx, RH_ERR_STATE, std::string(who) + ": array " + SAS_NAMES[a] + " is not held by this context");
    if (SAS_KIND[a] == K_DAILY) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": daily inputs are (forcing_days, n_cells); use the whole-array call");
    if (first < 0 || cnt <= 0 || first + cnt > ctx->cfg.n_cells) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": cell range out of bounds");
    const size_t row = (size_t)(ctx->elems[a] / ctx->cfg.n_cells) * (SAS_KIND[a] == K_MASK ? sizeof(int32_t) : sizeof(double));
    if (bytes != row * (size_t)cnt) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": size mismatch for array " + SAS_NAMES[a]);
    if (!host) return sfail(ctx, RH_ERR_ARG, std::string(who) + ": null host pointer");
    *offset = row * (size_t)first;
    return RH_OK;
}

int rh_sas_upload_cells(rh_sas_ctx *ctx, int a, int64_t first_cell, int64_t n_cells, const void *host, size_t bytes) {
    size_t off;
    const int rc = sas_check_cells(ctx, a, first_cell, n_cells, bytes, host, "rh_sas_upload_cells", &off);
    if (rc) return rc;
    SHIPCHK(ctx, hipMemcpyAsync((char *)ctx->arr[a] + off, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RH_OK;
}

int rh_sas_download_cells(rh_sas_ctx *ctx, int a, int64_t first_cell, int64_t n_cells, void *host, size_t bytes) {
    size_t off;
    const int rc = sas_check_cells(ctx, a, first_cell, n_cells, bytes, host, "rh_sas_download_cells", &off);
    if (rc) return rc;
    SHIPCHK(ctx, hipMemcpyAsync(host, (const char *)ctx->arr[a] + off, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return rh_sas_sync(ctx);
}

int rh_sas_set_daily_from_device(rh_sas_ctx *ctx, int a, int64_t day_row, const double *dev_src) {
    if (!ctx) return RH_ERR_ARG;
    if (a < 0 || a >= SA_COUNT || SAS_KIND[a] != K_DAILY) return sfail(ctx, RH_ERR_ARG, "rh_sas_set_daily_from_device: not a daily input array");
    if (day_row < 0 || day_row >= ctx->cfg.forcing_days || !dev_src) return sfail(ctx, RH_ERR_ARG, "rh_sas_set_daily_from_device: bad row or null source");
    SHIPCHK(ctx, hipMemcpyAsync((double *)ctx->arr[a] + day_row * ctx->cfg.n_cells, dev_src, (size_t)ctx->cfg.n_cells * sizeof(double),
                                hipMemcpyDeviceToDevice, ctx->stream));
    return RH_OK;
}

void *rh_sas_array_device_ptr(rh_sas_ctx *ctx, int a) { return (ctx && a >= 0 && a < SA_COUNT) ? ctx->arr[a] : nullptr; }

int rh_sas_stages(rh_sas_ctx *ctx, int64_t day, int stages) {
    if (!ctx) return RH_ERR_ARG;
    if (day < 0) return sfail(ctx, RH_ERR_ARG, "rh_sas_stages: negative day");
    if ((stages & ~(RH_SAS_ALL | RH_SAS_RESCALE)) || !stages) return sfail(ctx, RH_ERR_ARG, "rh_sas_stages: bad stage mask");
    const rh_sas_config &c = ctx->cfg;
    if ((stages & RH_SAS_STORAGE) && c.age_statistics && !c.keep_distributions &&
        (!(stages & RH_SAS_TRANSP) || !(stages & RH_SAS_Q_SS)))
        return sfail(ctx, RH_ERR_STATE,
                     "rh_sas_stages: age statistics need the transpiration and percolation stages in the same launch, or keep_distributions");
    SasArgs args;
    args.n = c.n_cells;
    args.day_off = (day % c.forcing_days) * c.n_cells;
    args.ages = c.ages;
    args.substeps = c.substeps;
    args.stages = stages;
    args.stats = c.age_statistics ? 1 : 0;
    args.diag = c.keep_distributions ? 1 : 0;
    args.tracer = c.tracer;
    args.vsmow = c.vsmow;
    args.dmin = c.d18O_min;
    args.dmax = c.d18O_max;
    args.unsupported = ctx->unsupported;
    for (int a = 0; a < SA_COUNT; ++a) args.a[a] = ctx->arr[a];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->timing) {
        if (ctx->ev_used + 2 > ctx->events.size()) {
            for (int k = 0; k < 2; ++k) {
                hipEvent_t ev;
                SHIPCHK(ctx, hipEventCreate(&ev));
                ctx->events.push_back(ev);
            }
        }
        ev0 = ctx->events[ctx->ev_used];
        ev1 = ctx->events[ctx->ev_used + 1];
        ctx->ev_used += 2;
        SHIPCHK(ctx, hipEventRecord(ev0, ctx->stream));
    }
    if (c.solver != RH_SAS_SOLVER_DETERMINISTIC && (stages & RH_SAS_ALL)) {
        // the explicit solvers evaluate all fluxes of a sub-step on one state: the day cannot be cut into stages
        if (stages != RH_SAS_ALL)
            return sfail(ctx, RH_ERR_ARG, "rh_sas_stages: with an explicit solver the day runs in one launch (RH_SAS_ALL); only RH_SAS_RESCALE may run on its own");
        const bool anion = c.tracer != RH_SAS_TRACER_OXYGEN18;
        const int rc = c.solver == RH_SAS_SOLVER_RK4 ? (anion ? rh_sas_launch_rk4_anion : rh_sas_launch_rk4_iso)(ctx->stream, args)
                                                     : (anion ? rh_sas_launch_euler_anion : rh_sas_launch_euler_iso)(ctx->stream, args);
        if (rc) return sfail(ctx, rc, "rh_sas_stages: unknown solver");
        SHIPCHK(ctx, hipGetLastError());
        if (ctx->timing) SHIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
        return RH_OK;
    }
    // smallest workgroup whose blocked layout covers the ages + 1 edges (rh_sas_kernels.h; the isotope and the anion kernels are
    // translation units of their own, rh_sas_det_iso.hip / rh_sas_det_anion.hip)
    static const bool e4 = std::getenv("RH_SAS_E4") != nullptr;
#ifdef RH_SAS_PHASES
    static unsigned long long *d_phases = nullptr;
    if (!d_phases) (void)hipMalloc((void **)&d_phases, 32 * sizeof(unsigned long long));
    (void)hipMemsetAsync(d_phases, 0, 32 * sizeof(unsigned long long), ctx->stream);
    args.phases = d_phases;
#endif
    const int lrc = (ctx->cfg.tracer != RH_SAS_TRACER_OXYGEN18 ? rh_sas_launch_det_anion : rh_sas_launch_det_iso)(ctx->stream, args, (unsigned)ctx->cfg.n_cells, c.ages + 1, e4);
#ifdef RH_SAS_PHASES
    {
        unsigned long long h[32];
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(h, d_phases, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long tot = 0;
        for (int k = 0; k < 16; ++k) tot += h[k];
        std::fprintf(stderr, "sas phases, cycles per column (day %lld):", (long long)day);
        for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %d:%.0f", k, (double)h[k] / (double)ctx->cfg.n_cells);
        std::fprintf(stderr, "  total %.0f\n", (double)tot / (double)ctx->cfg.n_cells);
    }
#endif
    if (lrc) return sfail(ctx, lrc, "rh_sas_stages: the age axis does not fit the kernel shapes of this build");
    SHIPCHK(ctx, hipGetLastError());
    if (ctx->timing) SHIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
    return RH_OK;
}

int rh_sas_step(rh_sas_ctx *ctx, int64_t day) { return rh_sas_stages(ctx, day, RH_SAS_ALL); }

int rh_sas_run_days(rh_sas_ctx *ctx, int64_t day0, int64_t ndays) {
    if (!ctx) return RH_ERR_ARG;
    if (ndays < 0) return sfail(ctx, RH_ERR_ARG, "rh_sas_run_days: negative ndays");
    for (int64_t d = 0; d < ndays; ++d) {
        const int rc = rh_sas_stages(ctx, day0 + d, RH_SAS_ALL);
        if (rc) return rc;
    }
    return RH_OK;
}

static int selftest2(const double *x, const double *k, double *out, int64_t n, int which);
int rh_sas_selftest_pow(const double *x, const double *k, double *out, int64_t n) { return selftest2(x, k, out, n, 0); }
int rh_sas_selftest_div(const double *a, const double *d, double *out, int64_t n) { return selftest2(a, d, out, n, 1); }
static int selftest2(const double *x, const double *k, double *out, int64_t n, int which) {
    if (!x || !k || !out || n <= 0) return RH_ERR_ARG;
    double *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * 3 * sizeof(double)) != hipSuccess) return RH_ERR_HIP;
    int rc = RH_OK;
    if (hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, k, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        rc = RH_ERR_HIP;
    if (rc == RH_OK) {
        if (which == 0)
            hipLaunchKernelGGL(k_selftest_pow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, d + n, d + 2 * n, n);
        else
            hipLaunchKernelGGL(k_selftest_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, d + n, d + 2 * n, n);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(out, d + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = RH_ERR_HIP;
    }
    (void)hipFree(d);
    return rc;
}

int rh_sas_enable_timing(rh_sas_ctx *ctx, int on) {
    if (!ctx) return RH_ERR_ARG;
    ctx->timing = on != 0;
    ctx->ev_used = 0;
    return RH_OK;
}

int rh_sas_timing_summary(rh_sas_ctx *ctx, double *total_ms, int64_t *launches) {
    if (!ctx || !total_ms || !launches) return RH_ERR_ARG;
    SHIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0;
    for (size_t k = 0; k + 1 < ctx->ev_used; k += 2) {
        float ms = 0;
        SHIPCHK(ctx, hipEventElapsedTime(&ms, ctx->events[k], ctx->events[k + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)(ctx->ev_used / 2);
    return RH_OK;
}

}  // extern "C"
