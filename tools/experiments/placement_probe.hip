// placement_probe.hip -- which access shape of the arena is insensitive to where hipMalloc put it?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/placement_probe.hip -o gpurun_out/placement_probe
//   placement_probe [arenas=8] [alloc=malloc|contig|carve|mimic|fine|uncached|managed] [slots=267]
// K arenas of the product's size (10^6 columns x `slots` 512-byte slots) are allocated one after the other in one process (the
// staircase of tools/arena_levels.py: the first few allocations are fast, the rest sit on a plateau).  On EVERY arena the same
// bytes (96 planes read + 96 planes written per column, k_calib_copy's shape) are moved with different address shapes:
//   tile64x   tiles of 64 columns (512-byte slots), workgroup -> XCD-contiguous block mapping   (the product)
//   tile64    the same, plain block mapping
//   tile256   tiles of 256 columns (one workgroup; 2 KiB slots)
//   tile1024  tiles of 1024 columns (8 KiB slots)
//   plane     plane-major (plane p at base + p * n * 8)
//   rd / wr   tile64x, loads only / stores only
// and the product's shape on each eighth of the arena alone (is the level a property of regions of the allocation?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NP = 96, SRC0 = 0, DST0 = 100;

// SHAPE 0 tile64 (xcd mapping by flag), 2 tile256, 3 tile1024, 4 plane-major.  MODE 0 copy, 1 loads only, 2 stores only
template <int SHAPE, int MODE>
__global__ __launch_bounds__(256, 2) void k_probe(char *base, long n, int slots, int xcd, long blk0, long nblk, double *sink) {
    long b = blockIdx.x;
    if (xcd) {
        const long per = (nblk + 7) / 8;
        b = (b & 7) * per + (b >> 3);
        if (b >= nblk) return;
    }
    b += blk0;
    const long i = b * 256 + threadIdx.x;
    if (i >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *p0;
    size_t pstride;
    if (SHAPE == 0) { p0 = base + (size_t)(i >> 6) * slots * 512 + lane * 8; pstride = 512; }
    else if (SHAPE == 2) { p0 = base + (size_t)b * slots * 2048 + wave * 512 + lane * 8; pstride = 2048; }
    else if (SHAPE == 3) { p0 = base + (size_t)(b >> 2) * slots * 8192 + (b & 3) * 2048 + wave * 512 + lane * 8; pstride = 8192; }
    else { pstride = ((size_t)n * 8 + 255) / 256 * 256; p0 = base + i * 8; }
    double acc = 0;
    for (int q = 0; q < NP; q += 32) {
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (MODE == 2) v[k] = (double)(q + k);
            else v[k] = __builtin_nontemporal_load(reinterpret_cast<const double *>(p0 + (size_t)(SRC0 + q + k) * pstride));
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (MODE == 1) acc += v[k];
            else __builtin_nontemporal_store(v[k], reinterpret_cast<double *>(p0 + (size_t)(DST0 + q + k) * pstride));
        }
    }
    if (MODE == 1 && acc == 12345.678) *sink = acc;
}

static hipEvent_t ev0, ev1;
template <int SHAPE, int MODE>
float time_shape(char *base, long n, int slots, int xcd, long blk0 = 0, long nblk = -1, int reps = 6) {
    const long allblk = (n + 255) / 256;
    if (nblk < 0) nblk = allblk;
    const long grid = xcd ? ((nblk + 7) / 8) * 8 : nblk;
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHK(hipEventRecord(ev0));
        hipLaunchKernelGGL((k_probe<SHAPE, MODE>), dim3(grid), dim3(256), 0, 0, base, n, slots, xcd, blk0, nblk, (double *)base);
        CHK(hipEventRecord(ev1));
        CHK(hipEventSynchronize(ev1));
        float ms;
        CHK(hipEventElapsedTime(&ms, ev0, ev1));
        if (r && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    const char *alloc = argc > 2 ? argv[2] : "malloc";
    const int slots = argc > 3 ? atoi(argv[3]) : 267;
    const long n = 1000000;   // divisible by 64; tile256 / tile1024 round up
    const size_t bytes = !strcmp(alloc, "mimic") ? (size_t)(n / 64) * slots * 512 : (size_t)((n + 1023) / 1024) * 1024 / 64 * slots * 512;
    CHK(hipEventCreate(&ev0)); CHK(hipEventCreate(&ev1));
    std::vector<char *> arena(K);
    char *big = nullptr;
    if (!strcmp(alloc, "carve")) CHK(hipMalloc((void **)&big, bytes * K + (size_t)K * (2 << 20)));
    for (int k = 0; k < K; ++k) {
        if (big) arena[k] = big + (size_t)k * ((bytes + (2 << 20) - 1) / (2 << 20) * (2 << 20));
        else if (!strcmp(alloc, "contig")) CHK(hipExtMallocWithFlags((void **)&arena[k], bytes, hipDeviceMallocContiguous));
        else if (!strcmp(alloc, "fine")) CHK(hipExtMallocWithFlags((void **)&arena[k], bytes, hipDeviceMallocFinegrained));
        else if (!strcmp(alloc, "uncached")) CHK(hipExtMallocWithFlags((void **)&arena[k], bytes, hipDeviceMallocUncached));
        else if (!strcmp(alloc, "managed")) CHK(hipMallocManaged((void **)&arena[k], bytes));
        else if (!strcmp(alloc, "mimic")) {   // rh_create's sequence: a stream, the arena (the product's exact size), a staging plane, the control block
            hipStream_t st; void *stage, *dev;
            CHK(hipStreamCreate(&st));
            CHK(hipMalloc((void **)&arena[k], bytes));
            CHK(hipMemsetAsync(arena[k], 0, bytes, st));
            CHK(hipMalloc(&stage, (size_t)n * 8));
            CHK(hipMalloc(&dev, 70000));
            CHK(hipStreamSynchronize(st));
            continue;
        }
        else CHK(hipMalloc((void **)&arena[k], bytes));
        CHK(hipMemset(arena[k], 0, bytes));
    }
    printf("alloc=%s arenas=%d slots=%d bytes=%zu (%.3f GB); moved per launch %.3f GB\n", alloc, K, slots, bytes, bytes / 1e9, 2.0 * NP * 8 * n / 1e9);
    printf("%-3s %-16s %8s %8s %8s %8s %8s %8s %8s | eighths of the arena, tile64 plain mapping (us)\n", "k", "address", "tile64x", "tile64", "tile256", "tile1024", "plane", "rd", "wr");
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int k = 0; k < K; ++k) {
            char *a = arena[k];
            const bool exact = !strcmp(alloc, "mimic");   // the product's exact size has no room for the rounded-up tile256 / tile1024 shapes
            const float t0 = time_shape<0, 0>(a, n, slots, 1), t1 = time_shape<0, 0>(a, n, slots, 0), t2 = exact ? 0 : time_shape<2, 0>(a, n, slots, 1),
                        t3 = exact ? 0 : time_shape<3, 0>(a, n, slots, 1), t4 = time_shape<4, 0>(a, n, slots, 1), t5 = time_shape<0, 1>(a, n, slots, 1),
                        t6 = time_shape<0, 2>(a, n, slots, 1);
            printf("%-3d %016zx %8.4f %8.4f %8.4f %8.4f %8.4f %8.4f %8.4f |", k, (size_t)a, t0, t1, t2, t3, t4, t5, t6);
            const long allblk = (n + 255) / 256, per = allblk / 8;
            for (int e = 0; e < 8; ++e) printf(" %5.1f", 1e3 * time_shape<0, 0>(a, n, slots, 0, e * per, per, 8));
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
