// xcd_phase.hip -- hypothesis: with the XCD-contiguous block mapping the eight XCDs stream through eight regions of the arena S = 267 MB
// apart, in lockstep; whether those eight streams fall onto different HBM channels / banks or onto the same ones depends on S (and, for
// an arena that is not physically contiguous, on where its pieces lie: the "placement level").  On PHYSICALLY CONTIGUOUS arenas
// (hipExtMallocWithFlags(hipDeviceMallocContiguous)) the level should then be a deterministic function of a padding inserted between
// the regions.      hipcc --offload-arch=gfx950 -O3 tools/experiments/xcd_phase.hip -o roger_amd/variants/xcd_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NP = 96, SRC0 = 0, DST0 = 100, SLOTS = 267;

template <int MODE>   // 0 copy, 2 stores only
__global__ __launch_bounds__(256, 2) void k_probe(char *base, long n, size_t pad) {
    const long nblk = gridDim.x, per = (nblk + 7) / 8;
    const long x = blockIdx.x & 7, b = x * per + (blockIdx.x >> 3);
    const long i = b * 256 + threadIdx.x;
    if (b >= nblk || i >= n) return;
    char *p0 = base + (size_t)(i >> 6) * SLOTS * 512 + (threadIdx.x & 63) * 8 + (size_t)x * pad;
    for (int q = 0; q < NP; q += 32) {
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            v[k] = MODE == 2 ? (double)(q + k) : __builtin_nontemporal_load(reinterpret_cast<const double *>(p0 + (size_t)(SRC0 + q + k) * 512));
#pragma unroll
        for (int k = 0; k < 32; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<double *>(p0 + (size_t)(DST0 + q + k) * 512));
    }
}
static hipEvent_t ev0, ev1;
template <int MODE>
float t_ms(char *base, long n, size_t pad, int reps = 6) {
    float best = 1e30f;
    const long grid = ((n + 255) / 256 + 7) / 8 * 8;
    for (int r = 0; r < reps; ++r) {
        CHK(hipEventRecord(ev0));
        hipLaunchKernelGGL((k_probe<MODE>), dim3(grid), dim3(256), 0, 0, base, n, pad);
        CHK(hipEventRecord(ev1));
        CHK(hipEventSynchronize(ev1));
        float ms;
        CHK(hipEventElapsedTime(&ms, ev0, ev1));
        if (r && ms < best) best = ms;
    }
    return best;
}
int main() {
    const long n = 1000000;
    const size_t KiB = 1024, MiB = 1 << 20, arena = (size_t)(n / 64 + 8) * SLOTS * 512, maxpad = 300 * MiB;
    CHK(hipEventCreate(&ev0)); CHK(hipEventCreate(&ev1));
    std::vector<size_t> pads = {0, 512, 4 * KiB, 8 * KiB, 16 * KiB, 32 * KiB, 64 * KiB, 128 * KiB, 136704, 256 * KiB, 512 * KiB, MiB, 2 * MiB, 4 * MiB, 8 * MiB, 16 * MiB,
                                32 * MiB, 64 * MiB, 128 * MiB, 256 * MiB,
                                // region spacing S = 267 003 750 B + pad rounded up to powers of two and odd multiples
                                (size_t)268435456 - 266998656 % 268435456, 1437 * KiB, 3 * MiB + 512 * KiB, 5 * MiB, 11 * MiB, 23 * MiB, 47 * MiB, 97 * MiB, 200 * MiB};
    for (int a = 0; a < 3; ++a) {
        char *base = nullptr;
        if (a < 2) CHK(hipExtMallocWithFlags((void **)&base, arena + 8 * maxpad, hipDeviceMallocContiguous));
        else CHK(hipMalloc((void **)&base, arena + 8 * maxpad));
        CHK(hipMemset(base, 0, arena + 8 * maxpad));
        printf("arena %d (%s) at %p\n", a, a < 2 ? "physically contiguous" : "hipMalloc", (void *)base);
        for (size_t pad : pads) {
            printf("  pad %10zu B (%8.2f MiB): copy %.4f  stores only %.4f\n", pad, pad / 1048576.0, t_ms<0>(base, n, pad), t_ms<2>(base, n, pad));
            fflush(stdout);
        }
    }
    return 0;
}
