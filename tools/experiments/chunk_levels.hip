// chunk_levels.hip -- is the fused kernel's speed level a property of the PHYSICAL chunks an arena is made of?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/chunk_levels.hip -o roger_amd/variants/chunk_levels
//   chunk_levels [chunks=16] [chunk_mib=1088]
// Physical chunks are created with the virtual memory management API (hipMemCreate) and mapped side by side into one reserved
// address range; an "arena" is two neighbouring chunks (2 x 1088 MiB >= the product's 2 037 MiB at 10^6 columns).  Measured with the
// fused kernel's access shape (tiles of 64 columns, 267 slots, XCD-contiguous block mapping; 96 planes read + 96 written per column):
//   1. every arena (chunk pair) as mapped first: copy / loads only / stores only
//   2. every chunk alone (the first 500 000 columns' worth of tiles fit one chunk)
//   3. the chunks re-mapped in the order of their own speed (fastest first): do the arenas made of fast chunks come out fast?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NP = 96, SRC0 = 0, DST0 = 100, SLOTS = 267;

template <int MODE>   // 0 copy, 1 loads only, 2 stores only
__global__ __launch_bounds__(256, 2) void k_probe(char *base, long n, double *sink) {
    const long nblk = gridDim.x, per = (nblk + 7) / 8;
    const long b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const long i = b * 256 + threadIdx.x;
    if (b >= nblk || i >= n) return;
    char *p0 = base + (size_t)(i >> 6) * SLOTS * 512 + (threadIdx.x & 63) * 8;
    double acc = 0;
    for (int q = 0; q < NP; q += 32) {
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            v[k] = MODE == 2 ? (double)(q + k) : __builtin_nontemporal_load(reinterpret_cast<const double *>(p0 + (size_t)(SRC0 + q + k) * 512));
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (MODE == 1) acc += v[k];
            else __builtin_nontemporal_store(v[k], reinterpret_cast<double *>(p0 + (size_t)(DST0 + q + k) * 512));
        }
    }
    if (MODE == 1 && acc == 12345.678) *sink = acc;
}
static hipEvent_t ev0, ev1;
template <int MODE>
float t_ms(char *base, long n, int reps = 7) {
    float best = 1e30f;
    const long grid = ((n + 255) / 256 + 7) / 8 * 8;
    for (int r = 0; r < reps; ++r) {
        CHK(hipEventRecord(ev0));
        hipLaunchKernelGGL((k_probe<MODE>), dim3(grid), dim3(256), 0, 0, base, n, (double *)base);
        CHK(hipEventRecord(ev1));
        CHK(hipEventSynchronize(ev1));
        float ms;
        CHK(hipEventElapsedTime(&ms, ev0, ev1));
        if (r && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 1088) << 20;
    const long n = 1000000, nhalf = 499968;   // 7812 tiles: 1.068 GB, inside one chunk
    CHK(hipEventCreate(&ev0)); CHK(hipEventCreate(&ev1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu, %d chunks of %zu MiB\n", gran, M, chunk >> 20);
    if (chunk % gran) { printf("chunk size is not a multiple of the granularity\n"); return 1; }
    std::vector<hipMemGenericAllocationHandle_t> h(M);
    for (int j = 0; j < M; ++j) CHK(hipMemCreate(&h[j], chunk, &prop, 0));
    void *va = nullptr;
    CHK(hipMemAddressReserve(&va, chunk * M, 0, nullptr, 0));
    char *base = (char *)va;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    auto map_in_order = [&](const std::vector<int> &order) {
        for (int j = 0; j < M; ++j) CHK(hipMemMap(base + (size_t)j * chunk, chunk, 0, h[order[j]], 0));
        CHK(hipMemSetAccess(base, chunk * M, &acc, 1));
    };
    std::vector<int> order(M);
    std::iota(order.begin(), order.end(), 0);
    map_in_order(order);
    CHK(hipMemset(base, 0, chunk * M));
    printf("1. arenas = chunk pairs as created (ms: copy, loads only, stores only)\n");
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int a = 0; a + 1 < M; a += 2) {
            char *p = base + (size_t)a * chunk;
            printf("   arena of chunks %2d,%2d: %.4f %.4f %.4f\n", a, a + 1, t_ms<0>(p, n), t_ms<1>(p, n), t_ms<2>(p, n));
            fflush(stdout);
        }
    printf("2. every chunk alone, %ld columns (ms: copy, stores only)\n", nhalf);
    std::vector<float> tc(M);
    for (int j = 0; j < M; ++j) {
        char *p = base + (size_t)j * chunk;
        tc[j] = t_ms<0>(p, nhalf);
        printf("   chunk %2d: %.4f %.4f\n", j, tc[j], t_ms<2>(p, nhalf));
        fflush(stdout);
    }
    CHK(hipDeviceSynchronize());
    CHK(hipMemUnmap(base, chunk * M));
    std::sort(order.begin(), order.end(), [&](int x, int y) { return tc[x] < tc[y]; });
    map_in_order(order);
    printf("3. re-mapped, fastest chunks first:");
    for (int j = 0; j < M; ++j) printf(" %d", order[j]);
    printf("\n");
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int a = 0; a + 1 < M; a += 2) {
            char *p = base + (size_t)a * chunk;
            printf("   arena of chunks %2d,%2d: %.4f %.4f %.4f   (sum of the two chunks alone %.4f)\n", order[a], order[a + 1], t_ms<0>(p, n), t_ms<1>(p, n),
                   t_ms<2>(p, n), tc[order[a]] + tc[order[a + 1]]);
            fflush(stdout);
        }
    return 0;
}
