// va_levels.hip -- the fused kernel's speed level follows the VIRTUAL address of the arena (tools/experiments/chunk_levels.hip: the
// same physical chunks re-mapped in another order -- the first position of the reserved range stays the fast one).  Which addresses?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/va_levels.hip -o roger_amd/variants/va_levels
//   va_levels [range_gib=64]
// ONE physical allocation of the arena's size (hipMemCreate) is mapped at a sequence of offsets inside a large reserved range, and
// the copy with the fused kernel's access shape is timed at each: same memory, same kernel, only the virtual address changes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NP = 96, SRC0 = 0, DST0 = 100, SLOTS = 267;

template <int MODE>   // 0 copy, 2 stores only
__global__ __launch_bounds__(256, 2) void k_probe(char *base, long n) {
    const long nblk = gridDim.x, per = (nblk + 7) / 8;
    const long b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const long i = b * 256 + threadIdx.x;
    if (b >= nblk || i >= n) return;
    char *p0 = base + (size_t)(i >> 6) * SLOTS * 512 + (threadIdx.x & 63) * 8;
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
float t_ms(char *base, long n, int reps = 6) {
    float best = 1e30f;
    const long grid = ((n + 255) / 256 + 7) / 8 * 8;
    for (int r = 0; r < reps; ++r) {
        CHK(hipEventRecord(ev0));
        hipLaunchKernelGGL((k_probe<MODE>), dim3(grid), dim3(256), 0, 0, base, n);
        CHK(hipEventRecord(ev1));
        CHK(hipEventSynchronize(ev1));
        float ms;
        CHK(hipEventElapsedTime(&ms, ev0, ev1));
        if (r && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const size_t range = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 30;
    const long n = 1000000;
    const size_t MiB = 1 << 20, size = 2040 * MiB;   // >= 15625 tiles x 136704 B
    CHK(hipEventCreate(&ev0)); CHK(hipEventCreate(&ev1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemGenericAllocationHandle_t h;
    CHK(hipMemCreate(&h, size, &prop, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int res = 0; res < 2; ++res) {
        void *va = nullptr;
        CHK(hipMemAddressReserve(&va, range, 0, nullptr, 0));
        char *base = (char *)va;
        printf("reservation %d: base %p (%zu GiB)\n", res, va, range >> 30);
        std::vector<size_t> offs;
        for (size_t o = 0; o <= 4096 * MiB; o += 128 * MiB) offs.push_back(o);          // fine scan of the first 4 GiB
        for (size_t o = 5120 * MiB; o + size <= range; o += 1024 * MiB) offs.push_back(o);   // then every GiB
        for (size_t o : {(size_t)2 * MiB, (size_t)4 * MiB, (size_t)32 * MiB, (size_t)64 * MiB}) offs.push_back(o);
        for (size_t o : offs) {
            if (o + size > range) continue;
            char *p = base + o;
            CHK(hipMemMap(p, size, 0, h, 0));
            CHK(hipMemSetAccess(p, size, &acc, 1));
            const float c = t_ms<0>(p, n), w = t_ms<2>(p, n);
            printf("  offset %6zu MiB  (va %p, va mod 4 GiB = %5zu MiB): copy %.4f  stores only %.4f%s\n", o >> 20, (void *)p, ((size_t)p & ((4096 * MiB) - 1)) >> 20, c, w,
                   c < 0.27 ? "   <-- fast" : "");
            fflush(stdout);
            CHK(hipDeviceSynchronize());
            CHK(hipMemUnmap(p, size));
        }
        // (the reservation is kept: the second one lands somewhere else)
    }
    return 0;
}
