// chunk_probe.hip -- is the bandwidth level of an allocation a property of the physical region it sits in?  Allocates K chunks of S MiB
// one after the other (all held), times an in-place streaming read + write (x = x + 1 over the chunk, 8 B per lane) on each chunk, three
// rounds; then the same on pairs / quadruples of neighbouring chunks launched as one kernel over several pointers.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/chunk_probe.hip -o gpurun_out/chunk_probe && gpurun_out/chunk_probe [K=24] [S=512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ptrs { double *p[8]; };
__global__ __launch_bounds__(256) void k_touch(Ptrs P, int np, size_t n_each) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t which = i / n_each, j = i % n_each;
    if ((int)which < np) {
        double *p = P.p[which];
        p[j] = p[j] + 1.0;
    }
}
static float time_touch(const Ptrs &P, int np, size_t n_each) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e30f;
    const size_t total = n_each * np;
    for (int rep = 0; rep < 5; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(k_touch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, P, np, n_each);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float t; CHK(hipEventElapsedTime(&t, a, b));
        if (rep > 0 && t < best) best = t;
    }
    CHK(hipEventDestroy(a)); CHK(hipEventDestroy(b));
    return best;
}
int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 24;
    const size_t S = (size_t)(argc > 2 ? atoi(argv[2]) : 512) << 20;
    std::vector<double *> c(K);
    for (int k = 0; k < K; ++k) { CHK(hipMalloc((void **)&c[k], S)); CHK(hipMemset(c[k], 0, S)); }
    const size_t n = S / 8;
    for (int rnd = 0; rnd < 3; ++rnd) {
        printf("round %d, GB/s (read + write) per chunk of %zu MiB:", rnd, S >> 20);
        for (int k = 0; k < K; ++k) { Ptrs P{}; P.p[0] = c[k]; printf(" %.0f", 2.0 * S / time_touch(P, 1, n) / 1e6); }
        printf("\n");
    }
    for (int g : {2, 4}) {
        printf("groups of %d neighbouring chunks as one launch, GB/s:", g);
        for (int k = 0; k + g <= K; k += g) { Ptrs P{}; for (int j = 0; j < g; ++j) P.p[j] = c[k + j]; printf(" %.0f", 2.0 * S * g / time_touch(P, g, n) / 1e6); }
        printf("\n");
    }
    for (int k = 0; k < K; ++k) printf("%s%p", k ? " " : "addresses: ", (void *)c[k]);
    printf("\n");
    return 0;
}
