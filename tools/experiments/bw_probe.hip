// bw_probe.hip -- how fast can a streaming read+write kernel with k_step's access shape go, as a function of the bytes one lane
// moves per instruction (8 B = one float64 plane slot per lane, 16 B = two planes interleaved) and of the loads in flight?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
// Layout: tiles of 64 columns; inside a tile, P "planes"; a wave reads NP planes of its tile and writes NP planes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename V, int INFLIGHT, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_copy(const char *src, char *dst, size_t tile_bytes, int nvec, long ntiles) {
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int lane = threadIdx.x & 63;
    const V *s = reinterpret_cast<const V *>(src + tile * tile_bytes) + lane;
    V *d = reinterpret_cast<V *>(dst + tile * tile_bytes) + lane;
    for (int p0 = 0; p0 < nvec; p0 += INFLIGHT) {
        V v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) if (p0 + k < nvec) v[k] = s[(size_t)(p0 + k) * 64];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) if (p0 + k < nvec) d[(size_t)(p0 + k) * 64] = v[k];
    }
}

// MODE 0 copy, 1 read only, 2 write only, 3 copy with nontemporal stores, 4 copy with nontemporal loads and stores
template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256, 2) void k_mode(const char *src, char *dst, size_t tile_bytes, int nvec, long ntiles, double *sink) {
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int lane = threadIdx.x & 63;
    const double *s = reinterpret_cast<const double *>(src + tile * tile_bytes) + lane;
    double *d = reinterpret_cast<double *>(dst + tile * tile_bytes) + lane;
    double acc = 0;
    for (int p0 = 0; p0 < nvec; p0 += INFLIGHT) {
        double v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            if (MODE == 2) v[k] = (double)(p0 + k);
            else if (MODE == 4) v[k] = __builtin_nontemporal_load(&s[(size_t)(p0 + k) * 64]);
            else v[k] = s[(size_t)(p0 + k) * 64];
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            if (MODE == 1) acc += v[k];
            else if (MODE >= 3) __builtin_nontemporal_store(v[k], &d[(size_t)(p0 + k) * 64]);
            else d[(size_t)(p0 + k) * 64] = v[k];
        }
    }
    if (MODE == 1 && acc == 12345.678) *sink = acc;
}
template <int MODE>
void run_mode(const char *src, char *dst, long ncols, int planes_rw, const char *name) {
    const size_t tile_bytes = (size_t)planes_rw * 512;
    const long ntiles = ncols / 64;
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((k_mode<MODE, 16>), dim3((ntiles + 3) / 4), dim3(256), 0, 0, src, dst, tile_bytes, planes_rw, ntiles, (double *)dst);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double gb = ((MODE == 1 || MODE == 2) ? 1.0 : 2.0) * planes_rw * 8.0 * ncols / 1e9;
    printf("%-40s %.3f ms  %.2f TB/s\n", name, best, gb / best);
}

template <typename V, int INFLIGHT, int WAVES>
double run(const char *src, char *dst, long ncols, int planes_rw, const char *name) {
    // planes_rw float64 planes read and as many written per column
    const size_t tile_bytes = (size_t)planes_rw * 512;
    const long ntiles = ncols / 64;
    const int nvec = planes_rw * 8 / (int)sizeof(V);
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((k_copy<V, INFLIGHT, WAVES>), dim3((ntiles + 3) / 4), dim3(256), 0, 0, src, dst, tile_bytes, nvec, ntiles);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double gb = 2.0 * planes_rw * 8.0 * ncols / 1e9;
    printf("%-34s %3d B/lane, %2d in flight, %d waves/SIMD: %.3f ms  %.2f TB/s\n", name, (int)sizeof(V), INFLIGHT, WAVES, best, gb / best);
    return best;
}

int main(int argc, char **argv) {
    const long ncols = argc > 1 ? atol(argv[1]) : 1000000;
    const int planes = 96;   // read 96 planes, write 96 planes per column: 1.536 GB at 10^6 columns (k_calib_copy's shape)
    const size_t bytes = (size_t)planes * 512 * (ncols / 64 + 1);
    char *src, *dst;
    CHK(hipMalloc(&src, bytes)); CHK(hipMalloc(&dst, bytes));
    CHK(hipMemset(src, 1, bytes)); CHK(hipMemset(dst, 0, bytes));
    printf("columns %ld, %d planes each way, %.3f GB per launch\n", ncols, planes, 2.0 * planes * 8 * ncols / 1e9);
    run<double, 8, 2>(src, dst, ncols, planes, "8 B");
    run<double, 16, 2>(src, dst, ncols, planes, "8 B");
    run<double, 32, 2>(src, dst, ncols, planes, "8 B");
    run<double2, 8, 2>(src, dst, ncols, planes, "16 B");
    run<double2, 16, 2>(src, dst, ncols, planes, "16 B");
    run<double2, 32, 2>(src, dst, ncols, planes, "16 B");
    run<double, 16, 4>(src, dst, ncols, planes, "8 B");
    run<double2, 16, 4>(src, dst, ncols, planes, "16 B");
    run<double, 16, 8>(src, dst, ncols, planes, "8 B");
    run<double2, 16, 8>(src, dst, ncols, planes, "16 B");
    run<double4, 8, 2>(src, dst, ncols, planes, "32 B (two x4 per lane)");
    run_mode<0>(src, dst, ncols, planes, "copy 8 B/lane");
    run_mode<1>(src, dst, ncols, planes, "read only");
    run_mode<2>(src, dst, ncols, planes, "write only");
    run_mode<3>(src, dst, ncols, planes, "copy, nontemporal stores");
    run_mode<4>(src, dst, ncols, planes, "copy, nontemporal loads + stores");
    return 0;
}
