// Microbenchmark: what would MORE RESIDENT TILES per CU buy the 480/500-point per-channel passes?
// A proxy of a strided FFT pass without the FFT: tile = 480 rows x 128 bytes at a row pitch of 512 points inside
// 245 760-point signals (1000 signals, 1.97 GB each way), every thread issues its 15 loads back to back, the tile goes
// through LDS once (write, barrier, transposed read), then `iters` dependent FMAs per component stand in for the
// butterflies (2 * iters VALU operations per point), then non-temporal stores.  The dynamic LDS size sets the number
// of workgroups a CU holds -- lds_residency.hip measures the real steps: 80 KiB -> 2, 53 KiB -> still 2, <= 52 KiB -> 3,
// <= 40 KiB -> 4 (hipOccupancyMaxActiveBlocksPerMultiprocessor, printed for comparison, is wrong on this part); the real
// kernels hold 2 (61-72 KiB tiles).
//   hipcc --offload-arch=gfx950 -O3 -o occupancy_proxy occupancy_proxy.hip && ./occupancy_proxy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int L = 480, W = 16, PITCH = 512;

template <int T>
__global__ __launch_bounds__(T) void k_proxy(const float2* __restrict__ in, float2* __restrict__ out, int iters, float c) {
    extern __shared__ float2 lds[];
    constexpr int K = L * W / T;
    const unsigned gx = gridDim.x, x = blockIdx.x;
    const unsigned tile = (x & 7u) * (gx >> 3) + (x >> 3);          // XCD-aware: an eighth of the row per XCD
    const long base = (long)blockIdx.y * L * PITCH + (long)tile * W;
    const int w = threadIdx.x & (W - 1), rg = threadIdx.x >> 4;
    float2 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = in[base + (long)(rg + (T / W) * k) * PITCH + w];
#pragma unroll
    for (int k = 0; k < K; ++k) lds[(rg + (T / W) * k) * W + w] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = lds[(rg * K + k) * W + w];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v[k].x = fmaf(v[k].x, c, v[k].y);
            v[k].y = fmaf(v[k].y, c, v[k].x);
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float2* p = &out[base + (long)(rg * K + k) * PITCH + w];
        __builtin_nontemporal_store(v[k].x, &p->x);
        __builtin_nontemporal_store(v[k].y, &p->y);
    }
}

template <int T>
double run(const float2* in, float2* out, int signals, int lds_bytes, int iters, int reps) {
    CK(hipFuncSetAttribute((const void*)k_proxy<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_proxy<T>, T, lds_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    dim3 grid(PITCH / W, signals);
    hipLaunchKernelGGL(k_proxy<T>, grid, dim3(T), lds_bytes, 0, in, out, iters, 0.999f);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_proxy<T>, grid, dim3(T), lds_bytes, 0, in, out, iters, 0.999f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double tbs = 2.0 * 8.0 * (double)signals * L * PITCH * reps / (ms * 1e-3) / 1e12;
    printf("  T=%4d  LDS %6d B  (occupancy API: %d/CU)   %2d VALU ops/point: %6.1f us  %5.2f TB/s\n", T, lds_bytes, per_cu,
           2 * iters, ms * 1e3 / reps, tbs);
    return tbs;
}

int main() {
    const int signals = 1000;
    const size_t n = (size_t)signals * L * PITCH;
    float2 *in, *out;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8));
    CK(hipMemset(in, 0, n * 8)); CK(hipMemset(out, 0, n * 8));
    for (int iters : {0, 20, 35, 50}) {
        printf("%d VALU operations per point\n", 2 * iters);
        for (int lds : {80 * 1024, 52 * 1024, 40 * 1024}) run<512>(in, out, signals, lds, iters, 5);
        for (int lds : {80 * 1024, 52 * 1024, 40 * 1024}) run<256>(in, out, signals, lds, iters, 5);
    }
    return 0;
}
