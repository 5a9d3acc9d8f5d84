// Microbenchmark: how many workgroups does a gfx950 CU really hold as a function of their LDS allocation?
// One-wave workgroups that run a fixed latency-bound chain (no memory): with r resident per CU a grid of 256 * 96
// workgroups takes ceil(96 / r) rounds, so time steps expose r (up to the 32-wave limit of the CU).
//   hipcc --offload-arch=gfx950 -O3 -o lds_residency lds_residency.hip && ./lds_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(64) void k_chain(float* out, int iters, float c) {
    extern __shared__ float lds[];
    float v = threadIdx.x;
    lds[threadIdx.x] = v;
    for (int i = 0; i < iters; ++i) v = fmaf(v, c, 1.0f);
    if (v == 12345.f) out[blockIdx.x] = v + lds[(threadIdx.x + 1) & 63];
}

int main() {
    float* out;
    CK(hipMalloc(&out, 1 << 20));
    const int per_cu = 96, grid = 256 * per_cu, iters = 20000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double t1 = 0;
    for (int kb : {160, 81, 80, 64, 54, 53, 52, 48, 41, 40, 33, 32, 27, 26, 21, 20, 16, 13, 10, 8, 5, 4}) {
        const int bytes = kb * 1024;
        if (hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            printf("%3d KiB: refused\n", kb); (void)hipGetLastError(); continue;
        }
        hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), bytes, 0, out, iters, 0.5f);
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), bytes, 0, out, iters, 0.5f);
        CK(hipEventRecord(b));
        if (hipEventSynchronize(b) != hipSuccess) { printf("%3d KiB: launch failed\n", kb); (void)hipGetLastError(); continue; }
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (t1 == 0) t1 = ms / per_cu;     // 160 KiB: one resident workgroup, 96 rounds
        printf("%3d KiB per workgroup: %8.2f ms  -> %.1f rounds -> about %.1f resident per CU (160 KiB / size = %.2f)\n", kb, ms,
               ms / t1, per_cu / (ms / t1), 160.0 / kb);
    }
    return 0;
}
