// Microbenchmark: how fast a buffer can be WRITTEN depends on where hipMalloc put it.
//
// pitch_sweep.hip (placement mode) showed that the rate of one and the same tile copy varies by 20 % between buffer pairs
// allocated in one process, that the spread follows the OUTPUT buffer, and that a buffer's plain write rate predicts it
// (read rates hardly differ).  This file takes the census: K buffers of S MiB, each probed slice by slice (write-only and
// read-only sweeps of 256 MiB), to see how the fast and the slow memory is distributed and at what granularity.
//   placement_census [K = 40] [S MiB = 2048] [slice MiB = 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_write(float4* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 40;
    const size_t S = (size_t)(argc > 2 ? atoi(argv[2]) : 2048) << 20, slice = (size_t)(argc > 3 ? atoi(argv[3]) : 256) << 20;
    const int reps = 6;
    float* sink; CK(hipMalloc(&sink, 4));
    std::vector<char*> bufs;
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("free %.1f GiB of %.1f GiB; %d buffers of %zu MiB, slices of %zu MiB\n", free_b / 1073741824.0, total_b / 1073741824.0, K,
           S >> 20, slice >> 20);
    for (int k = 0; k < K; ++k) {
        char* p = nullptr;
        if (hipMalloc(&p, S) != hipSuccess) { printf("allocation %d failed\n", k); break; }
        bufs.push_back(p);
        CK(hipMemset(p, 0, S));
    }
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t k = 0; k < bufs.size(); ++k) {
        printf("buf %2zu %p  write GB/s per slice:", k, (void*)bufs[k]);
        double wsum = 0, rsum = 0; int ns = 0;
        std::vector<double> rr;
        for (size_t off = 0; off + slice <= S; off += slice, ++ns) {
            float4* q = reinterpret_cast<float4*>(bufs[k] + off);
            const size_t n = slice / 16;
            float ms;
            hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, 0, q, n);
            CK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, 0, q, n);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            const double w = slice * reps / (ms * 1e-3) / 1e9;
            hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(256), 0, 0, q, n, sink);
            CK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(256), 0, 0, q, n, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            rr.push_back(slice * reps / (ms * 1e-3) / 1e9);
            printf(" %5.0f", w);
            wsum += w; rsum += rr.back();
        }
        printf("  | mean write %5.0f read %5.0f\n", wsum / ns, rsum / ns);
    }
    return 0;
}
