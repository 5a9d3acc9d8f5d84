// Microbenchmark 2: tile copies with the memory-level parallelism of the FFT tile kernels -- every thread
// issues all its loads back to back (K per thread), then stores.  Tile = L rows x SEG bytes at row pitch P,
// one tile per workgroup (512 threads), XCD-aware order.  Question: do 256-byte segments (32-line tiles)
// stream faster than 128-byte ones (16-line tiles) at HBM scale?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int SEGE, int L, int T>
__global__ __launch_bounds__(T) void k_tile(const float2* __restrict__ in, float2* __restrict__ out, long pitch,
                                            long tiles_per_row) {
    constexpr int K = (L * SEGE + T - 1) / T;
    const unsigned gx = gridDim.x, x = blockIdx.x;
    const unsigned tix = (gx & 7u) ? x : (x & 7u) * (gx >> 3) + (x >> 3);
    const long col0 = (long)(tix % tiles_per_row) * SEGE;
    const long slab = (long)(tix / tiles_per_row) * (long)L * pitch;
    float2 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int e = threadIdx.x + T * k;
        e = e < L * SEGE ? e : 0;
        const int l = e / SEGE, w = e % SEGE;
        v[k] = in[slab + (long)l * pitch + col0 + w];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = threadIdx.x + T * k;
        if (e < L * SEGE) {
            const int l = e / SEGE, w = e % SEGE;
            float2 t = v[k];
            t.x += 1.f;
            __builtin_nontemporal_store(t.x, &out[slab + (long)l * pitch + col0 + w].x);
            __builtin_nontemporal_store(t.y, &out[slab + (long)l * pitch + col0 + w].y);
        }
    }
}

template <int SEGE, int L, int T>
double run(const float2* in, float2* out, long n, long pitch, int reps) {
    const long tiles_per_row = pitch / SEGE, slabs = n / ((long)L * pitch);
    const long total = slabs * tiles_per_row;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_tile<SEGE, L, T>), dim3((unsigned)total), dim3(T), 0, 0, in, out, pitch, tiles_per_row);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_tile<SEGE, L, T>), dim3((unsigned)total), dim3(T), 0, 0, in, out, pitch, tiles_per_row);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 2.0 * 8.0 * (double)(slabs * (long)L * pitch) * reps / (ms * 1e-3) / 1e9;
}

int main() {
    const long n = 240000000L;
    float2 *in, *out;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8));
    CK(hipMemset(in, 0, n * 8)); CK(hipMemset(out, 0, n * 8));
    // FFT_N pass 0 like: L = 600 rows, pitch 400000;  pass 1 like: L = 625, pitch 640 inside slabs
    printf("L=600 pitch=400000: seg 128B T=512 %7.1f | seg 128B T=1024 %7.1f | seg 256B T=1024 %7.1f GB/s\n",
           run<16, 600, 512>(in, out, n, 400000, 5), run<16, 600, 1024>(in, out, n, 400000, 5),
           run<32, 600, 1024>(in, out, n, 400000, 5));
    printf("L=625 pitch=640   : seg 128B T=512 %7.1f | seg 128B T=1024 %7.1f | seg 256B T=1024 %7.1f GB/s\n",
           run<16, 625, 512>(in, out, n, 640, 5), run<16, 625, 1024>(in, out, n, 640, 5),
           run<32, 625, 1024>(in, out, n, 640, 5));
    printf("L=480 pitch=500(ish 512): seg 128B T=512 %7.1f | seg 256B T=512 %7.1f | seg 256B T=1024 %7.1f GB/s\n",
           run<16, 480, 512>(in, out, n, 512, 5), run<32, 480, 512>(in, out, n, 512, 5),
           run<32, 480, 1024>(in, out, n, 512, 5));
    return 0;
}
