// Microbenchmark: does the RELATIVE placement of a streaming kernel's input and output buffers matter?
// hipMalloc returns 2 MiB-aligned blocks, so in[i] and out[i] of every pass share all their low address bits; if the
// DRAM channel / bank hash used only those, every read would meet its own write in one bank.  Tile copy in the shape of
// the wideband FFT's middle pass (625 rows x 128 bytes at a pitch of 640 points, every load of a thread in flight), the
// output displaced by `off` bytes inside a larger allocation; plus the in-place form (out == in).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int L, int T>
__global__ __launch_bounds__(T) void k_tile(const float2* __restrict__ in, float2* __restrict__ out, long pitch, long tiles_per_row) {
    constexpr int K = (L * 16 + T - 1) / T;
    extern __shared__ char lds[];
    const unsigned gx = gridDim.x, x = blockIdx.x;
    const unsigned tix = (gx & 7u) ? x : (x & 7u) * (gx >> 3) + (x >> 3);
    const long col0 = (long)(tix % tiles_per_row) * 16;
    const long slab = (long)(tix / tiles_per_row) * (long)L * pitch;
    float2 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int e = threadIdx.x + T * k;
        e = e < L * 16 ? e : 0;
        v[k] = in[slab + (long)(e >> 4) * pitch + col0 + (e & 15)];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = threadIdx.x + T * k;
        if (e < L * 16) {
            using v2 = __attribute__((ext_vector_type(2))) float;
            v2 t; t.x = v[k].x + 1.f; t.y = v[k].y;
            __builtin_nontemporal_store(t, reinterpret_cast<v2*>(&out[slab + (long)(e >> 4) * pitch + col0 + (e & 15)]));
        }
    }
    if (lds[0] == 77 && threadIdx.x == 12345) out[0].x = 0.f;
}

template <int L, int T>
double run(const float2* in, float2* out, long n, long pitch, int reps, size_t lds) {
    const long tiles_per_row = pitch / 16, slabs = n / ((long)L * pitch);
    const long total = slabs * tiles_per_row;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)k_tile<L, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_tile<L, T>), dim3((unsigned)total), dim3(T), lds, 0, in, out, pitch, tiles_per_row);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_tile<L, T>), dim3((unsigned)total), dim3(T), lds, 0, in, out, pitch, tiles_per_row);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 2.0 * 8.0 * (double)(slabs * (long)L * pitch) * reps / (ms * 1e-3) / 1e9;
}

int main() {
    const long n = 240000000L;
    const size_t slack = 64u << 20;
    float2 *in; char* outbuf;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&outbuf, n * 8 + slack));
    CK(hipMemset(in, 0, n * 8)); CK(hipMemset(outbuf, 0, n * 8 + slack));
    printf("in %p out %p\n", (void*)in, (void*)outbuf);
    const size_t offs[] = {0, 64, 128, 192, 256, 384, 512, 640, 768, 896, 1024, 1152, 2048 + 128, 4096 + 256, 8192 + 384, 65536 + 128, 2097152 + 128, 33554432 + 8192 + 256};
    for (int rep = 0; rep < 2; ++rep)
        for (size_t off : offs) {
            float2* out = reinterpret_cast<float2*>(outbuf + off);
            printf("off %9zu B: 625 rows pitch 640, 1024 thr x2/CU %7.1f | 600 rows pitch 400000 %7.1f GB/s\n", off,
                   run<625, 1024>(in, out, n, 640, 5, 81920), run<600, 1024>(in, out, n, 400000, 5, 81920));
        }
    printf("in place               : 625 rows pitch 640 %7.1f | 600 rows pitch 400000 %7.1f GB/s\n",
           run<625, 1024>(in, in, n, 640, 5, 81920), run<600, 1024>(in, in, n, 400000, 5, 81920));
    return 0;
}
