// Microbenchmark: cache-policy bits of the streaming loads and stores of a tile copy (625 rows x 128 B at a pitch of 640 points,
// 600 rows at 400 000; two 1024-thread workgroups per CU; output displaced by 128 bytes).  Stores: plain, nt (what the FFT
// passes use), sc1, sc0 sc1, sc1 nt, sc0.  Loads: plain, nt, sc1, sc0 sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int L, int T, int ST, int LD>
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
        const float2* src = &in[slab + (long)(e >> 4) * pitch + col0 + (e & 15)];
        if (LD == 0) v[k] = *src;
        else if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v[k]) : "v"(src) : "memory");
        else if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[k]) : "v"(src) : "memory");
        else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v[k]) : "v"(src) : "memory");
    }
    if (LD != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = threadIdx.x + T * k;
        if (e < L * 16) {
            using v2 = __attribute__((ext_vector_type(2))) float;
            v2 t; t.x = v[k].x + 1.f; t.y = v[k].y;
            float2* dst = &out[slab + (long)(e >> 4) * pitch + col0 + (e & 15)];
            if (ST == 0) *reinterpret_cast<v2*>(dst) = t;
            else if (ST == 1) __builtin_nontemporal_store(t, reinterpret_cast<v2*>(dst));
            else if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(t) : "memory");
            else if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(dst), "v"(t) : "memory");
            else if (ST == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" :: "v"(dst), "v"(t) : "memory");
            else asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(dst), "v"(t) : "memory");
        }
    }
    if (lds[0] == 77 && threadIdx.x == 12345) out[0].x = 0.f;
}

template <int L, int T, int ST, int LD>
double run(const float2* in, float2* out, long n, long pitch, int reps, size_t lds) {
    const long tiles_per_row = pitch / 16, slabs = n / ((long)L * pitch);
    const long total = slabs * tiles_per_row;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)k_tile<L, T, ST, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_tile<L, T, ST, LD>), dim3((unsigned)total), dim3(T), lds, 0, in, out, pitch, tiles_per_row);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_tile<L, T, ST, LD>), dim3((unsigned)total), dim3(T), lds, 0, in, out, pitch, tiles_per_row);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 2.0 * 8.0 * (double)(slabs * (long)L * pitch) * reps / (ms * 1e-3) / 1e9;
}

int main() {
    const long n = 240000000L;
    float2 *in; char* outbuf;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&outbuf, n * 8 + 4096));
    CK(hipMemset(in, 0, n * 8)); CK(hipMemset(outbuf, 0, n * 8 + 4096));
    float2* out = reinterpret_cast<float2*>(outbuf + 128);
    for (int rep = 0; rep < 2; ++rep) {
#define ROW(ST, LD, name) printf("%-22s 625 x pitch 640 %7.1f | 600 x pitch 400000 %7.1f GB/s\n", name, \
        run<625, 1024, ST, LD>(in, out, n, 640, 5, 81920), run<600, 1024, ST, LD>(in, out, n, 400000, 5, 81920));
        ROW(0, 0, "store plain") ROW(1, 0, "store nt") ROW(2, 0, "store sc1") ROW(3, 0, "store sc0 sc1") ROW(4, 0, "store sc1 nt") ROW(5, 0, "store sc0")
        ROW(1, 1, "load nt, store nt") ROW(1, 2, "load sc1, store nt") ROW(1, 3, "load sc0 sc1, store nt")
    }
    return 0;
}
