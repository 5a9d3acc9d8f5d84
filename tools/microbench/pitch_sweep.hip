// Microbenchmark: a strided pass's rate against its ROW PITCH.  Tile copy of 600 rows x 128 bytes (the wideband FFT's first
// pass), every load of a thread in flight, two 1024-thread workgroups per CU; the pitch goes from 640 points (a tile spans
// 3 MB contiguous) to 400 000 points (every row of a tile in another 2 MiB page).  Question: is the 8-11 % between the
// first / last pass's pattern and the middle pass's a matter of address translation (one page per row: a step at a pitch
// of 2 MiB) or of DRAM locality (gradual)?  Output displaced by 128 bytes (dst_offset.hip).
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


// placement probes: what does ONE buffer tell about itself?
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
static double time_ms(void (*launch)(void*, size_t), void* p, size_t n, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(p, n);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch(p, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
static float* g_sink;
static void l_read(void* p, size_t n) { hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, (const float4*)p, n, g_sink); }
static void l_write(void* p, size_t n) { hipLaunchKernelGGL(k_write, dim3(256 * 16), dim3(256), 0, 0, (float4*)p, n); }

int main(int argc, char** argv) {
    const long n = 240000000L;
    if (argc > 1 && argv[1][0] == 'd') {   // distance: in and out inside ONE big allocation, out = in + D
        const size_t G = (size_t)1 << 30, big_b = 56 * G;
        char* big; CK(hipMalloc(&big, big_b)); CK(hipMemset(big, 0, big_b));
        printf("one allocation of 56 GiB at %p; copy of 1.92 GB: in at A GiB, out at A + D GiB + 128 B\n", (void*)big);
        for (int A = 0; A <= 8; A += 8)
            for (int D = 2; D <= 44; ++D) {
                float2* in2 = reinterpret_cast<float2*>(big + A * G);
                float2* out2 = reinterpret_cast<float2*>(big + (A + D) * G + 128);
                printf("A %2d D %2d: pitch 640 %7.1f | pitch 400000 %7.1f GB/s\n", A, D, run<600, 1024>(in2, out2, n, 640, 3, 81920),
                       run<600, 1024>(in2, out2, n, 400000, 3, 81920));
            }
        return 0;
    }
    if (argc > 1) {   // placement: several (in, out) pairs allocated in ONE process, each measured twice
        float2* ins[6]; char* outs[6];
        for (int a = 0; a < 6; ++a) {
            CK(hipMalloc(&ins[a], n * 8)); CK(hipMalloc(&outs[a], n * 8 + 4096));
            CK(hipMemset(ins[a], 0, n * 8)); CK(hipMemset(outs[a], 0, n * 8 + 4096));
        }
        CK(hipMalloc(&g_sink, 4));
        for (int a = 0; a < 6; ++a) {
            const size_t nf4 = (size_t)n / 2;
            printf("buffer in%d : read %6.0f write %6.0f | in place 640: %6.0f 400000: %6.0f GB/s\n", a, n * 8 / time_ms(l_read, ins[a], nf4, 5) / 1e6,
                   n * 8 / time_ms(l_write, ins[a], nf4, 5) / 1e6, run<600, 1024>(ins[a], ins[a], n, 640, 5, 81920), run<600, 1024>(ins[a], ins[a], n, 400000, 5, 81920));
            float2* o = reinterpret_cast<float2*>(outs[a]);
            printf("buffer out%d: read %6.0f write %6.0f | in place 640: %6.0f 400000: %6.0f GB/s\n", a, n * 8 / time_ms(l_read, o, nf4, 5) / 1e6,
                   n * 8 / time_ms(l_write, o, nf4, 5) / 1e6, run<600, 1024>(o, o, n, 640, 5, 81920), run<600, 1024>(o, o, n, 400000, 5, 81920));
        }
        for (int rep = 0; rep < 1; ++rep)
            for (int a = 0; a < 6; ++a) {
                float2* out = reinterpret_cast<float2*>(outs[a] + 128);
                printf("pair %d (in %p out %p): pitch 640 %7.1f | pitch 400000 %7.1f GB/s\n", a, (void*)ins[a], (void*)outs[a],
                       run<600, 1024>(ins[a], out, n, 640, 5, 81920), run<600, 1024>(ins[a], out, n, 400000, 5, 81920));
            }
        // crossed: input of pair 0 with the outputs of the others
        for (int a = 0; a < 6; ++a) {
            float2* out = reinterpret_cast<float2*>(outs[a] + 128);
            printf("in 0 -> out %d: pitch 640 %7.1f | pitch 400000 %7.1f GB/s\n", a,
                   run<600, 1024>(ins[0], out, n, 640, 5, 81920), run<600, 1024>(ins[0], out, n, 400000, 5, 81920));
        }
        return 0;
    }
    float2 *in; char* outbuf;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&outbuf, n * 8 + 4096));
    CK(hipMemset(in, 0, n * 8)); CK(hipMemset(outbuf, 0, n * 8 + 4096));
    float2* out = reinterpret_cast<float2*>(outbuf + 128);
    const long pitches[] = {640, 2048, 8192, 16384, 32768, 65536, 131072, 200000, 262144, 300000, 400000};
    for (int rep = 0; rep < 2; ++rep)
        for (long p : pitches)
            printf("pitch %7ld points = %8.1f KiB: 600 rows %7.1f GB/s\n", p, p * 8 / 1024.0, run<600, 1024>(in, out, n, p, 5, 81920));
    return 0;
}
