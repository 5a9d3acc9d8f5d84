// Microbenchmark: streaming kernels with the pilot stage's traffic shape -- read one float array, write two of the same
// size (0.98 GB in, 1.97 GB out at cfg4) -- and with the de-emphasis shape (read one, write one, 0.39 GB each).
// K float4 loads per thread issued back to back, then the stores (plain or non-temporal).
//   hipcc --offload-arch=gfx950 -O3 -o stream_1r2w stream_1r2w.hip && ./stream_1r2w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int K, int NW, bool NT>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ in, float4* __restrict__ o1, float4* __restrict__ o2) {
    const size_t base = (size_t)blockIdx.x * 256 * K + threadIdx.x;
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = in[base + 256 * k];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float4 a = v[k];
        a.x += 1.f;
        if (NT) {
            __builtin_nontemporal_store(a.x, &o1[base + 256 * k].x); __builtin_nontemporal_store(a.y, &o1[base + 256 * k].y);
            __builtin_nontemporal_store(a.z, &o1[base + 256 * k].z); __builtin_nontemporal_store(a.w, &o1[base + 256 * k].w);
        } else o1[base + 256 * k] = a;
        if (NW == 2) {
            a.y += 1.f;
            if (NT) {
                __builtin_nontemporal_store(a.x, &o2[base + 256 * k].x); __builtin_nontemporal_store(a.y, &o2[base + 256 * k].y);
                __builtin_nontemporal_store(a.z, &o2[base + 256 * k].z); __builtin_nontemporal_store(a.w, &o2[base + 256 * k].w);
            } else o2[base + 256 * k] = a;
        }
    }
}

// the stencil kernels' store shape: thread t owns 8 consecutive floats (two float4 at a lane stride of 32 bytes), loads
// coalesced (LD8 = false: one dword per lane and sweep, like k_pilot_stage_h40) or in the same 32-byte shape
template <bool LD8, bool NT>
__global__ __launch_bounds__(256) void k_per8(const float* __restrict__ in, float* __restrict__ o1, float* __restrict__ o2) {
    const size_t base = (size_t)blockIdx.x * 2048;
    float v[8];
    if (LD8) {
        const float4* p = reinterpret_cast<const float4*>(in + base + threadIdx.x * 8);
        const float4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        __shared__ float sh[2048];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = in[base + threadIdx.x + 256 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) sh[threadIdx.x + 256 * k] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sh[threadIdx.x * 8 + k];
    }
    float4* d1 = reinterpret_cast<float4*>(o1 + base + threadIdx.x * 8);
    float4* d2 = reinterpret_cast<float4*>(o2 + base + threadIdx.x * 8);
    const float4 a = make_float4(v[0] + 1.f, v[1], v[2], v[3]), b = make_float4(v[4], v[5], v[6], v[7]);
    if (NT) {
        __builtin_nontemporal_store(a.x, &d1[0].x); __builtin_nontemporal_store(a.y, &d1[0].y); __builtin_nontemporal_store(a.z, &d1[0].z); __builtin_nontemporal_store(a.w, &d1[0].w);
        __builtin_nontemporal_store(b.x, &d1[1].x); __builtin_nontemporal_store(b.y, &d1[1].y); __builtin_nontemporal_store(b.z, &d1[1].z); __builtin_nontemporal_store(b.w, &d1[1].w);
        __builtin_nontemporal_store(a.x, &d2[0].x); __builtin_nontemporal_store(a.y, &d2[0].y); __builtin_nontemporal_store(a.z, &d2[0].z); __builtin_nontemporal_store(a.w, &d2[0].w);
        __builtin_nontemporal_store(b.x, &d2[1].x); __builtin_nontemporal_store(b.y, &d2[1].y); __builtin_nontemporal_store(b.z, &d2[1].z); __builtin_nontemporal_store(b.w, &d2[1].w);
    } else {
        d1[0] = a; d1[1] = b; d2[0] = a; d2[1] = b;
    }
}

template <bool LD8, bool NT>
void run8(const float* in, float* o1, float* o2, size_t n) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)(n / 2048);
    hipLaunchKernelGGL((k_per8<LD8, NT>), dim3(grid), dim3(256), 0, 0, in, o1, o2);
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_per8<LD8, NT>), dim3(grid), dim3(256), 0, 0, in, o1, o2);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("  8 floats per thread, %s loads, %s stores at a lane stride of 32 B: %7.1f us  %5.2f TB/s\n",
           LD8 ? "32-byte-per-lane" : "dword (via LDS) ", NT ? "non-temporal" : "plain       ", ms * 1e3 / 5,
           3.0 * 4.0 * grid * 2048 * 5 / (ms * 1e-3) / 1e12);
}

// read only (the value is needed, nothing is written) and write only
template <int K>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ in, float4* __restrict__ o1) {
    const size_t base = (size_t)blockIdx.x * 256 * K + threadIdx.x;
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = in[base + 256 * k];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k].x + v[k].y + v[k].z + v[k].w;
    if (s == 12345.f) o1[base] = v[0];
}
template <int NW>
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ o1, float4* __restrict__ o2, float c) {
    const size_t base = (size_t)blockIdx.x * 256 + threadIdx.x;
    o1[base] = make_float4(c, c, c, c);
    if (NW == 2) o2[base] = make_float4(c, c, c, c);
}
template <class F>
void timeit(const char* what, double bytes, F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("  %-46s %7.1f us  %5.2f TB/s\n", what, ms * 1e3 / 5, bytes * 5 / (ms * 1e-3) / 1e12);
}

template <int K, int NW, bool NT>
void run(const float4* in, float4* o1, float4* o2, size_t n4) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)(n4 / (256 * K));
    hipLaunchKernelGGL((k_stream<K, NW, NT>), dim3(grid), dim3(256), 0, 0, in, o1, o2);
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_stream<K, NW, NT>), dim3(grid), dim3(256), 0, 0, in, o1, o2);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("  %d write%s, K=%d float4 per thread, %s stores: %7.1f us  %5.2f TB/s\n", NW, NW == 2 ? "s" : " ", K,
           NT ? "non-temporal" : "plain       ", ms * 1e3 / 5, (1.0 + NW) * 16.0 * grid * 256 * K * 5 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t n4 = (size_t)1024 * 240000 / 4;      // float4 elements of one [1024][240000] float array
    float4 *in, *o1, *o2;
    CK(hipMalloc(&in, n4 * 16)); CK(hipMalloc(&o1, n4 * 16)); CK(hipMalloc(&o2, n4 * 16));
    CK(hipMemset(in, 0, n4 * 16));
    printf("pilot-stage shape: read 0.98 GB, write 2 x 0.98 GB\n");
    run<1, 2, false>(in, o1, o2, n4); run<1, 2, true>(in, o1, o2, n4);
    run<2, 2, false>(in, o1, o2, n4); run<2, 2, true>(in, o1, o2, n4);
    run<4, 2, true>(in, o1, o2, n4); run<8, 2, true>(in, o1, o2, n4);
    run8<true, false>((const float*)in, (float*)o1, (float*)o2, n4 * 4); run8<true, true>((const float*)in, (float*)o1, (float*)o2, n4 * 4);
    run8<false, false>((const float*)in, (float*)o1, (float*)o2, n4 * 4); run8<false, true>((const float*)in, (float*)o1, (float*)o2, n4 * 4);
    printf("one direction only, 0.98 GB per array\n");
    timeit("read only, 1 float4 per thread", 16.0 * n4, [&] { hipLaunchKernelGGL(k_read<1>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, in, o1); });
    timeit("read only, 4 float4 per thread", 16.0 * n4, [&] { hipLaunchKernelGGL(k_read<4>, dim3((unsigned)(n4 / 1024)), dim3(256), 0, 0, in, o1); });
    timeit("read only, 8 float4 per thread", 16.0 * n4, [&] { hipLaunchKernelGGL(k_read<8>, dim3((unsigned)(n4 / 2048)), dim3(256), 0, 0, in, o1); });
    timeit("write only, one array", 16.0 * n4, [&] { hipLaunchKernelGGL(k_write<1>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, o1, o2, 1.f); });
    timeit("write only, two arrays", 32.0 * n4, [&] { hipLaunchKernelGGL(k_write<2>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, o1, o2, 1.f); });
    printf("copy shape: read 0.98 GB, write 0.98 GB\n");
    run<1, 1, false>(in, o1, o2, n4); run<1, 1, true>(in, o1, o2, n4); run<4, 1, true>(in, o1, o2, n4); run<8, 1, true>(in, o1, o2, n4);
    const size_t a4 = (size_t)1024 * 48000 * 2 / 4;
    printf("de-emphasis shape: read 0.39 GB, write 0.39 GB\n");
    run<1, 1, true>(in, o1, o2, a4); run<4, 1, true>(in, o1, o2, a4);
    return 0;
}
