// Microbenchmark: is a buffer built from SCATTERED physical chunks faster to write than what hipMalloc returns?
//
// pitch_sweep.hip / placement_census.hip: the rate at which a 1.92 GB buffer can be written -- and with it the rate of every tile
// copy into it -- depends on where hipMalloc put it (20 %), 2 GiB buffers and one 56 GiB block are uniformly slow, 1 832 MiB
// buffers sometimes fast.  Hypothesis: large physically contiguous, naturally aligned blocks are the slow ones.  Here the same
// virtual range is backed through the virtual memory API (hipMemAddressReserve / hipMemCreate / hipMemMap) by chunks of a
// given size, mapped in linear or in shuffled order, and probed like the census does (write sweep, read sweep, the first and
// the middle pass's tile copies from a hipMalloc'ed input).
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



#include <vector>
#include <algorithm>
#include <random>
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}
static float* g_sink;
template <class F> static double gbps(F launch, size_t bytes) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return bytes * 5 / (ms * 1e-3) / 1e9;
}

struct Vmm {
    void* va = nullptr; size_t size = 0;
    std::vector<hipMemGenericAllocationHandle_t> h;
    bool make(size_t bytes, size_t chunk, bool shuffle, unsigned seed) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return false;
        if (chunk < gran) chunk = gran;
        chunk = (chunk + gran - 1) / gran * gran;
        const size_t nchunk = (bytes + chunk - 1) / chunk;
        size = nchunk * chunk;
        if (hipMemAddressReserve(&va, size, 0, nullptr, 0) != hipSuccess) return false;
        h.resize(nchunk);
        for (size_t i = 0; i < nchunk; ++i)
            if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at chunk %zu\n", i); return false; }
        std::vector<size_t> pos(nchunk);
        for (size_t i = 0; i < nchunk; ++i) pos[i] = i;
        if (shuffle) { std::mt19937 g(seed); std::shuffle(pos.begin(), pos.end(), g); }
        for (size_t i = 0; i < nchunk; ++i)
            if (hipMemMap((char*)va + pos[i] * chunk, chunk, 0, h[i], 0) != hipSuccess) { printf("hipMemMap failed\n"); return false; }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(va, size, &acc, 1) != hipSuccess) { printf("hipMemSetAccess failed\n"); return false; }
        return true;
    }
};

static void probe(const char* name, const float2* in, char* outbase, long n) {
    float2* out = reinterpret_cast<float2*>(outbase + 128);
    float4* o4 = reinterpret_cast<float4*>(outbase);
    const size_t nf4 = (size_t)n / 2;
    const double w = gbps([&] { hipLaunchKernelGGL(k_write, dim3(256 * 16), dim3(256), 0, 0, o4, nf4); }, (size_t)n * 8);
    const double r = gbps([&] { hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, (const float4*)o4, nf4, g_sink); }, (size_t)n * 8);
    printf("%-34s write %5.0f read %5.0f | copy into it: pitch 640 %6.0f, pitch 400000 %6.0f | in place %6.0f %6.0f GB/s\n", name, w, r,
           run<600, 1024>(in, out, n, 640, 5, 81920), run<600, 1024>(in, out, n, 400000, 5, 81920),
           run<600, 1024>(out, out, n, 640, 5, 81920), run<600, 1024>(out, out, n, 400000, 5, 81920));
}

int main() {
    const long n = 240000000L;
    const size_t bytes = (size_t)n * 8 + 4096;
    CK(hipMalloc(&g_sink, 4));
    float2* in; CK(hipMalloc(&in, n * 8)); CK(hipMemset(in, 0, n * 8));
    for (int k = 0; k < 4; ++k) {
        char* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes));
        char name[64]; snprintf(name, sizeof(name), "hipMalloc #%d", k);
        probe(name, in, p, n);
    }
    const size_t M = (size_t)1 << 20;
    const size_t chunks[] = {2 * M, 8 * M, 32 * M, 128 * M, 512 * M, 2048 * M};
    for (size_t c : chunks)
        for (int sh = 0; sh < 2; ++sh) {
            if (c >= 2048 * M && sh) continue;
            Vmm v;
            char name[64]; snprintf(name, sizeof(name), "vmm chunks of %4zu MiB %s", c >> 20, sh ? "shuffled" : "linear");
            if (!v.make(bytes, c, sh != 0, 17)) { printf("%s: not available\n", name); continue; }
            CK(hipMemset(v.va, 0, bytes));
            probe(name, in, (char*)v.va, n);
        }
    // and as the INPUT: a scattered input with a hipMalloc'ed output
    Vmm vin;
    if (vin.make(bytes, 2 * M, true, 5)) {
        CK(hipMemset(vin.va, 0, bytes));
        char* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes));
        probe("scattered 2 MiB INPUT -> hipMalloc", (const float2*)vin.va, p, n);
        Vmm vout;
        if (vout.make(bytes, 2 * M, true, 9)) { CK(hipMemset(vout.va, 0, bytes)); probe("scattered INPUT -> scattered", (const float2*)vin.va, (char*)vout.va, n); }
    }
    return 0;
}
