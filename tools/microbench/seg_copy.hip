// Microbenchmark: how fast can MI355X move "FFT pass" tiles?
//   mode 0: tile = L rows x SEG bytes, rows at stride P bytes (column pass: strided read, strided write same place)
//   mode 1: plain contiguous float4 copy (ceiling)
// Reports GB/s (read+write bytes) for several SEG and working-set sizes (HBM vs Infinity Cache resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// Each block handles tiles; a tile = L rows, each row a SEG-byte segment at row stride `pitch` elements (float2).
// Lanes run along the segment (SEG/8 float2 per row), then rows.
// XCD: workgroups go to the 8 XCDs round-robin; with XCD = true each XCD walks a contiguous eighth of the
// tile list, so neighbouring tiles (which share 128-byte lines when SEG < 128 B) meet in one L2.
template <int SEGE, bool XCD = false>  // segment length in float2 elements (8 B each)
__global__ __launch_bounds__(256) void k_tile_copy(const float2* __restrict__ in, float2* __restrict__ out,
                                                   long pitch, int L, long ntiles_per_row, long total_tiles) {
    const long per_xcd = (total_tiles + 7) / 8, lanes = gridDim.x / 8;
    for (long it = 0;; ++it) {
        long t;
        if (XCD) {
            const long j = (long)(blockIdx.x >> 3) + lanes * it;
            if (j >= per_xcd) break;
            t = (long)(blockIdx.x & 7) * per_xcd + j;
            if (t >= total_tiles) break;
        } else {
            t = blockIdx.x + it * (long)gridDim.x;
            if (t >= total_tiles) break;
        }
        // tile t covers columns [t % ntiles_per_row * SEGE, +SEGE) of block-row (t / ntiles_per_row)
        const long col0 = (t % ntiles_per_row) * SEGE;
        const long slab = (t / ntiles_per_row) * (long)L * pitch;
        for (int e = threadIdx.x; e < L * SEGE; e += 256) {
            const int l = e / SEGE, w = e % SEGE;
            const long a = slab + (long)l * pitch + col0 + w;
            float2 v = in[a];
            v.x += 1.0f;
            out[a] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_copy4(const float4* __restrict__ in, float4* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float4 v = in[i];
        v.x += 1.0f;
        out[i] = v;
    }
}

template <int SEGE, bool XCD = false>
double run_tile(const float2* in, float2* out, long n_elems, long pitch, int L, int reps) {
    const long ntiles_per_row = pitch / SEGE;
    const long slabs = n_elems / ((long)L * pitch);
    const long total = slabs * ntiles_per_row;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8;
    hipLaunchKernelGGL((k_tile_copy<SEGE, XCD>), dim3(grid), dim3(256), 0, 0, in, out, pitch, L, ntiles_per_row, total);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_tile_copy<SEGE, XCD>), dim3(grid), dim3(256), 0, 0, in, out, pitch, L, ntiles_per_row, total);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 2.0 * 8.0 * (double)(slabs * (long)L * pitch) * reps / (ms * 1e-3) / 1e9;
}

int main() {
    const long big = 240000000L;       // 1.92 GB per buffer: HBM
    float2 *in, *out;
    CK(hipMalloc(&in, big * 8)); CK(hipMalloc(&out, big * 8));
    CK(hipMemset(in, 0, big * 8)); CK(hipMemset(out, 0, big * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (long n : {big, 16000000L, 4000000L, 1000000L}) {   // 1.92 GB, 128 MB, 32 MB, 8 MB per buffer
        const int reps = n == big ? 5 : 50;
        hipLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, 0, (const float4*)in, (float4*)out, n / 2);
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, 0, (const float4*)in, (float4*)out, n / 2);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("copy4      n=%10ld (%7.1f MB/buf)                 : %8.1f GB/s\n", n, n * 8 / 1e6, 2.0 * 8 * n * reps / (ms * 1e-3) / 1e9);
        // column-pass like tiles: L = 500 rows, pitch = 480000 elements (3.84 MB) for the big one, smaller for cache-resident
        const long pitch = n >= 16000000L ? 400000 : 20000;
        const int L = (int)(n / pitch >= 500 ? 500 : n / pitch);
        printf("tile seg 32B  L=%d pitch=%ld : %8.1f GB/s\n", L, pitch, run_tile<4>(in, out, n, pitch, L, reps));
        printf("tile seg 64B  L=%d pitch=%ld : %8.1f GB/s\n", L, pitch, run_tile<8>(in, out, n, pitch, L, reps));
        printf("tile seg 64B  L=%d pitch=%ld XCD-aware order : %8.1f GB/s\n", L, pitch, (run_tile<8, true>(in, out, n, pitch, L, reps)));
        printf("tile seg 128B L=%d pitch=%ld : %8.1f GB/s\n", L, pitch, run_tile<16>(in, out, n, pitch, L, reps));
        printf("tile seg 128B L=%d pitch=%ld XCD-aware order : %8.1f GB/s\n", L, pitch, (run_tile<16, true>(in, out, n, pitch, L, reps)));
        printf("tile seg 256B L=%d pitch=%ld : %8.1f GB/s\n", L, pitch, run_tile<32>(in, out, n, pitch, L, reps));
    }
    return 0;
}
