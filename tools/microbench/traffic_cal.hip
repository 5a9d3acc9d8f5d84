// Calibration kernels for the memory-side PMC counters (FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_* / TCC_EA0_WRREQ_*):
// every kernel moves a KNOWN number of bytes from a 1 GiB source to a 1 GiB destination (well beyond the 256 MiB
// Infinity Cache), in the access shapes the radio path uses.  tools/profile_traffic.sh runs it under rocprofv3 and
// tools/traffic_summary.py turns counter values into a per-shape factor "true bytes / reported FETCH_SIZE bytes".
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/traffic_cal.hip -o tools/microbench/traffic_cal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// T-byte-per-lane streaming copy: 4 (float), 8 (float2), 16 (float4)
template <class T>
__global__ __launch_bounds__(256) void cal_stream(const T* __restrict__ in, T* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = in[i];
}

// FFT-pass shaped copy: tiles of L rows x SEG float2 (SEG * 8 bytes per row segment), rows `pitch` float2 apart,
// OFF float2 of misalignment (OFF = 8: every 128-byte segment straddles two 128-byte lines).
template <int SEG, int OFF>
__global__ __launch_bounds__(256) void cal_tile(const float2* __restrict__ in, float2* __restrict__ out, long pitch,
                                                int L, long tiles_per_row, long total) {
    for (long t = blockIdx.x; t < total; t += gridDim.x) {
        const long col0 = (t % tiles_per_row) * SEG + OFF;
        const long slab = (t / tiles_per_row) * (long)L * pitch;
        for (int e = threadIdx.x; e < L * SEG; e += 256) {
            const int l = e / SEG, w = e % SEG;
            const long a = slab + (long)l * pitch + col0 + w;
            out[a] = in[a];
        }
    }
}

// two real channels read with 4 bytes per lane (16 lanes = 64 bytes per channel and row), written as one complex
// row segment of 128 bytes: the LoadRealPair shape of the pilot pair FFT
__global__ __launch_bounds__(256) void cal_real_pair(const float* __restrict__ a, const float* __restrict__ b,
                                                     float2* __restrict__ out, long pitch, int L, long tiles_per_row,
                                                     long total) {
    for (long t = blockIdx.x; t < total; t += gridDim.x) {
        const long col0 = (t % tiles_per_row) * 16;
        const long slab = (t / tiles_per_row) * (long)L * pitch;
        for (int e = threadIdx.x; e < L * 16; e += 256) {
            const int l = e / 16, w = e % 16;
            const long i = slab + (long)l * pitch + col0 + w;
            out[i] = make_float2(a[i], b[i]);
        }
    }
}

int main() {
    const long n2 = 1L << 27;   // float2 elements: 1 GiB
    float2 *in, *out;
    CK(hipMalloc(&in, n2 * 8));
    CK(hipMalloc(&out, n2 * 8));
    CK(hipMemset(in, 0, n2 * 8));
    CK(hipMemset(out, 0, n2 * 8));
    const int grid = 256 * 16;
    // expected bytes: every kernel reads B_r and writes B_w as printed
    hipLaunchKernelGGL(cal_stream<float>, dim3(grid), dim3(256), 0, 0, (const float*)in, (float*)out, n2 * 2);
    printf("cal_stream<float>   read %ld write %ld\n", n2 * 8, n2 * 8);
    hipLaunchKernelGGL(cal_stream<float2>, dim3(grid), dim3(256), 0, 0, in, out, n2);
    printf("cal_stream<float2>  read %ld write %ld\n", n2 * 8, n2 * 8);
    hipLaunchKernelGGL(cal_stream<float4>, dim3(grid), dim3(256), 0, 0, (const float4*)in, (float4*)out, n2 / 2);
    printf("cal_stream<float4>  read %ld write %ld\n", n2 * 8, n2 * 8);
    const long pitch = 400000;            // 3.2 MB rows, like pass 1 of the wideband FFT
    const int L = 320;
    const long slabs = n2 / ((long)L * pitch);          // 1
    const long bytes = slabs * (long)L * pitch * 8;
    hipLaunchKernelGGL((cal_tile<16, 0>), dim3(grid), dim3(256), 0, 0, in, out, pitch, L, pitch / 16, slabs * (pitch / 16));
    printf("cal_tile<16,0>      read %ld write %ld   (128-byte aligned segments)\n", bytes, bytes);
    hipLaunchKernelGGL((cal_tile<16, 8>), dim3(grid), dim3(256), 0, 0, in, out, pitch, L, pitch / 16 - 1,
                       slabs * (pitch / 16 - 1));
    printf("cal_tile<16,8>      read %ld write %ld   (128-byte segments straddling two lines)\n",
           slabs * L * (pitch - 16) * 8, slabs * L * (pitch - 16) * 8);
    hipLaunchKernelGGL((cal_tile<8, 0>), dim3(grid), dim3(256), 0, 0, in, out, pitch, L, pitch / 8, slabs * (pitch / 8));
    printf("cal_tile<8,0>       read %ld write %ld   (64-byte segments)\n", bytes, bytes);
    hipLaunchKernelGGL((cal_tile<4, 0>), dim3(grid), dim3(256), 0, 0, in, out, pitch, L, pitch / 4, slabs * (pitch / 4));
    printf("cal_tile<4,0>       read %ld write %ld   (32-byte segments)\n", bytes, bytes);
    {   // real pair: two float planes of L x pitch, one complex output
        const float* a = (const float*)in;
        const float* b = a + (long)L * pitch + 4096;
        hipLaunchKernelGGL(cal_real_pair, dim3(grid), dim3(256), 0, 0, a, b, out, pitch, L, pitch / 16, pitch / 16);
        printf("cal_real_pair       read %ld write %ld   (2 x 64-byte real segments -> 128-byte complex)\n",
               2L * L * pitch * 4, (long)L * pitch * 8);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
