// Timing-only proxy (round 6): WBFM's pilot stage fused into the first pass of the pilot pair FFT.
//
// Today: k_pilot_stage_h40 (theta -> m, p; 0.61-0.63 ms at cfg4) + the pair FFT's first pass (p -> T; 0.37-0.38 ms): 0.99 ms,
// 5 GB of traffic.  Fused, the first pass would compute p itself and the read of p (1 GB) would go.  The first pass
// owns a tile of 16 ADJACENT time samples x 480 rows (time t = 500 row + col); the pilot band-pass is an 81-tap FIR
// along t behind the discriminator and a 3-tap FIR, so each 16-sample run needs theta[t0 - 42 .. t0 + 57]: 100 samples
// for 16 outputs -- 6.25 x the loads (from L2 when neighbouring tiles run on one XCD), 6.25 x the discriminator / 3-tap
// work, and FIR windows of 96 values that feed only 16 outputs (the stand-alone stage feeds 2048 from one window).
//
// This kernel does exactly that data movement and arithmetic -- window loads, discriminator + 3-tap in LDS, 81-tap FIR
// from LDS, m and p stored tile-blocked, p pair-packed into the [480][16] complex tile -- and then only COPIES the tile
// out (no butterflies, no twiddles): a LOWER bound of the fused kernel.  CR rows per chunk, OP outputs per FIR thread.
//   hipcc --offload-arch=gfx950 -O3 -o pilot_fused_proxy pilot_fused_proxy.hip && ./pilot_fused_proxy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int B = 240000, ROWS = 480, COLS = 500, TILES = 32, WIN = 104;
__constant__ float g_taps[41];

__device__ __forceinline__ float wrap1(float d) { return d - 2.f * rintf(0.5f * d); }

template <int CR, int OP>
__global__ __launch_bounds__(512) void k_fused_proxy(const float* __restrict__ theta, float* __restrict__ m_out,
                                                     float* __restrict__ p_out, float2* __restrict__ t_out, int pairs) {
    constexpr int RM = 2 * CR;
    __shared__ __attribute__((aligned(16))) float tile[ROWS * 16 * 2];
    __shared__ __attribute__((aligned(16))) float win[RM * WIN];
    __shared__ __attribute__((aligned(16))) float mwin[RM * WIN];
    const int tid = threadIdx.x;
    // XCD-aware: XCD (id & 7) owns whole pairs, the 32 tiles of a pair run back to back on it (shared windows hit its L2)
    const unsigned id = blockIdx.x, xcd = id & 7u, slot = id >> 3;
    const int pair = (int)((slot / TILES) * 8 + xcd), tl = (int)(slot % TILES);
    if (pair >= pairs) return;
    const int col0 = tl * 16;
    const size_t blk = (size_t)ROWS * 16;                         // floats of one tile of one channel (tile-blocked m, p)
    for (int chunk = 0; chunk < ROWS / CR; ++chunk) {
        constexpr int NQ = RM * 26, NL = (NQ + 511) / 512;
        float4 v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            int idx = tid + 512 * j;
            idx = idx < NQ ? idx : NQ - 1;
            const int rm = idx / 26, q = idx - rm * 26;
            const int row = chunk * CR + (rm >> 1), ch = 2 * pair + (rm & 1);
            int t = row * COLS + col0 - 44 + 4 * q;
            t = t < 0 ? 0 : (t > B - 4 ? B - 4 : t);
            v[j] = *reinterpret_cast<const float4*>(theta + (size_t)ch * B + t);
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int idx = tid + 512 * j;
            if (idx < NQ) *reinterpret_cast<float4*>(&win[4 * idx]) = v[j];      // rm * 104 + 4 q = 4 idx
        }
        __syncthreads();
        // discriminator + 3-tap FIR (wbfm.py:77-79) on window positions 2 .. 101
        for (int idx = tid; idx < RM * 100; idx += 512) {
            const int rm = idx / 100, e = idx - rm * 100 + 2;
            const float* w = &win[rm * WIN + e];
            const float d0 = wrap1(w[-1] - w[-2]), d1 = wrap1(w[0] - w[-1]), d2 = wrap1(w[1] - w[0]);
            mwin[rm * WIN + e] = 0.54f * d1 + 0.23f * (d0 + d2);
        }
        __syncthreads();
        // 81-tap symmetric FIR: outputs at window positions 44 .. 59
        constexpr int TASKS = RM * 16 / OP;
        for (int task = tid; task < TASKS; task += 512) {
            const int rm = task / (16 / OP), o0 = (task - rm * (16 / OP)) * OP;
            const float* w = &mwin[rm * WIN + 4 + o0];
            float acc[OP];
#pragma unroll
            for (int r = 0; r < OP; ++r) acc[r] = 0.f;
#pragma unroll
            for (int k = 0; k < 80 + OP; ++k) {
                const float x = w[k];
#pragma unroll
                for (int r = 0; r < OP; ++r) {
                    const int j = k - r;
                    if (j >= 0 && j <= 80) acc[r] = fmaf(g_taps[j < 40 ? 40 - j : j - 40], x, acc[r]);
                }
            }
            const int row = chunk * CR + (rm >> 1), mem = rm & 1, ch = 2 * pair + mem;
            float* md = m_out + (size_t)ch * (blk * TILES) + (size_t)tl * blk + row * 16 + o0;
            float* pd = p_out + (size_t)ch * (blk * TILES) + (size_t)tl * blk + row * 16 + o0;
#pragma unroll
            for (int r = 0; r < OP; ++r) {
                md[r] = w[40 + r];
                pd[r] = acc[r];
                tile[(row * 16 + o0 + r) * 2 + mem] = acc[r];
            }
        }
        __syncthreads();
    }
    // "first pass": the tile leaves as 128-byte segments (a real pass would transform the 480 rows first)
    const float2* tc = reinterpret_cast<const float2*>(tile);
    for (int e = tid; e < ROWS * 16; e += 512) {
        const int row = e >> 4, w = e & 15;
        if (col0 + w < COLS) t_out[(size_t)pair * (ROWS * 512) + row * 512 + col0 + w] = tc[e];
    }
}

template <int CR, int OP>
static void run(const float* th, float* m, float* p, float2* t, int pairs) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)((pairs + 7) / 8 * 8 * TILES);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_fused_proxy<CR, OP>), dim3(grid), dim3(512), 0, 0, th, m, p, t, pairs);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_fused_proxy<CR, OP>), dim3(grid), dim3(512), 0, 0, th, m, p, t, pairs);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fused_proxy<CR, OP>, 512, 0));
    printf("rows per chunk %2d, outputs per FIR thread %d, LDS %6zu B, %d workgroup(s) per CU: %.3f ms per 1024 channels\n", CR,
           OP, sizeof(float) * (ROWS * 32 + 2 * 2 * CR * WIN), occ, ms / 5);
}

int main() {
    const int C = 1024, pairs = C / 2;
    float *th, *m, *p;
    float2* t;
    CK(hipMalloc(&th, (size_t)C * B * 4));
    CK(hipMalloc(&m, (size_t)C * ROWS * 16 * TILES * 4));
    CK(hipMalloc(&p, (size_t)C * ROWS * 16 * TILES * 4));
    CK(hipMalloc(&t, (size_t)pairs * ROWS * 512 * 8));
    CK(hipMemset(th, 0, (size_t)C * B * 4));
    float g[41];
    for (int i = 0; i < 41; ++i) g[i] = 1.f / (1.f + i);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_taps), g, sizeof(g)));
    printf("today: k_pilot_stage_h40 0.61-0.63 ms + pair FFT first pass 0.37-0.38 ms = 0.99 ms (cfg4, profiles/r06_*)\n");
    run<8, 1>(th, m, p, t, pairs);
    run<8, 2>(th, m, p, t, pairs);
    run<8, 4>(th, m, p, t, pairs);
    run<16, 2>(th, m, p, t, pairs);
    run<16, 4>(th, m, p, t, pairs);
    run<16, 8>(th, m, p, t, pairs);
    run<32, 4>(th, m, p, t, pairs);
    run<32, 8>(th, m, p, t, pairs);
    return 0;
}
