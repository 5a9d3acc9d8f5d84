// Streaming passes of long transforms with the tile LOADED BY LDS-DMA (global_load_lds_dwordx4) and
// double-buffered: one persistent 1024-thread workgroup per CU owns both 80 KiB halves of the LDS; while it
// transforms and stores tile i out of one half, the 1 KiB wave-instructions that fetch tile i+1 into the
// other half are in flight -- they cost no VGPR and no ds_write, and nothing in the iteration waits for
// them until the next one starts.  (Round 3's persistent form prefetched through registers, which a
// 64-VGPR kernel does not have; this one has 128 and needs none for the prefetch.)
//
// Everything that touches the VM counter inside the tile loop is either an LDS-DMA / asm load whose wait
// is written here by hand, or a store, which nobody waits for.  gfx9 retires vector-memory operations in
// issue order (one counter for loads and stores), so every wait is a COUNTED s_waitcnt vmcnt(K):
//   queue of one wave, iteration i:  [stores(i-1)] [twiddle loads(i)] [DMA(i+1)] [stores(i)] ...
//   * before the last stage (first use of the inter-pass twiddles): vmcnt(<= DMA instructions of the wave)
//   * top of iteration i+1 (tile i+1 must have landed): vmcnt(<= stores of the wave in iteration i)
// A compiler-visible load inside the loop would end in a compiler-made vmcnt(0) at its first use and drain
// the prefetch: the stage twiddles are loop-invariant and live in registers, the inter-pass twiddle's
// table entries come through asm loads.
//
// LDS images (what a lane-linear 1 KiB DMA write can produce; the swizzle is on the SOURCE address):
//   strided pass: [l][16 lines], exactly the image of k_fft_tile -- a wave-instruction is 8 rows of 128 B;
//   rows pass (lines contiguous in memory): [l / 8][line][4 chunks of 2 points], chunk c of line w holds
//   points 8 (l / 8) + 2 (c ^ (w >> 2)) + {0, 1}: a 16-lane row access covers sixteen distinct 16-byte
//   groups of the 64 banks and the neighbouring butterfly row (the other half-wave) takes the other 8 bytes
//   of each group.
#pragma once

namespace rcfm {
RCFM_NS_OPEN
namespace fftk {

#ifndef RCFM_FFT_DMA_NT
#define RCFM_FFT_DMA_NT 0
#endif
// RCFM_DMA_ABLATE (timing experiments, wrong results): 1 = skip every stage but the last (data movement only)
#ifndef RCFM_DMA_ABLATE
#define RCFM_DMA_ABLATE 0
#endif

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
#if RCFM_FFT_DMA_NT
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}

typedef float dma_v2f __attribute__((ext_vector_type(2)));

// -DRCFM_DMA_TRACE (timing experiments): workgroup 0 records s_memtime at the phase boundaries of its first 32 tiles;
// the launcher prints the differences (fft_kernel.h, launch_fft_tile_one).
#ifdef RCFM_DMA_TRACE
static __device__ long long g_dma_trace[32 * 8];
#define RCFM_DMA_TRACE_POINT(i) \
    do { if (blockIdx.x == 0 && threadIdx.x == 0 && trace_it < 32) g_dma_trace[trace_it * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define RCFM_DMA_TRACE_POINT(i) do { } while (0)
#endif

// big_twiddle() with the table entry already in hand.
__device__ __forceinline__ float2 big_twiddle_from(const FftPassDev& d, unsigned e, dma_v2f c) {
    const float th = d.fine_step * (float)(e & ((1u << d.fine_bits) - 1u));
    const float t2 = th * th;
    const float s = th * (1.f - t2 * (1.f / 6.f) * (1.f - t2 * (1.f / 20.f)));
    const float co = 1.f - t2 * 0.5f * (1.f - t2 * (1.f / 12.f) * (1.f - t2 * (1.f / 30.f)));
    return cmul(make_float2(c.x, c.y), make_float2(co, -s));
}

template <int K>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}

template <bool ROWS>
struct DmaSlot {
    static __device__ __forceinline__ int at(int row, int w) {
        if constexpr (ROWS) return (row >> 3) * 128 + w * 8 + ((((row >> 1) ^ (w >> 2)) & 3) << 1) + (row & 1);
        else return row * W + w;
    }
};

// One stage, LDS -> LDS, twiddle W_L^(kp step) of butterfly `it` of this thread in twr[it] (loop-invariant).
template <int L, int R, int MT, bool ROWS, int RG, bool SWAPIN>
__device__ __forceinline__ void dma_stage(float2* tile, const float2* twr, int w, int rg) {
    constexpr int m = MT / R, rows = L / R, nit = (rows + RG - 1) / RG;
#pragma unroll
    for (int it = 0; it < nit; ++it) {
        const int b = rg + RG * it;
        if ((rows % RG == 0) || b < rows) {
            const int g = b / m, kp = b - g * m;
            const int base = g * MT + kp;
            float2 v[R];
            float2 pw[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                v[q] = tile[DmaSlot<ROWS>::at(base + q * m, w)];
                if constexpr (SWAPIN) v[q] = make_float2(v[q].y, v[q].x);
            }
            dft_pa<R>(v);
            twiddle_powers<R>(twr[it], pw);
#pragma unroll
            for (int q = 1; q < R; ++q) v[dft_slot<R>(q)] = cmul(v[dft_slot<R>(q)], pw[q]);
#pragma unroll
            for (int q = 0; q < R; ++q) tile[DmaSlot<ROWS>::at(base + q * m, w)] = v[dft_slot<R>(q)];
        }
    }
}

template <int L, int R, int MT, int RG>
__device__ __forceinline__ void dma_stage_twiddles(const float2* table, float2* twr, int rg) {
    constexpr int m = MT / R, step = L / MT, rows = L / R, nit = (rows + RG - 1) / RG;
#pragma unroll
    for (int it = 0; it < nit; ++it) {
        int b = rg + RG * it;
        if (rows % RG != 0) b = b < rows ? b : 0;
        twr[it] = table[(b % m) * step];
    }
}

constexpr int dma_nit(int L, int R, int RG) { return (L / R + RG - 1) / RG; }

// STORE_MIN: a lower bound of the store instructions every wave issues per tile (0 = unknown: the top of the loop
// then waits for the stores too).
template <int L, int R0, int R1, int R2, int R3, bool ROWS, bool SWAPIN, class StoreOp, bool COUNT_STORES>
__global__ __launch_bounds__(1024, 4) void k_fft_tile_dma(FftPassDev d, const float2* __restrict__ in, StoreOp store,
                                                          dim3 vgrid) {
    constexpr int T = 1024;
    constexpr int S = (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1);
    static_assert(S >= 3 && R0 * R1 * R2 * R3 == L, "bad radix list");
    constexpr int RL = (S == 3) ? R2 : R3;
    constexpr int RG = T / W;
    constexpr int LB = ROWS ? (L + 7) / 8 * 8 : L;          // rows of the LDS image
    constexpr int kTile = LB * W;                            // float2 per buffer
    constexpr int NG = (LB + 7) / 8;                         // 1 KiB wave-instructions per tile (8 rows each; 625 rows: the last one is one row)
    constexpr int ND_MIN = NG / 16;                          // every wave issues at least this many
    constexpr int ND_MAX = (NG + 15) / 16;
    __shared__ __attribute__((aligned(1024))) float2 lds[2 * kTile];
    const FftPass& p = d.p;

    const int tid = threadIdx.x;
    const int w = tid & (W - 1), rg = tid >> 4;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)lds;

    // loop-invariant stage twiddles (compiler-visible loads, complete before the first DMA is issued)
    float2 tw0[dma_nit(L, R0, RG)], tw1[dma_nit(L, R1, RG)], tw2[S == 4 ? dma_nit(L, R2, RG) : 1];
    dma_stage_twiddles<L, R0, L, RG>(d.stage_tw, tw0, rg);
    dma_stage_twiddles<L, R1, L / R0, RG>(d.stage_tw, tw1, rg);
    if constexpr (S == 4) dma_stage_twiddles<L, R2, L / (R0 * R1), RG>(d.stage_tw, tw2, rg);
    else tw2[0] = make_float2(1.f, 0.f);
    // a use the compiler sees: its wait for those loads happens HERE, not as a vmcnt(0) at their first use inside the loop
#pragma unroll
    for (int i = 0; i < dma_nit(L, R0, RG); ++i) asm volatile("" : "+v"(tw0[i].x), "+v"(tw0[i].y));
#pragma unroll
    for (int i = 0; i < dma_nit(L, R1, RG); ++i) asm volatile("" : "+v"(tw1[i].x), "+v"(tw1[i].y));
#pragma unroll
    for (int i = 0; i < (S == 4 ? dma_nit(L, R2, RG) : 1); ++i) asm volatile("" : "+v"(tw2[i].x), "+v"(tw2[i].y));

    constexpr int rowsL = L / RL, nitL = (rowsL + RG - 1) / RG;
    // stores every wave is sure to issue per tile: the sweeps in which all of its butterfly rows exist
    constexpr int kStoreMin = COUNT_STORES ? RL * (rowsL / RG) : 0;
    auto kbase = [](int g) -> int {
        if constexpr (S == 3) {
            constexpr int w1 = L / (R0 * RL);
            const int q1 = g / w1, q2 = g - q1 * w1;
            return q1 + R0 * q2;
        } else {
            constexpr int w1 = L / (R0 * RL), w2 = L / (R0 * R1 * RL);
            const int q1 = g / w1, r1 = g - q1 * w1;
            const int q2 = r1 / w2, q3 = r1 - q2 * w2;
            return q1 + R0 * (q2 + R1 * q3);
        }
    };

    struct Coord {
        int64_t in_base, out_base;
        int i0, wvalid, o1, o2, batch;
    };
    auto coord = [&](unsigned vid) -> Coord {
        const VBlock vb = vblock_of(vid, vgrid.x, vgrid.y, vgrid.z);
        const BlockPos bp = block_pos(vb);
        Coord c;
        c.batch = (int)bp.batch;
        const unsigned o = vb.y, n_o2 = (unsigned)p.n_o2;
        c.o1 = (int)(n_o2 == 1 ? o : o / n_o2);
        c.o2 = (int)(n_o2 == 1 ? 0 : o - (unsigned)c.o1 * n_o2);
        c.i0 = (int)bp.tile * W;
        const int left = (int)p.n_inner - c.i0;
        c.wvalid = left < W ? left : W;
        c.in_base = (int64_t)c.batch * d.in_batch + c.o1 * p.in_o1 + c.o2 * p.in_o2 + (int64_t)c.i0 * p.in_i;
        c.out_base = (int64_t)c.batch * d.out_batch + c.o1 * p.out_o1 + c.o2 * p.out_o2 + c.i0;
        return c;
    };
    // the 1 KiB pieces of a tile: wave v issues pieces v, v + 16, ...
    auto issue = [&](const Coord& c, int buf) {
        const unsigned dst = lds0 + (unsigned)buf * (unsigned)(kTile * sizeof(float2));
#pragma unroll
        for (int j = 0; j < ND_MAX; ++j) {
            const int g = wave + 16 * j;
            if (ND_MIN == ND_MAX || g < NG) {   // wave-uniform
                const float2* src;
                if constexpr (ROWS) {
                    const int wl = lane >> 2, ch = lane & 3;
                    const int wc = wl < c.wvalid ? wl : 0;
                    src = in + c.in_base + (int64_t)wc * p.in_i + g * 8 + 2 * (ch ^ (wl >> 2));
                } else {
                    const int r = lane >> 3, ch = lane & 7;
                    const int row = g * 8 + r;
                    src = in + c.in_base + (int64_t)row * p.in_l + 2 * ch;
                    if (L % 8 != 0 && row >= L) continue;   // the last piece of a 625-row tile is one row: the other lanes sit out
                }
                glds16(src, __builtin_amdgcn_readfirstlane(dst + (unsigned)g * 1024u));
            }
        }
    };

    const unsigned vtotal = vgrid.x * vgrid.y * vgrid.z;
    unsigned vid = blockIdx.x;
    if (vid >= vtotal) return;
    Coord cur = coord(vid);
    issue(cur, 0);
    int buf = 0;
    [[maybe_unused]] int trace_it = 0;
#pragma unroll 1
    for (; vid < vtotal; vid += gridDim.x, buf ^= 1) {
        float2* tile = lds + buf * kTile;
        RCFM_DMA_TRACE_POINT(0);
        // inter-pass twiddle factors of this tile (asm loads: their wait is the counted one before the last stage)
        const unsigned f = (unsigned)(cur.o1 * p.tw_o1 + cur.o2 * p.tw_o2 + (int64_t)(cur.i0 + w) * p.tw_i);
        constexpr bool PTW = !ROWS;
        dma_v2f cD, cT[nitL];
        unsigned eD = 0, eT[nitL];
        if constexpr (PTW) {
            eD = f * (unsigned)(L / RL);
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(cD) : "v"(d.coarse + (eD >> d.fine_bits)) : "memory");
#pragma unroll
            for (int it = 0; it < nitL; ++it) {
                int g = rg + RG * it;
                if (rowsL % RG != 0) g = g < rowsL ? g : 0;
                eT[it] = f * (unsigned)kbase(g);
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(cT[it]) : "v"(d.coarse + (eT[it] >> d.fine_bits)) : "memory");
            }
        }
        // (the next tile's coordinates cost four integer divisions: under the wait, not between the barrier and the stages)
        const unsigned vnext = vid + gridDim.x;
        Coord nxt = cur;
        if (vnext < vtotal) nxt = coord(vnext);
        // tile i has landed: this wave's pieces by its own count, the others' by the barrier
        if (vid == blockIdx.x) {
            if constexpr (PTW) wait_vm<1 + nitL>(); else wait_vm<0>();
        } else {
            wait_vm<(PTW ? 1 + nitL : 0) + kStoreMin>();
        }
        lds_barrier();
        RCFM_DMA_TRACE_POINT(1);
        if (vnext < vtotal) issue(nxt, buf ^ 1);

        RCFM_DMA_TRACE_POINT(2);
#if RCFM_DMA_ABLATE & 1
        if (d.debug == 12345) {   // never true: keeps the stages' code and registers, skips their time
#endif
        dma_stage<L, R0, L, ROWS, RG, SWAPIN>(tile, tw0, w, rg);
        lds_barrier();
        RCFM_DMA_TRACE_POINT(3);
        dma_stage<L, R1, L / R0, ROWS, RG, false>(tile, tw1, w, rg);
        lds_barrier();
        RCFM_DMA_TRACE_POINT(4);
        if constexpr (S == 4) {
            dma_stage<L, R2, L / (R0 * R1), ROWS, RG, false>(tile, tw2, w, rg);
            lds_barrier();
        }
#if RCFM_DMA_ABLATE & 1
        }
#endif
        RCFM_DMA_TRACE_POINT(5);

        // ---- last stage: LDS -> registers -> memory
        float2 D = make_float2(1.f, 0.f);
        if constexpr (PTW) {
            // the table entries are older than the prefetch: leave the prefetch in flight
            if (vnext < vtotal) {
                if constexpr (nitL == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(cD), "+v"(cT[0]) : "n"(ND_MIN) : "memory");
                else if constexpr (nitL == 2)
                    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(cD), "+v"(cT[0]), "+v"(cT[1]) : "n"(ND_MIN) : "memory");
                else static_assert(nitL <= 2, "more butterfly sweeps than the wait statement names");
            } else {
                if constexpr (nitL == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cD), "+v"(cT[0]) : : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(cD), "+v"(cT[0]), "+v"(cT[1]) : : "memory");
            }
            D = big_twiddle_from(d, eD, cD);
        }
        LineId id;
        id.batch = cur.batch;
        id.o1 = cur.o1;
        id.o2 = cur.o2;
        id.i = cur.i0 + w;
        const bool lane_ok = w < cur.wvalid;
        const unsigned out_k = (unsigned)p.out_k;
#pragma unroll
        for (int it = 0; it < nitL; ++it) {
            const int g = rg + RG * it;
            if ((rowsL % RG == 0) || g < rowsL) {
                float2 x[RL];
#pragma unroll
                for (int q = 0; q < RL; ++q) x[q] = tile[DmaSlot<ROWS>::at(g * RL + q, w)];
                dft_pa<RL>(x);
                const int kb = kbase(g);
                float2 Tw = make_float2(1.f, 0.f);
                if constexpr (PTW) Tw = big_twiddle_from(d, eT[it], cT[it]);
                if (lane_ok) {
#pragma unroll
                    for (int q = 0; q < RL; ++q) {
                        const int k = kb + (L / RL) * q;
                        float2 y = x[dft_slot<RL>(q)];
                        if constexpr (PTW) {
                            y = cmul(y, Tw);
                            Tw = cmul(Tw, D);
                        }
                        store(id, k, cur.out_base, (unsigned)k * out_k + (unsigned)w, y);
                    }
                }
            }
        }
        RCFM_DMA_TRACE_POINT(6);
        ++trace_it;
        cur = nxt;
    }
}

}  // namespace fftk
RCFM_NS_CLOSE
}  // namespace rcfm
