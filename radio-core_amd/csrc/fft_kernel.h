// Device side of the multi-pass FFT (see fft_engine.h): one workgroup = one tile of
// 16 lines x L points.  Templated on load / store functors so element-wise stages of
// the radio path can be fused into a pass.
//
// LDS image: row r (= point index, later digit-reversed output slot) x 16 lanes of
// float2, pitch 16.  With lanes running along the 16 lines every ds_read_b64 /
// ds_write_b64 of a 16-lane group covers one 128-byte row: conflict-free by
// construction (MI355X_MICROARCH.md, LDS table).  Passes whose lines are contiguous in
// memory load with lanes along l instead; their rows are XOR-swizzled
// (lane ^ (row & 15)) so that the transposing store is conflict-free as well.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fft_engine.h"

namespace rcfm {
namespace fftk {

constexpr int kThreads = 256;
constexpr int W = kFftTileW;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i / +i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }

// ---- small forward DFTs, y[q'] = sum_q x[q] exp(-2 pi i q q' / R), in place --------

__device__ __forceinline__ void dft2(float2& a, float2& b) {
    const float2 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

__device__ __forceinline__ void dft3(float2& x0, float2& x1, float2& x2) {
    const float2 s = cadd(x1, x2);
    const float2 d = csub(x1, x2);
    const float2 t = make_float2(x0.x - 0.5f * s.x, x0.y - 0.5f * s.y);
    const float2 u = make_float2(0.86602540378443864676f * d.x, 0.86602540378443864676f * d.y);
    x0 = cadd(x0, s);
    x1 = cadd(t, mul_mi(u));
    x2 = cadd(t, mul_pi(u));
}

__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 a = cadd(x0, x2), b = csub(x0, x2);
    const float2 c = cadd(x1, x3), d = mul_mi(csub(x1, x3));
    x0 = cadd(a, c);
    x1 = cadd(b, d);
    x2 = csub(a, c);
    x3 = csub(b, d);
}

__device__ __forceinline__ void dft5(float2& x0, float2& x1, float2& x2, float2& x3, float2& x4) {
    constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 a = cadd(x1, x4), b = cadd(x2, x3), c = csub(x1, x4), d = csub(x2, x3);
    const float2 t1 = make_float2(x0.x + c1 * a.x + c2 * b.x, x0.y + c1 * a.y + c2 * b.y);
    const float2 t2 = make_float2(x0.x + c2 * a.x + c1 * b.x, x0.y + c2 * a.y + c1 * b.y);
    const float2 u1 = make_float2(s1 * c.x + s2 * d.x, s1 * c.y + s2 * d.y);
    const float2 u2 = make_float2(s2 * c.x - s1 * d.x, s2 * c.y - s1 * d.y);
    x0 = cadd(x0, cadd(a, b));
    x1 = cadd(t1, mul_mi(u1));
    x4 = cadd(t1, mul_pi(u1));
    x2 = cadd(t2, mul_mi(u2));
    x3 = cadd(t2, mul_pi(u2));
}

// Composite radices: R = R1 * R2, input q = q1 R2 + q2, output q' = k1 + R1 k2:
// R2 DFTs of length R1, constant twiddles W_R^(q2 k1), R1 DFTs of length R2.

__device__ __forceinline__ void dft6(float2* v) {
    // R1 = 3 (over q1, stride 2), R2 = 2
    dft3(v[0], v[2], v[4]);   // q2 = 0 -> t[k1][0]
    dft3(v[1], v[3], v[5]);   // q2 = 1 -> t[k1][1]
    // twiddles W_6^(k1) on the q2 = 1 column
    v[3] = cmul(v[3], make_float2(0.5f, -0.86602540378443864676f));
    v[5] = cmul(v[5], make_float2(-0.5f, -0.86602540378443864676f));
    // DFT2 over q2 for each k1: outputs y[k1], y[k1 + 3]
    dft2(v[0], v[1]);
    dft2(v[2], v[3]);
    dft2(v[4], v[5]);
    // now v[2 k1 + k2] = y[k1 + 3 k2]  -> reorder to natural
    const float2 y0 = v[0], y3 = v[1], y1 = v[2], y4 = v[3], y2 = v[4], y5 = v[5];
    v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3; v[4] = y4; v[5] = y5;
}

__device__ __forceinline__ void dft8(float2* v) {
    constexpr float h = 0.70710678118654752440f;
    // R1 = 4 (stride 2), R2 = 2
    dft4(v[0], v[2], v[4], v[6]);
    dft4(v[1], v[3], v[5], v[7]);
    v[3] = cmul(v[3], make_float2(h, -h));
    v[5] = mul_mi(v[5]);
    v[7] = cmul(v[7], make_float2(-h, -h));
    dft2(v[0], v[1]);
    dft2(v[2], v[3]);
    dft2(v[4], v[5]);
    dft2(v[6], v[7]);
    const float2 y0 = v[0], y4 = v[1], y1 = v[2], y5 = v[3], y2 = v[4], y6 = v[5], y3 = v[6], y7 = v[7];
    v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3; v[4] = y4; v[5] = y5; v[6] = y6; v[7] = y7;
}

__device__ __forceinline__ void dft10(float2* v) {
    // R1 = 5 (stride 2), R2 = 2; twiddles W_10^(k1), k1 = 1..4
    dft5(v[0], v[2], v[4], v[6], v[8]);
    dft5(v[1], v[3], v[5], v[7], v[9]);
    v[3] = cmul(v[3], make_float2(0.80901699437494742410f, -0.58778525229247312917f));
    v[5] = cmul(v[5], make_float2(0.30901699437494742410f, -0.95105651629515357212f));
    v[7] = cmul(v[7], make_float2(-0.30901699437494742410f, -0.95105651629515357212f));
    v[9] = cmul(v[9], make_float2(-0.80901699437494742410f, -0.58778525229247312917f));
    dft2(v[0], v[1]);
    dft2(v[2], v[3]);
    dft2(v[4], v[5]);
    dft2(v[6], v[7]);
    dft2(v[8], v[9]);
    const float2 y0 = v[0], y5 = v[1], y1 = v[2], y6 = v[3], y2 = v[4], y7 = v[5], y3 = v[6], y8 = v[7],
                 y4 = v[8], y9 = v[9];
    v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3; v[4] = y4;
    v[5] = y5; v[6] = y6; v[7] = y7; v[8] = y8; v[9] = y9;
}

// One DIF stage of radix R over the tile: block length mt, sub-length m = mt / R.
// Points base + q m (q < R) -> DFT_R -> times W_mt^(q' k') -> same slots.
template <int R>
__device__ __forceinline__ void dif_stage(float2* tile, const float2* tw, int L, int mt, int swz) {
    const int m = mt / R;
    const int step = L / mt;
    const int nb = (L / R) * W;
    for (int e = threadIdx.x; e < nb; e += kThreads) {
        const int w = e & (W - 1);
        const int b = e / W;
        const int g = b / m;
        const int kp = b - g * m;
        const int base = g * mt + kp;
        float2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int row = base + q * m;
            v[q] = tile[row * W + (w ^ (swz & row & (W - 1)))];
        }
        if constexpr (R == 2) dft2(v[0], v[1]);
        if constexpr (R == 3) dft3(v[0], v[1], v[2]);
        if constexpr (R == 4) dft4(v[0], v[1], v[2], v[3]);
        if constexpr (R == 5) dft5(v[0], v[1], v[2], v[3], v[4]);
        if constexpr (R == 6) dft6(v);
        if constexpr (R == 8) dft8(v);
        if constexpr (R == 10) dft10(v);
        const int tstep = kp * step;
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmul(v[q], tw[q * tstep]);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int row = base + q * m;
            tile[row * W + (w ^ (swz & row & (W - 1)))] = v[q];
        }
    }
}

// W_n^e = coarse[e >> fb] * cis(-2 pi (e & mask) / n); the fine angle is < 0.03 rad.
__device__ __forceinline__ float2 big_twiddle(const FftPassDev& d, unsigned e) {
    const float2 c = d.coarse[e >> d.fine_bits];
    const float th = d.fine_step * (float)(e & ((1u << d.fine_bits) - 1u));
    const float t2 = th * th;
    const float s = th * (1.f - t2 * (1.f / 6.f) * (1.f - t2 * (1.f / 20.f)));
    const float co = 1.f - t2 * 0.5f * (1.f - t2 * (1.f / 12.f) * (1.f - t2 * (1.f / 30.f)));
    return cmul(c, make_float2(co, -s));
}

// RCFM_FFT_GENERIC=1 forces the runtime-radix kernel (A/B testing).
inline bool getenv_generic_fft() {
    static const bool v = [] {
        const char* e = std::getenv("RCFM_FFT_GENERIC");
        return e && e[0] == '1';
    }();
    return v;
}

struct LineId {
    int batch;
    int64_t o1, o2, i;
};

// LoadOp:  float2 operator()(const LineId&, int l, int64_t addr)      addr = default input address
// StoreOp: void   operator()(const LineId&, int k, int64_t addr, float2 v)
template <class LoadOp, class StoreOp>
__global__ __launch_bounds__(kThreads) void k_fft_pass(FftPassDev d, LoadOp load, StoreOp store) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const FftPass& p = d.p;
    const int L = p.L;
    float2* tile = reinterpret_cast<float2*>(smem_raw);           // L x 16
    float2* tw = tile + L * W;                                    // L
    uint16_t* pos = reinterpret_cast<uint16_t*>(tw + L);          // L

    const int tid = threadIdx.x;
    const int64_t tiles_inner = (p.n_inner + W - 1) / W;
    int64_t t = blockIdx.x;
    const int64_t ti = t % tiles_inner;
    t /= tiles_inner;
    LineId id;
    id.batch = blockIdx.y;
    id.o2 = t % p.n_o2;
    id.o1 = t / p.n_o2;
    const int64_t i0 = ti * W;
    const int wvalid = (int)((p.n_inner - i0) < W ? (p.n_inner - i0) : W);
    const int64_t in_base = (int64_t)id.batch * d.in_batch + id.o1 * p.in_o1 + id.o2 * p.in_o2 + i0 * p.in_i;
    const int64_t out_base = (int64_t)id.batch * d.out_batch + id.o1 * p.out_o1 + id.o2 * p.out_o2 + i0 * p.out_i;
    const int swz = p.load_along_l ? (W - 1) : 0;

    for (int e = tid; e < L; e += kThreads) {
        tw[e] = d.stage_tw[e];
        pos[e] = d.pos[e];
    }
    if (p.load_along_l) {
        for (int e = tid; e < L * W; e += kThreads) {
            const int w = e / L, l = e - w * L;
            float2 v = make_float2(0.f, 0.f);
            if (w < wvalid) {
                id.i = i0 + w;
                v = load.post(id, l, load(id, l, in_base + (int64_t)w * p.in_i + l));
            }
            tile[l * W + (w ^ (l & (W - 1)))] = v;
        }
    } else {
        for (int e = tid; e < L * W; e += kThreads) {
            const int l = e / W, w = e & (W - 1);
            float2 v = make_float2(0.f, 0.f);
            if (w < wvalid) {
                id.i = i0 + w;
                v = load.post(id, l, load(id, l, in_base + (int64_t)l * p.in_l + (int64_t)w * p.in_i));
            }
            tile[l * W + w] = v;
        }
    }
    __syncthreads();

    int mt = L;
    for (int s = 0; s < p.nstages; ++s) {
        const int r = p.radix[s];
        switch (r) {
            case 2: dif_stage<2>(tile, tw, L, mt, swz); break;
            case 3: dif_stage<3>(tile, tw, L, mt, swz); break;
            case 4: dif_stage<4>(tile, tw, L, mt, swz); break;
            case 5: dif_stage<5>(tile, tw, L, mt, swz); break;
            case 6: dif_stage<6>(tile, tw, L, mt, swz); break;
            case 8: dif_stage<8>(tile, tw, L, mt, swz); break;
            default: dif_stage<10>(tile, tw, L, mt, swz); break;
        }
        mt /= r;
        __syncthreads();
    }

    const unsigned tw_base = (unsigned)(id.o1 * p.tw_o1 + id.o2 * p.tw_o2 + i0 * p.tw_i);
    for (int e = tid; e < L * W; e += kThreads) {
        const int k = e / W, w = e & (W - 1);
        if (w >= wvalid) continue;
        const int row = pos[k];
        float2 v = tile[row * W + (w ^ (swz & row & (W - 1)))];
        if (p.has_twiddle) v = cmul(v, big_twiddle(d, (tw_base + (unsigned)w * (unsigned)p.tw_i) * (unsigned)k));
        id.i = i0 + w;
        store(id, k, out_base + (int64_t)k * p.out_k + (int64_t)w * p.out_i, v);
    }
}

// ---- compile-time specialised pass ------------------------------------------------
//
// Same algorithm with L and the radices known at compile time: every index division
// has a constant divisor, all loops unroll, each thread issues all of its global loads
// before the first use (L/16 independent 8-byte loads per thread in flight), the first
// DFT stage of a strided pass runs straight from those registers and the last stage
// stores straight to memory, so a 3-stage pass makes 2 LDS round trips instead of 4.
// Thread layout: lane w = tid & 15 is the line inside the tile, rg = tid >> 4 picks the
// butterfly; the 16 lanes of a row always touch one 128-byte LDS row / global segment.

template <int R>
__device__ __forceinline__ void dft_r(float2* v) {
    if constexpr (R == 2) dft2(v[0], v[1]);
    if constexpr (R == 3) dft3(v[0], v[1], v[2]);
    if constexpr (R == 4) dft4(v[0], v[1], v[2], v[3]);
    if constexpr (R == 5) dft5(v[0], v[1], v[2], v[3], v[4]);
    if constexpr (R == 6) dft6(v);
    if constexpr (R == 8) dft8(v);
    if constexpr (R == 10) dft10(v);
}

__device__ __forceinline__ int lds_slot(int row, int w, int swz) { return row * W + (w ^ (swz & row & (W - 1))); }

// Middle stage: LDS -> LDS.  MT = block length entering the stage.
template <int L, int R, int MT>
__device__ __forceinline__ void stage_lds(float2* tile, const float2* tw, int w, int rg, int swz) {
    constexpr int m = MT / R, step = L / MT, rows = L / R, nit = (rows + 15) / 16;
#pragma unroll
    for (int it = 0; it < nit; ++it) {
        const int b = rg + 16 * it;
        if ((rows % 16 == 0) || b < rows) {
            const int g = b / m, kp = b - g * m;
            const int base = g * MT + kp;
            float2 v[R];
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = tile[lds_slot(base + q * m, w, swz)];
            dft_r<R>(v);
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul(v[q], tw[q * kp * step]);
#pragma unroll
            for (int q = 0; q < R; ++q) tile[lds_slot(base + q * m, w, swz)] = v[q];
        }
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (global
// loads and stores share that counter on gfx9), which would wait for the prefetched tile at
// every stage boundary; LDS hand-offs only need lgkmcnt(0) + s_barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Per-tile addressing, decoded once per tile per thread.
struct TileCtx {
    LineId id;
    int64_t i0, in_base, out_base;
    int wvalid;
};

__device__ __forceinline__ void decode_tile(const FftPassDev& d, int64_t tile, int64_t tiles_inner,
                                            int64_t tiles_per_batch, TileCtx& c) {
    const FftPass& p = d.p;
    const int64_t batch = tile / tiles_per_batch;
    int64_t t = tile - batch * tiles_per_batch;
    const int64_t ti = t % tiles_inner;
    t /= tiles_inner;
    c.id.batch = (int)batch;
    c.id.o2 = t % p.n_o2;
    c.id.o1 = t / p.n_o2;
    c.id.i = 0;
    c.i0 = ti * W;
    c.wvalid = (int)((p.n_inner - c.i0) < W ? (p.n_inner - c.i0) : W);
    c.in_base = batch * d.in_batch + c.id.o1 * p.in_o1 + c.id.o2 * p.in_o2 + c.i0 * p.in_i;
    c.out_base = batch * d.out_batch + c.id.o1 * p.out_o1 + c.id.o2 * p.out_o2 + c.i0 * p.out_i;
}

// Persistent workgroups: each loops over tiles (grid-stride) and issues the global loads of
// its NEXT tile right after the first DFT stage has consumed the registers of the current
// one, so HBM latency hides behind the remaining stages instead of being paid per tile
// (with 65 KiB of LDS only two workgroups fit a CU: there is no other latency hiding).
// PERSIST = false: one tile per workgroup, no loop (short lengths: 8 workgroups per CU hide
// the latency by themselves).
template <int L, int R0, int R1, int R2, int R3, bool ROWS, bool PERSIST, class LoadOp, class StoreOp>
__global__ __launch_bounds__(kThreads) void k_fft_pass_t(FftPassDev d, LoadOp load, StoreOp store,
                                                         int64_t total_tiles, int64_t tiles_per_batch) {
    constexpr int S = (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1);
    static_assert(S >= 2 && R0 * R1 * R2 * R3 == L, "bad radix list");
    constexpr int RL = (S == 2) ? R1 : (S == 3) ? R2 : R3;   // last radix
    constexpr int m0 = L / R0;
    constexpr int nit0 = (m0 + 15) / 16;                       // first-stage butterflies per thread
    constexpr int nld = ROWS ? (L * W + kThreads - 1) / kThreads : nit0 * R0;
    constexpr int swz = ROWS ? (W - 1) : 0;
    __shared__ __attribute__((aligned(16))) float2 tile[L * W];
    __shared__ __attribute__((aligned(16))) float2 tw[L];
    const FftPass& p = d.p;
    const int tid = threadIdx.x;
    int w = tid & (W - 1), rg = tid >> 4;
    unsigned in_l = (unsigned)p.in_l, in_i = (unsigned)p.in_i, out_k = (unsigned)p.out_k;
    const int64_t tiles_inner = (p.n_inner + W - 1) / W;

    // XCD-aware order: workgroup b runs on XCD b % 8 (observed, speed only); give each XCD a
    // contiguous run of tiles so neighbouring 128-byte segments meet in one L2.
    const int64_t G = gridDim.x;
    const int64_t vb = (G % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;

    float2 v[nld];
    // Loads are issued unconditionally from clamped (always valid) addresses and masked
    // afterwards: a load under `if` costs a branch plus an immediate s_waitcnt vmcnt(0),
    // i.e. one serialized HBM round trip per element.
    auto issue_loads = [&](TileCtx& c) {
        if constexpr (ROWS) {
#pragma unroll
            for (int it = 0; it < nld; ++it) {
                int e = tid + kThreads * it;
                if ((L * W) % kThreads != 0) e = e < L * W ? e : 0;
                const int wl = e / L, l = e - wl * L;
                const int wc = wl < c.wvalid ? wl : 0;
                c.id.i = c.i0 + wc;
                v[it] = load(c.id, l, c.in_base + (int64_t)((unsigned)wc * in_i + (unsigned)l));
            }
        } else {
            const int wc = w < c.wvalid ? w : 0;
            c.id.i = c.i0 + wc;
#pragma unroll
            for (int it = 0; it < nit0; ++it) {
                int b = rg + 16 * it;
                if (m0 % 16 != 0) b = b < m0 ? b : 0;
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    const int l = b + q * m0;
                    v[it * R0 + q] = load(c.id, l, c.in_base + (int64_t)((unsigned)l * in_l + (unsigned)wc));
                }
            }
        }
    };
    // At consume time: the functor's arithmetic (LoadOp::post -- kept out of issue_loads so the
    // prefetch has no consumer until here), then zero the lanes of a partial tile (their
    // clamped loads fetched line 0).
    auto mask_loads = [&](TileCtx& c) {
        if constexpr (ROWS) {
#pragma unroll
            for (int it = 0; it < nld; ++it) {
                const int e = tid + kThreads * it;
                const int wl = e / L, l = e - wl * L;
                c.id.i = c.i0 + wl;
                v[it] = load.post(c.id, l, v[it]);
                if (wl >= c.wvalid) v[it] = make_float2(0.f, 0.f);
            }
        } else {
            c.id.i = c.i0 + w;
#pragma unroll
            for (int it = 0; it < nit0; ++it) {
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    v[it * R0 + q] = load.post(c.id, rg + 16 * it + q * m0, v[it * R0 + q]);
                    if (w >= c.wvalid) v[it * R0 + q] = make_float2(0.f, 0.f);
                }
            }
        }
    };

    TileCtx cur;
    int64_t tile_id = vb;
    if (tile_id < total_tiles) {
        decode_tile(d, tile_id, tiles_inner, tiles_per_batch, cur);
        issue_loads(cur);
    }
    for (int e = tid; e < L; e += kThreads) tw[e] = d.stage_tw[e];
    __syncthreads();

    while (tile_id < total_tiles) {
        if constexpr (PERSIST) {
            // Keep per-thread offsets (l * in_l + w, LDS slots, k * out_k) out of registers across
            // iterations: hoisted, they cost ~150 VGPRs and halve the occupancy.
            asm volatile("" : "+v"(w), "+v"(rg));
            asm volatile("" : "+s"(in_l), "+s"(in_i), "+s"(out_k));
        }
        // ---- first stage -------------------------------------------------------
        mask_loads(cur);
        if constexpr (ROWS) {
#pragma unroll
            for (int it = 0; it < nld; ++it) {
                const int e = tid + kThreads * it;
                const int wl = e / L, l = e - wl * L;
                if ((L * W) % kThreads == 0 || e < L * W) tile[lds_slot(l, wl, swz)] = v[it];
            }
            lds_barrier();
        } else {
#pragma unroll
            for (int it = 0; it < nit0; ++it) {
                const int b = rg + 16 * it;
                if ((m0 % 16 == 0) || b < m0) {
                    dft_r<R0>(&v[it * R0]);
#pragma unroll
                    for (int q = 1; q < R0; ++q) v[it * R0 + q] = cmul(v[it * R0 + q], tw[q * b]);
#pragma unroll
                    for (int q = 0; q < R0; ++q) tile[lds_slot(b + q * m0, w, swz)] = v[it * R0 + q];
                }
            }
            lds_barrier();
        }
        // ---- prefetch the next tile (registers are free again) -------------------
        const int64_t next_id = tile_id + G;
        TileCtx nxt = cur;
        if (PERSIST && next_id < total_tiles) {
            decode_tile(d, next_id, tiles_inner, tiles_per_batch, nxt);
            issue_loads(nxt);
        }
        if constexpr (ROWS) {
            stage_lds<L, R0, L>(tile, tw, w, rg, swz);
            lds_barrier();
        }
        if constexpr (S >= 3) {
            stage_lds<L, R1, L / R0>(tile, tw, w, rg, swz);
            lds_barrier();
        }
        if constexpr (S >= 4) {
            stage_lds<L, R2, L / (R0 * R1)>(tile, tw, w, rg, swz);
            lds_barrier();
        }

        // ---- last stage (block length RL, no stage twiddle): LDS -> registers -> memory.
        // Block g holds outputs k = kb(g) + (L / RL) q': g's digits are q_1 .. q_{S-1} with
        // weights L/(r_1 RL), L/(r_1 r_2 RL), ...; kb = q_1 + r_1 q_2 + r_1 r_2 q_3.
        constexpr int rowsL = L / RL, nitL = (rowsL + 15) / 16;
        const unsigned f = (unsigned)(cur.id.o1 * p.tw_o1 + cur.id.o2 * p.tw_o2 + (cur.i0 + w) * p.tw_i);
        cur.id.i = cur.i0 + w;
        const bool lane_ok = w < cur.wvalid;
#pragma unroll
        for (int it = 0; it < nitL; ++it) {
            const int g = rg + 16 * it;
            if ((rowsL % 16 == 0) || g < rowsL) {
                float2 o[RL];
#pragma unroll
                for (int q = 0; q < RL; ++q) o[q] = tile[lds_slot(g * RL + q, w, swz)];
                dft_r<RL>(o);
                int kb;
                if constexpr (S == 2) {
                    kb = g;
                } else if constexpr (S == 3) {
                    constexpr int w1 = L / (R0 * RL);
                    const int q1 = g / w1, q2 = g - q1 * w1;
                    kb = q1 + R0 * q2;
                } else {
                    constexpr int w1 = L / (R0 * RL), w2 = L / (R0 * R1 * RL);
                    const int q1 = g / w1, r1 = g - q1 * w1;
                    const int q2 = r1 / w2, q3 = r1 - q2 * w2;
                    kb = q1 + R0 * (q2 + R1 * q3);
                }
                if (lane_ok) {
#pragma unroll
                    for (int q = 0; q < RL; ++q) {
                        const int k = kb + (L / RL) * q;
                        float2 x = o[q];
                        if constexpr (!ROWS) x = cmul(x, big_twiddle(d, f * (unsigned)k));   // ROWS = last pass
                        store(cur.id, k, cur.out_base + (int64_t)((unsigned)k * out_k + (unsigned)w), x);
                    }
                }
            }
        }
        if constexpr (!PERSIST) break;
        lds_barrier();   // all reads of this tile are done before the next one lands in LDS
        cur = nxt;
        tile_id = next_id;
    }
}

// ---- plain functors ------------------------------------------------------------

// LoadOp contract: operator() only fetches (no arithmetic on the result: the specialised
// kernel issues it one tile ahead); post() runs when the tile is consumed.
struct LoadPlain {
    const float2* in;
    int swap;   // 1: exchange re/im (inverse transform by the swap identity)
    __device__ __forceinline__ float2 operator()(const LineId&, int, int64_t a) const { return in[a]; }
    __device__ __forceinline__ float2 post(const LineId&, int, float2 v) const {
        return swap ? make_float2(v.y, v.x) : v;
    }
};

struct StorePlain {
    float2* out;
    int swap;
    float scale;
    __device__ __forceinline__ void operator()(const LineId&, int, int64_t a, float2 v) const {
        out[a] = swap ? make_float2(v.y * scale, v.x * scale) : make_float2(v.x * scale, v.y * scale);
    }
};

// Lengths with a compile-time specialisation (radices listed first stage first; they must
// match choose_radices() in fft_engine.hip).  Other lengths run the generic kernel.
#define RCFM_FFT_FAST_LENGTHS(X) \
    X(80, 10, 8, 1, 1)           \
    X(100, 10, 10, 1, 1)         \
    X(120, 10, 6, 2, 1)          \
    X(125, 5, 5, 5, 1)           \
    X(128, 8, 8, 2, 1)           \
    X(160, 10, 8, 2, 1)          \
    X(200, 10, 10, 2, 1)         \
    X(240, 10, 8, 3, 1)          \
    X(250, 10, 5, 5, 1)          \
    X(256, 8, 8, 4, 1)           \
    X(320, 10, 8, 4, 1)          \
    X(400, 10, 10, 4, 1)         \
    X(480, 10, 8, 6, 1)          \
    X(500, 10, 10, 5, 1)         \
    X(512, 8, 8, 8, 1)

inline bool getenv_flag(const char* name) {
    const char* e = std::getenv(name);
    return e && e[0] == '1';
}

template <int LEN, int A, int B, int C, int D, class LoadOp, class StoreOp>
inline void launch_variant(bool rows, bool persist, dim3 grid, hipStream_t s, const FftPassDev& d,
                           const LoadOp& ld, const StoreOp& st, int64_t total, int64_t tpb) {
    // short lengths never loop; long ones only exist in the persistent form
    constexpr bool kLong = LEN >= 200;
    if (rows) {
        if (kLong && persist)
            hipLaunchKernelGGL((k_fft_pass_t<LEN, A, B, C, D, true, kLong, LoadOp, StoreOp>), grid, dim3(kThreads),
                               0, s, d, ld, st, total, tpb);
        else
            hipLaunchKernelGGL((k_fft_pass_t<LEN, A, B, C, D, true, false, LoadOp, StoreOp>), grid,
                               dim3(kThreads), 0, s, d, ld, st, total, tpb);
    } else {
        if (kLong && persist)
            hipLaunchKernelGGL((k_fft_pass_t<LEN, A, B, C, D, false, kLong, LoadOp, StoreOp>), grid,
                               dim3(kThreads), 0, s, d, ld, st, total, tpb);
        else
            hipLaunchKernelGGL((k_fft_pass_t<LEN, A, B, C, D, false, false, LoadOp, StoreOp>), grid,
                               dim3(kThreads), 0, s, d, ld, st, total, tpb);
    }
}

template <class LoadOp, class StoreOp>
inline void launch_fft_pass(const FftPassDev& d, int batch, const LoadOp& ld, const StoreOp& st, hipStream_t s) {
    bool done = false;
    const bool strided_ok = d.p.load_along_l || (d.p.in_i == 1);
    const bool store_ok = (d.p.out_i == 1) && ((d.p.has_twiddle != 0) == (d.p.load_along_l == 0));
    if (strided_ok && store_ok && !getenv_generic_fft()) {
        const int64_t tiles_per_batch = d.p.n_o1 * d.p.n_o2 * ((d.p.n_inner + W - 1) / W);
        const int64_t total = tiles_per_batch * batch;
        // persistent grid: as many workgroups as fit the chip at once (LDS-limited), multiple of 8
        const size_t lds = (size_t)d.p.L * (W + 1) * sizeof(float2);
        int per_cu = (int)(160 * 1024 / lds);
        per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
        int64_t g = (int64_t)FftEngine::compute_units() * per_cu;
        if (g > total) g = total;
        if (g >= 8) g -= g % 8;
        const bool persist = d.p.L >= 200 && !getenv_flag("RCFM_FFT_NOPERSIST");
        const dim3 grid((unsigned)(persist ? g : total), 1, 1);
        switch (d.p.L) {
#define RCFM_CASE(LEN, A, B, C, D)                                                                              \
    case LEN:                                                                                                   \
        launch_variant<LEN, A, B, C, D>(d.p.load_along_l != 0, persist, grid, s, d, ld, st, total,               \
                                        tiles_per_batch);                                                       \
        done = true;                                                                                            \
        break;
            RCFM_FFT_FAST_LENGTHS(RCFM_CASE)
#undef RCFM_CASE
            default: break;
        }
    }
    if (!done)
        hipLaunchKernelGGL((k_fft_pass<LoadOp, StoreOp>), FftEngine::grid(d.p, batch), dim3(kThreads),
                           FftEngine::lds_bytes(d.p.L), s, d, ld, st);
    RC_HIP(hipGetLastError());
}

}  // namespace fftk
}  // namespace rcfm
