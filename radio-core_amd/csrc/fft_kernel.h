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
                v = load(id, l, in_base + (int64_t)w * p.in_i + l);
            }
            tile[l * W + (w ^ (l & (W - 1)))] = v;
        }
    } else {
        for (int e = tid; e < L * W; e += kThreads) {
            const int l = e / W, w = e & (W - 1);
            float2 v = make_float2(0.f, 0.f);
            if (w < wvalid) {
                id.i = i0 + w;
                v = load(id, l, in_base + (int64_t)l * p.in_l + (int64_t)w * p.in_i);
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

// ---- plain functors ------------------------------------------------------------

struct LoadPlain {
    const float2* in;
    int swap;   // 1: exchange re/im (inverse transform by the swap identity)
    __device__ __forceinline__ float2 operator()(const LineId&, int, int64_t a) const {
        const float2 v = in[a];
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

template <class LoadOp, class StoreOp>
inline void launch_fft_pass(const FftPassDev& d, int batch, const LoadOp& ld, const StoreOp& st, hipStream_t s) {
    hipLaunchKernelGGL((k_fft_pass<LoadOp, StoreOp>), FftEngine::grid(d.p, batch), dim3(kThreads),
                       FftEngine::lds_bytes(d.p.L), s, d, ld, st);
    RC_HIP(hipGetLastError());
}

}  // namespace fftk
}  // namespace rcfm
