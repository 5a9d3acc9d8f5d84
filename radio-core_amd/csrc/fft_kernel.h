// Device side of the multi-pass FFT (see fft_engine.h): one workgroup = one tile of
// 16 lines x L points.  Templated on load / store functors so element-wise stages of
// the radio path can be fused into a pass.
//
// LDS image: row r (= point index, later digit-reversed output slot) x 16 lanes of
// float2, pitch 16.  With lanes running along the 16 lines every ds_read_b64 /
// ds_write_b64 of a 16-lane group covers one 128-byte row: conflict-free by
// construction (MI355X_MICROARCH.md, LDS table).  Passes whose lines are contiguous in
// memory load with lanes along l instead; their rows have a pitch of 17 points
// (kRowsPitch) so that the transposing store is conflict-free as well.
//
// Kernels in this file:
//   k_fft_pass          runtime-radix fallback for lengths without a specialisation
//   k_fft_tile          one pass, compile-time length and radices (the workhorse)
//   k_fft_tile2         last pass of one plan + point-wise stage + first pass of the swapped plan
//   k_fft_tile2_pair    the same for a transform that carried two real signals (two second transforms)
//   k_fft_tile2_decim   last pass of a long plan + spectral decimation + first pass of a short one
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "fft_engine.h"
#include "tile_ns.h"

namespace rcfm {
RCFM_NS_OPEN
namespace fftk {

constexpr int W = RCFM_TILE_W;                 // lines per tile (tile_ns.h)
constexpr int kLogW = W == 16 ? 4 : W == 8 ? 3 : 2;
static_assert((1 << kLogW) == W, "tile width: 4, 8 or 16 lines");
constexpr int kThreads = 256 * W / 16;

// Complex product: two packed instructions whose op_sel / neg modifiers pick the
// halves (a.x b, then a.y (-b.y, b.x) + ...) -- the compiler's own packing builds those operand
// pairs with v_mov's (19 % of the tile kernel's VALU instructions).
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f A, B, D;
    A.x = a.x; A.y = a.y; B.x = b.x; B.y = b.y;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(D) : "v"(A), "v"(B));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(D) : "v"(A), "v"(B));
    return make_float2(D.x, D.y);
}
// Complex add / subtract: always ONE v_pk_add_f32 on aligned register pairs (left to the compiler, some were packed and
// paid for their operand pairs with v_mov: cfg4 -0.5 %, the tuner's last pass lost 69 of 819 scalar VALU instructions).
__device__ __forceinline__ float2 cadd(float2 a, float2 b) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f A, B, D;
    A.x = a.x; A.y = a.y; B.x = b.x; B.y = b.y;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B));
    return make_float2(D.x, D.y);
}
__device__ __forceinline__ float2 csub(float2 a, float2 b) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f A, B, D;
    A.x = a.x; A.y = a.y; B.x = b.x; B.y = b.y;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(D) : "v"(A), "v"(B));
    return make_float2(D.x, D.y);
}
// multiply by -i / +i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }
// t + (-i) u and t + (+i) u in one packed add (half-swap and sign via op_sel / neg_hi / neg_lo).
__device__ __forceinline__ float2 cadd_mi(float2 t, float2 u) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f T, U, D;
    T.x = t.x; T.y = t.y; U.x = u.x; U.y = u.y;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(D) : "v"(T), "v"(U));
    return make_float2(D.x, D.y);
}
__device__ __forceinline__ float2 cadd_pi(float2 t, float2 u) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f T, U, D;
    T.x = t.x; T.y = t.y; U.x = u.x; U.y = u.y;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(D) : "v"(T), "v"(U));
    return make_float2(D.x, D.y);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (global loads and
// stores share that counter on gfx9): in the multi-transform tile kernels that would wait, at every
// stage boundary, for the point-wise stage's prefetched inputs and for the stores of the transform just
// finished.  LDS hand-offs only need lgkmcnt(0) + s_barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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
    x1 = cadd_mi(t, u);
    x2 = cadd_pi(t, u);
}

__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 a = cadd(x0, x2), b = csub(x0, x2);
    const float2 c = cadd(x1, x3), e = csub(x1, x3);
    x0 = cadd(a, c);
    x1 = cadd_mi(b, e);
    x2 = csub(a, c);
    x3 = cadd_pi(b, e);
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
    x1 = cadd_mi(t1, u1);
    x4 = cadd_pi(t1, u1);
    x2 = cadd_mi(t2, u2);
    x3 = cadd_pi(t2, u2);
}

// Radix 7 (round 6: audio rates with a factor 7 -- 44 100 = 2^2 3^2 5^2 7^2 = 210 x 210 -- stay inside the engine instead
// of sending every demodulator transform to rocFFT): y[k] = x0 + sum_j c(jk) a_j - i sum_j s(jk) b_j with a_j = x_j + x_{7-j},
// b_j = x_j - x_{7-j}, c / s = cos / sin(2 pi m / 7).
__device__ __forceinline__ void dft7(float2& x0, float2& x1, float2& x2, float2& x3, float2& x4, float2& x5, float2& x6) {
    constexpr float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
    constexpr float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
    const float2 a1 = cadd(x1, x6), a2 = cadd(x2, x5), a3 = cadd(x3, x4);
    const float2 b1 = csub(x1, x6), b2 = csub(x2, x5), b3 = csub(x3, x4);
    const float2 t1 = make_float2(x0.x + c1 * a1.x + c2 * a2.x + c3 * a3.x, x0.y + c1 * a1.y + c2 * a2.y + c3 * a3.y);
    const float2 t2 = make_float2(x0.x + c2 * a1.x + c3 * a2.x + c1 * a3.x, x0.y + c2 * a1.y + c3 * a2.y + c1 * a3.y);
    const float2 t3 = make_float2(x0.x + c3 * a1.x + c1 * a2.x + c2 * a3.x, x0.y + c3 * a1.y + c1 * a2.y + c2 * a3.y);
    const float2 u1 = make_float2(s1 * b1.x + s2 * b2.x + s3 * b3.x, s1 * b1.y + s2 * b2.y + s3 * b3.y);
    const float2 u2 = make_float2(s2 * b1.x - s3 * b2.x - s1 * b3.x, s2 * b1.y - s3 * b2.y - s1 * b3.y);
    const float2 u3 = make_float2(s3 * b1.x - s1 * b2.x + s2 * b3.x, s3 * b1.y - s1 * b2.y + s2 * b3.y);
    x0 = cadd(x0, cadd(a1, cadd(a2, a3)));
    x1 = cadd_mi(t1, u1);
    x6 = cadd_pi(t1, u1);
    x2 = cadd_mi(t2, u2);
    x5 = cadd_pi(t2, u2);
    x3 = cadd_mi(t3, u3);
    x4 = cadd_pi(t3, u3);
}

// Composite radices (12, 15, 16, 20, 24, 25, 32, ...): a natural-order in-register DFT built from the
// radix-2/3/4/5 kernels, R = Ra Rb: Rb transforms of length Ra over the strided sub-sequences, constant
// twiddles W_R^(n2 k1) (folded at compile time), Ra transforms of length Rb (recursively).  Two LDS stages
// of radix ~22 replace three of radix ~8 for the 480..640-point tiles: a third fewer LDS round trips and
// one barrier phase less per transform.
template <int R>
__device__ __forceinline__ void dft_nat(float2* v) {
    if constexpr (R == 2) {
        dft2(v[0], v[1]);
    } else if constexpr (R == 3) {
        dft3(v[0], v[1], v[2]);
    } else if constexpr (R == 4) {
        dft4(v[0], v[1], v[2], v[3]);
    } else if constexpr (R == 5) {
        dft5(v[0], v[1], v[2], v[3], v[4]);
    } else if constexpr (R == 7) {
        dft7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
    } else {
        constexpr int Ra = (R % 7 == 0) ? 7 : (R % 5 == 0) ? 5 : (R % 4 == 0) ? 4 : (R % 3 == 0) ? 3 : 2;
        constexpr int Rb = R / Ra;
        static_assert(Ra * Rb == R && Rb > 1, "radix must be 2-3-5-7 smooth");
        float2 y[R];
#pragma unroll
        for (int n2 = 0; n2 < Rb; ++n2) {
            float2 t[Ra];
#pragma unroll
            for (int n1 = 0; n1 < Ra; ++n1) t[n1] = v[Rb * n1 + n2];
            dft_nat<Ra>(t);
#pragma unroll
            for (int k1 = 0; k1 < Ra; ++k1) {
                const int m = (n2 * k1) % R;
                if (m == 0) {
                    y[k1 * Rb + n2] = t[k1];
                } else {
                    const double a = -6.28318530717958647692 * (double)m / (double)R;
                    y[k1 * Rb + n2] = cmul(t[k1], make_float2((float)__builtin_cos(a), (float)__builtin_sin(a)));
                }
            }
        }
#pragma unroll
        for (int k1 = 0; k1 < Ra; ++k1) {
            dft_nat<Rb>(&y[k1 * Rb]);
#pragma unroll
            for (int k2 = 0; k2 < Rb; ++k2) v[k1 + Ra * k2] = y[k1 * Rb + k2];
        }
    }
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
        if constexpr (R == 7) dft7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
        if constexpr (R == 8) dft8(v);
        if constexpr (R == 10) dft10(v);
        if constexpr (R > 10) dft_nat<R>(v);
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

// A LoadOp may need two input values per point (kFetches = 2: fetch + fetch2, post(id, l, a, b)), or
// two and a real coefficient (kFetches = 3: + float fetch3, post(id, l, a, b, c)).
template <class T, class = void>
struct fetch_count { static constexpr int value = 1; };
template <class T>
struct fetch_count<T, std::void_t<decltype(T::kFetches)>> { static constexpr int value = T::kFetches; };

// A LoadOp of a strided pass may carry workgroup-uniform state (kCtx: Ctx prepare(id) is evaluated
// once with the tile's loads, post(id, l, v, ctx) receives it).
template <class T, class = void>
struct has_ctx : std::false_type {};
template <class T>
struct has_ctx<T, std::void_t<decltype(T::kCtx)>> : std::true_type {};

template <class LoadOp>
__device__ __forceinline__ float2 load_now(const LoadOp& load, const LineId& id, int l, int64_t base, unsigned off) {
    if constexpr (fetch_count<LoadOp>::value == 3)
        return load.post(id, l, load.fetch(id, l, base, off), load.fetch2(id, l, base, off),
                         load.fetch3(id, l, base, off));
    else if constexpr (fetch_count<LoadOp>::value == 2)
        return load.post(id, l, load.fetch(id, l, base, off), load.fetch2(id, l, base, off));
    else if constexpr (has_ctx<LoadOp>::value)
        return load.post(id, l, load.fetch(id, l, base, off), load.prepare(id));
    else
        return load.post(id, l, load.fetch(id, l, base, off));
}

// A LoadOp of a strided pass whose input is zero above the middle of the signal (one-sided spectra:
// point k = l * stride + i with n = L * stride, zero for k > n/2) declares kHalfZones = true.  The
// kernel then tells fetch / fetch2 / post which zone the point's l lies in -- 0: l < L/2 for the
// whole butterfly row, 1: mixed -- and neither loads nor post-processes zone 2 (l > L/2: zero).
template <class T, class = void>
struct has_half_zones : std::false_type {};
template <class T>
struct has_half_zones<T, std::void_t<decltype(T::kHalfZones)>> : std::bool_constant<T::kHalfZones> {};

// A StoreOp may need one auxiliary input value per output point (kAux: fetch_aux(id, k, base, off)
// is issued with the tile's loads, operator() receives the value as a sixth argument).
template <class T, class = void>
struct has_aux : std::false_type {};
template <class T>
struct has_aux<T, std::void_t<decltype(T::kAux)>> : std::true_type {};

template <class StoreOp>
__device__ __forceinline__ void store_now(const StoreOp& store, const LineId& id, int k, int64_t base, unsigned off,
                                          float2 v) {
    if constexpr (has_aux<StoreOp>::value) store(id, k, base, off, v, store.fetch_aux(id, k, base, off));
    else store(id, k, base, off, v);
}

// Runtime-radix fallback for lengths without a specialisation (same functor contracts).
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
    const int64_t in_base = (int64_t)id.batch * d.in_batch + id.o1 * p.in_o1 + id.o2 * p.in_o2 +
                            (p.in_t ? (i0 >> 4) * p.in_t + (i0 & 15) : i0 * p.in_i);
    const int64_t out_base = (int64_t)id.batch * d.out_batch + id.o1 * p.out_o1 + id.o2 * p.out_o2 +
                             (p.out_t ? (i0 >> 4) * p.out_t + (i0 & 15) : i0 * p.out_i);
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
                v = load_now(load, id, l, in_base, (unsigned)(w * p.in_i + l));
            }
            tile[l * W + (w ^ (l & (W - 1)))] = v;
        }
    } else {
        for (int e = tid; e < L * W; e += kThreads) {
            const int l = e / W, w = e & (W - 1);
            float2 v = make_float2(0.f, 0.f);
            if (w < wvalid) {
                id.i = i0 + w;
                v = load_now(load, id, l, in_base, (unsigned)(l * p.in_l + w * p.in_i));
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
            case 7: dif_stage<7>(tile, tw, L, mt, swz); break;
            case 8: dif_stage<8>(tile, tw, L, mt, swz); break;
            case 12: dif_stage<12>(tile, tw, L, mt, swz); break;
            case 15: dif_stage<15>(tile, tw, L, mt, swz); break;
            case 16: dif_stage<16>(tile, tw, L, mt, swz); break;
            case 20: dif_stage<20>(tile, tw, L, mt, swz); break;
            case 24: dif_stage<24>(tile, tw, L, mt, swz); break;
            case 25: dif_stage<25>(tile, tw, L, mt, swz); break;
            case 32: dif_stage<32>(tile, tw, L, mt, swz); break;
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
        store_now(store, id, k, out_base, (unsigned)(k * p.out_k + w * p.out_i), v);
    }
}

// ---- compile-time specialised pass ------------------------------------------------
//
// Same algorithm with L and the radices known at compile time: every index division has
// a constant divisor, all loops unroll, each thread issues all of its global loads back to
// back and unconditionally (L/16 independent 8-byte loads per thread in flight), the
// first DFT stage of a strided pass runs straight from those registers and the last stage
// stores straight to memory, so a 3-stage pass makes 2 LDS round trips instead of 4.
// Thread layout: lane w = tid & 15 is the line inside the tile, rg = tid >> 4 picks the
// butterfly; the 16 lanes of a row always touch one 128-byte LDS row / global segment.
// Grid: x = tile along the inner line index, y = outer line index, z = signal in the batch.

// XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin by linear id (x fastest), and
// each XCD has its own L2.  Neighbouring tiles of a row share 128-byte lines whenever a functor reads
// 64-byte or unaligned segments (real inputs, the tuner's rolled spectrum); mapping x -> tile so that
// each XCD owns a contiguous eighth of the row lets the second reader hit the first one's L2 line.
// Rows of at most this many tiles go to the XCDs as WHOLE rows instead: the blocks of eight consecutive signals form
// one group, and XCD k takes every tile of the group's k-th signal (an eighth of a short row is one or two tiles --
// with 8 tiles per row, cfg5's B = 12 500, every neighbour sat on another XCD and each shared line came from HBM twice).
constexpr unsigned kXcdGroupMax = 15;
struct BlockPos {
    unsigned tile, batch;
};
// A workgroup's place in the launch (blockIdx / gridDim).
struct VBlock {
    unsigned x, y, z, gx, gy, gz;
};
__device__ __forceinline__ VBlock vblock_hw() {
    return VBlock{blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z};
}
__device__ __forceinline__ BlockPos block_pos(const VBlock& vb);
__device__ __forceinline__ BlockPos block_pos() { return block_pos(vblock_hw()); }
__device__ __forceinline__ BlockPos block_pos(const VBlock& vb) {
    const unsigned gx = vb.gx, x = vb.x, z = vb.z;
    if (gx <= kXcdGroupMax) {
        // linear workgroup id = x + gx (y + gy z); with gy = 1 and z0 = z & ~7 its low three bits are those of
        // q = x + gx (z & 7), the position inside the group: XCD (q & 7) takes signal z0 + (q & 7), tile q >> 3
        if (vb.gy == 1 && (z | 7u) < vb.gz) {
            const unsigned q = x + gx * (z & 7u);
            return BlockPos{q >> 3, (z & ~7u) + (q & 7u)};
        }
        return BlockPos{x, z};
    }
    return BlockPos{(gx & 7u) ? x : (x & 7u) * (gx >> 3) + (x >> 3), z};
}

// Small DFTs that leave output q' in slot perm<R>(q') (no register shuffling afterwards).
template <int R>
__device__ __forceinline__ constexpr int dft_slot(int q) {
    if (R == 6) return (q % 3) * 2 + q / 3;     // dft6 leaves y[k1 + 3 k2] in v[2 k1 + k2]
    if (R == 8) return (q % 4) * 2 + q / 4;
    if (R == 10) return (q % 5) * 2 + q / 5;
    return q;
}

__device__ __forceinline__ void dft6p(float2* v) {
    dft3(v[0], v[2], v[4]);
    dft3(v[1], v[3], v[5]);
    v[3] = cmul(v[3], make_float2(0.5f, -0.86602540378443864676f));
    v[5] = cmul(v[5], make_float2(-0.5f, -0.86602540378443864676f));
    dft2(v[0], v[1]);
    dft2(v[2], v[3]);
    dft2(v[4], v[5]);
}

__device__ __forceinline__ void dft8p(float2* v) {
    constexpr float h = 0.70710678118654752440f;
    dft4(v[0], v[2], v[4], v[6]);
    dft4(v[1], v[3], v[5], v[7]);
    v[3] = cmul(v[3], make_float2(h, -h));
    v[5] = mul_mi(v[5]);
    v[7] = cmul(v[7], make_float2(-h, -h));
    dft2(v[0], v[1]);
    dft2(v[2], v[3]);
    dft2(v[4], v[5]);
    dft2(v[6], v[7]);
}

__device__ __forceinline__ void dft10p(float2* v) {
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
}

template <int R>
__device__ __forceinline__ void dft_p(float2* v) {
    if constexpr (R == 2) dft2(v[0], v[1]);
    if constexpr (R == 3) dft3(v[0], v[1], v[2]);
    if constexpr (R == 4) dft4(v[0], v[1], v[2], v[3]);
    if constexpr (R == 5) dft5(v[0], v[1], v[2], v[3], v[4]);
    if constexpr (R == 6) dft6p(v);
    if constexpr (R == 8) dft8p(v);
    if constexpr (R == 10) dft10p(v);
    if constexpr (R > 10 || R == 7 || R == 9) dft_nat<R>(v);
}

// Rows-type tiles use a padded pitch of 17 points instead of the XOR swizzle: constant LDS offsets instead of integer
// work per access; the transposing store stays conflict-free (34-dword stride), 32-lane reads pay one extra LDS cycle.
// Measured +1.4 % on cfg4 (the kernels are VALU-bound: SQ_ACTIVE_INST_VALU x waves per SIMD ~ 100 %, LDS far from busy).
constexpr int kRowsPitch = W + 1;

// P17 = false (big tiles: the whole 80 KiB budget of a workgroup is tile) keeps the XOR swizzle for rows-type
// tiles: same conflict-free accesses, a few integer instructions per access instead of 1/16 more LDS.
template <bool SWZ, bool P17 = true>
__device__ __forceinline__ int lds_slot(int row, int w) {
    if (SWZ && P17) return row * (W + 1) + w;
    return SWZ ? row * W + (w ^ (row & (W - 1))) : row * W + w;
}

// Stage twiddles W_L^(q e), q = 1 .. R-1, as powers of one table entry (big tiles keep no copy of the table
// in LDS: one global load -- an L1 hit, the table is 5 KiB -- per butterfly, then a product tree of depth
// <= log2 R: w_q = w_hb(q) w_(q - hb(q)), hb = highest power of two <= q).
template <int R>
__device__ __forceinline__ void twiddle_powers(float2 w1, float2* pw) {
    pw[1] = w1;
#pragma unroll
    for (int q = 2; q < R; ++q) {
        int hb = 1;
        while (hb * 2 <= q) hb *= 2;
        pw[q] = (hb == q) ? cmul(pw[q / 2], pw[q / 2]) : cmul(pw[hb], pw[q - hb]);
    }
}

// Middle stage: LDS -> LDS.  MT = block length entering the stage.
// GTW: `tw` is the table in global memory and the q-th twiddle is a power of entry kp * step (twiddle_powers).
// ZQ > 0 (first stage only, MT == L): the rows from ZQ * m on hold zeros that nobody wrote -- inputs q >= ZQ of every
// butterfly are literal zeros: no LDS write and no LDS read for them (a tenth of the mask tile's LDS traffic; the
// butterflies themselves are packed inline assembly and a radix-5 kernel with two or three live inputs is no cheaper
// than a full one, so the arithmetic stays: cfg4 -0.1...0.2 %).
template <int L, int R, int MT, bool SWZ, int RG, bool P17 = true, bool GTW = false, int ZQ = 0>
__device__ __forceinline__ void stage_lds(float2* tile, const float2* tw, int w, int rg) {
    static_assert(ZQ == 0 || MT == L, "zero rows are declared for the first stage only");
    constexpr int m = MT / R, step = L / MT, rows = L / R, nit = (rows + RG - 1) / RG;
#pragma unroll
    for (int it = 0; it < nit; ++it) {
        const int b = rg + RG * it;
        if ((rows % RG == 0) || b < rows) {
            const int g = b / m, kp = b - g * m;
            const int base = g * MT + kp;
            float2 v[R];
            float2 pw[GTW ? R : 1];
            if constexpr (GTW) pw[0] = tw[kp * step];   // issued ahead of the LDS reads it will meet
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (ZQ > 0 && q >= ZQ) v[q] = make_float2(0.f, 0.f);
                else v[q] = tile[lds_slot<SWZ, P17>(base + q * m, w)];
            }
            dft_p<R>(v);
            if constexpr (GTW) {
                twiddle_powers<R>(pw[0], pw);
#pragma unroll
                for (int q = 1; q < R; ++q) v[dft_slot<R>(q)] = cmul(v[dft_slot<R>(q)], pw[q]);
            } else {
#pragma unroll
                for (int q = 1; q < R; ++q) v[dft_slot<R>(q)] = cmul(v[dft_slot<R>(q)], tw[q * kp * step]);
            }
#pragma unroll
            for (int q = 0; q < R; ++q) tile[lds_slot<SWZ, P17>(base + q * m, w)] = v[dft_slot<R>(q)];
        }
    }
}

// The big tiles (600 / 625 / 640 points, 75 .. 80 KiB) run as TWO 1024-thread workgroups per CU (32 waves, <= 64 VGPRs):
// while one workgroup waits for its tile the other one transforms.  (Moving the 480/500-point tiles over too measured
// no change: those already have two workgroups per CU.)
constexpr bool big_tile_pair(int L) { return L > kFftMaxL; }
// The 400-point tile without its twiddle table in LDS is 51 200 B, so THREE 512-thread workgroups fit a CU (<= 80 VGPRs;
// the rows form spills six dwords).  With the table (54 400 B, 57 600 with the 17-point pitch) only two do.  Measured:
// cfg5 (400 . 625 . 400 wideband plan) 2.095 -> 2.067 ms, same-box alternation.
constexpr bool triple_tile(int L) { return L == 400; }

// LoadOp contract:  fetch(id, l, tile_base, off) returns element tile_base + off of the input
//                   (tile_base is workgroup-uniform, off a 32-bit per-lane offset) and does NO
//                   arithmetic on the value; post(id, l, v) runs when the tile is consumed.
// StoreOp contract: operator()(id, k, tile_base, off, v).
template <int L, int R0, int R1, int R2, int R3, bool ROWS, int T, class LoadOp, class StoreOp>
__global__ __launch_bounds__(T, (big_tile_pair(L) ? 8 : triple_tile(L) ? 6 : T * 16 / W >= 512 ? 4 : 1)) void k_fft_tile(FftPassDev d, LoadOp load,
                                                                                          StoreOp store) {
    // BIG: two 1024-thread workgroups per CU -- the tile is the whole LDS budget of the workgroup (80 KiB), so the
    // stage twiddles come from the table in global memory (twiddle_powers) and rows tiles use the XOR swizzle.
    constexpr bool BIG = big_tile_pair(L) || triple_tile(L);
    constexpr int S = (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1);
    static_assert(S >= 2 && R0 * R1 * R2 * R3 == L, "bad radix list");
    constexpr int RL = (S == 2) ? R1 : (S == 3) ? R2 : R3;   // last radix
    constexpr int m0 = L / R0;
    constexpr int RG = T / W;                                  // butterfly rows handled per sweep
    constexpr int nit0 = (m0 + RG - 1) / RG;                   // first-stage butterflies per thread
    constexpr int nld = ROWS ? (L * W + T - 1) / T : nit0 * R0;
    __shared__ __attribute__((aligned(16))) float2 tile[L * (ROWS && !BIG ? kRowsPitch : W)];
    __shared__ __attribute__((aligned(16))) float2 tw_lds[BIG ? 1 : L];
    const float2* tw = BIG ? d.stage_tw : tw_lds;
    const FftPass& p = d.p;

    const int tid = threadIdx.x;
    const int w = tid & (W - 1), rg = tid >> kLogW;
    const VBlock vb = vblock_hw();
    LineId id;
    BlockPos bp = block_pos(vb);
    id.batch = bp.batch;
    unsigned o = vb.y;
    const unsigned n_o2 = (unsigned)p.n_o2;
    if (p.flat_outer) {
        // one flat tile index over (outer lines, tiles of a row), dealt to the XCDs in contiguous eighths by block_pos
        // (the launch rounds the count up to a multiple of 8: the surplus workgroups leave here, before any barrier)
        const unsigned tiles_inner = ((unsigned)p.n_inner + W - 1) / W;
        o = bp.tile / tiles_inner;
        if (o >= (unsigned)(p.n_o1 * p.n_o2)) return;
        bp.tile -= o * tiles_inner;
    }
    id.o1 = n_o2 == 1 ? o : o / n_o2;
    id.o2 = n_o2 == 1 ? 0 : o - (unsigned)id.o1 * n_o2;
    const int i0 = (int)bp.tile * W;
    const int left = (int)p.n_inner - i0;
    // Lanes past the end of a row (last tile only) load line 0 of the tile again and are never stored:
    // every lane transforms its own line, so nothing has to be masked in between.
    const int wvalid = left < W ? left : W;
    // (tile-blocked hand-over between strided passes: FftPass::in_t / out_t, 0 = the plain layout; the blocks are 16
    // lines wide whatever W is: an 8-line tile is the lower or upper half of one)
    const int64_t in_base = (int64_t)id.batch * d.in_batch + id.o1 * p.in_o1 + id.o2 * p.in_o2 +
                            (p.in_t ? (int64_t)(i0 >> 4) * p.in_t + (i0 & 15) : (int64_t)i0 * p.in_i);
    const int64_t out_base = (int64_t)id.batch * d.out_batch + id.o1 * p.out_o1 + id.o2 * p.out_o2 +
                             (p.out_t ? (int64_t)(i0 >> 4) * p.out_t + (i0 & 15) : (int64_t)i0);
    const unsigned in_l = (unsigned)p.in_l, in_i = (unsigned)p.in_i, out_k = (unsigned)p.out_k;

    // ---- global loads: all issued before anything waits -----------------------------------
    constexpr int NF = fetch_count<LoadOp>::value;
    constexpr bool HZ = !ROWS && has_half_zones<LoadOp>::value && L % 2 == 0;
    auto zone = [](int q) -> int { return (q + 1) * m0 <= L / 2 ? 0 : (q * m0 > L / 2 ? 2 : 1); };
    float2 v[nld];
    float2 v2[NF >= 2 ? nld : 1];
    float v3[NF == 3 ? nld : 1];
    static_assert(NF <= 2 || !ROWS, "three-fetch load functors: strided passes only");
    if constexpr (ROWS) {
#pragma unroll
        for (int it = 0; it < nld; ++it) {
            int e = tid + T * it;
            if ((L * W) % T != 0) e = e < L * W ? e : 0;
            const int wl = e / L, l = e - wl * L;
            const int wc = wl < wvalid ? wl : 0;     // clamped: always a valid line
            id.i = i0 + wc;
            v[it] = load.fetch(id, l, in_base, (unsigned)wc * in_i + (unsigned)l);
            if constexpr (NF == 2) v2[it] = load.fetch2(id, l, in_base, (unsigned)wc * in_i + (unsigned)l);
        }
    } else {
        const int wc = w < wvalid ? w : 0;
        id.i = i0 + wc;
#pragma unroll
        for (int it = 0; it < nit0; ++it) {
            int b = rg + RG * it;
            if (m0 % RG != 0) b = b < m0 ? b : 0;
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                const int l = b + q * m0;
                if constexpr (HZ) {
                    const int z = zone(q);
                    if (z == 2) {
                        v[it * R0 + q] = make_float2(0.f, 0.f);
                        if constexpr (NF == 2) v2[it * R0 + q] = make_float2(0.f, 0.f);
                    } else {
                        v[it * R0 + q] = load.fetch(id, l, in_base, (unsigned)l * in_l + (unsigned)wc, z);
                        if constexpr (NF == 2)
                            v2[it * R0 + q] = load.fetch2(id, l, in_base, (unsigned)l * in_l + (unsigned)wc, z);
                    }
                } else {
                    v[it * R0 + q] = load.fetch(id, l, in_base, (unsigned)l * in_l + (unsigned)wc);
                    if constexpr (NF >= 2)
                        v2[it * R0 + q] = load.fetch2(id, l, in_base, (unsigned)l * in_l + (unsigned)wc);
                    if constexpr (NF == 3)
                        v3[it * R0 + q] = load.fetch3(id, l, in_base, (unsigned)l * in_l + (unsigned)wc);
                }
            }
        }
    }
    if constexpr (!BIG)
        for (int e = tid; e < L; e += T) tw_lds[e] = d.stage_tw[e];
    constexpr bool CTX = has_ctx<LoadOp>::value;
    static_assert(!CTX || (!ROWS && NF == 1), "load context: strided single-fetch passes only");
    auto ctx = [&] {
        if constexpr (CTX) return load.prepare(id);
        else return 0;
    }();

    // Output index of slot 0 of last-stage block g (see the last stage below).
    constexpr int rowsL = L / RL, nitL = (rowsL + RG - 1) / RG;
    auto kbase = [](int g) -> int {
        if constexpr (S == 2) {
            return g;
        } else if constexpr (S == 3) {
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
    // Auxiliary inputs of the store functor (e.g. m[n] of the stereo mix) travel with the tile's
    // loads instead of costing a dependent round trip inside the store loop.
    constexpr bool AUX = has_aux<StoreOp>::value;
    float aux[AUX ? nitL * RL : 1];
    if constexpr (AUX) {
        const int wc = w < wvalid ? w : 0;
        id.i = i0 + wc;
#pragma unroll
        for (int it = 0; it < nitL; ++it) {
            int g = rg + RG * it;
            if (rowsL % RG != 0) g = g < rowsL ? g : 0;
            const int kb = kbase(g);
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                const int k = kb + (L / RL) * q;
                aux[it * RL + q] = store.fetch_aux(id, k, out_base, (unsigned)k * out_k + (unsigned)wc);
            }
        }
    }

    // ---- first stage ---------------------------------------------------------------------
    if constexpr (ROWS) {
#pragma unroll
        for (int it = 0; it < nld; ++it) {
            const int e = tid + T * it;
            const int wl = e / L, l = e - wl * L;
            id.i = i0 + wl;
            float2 x;
            if constexpr (NF == 2) x = load.post(id, l, v[it], v2[it]);
            else x = load.post(id, l, v[it]);
            if ((L * W) % T == 0 || e < L * W) tile[lds_slot<true, !BIG>(l, wl)] = x;
        }
        __syncthreads();
        stage_lds<L, R0, L, true, RG, !BIG, BIG>(tile, tw, w, rg);
    } else {
        if constexpr (!BIG) __syncthreads();   // tw[] complete
        id.i = i0 + (w < wvalid ? w : 0);   // same LineId as the fetch: index work is shared (masked below)
#pragma unroll
        for (int it = 0; it < nit0; ++it) {
            const int b = rg + RG * it;
            if ((m0 % RG == 0) || b < m0) {
                float2* x = &v[it * R0];
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    if constexpr (HZ) {
                        if constexpr (NF == 2) {
                            if (zone(q) != 2) x[q] = load.post(id, b + q * m0, x[q], v2[it * R0 + q], zone(q));
                        } else {
                            if (zone(q) != 2) x[q] = load.post(id, b + q * m0, x[q], zone(q));
                        }
                    } else if constexpr (NF == 3) x[q] = load.post(id, b + q * m0, x[q], v2[it * R0 + q], v3[it * R0 + q]);
                    else if constexpr (NF == 2) x[q] = load.post(id, b + q * m0, x[q], v2[it * R0 + q]);
                    else if constexpr (CTX) x[q] = load.post(id, b + q * m0, x[q], ctx);
                    else x[q] = load.post(id, b + q * m0, x[q]);
                }
                dft_p<R0>(x);
                if constexpr (BIG) {
                    float2 pw[R0];
                    twiddle_powers<R0>(tw[b], pw);
#pragma unroll
                    for (int q = 1; q < R0; ++q) x[dft_slot<R0>(q)] = cmul(x[dft_slot<R0>(q)], pw[q]);
                } else {
#pragma unroll
                    for (int q = 1; q < R0; ++q) x[dft_slot<R0>(q)] = cmul(x[dft_slot<R0>(q)], tw[q * b]);
                }
#pragma unroll
                for (int q = 0; q < R0; ++q) tile[lds_slot<false>(b + q * m0, w)] = x[dft_slot<R0>(q)];
            }
        }
    }
    __syncthreads();
    if constexpr (S >= 3) {
        stage_lds<L, R1, L / R0, ROWS, RG, !BIG, BIG>(tile, tw, w, rg);
        __syncthreads();
    }
    if constexpr (S >= 4) {
        stage_lds<L, R2, L / (R0 * R1), ROWS, RG, !BIG, BIG>(tile, tw, w, rg);
        __syncthreads();
    }

    // ---- last stage (block length RL, no stage twiddle): LDS -> registers -> memory ---------
    // Block g holds outputs k = kb(g) + (L / RL) q': g's digits are q_1 .. q_{S-1} with weights
    // L/(r_1 RL), L/(r_1 r_2 RL), ...; kb = q_1 + r_1 q_2 + r_1 r_2 q_3.
    // Inter-pass twiddle W_n^(f k), f = this lane's line factor: W_n^(f kb) once per butterfly,
    // then successive powers of D = W_n^(f L / RL) (RL - 1 <= 9 multiplications).
    const unsigned f = (unsigned)(id.o1 * p.tw_o1 + id.o2 * p.tw_o2 + (int64_t)(i0 + w) * p.tw_i);
    float2 D = make_float2(1.f, 0.f);
    constexpr bool PTW = !ROWS;
    if constexpr (PTW) D = big_twiddle(d, f * (unsigned)(L / RL));
    id.i = i0 + w;
    const bool lane_ok = w < wvalid;
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
            float2 x[RL];
#pragma unroll
            for (int q = 0; q < RL; ++q) x[q] = tile[lds_slot<ROWS, !BIG>(g * RL + q, w)];
            dft_p<RL>(x);
            const int kb = kbase(g);
            float2 Tw = make_float2(1.f, 0.f);
            if constexpr (PTW) Tw = big_twiddle(d, f * (unsigned)kb);
            if (lane_ok) {
#pragma unroll
                for (int q = 0; q < RL; ++q) {
                    const int k = kb + (L / RL) * q;
                    float2 y = x[dft_slot<RL>(q)];
                    if constexpr (PTW) {
                        y = cmul(y, Tw);
                        Tw = cmul(Tw, D);
                    }
                    if constexpr (AUX) store(id, k, out_base, (unsigned)k * out_k + (unsigned)w, y, aux[it * RL + q]);
                    else store(id, k, out_base, (unsigned)k * out_k + (unsigned)w, y);
                }
            }
        }
    }
}

template <class T, class = void>
struct mid_upper_rows_zero : std::false_type {};
template <class T>
struct mid_upper_rows_zero<T, std::void_t<decltype(T::kUpperRowsZero)>> : std::bool_constant<T::kUpperRowsZero> {};

// ---- two transforms on one tile -----------------------------------------------------
//
// When plan P1 = (n_2, n_1) ends where plan P2 = (n_1, n_2) begins, the last (rows) pass of P1
// and the first (strided) pass of P2 own the SAME tile: output k of line j of the first is
// input l = k of line j of the second (both are point j + n_2 k of the natural-order signal).
// This kernel runs both length-L transforms back to back with a point-wise stage in between
// (MidOp, natural order), so the signal between them never travels to memory: one read and
// one write instead of two of each.  Used for ifft -> stereo mix -> fft in WBFM.
//
// MidOp contract: kAux + fetch_aux(id, k, base, off) like StoreOp; float2 operator()(id, k, v, aux).
template <int L, int R0, int R1, int R2, int R3, int T, class LoadOp, class MidOp, class StoreOp>
__global__ __launch_bounds__(T, (T * 16 / W >= 512 ? 4 : 1)) void k_fft_tile2(FftPassDev d1, FftPassDev d2, LoadOp load,
                                                                     MidOp mid, StoreOp store) {
    constexpr int S = (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1);
    static_assert(S >= 2 && R0 * R1 * R2 * R3 == L, "bad radix list");
    constexpr int RL = (S == 2) ? R1 : (S == 3) ? R2 : R3;
    constexpr int RG = T / W;
    constexpr int nld = (L * W + T - 1) / T;
    constexpr int rowsL = L / RL, nitL = (rowsL + RG - 1) / RG;
    // one-sided point-wise stages (kUpperRowsZero): input q of the second transform's first-stage butterflies covers the
    // rows [q m, (q + 1) m), m = L / R0; from q = kZeroQ on all of them lie above L / 2 and are zero
    constexpr int kZeroQ = mid_upper_rows_zero<MidOp>::value ? (L / 2) / (L / R0) + 1 : 0;
    constexpr int kZeroFrom = kZeroQ > 0 && kZeroQ < R0 ? kZeroQ * (L / R0) : L;
    __shared__ __attribute__((aligned(16))) float2 tile[L * kRowsPitch];
    __shared__ __attribute__((aligned(16))) float2 tw[L];
    const FftPass& p1 = d1.p;
    const FftPass& p2 = d2.p;
    const int tid = threadIdx.x;
    const int w = tid & (W - 1), rg = tid >> kLogW;

    LineId id;
    const BlockPos bp = block_pos();
    id.batch = bp.batch;
    id.o1 = 0;
    id.o2 = 0;
    const int i0 = (int)bp.tile * W;
    const int left = (int)p1.n_inner - i0;
    const int wvalid = left < W ? left : W;
    const int64_t in_base = (int64_t)id.batch * d1.in_batch + (int64_t)i0 * p1.in_i;
    const int64_t mid_base = (int64_t)id.batch * d1.out_batch + i0;      // natural-order signal
    const int64_t out_base = (int64_t)id.batch * d2.out_batch + i0;
    const unsigned in_i = (unsigned)p1.in_i, mid_k = (unsigned)p1.out_k, out_k = (unsigned)p2.out_k;

    auto kbase = [](int g) -> int {
        if constexpr (S == 2) {
            return g;
        } else if constexpr (S == 3) {
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

    // ---- loads of the first transform (contiguous lines) + the mid stage's inputs ----------
    float2 v[nld];
#pragma unroll
    for (int it = 0; it < nld; ++it) {
        int e = tid + T * it;
        if ((L * W) % T != 0) e = e < L * W ? e : 0;
        const int wl = e / L, l = e - wl * L;
        const int wc = wl < wvalid ? wl : 0;
        id.i = i0 + wc;
        v[it] = load.fetch(id, l, in_base, (unsigned)wc * in_i + (unsigned)l);
    }
    float aux[nitL * RL];
    {
        const int wc = w < wvalid ? w : 0;
        id.i = i0 + wc;
#pragma unroll
        for (int it = 0; it < nitL; ++it) {
            int g = rg + RG * it;
            if (rowsL % RG != 0) g = g < rowsL ? g : 0;
            const int kb = kbase(g);
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                const int k = kb + (L / RL) * q;
                aux[it * RL + q] = mid.fetch_aux(id, k, mid_base, (unsigned)k * mid_k + (unsigned)wc);
            }
        }
    }
    for (int e = tid; e < L; e += T) tw[e] = d1.stage_tw[e];

#pragma unroll
    for (int it = 0; it < nld; ++it) {
        const int e = tid + T * it;
        const int wl = e / L, l = e - wl * L;
        id.i = i0 + wl;
        float2 x = load.post(id, l, v[it]);
        if ((L * W) % T == 0 || e < L * W) tile[lds_slot<true>(l, wl)] = x;
    }
    lds_barrier();
    stage_lds<L, R0, L, true, RG>(tile, tw, w, rg);
    lds_barrier();
    if constexpr (S >= 3) {
        stage_lds<L, R1, L / R0, true, RG>(tile, tw, w, rg);
        lds_barrier();
    }
    if constexpr (S >= 4) {
        stage_lds<L, R2, L / (R0 * R1), true, RG>(tile, tw, w, rg);
        lds_barrier();
    }

    // ---- last stage of the first transform: results leave their digit-reversed slots, pass the
    // mid stage, and land in natural order (row k) for the second transform -----------------
    id.i = i0 + w;
    float2 xr[nitL * RL];
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
#pragma unroll
            for (int q = 0; q < RL; ++q) xr[it * RL + q] = tile[lds_slot<true>(g * RL + q, w)];
            dft_p<RL>(&xr[it * RL]);
        }
    }
    lds_barrier();   // every slot has been read
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
            const int kb = kbase(g);
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                const int k = kb + (L / RL) * q;
                // A MidOp that declares kUpperRowsZero maps every row k > L / 2 to zero (one-sided spectra).  kb < L / RL,
                // so the rows of this q start at (L / RL) q: decided at compile time, and the outputs of the last-stage
                // butterfly that nobody reads are not computed at all (dead code in the unrolled DFT).  Rows from
                // kZeroFrom on are not even written: the second transform's first stage takes them as literal zeros.
                if constexpr (mid_upper_rows_zero<MidOp>::value) {
                    if ((L / RL) * q > L / 2) {
                        if ((L / RL) * q < kZeroFrom) tile[lds_slot<true>(k, w)] = make_float2(0.f, 0.f);
                        continue;
                    }
                }
                float2 y = mid(id, k, xr[it * RL + dft_slot<RL>(q)], aux[it * RL + q]);
                tile[lds_slot<true>(k, w)] = y;
            }
        }
    }
    lds_barrier();

    // ---- second transform: a strided pass of plan 2 whose input already sits in LDS ----------
    stage_lds<L, R0, L, true, RG, true, false, kZeroQ>(tile, tw, w, rg);
    lds_barrier();
    if constexpr (S >= 3) {
        stage_lds<L, R1, L / R0, true, RG>(tile, tw, w, rg);
        lds_barrier();
    }
    if constexpr (S >= 4) {
        stage_lds<L, R2, L / (R0 * R1), true, RG>(tile, tw, w, rg);
        lds_barrier();
    }
    const bool lane_ok = w < wvalid;
    const unsigned f = (unsigned)((int64_t)(i0 + w) * p2.tw_i);
    const float2 D = big_twiddle(d2, f * (unsigned)(L / RL));
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
            float2 x[RL];
#pragma unroll
            for (int q = 0; q < RL; ++q) x[q] = tile[lds_slot<true>(g * RL + q, w)];
            dft_p<RL>(x);
            const int kb = kbase(g);
            float2 Tw = big_twiddle(d2, f * (unsigned)kb);
            if (lane_ok) {
#pragma unroll
                for (int q = 0; q < RL; ++q) {
                    const int k = kb + (L / RL) * q;
                    const float2 y = cmul(x[dft_slot<RL>(q)], Tw);
                    Tw = cmul(Tw, D);
                    store(id, k, out_base, (unsigned)k * out_k + (unsigned)w, y);
                }
            }
        }
    }
}

// ---- one inverse transform in, two forward transforms out ------------------------------
//
// k_fft_tile2 for PAIRS of real signals that travelled through one complex transform: the first
// transform's line belongs to pair P = blockIdx.z; the point-wise stage turns each of its values w
// into one value per member (channels 2P and 2P+1), and each member gets its own second transform
// (same tile, one after the other; the second member waits in registers).  Used by WBFM: the
// analytic signals of two pilots come out of ONE masked inverse FFT (fused_passes.h).
//
// MidOp contract (natural-order point `off` of the tile, channel bases b0 / b1 in signal samples):
//   In0 fetch0(b0, b1, off)          inputs needed for member 0 and for the split (issued early)
//   float fetch1(b1, off)            what member 1 still needs (issued while member 0 transforms)
//   float2 first(w, In0, Keep&)      member 0's value; Keep = what member 1 needs later
//   float2 second(Keep, float)       member 1's value
template <int L, int R0, int R1, int R2, int R3, int T, class LoadOp, class MidOp, class StoreOp>
__global__ __launch_bounds__(T, (T * 16 / W >= 512 ? 4 : 1)) void k_fft_tile2_pair(FftPassDev d1, FftPassDev d2, LoadOp load,
                                                                          MidOp mid, StoreOp store, int count) {
    constexpr int S = (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1);
    static_assert(S >= 2 && R0 * R1 * R2 * R3 == L, "bad radix list");
    constexpr int RL = (S == 2) ? R1 : (S == 3) ? R2 : R3;
    constexpr int RG = T / W;
    constexpr int nld = (L * W + T - 1) / T;
    constexpr int rowsL = L / RL, nitL = (rowsL + RG - 1) / RG;
    __shared__ __attribute__((aligned(16))) float2 tile[L * kRowsPitch];
    __shared__ __attribute__((aligned(16))) float2 tw[L];
    const FftPass& p1 = d1.p;
    const FftPass& p2 = d2.p;
    const int tid = threadIdx.x;
    const int w = tid & (W - 1), rg = tid >> kLogW;

    LineId id;
    const BlockPos bp = block_pos();
    id.batch = bp.batch;   // pair index
    id.o1 = 0;
    id.o2 = 0;
    const int c0 = 2 * (int)bp.batch;
    const bool has1 = c0 + 1 < count;          // an odd count leaves the last pair with one member
    const int c1 = has1 ? c0 + 1 : c0;
    const int i0 = (int)bp.tile * W;
    const int left = (int)p1.n_inner - i0;
    const int wvalid = left < W ? left : W;
    const int64_t in_base = (int64_t)id.batch * d1.in_batch + (int64_t)i0 * p1.in_i;
    // natural-order signals of the members -- or tile-blocked ones (MidOp::blk16: the tile's 16 L values contiguous)
    const int blk16 = mid.blk16;
    const int64_t mid_tile = blk16 ? (int64_t)(i0 >> 4) * blk16 + (i0 & 15) : (int64_t)i0;
    const int64_t mid_base0 = (int64_t)c0 * (blk16 ? mid.stride : d1.out_batch) + mid_tile;
    const int64_t mid_base1 = (int64_t)c1 * (blk16 ? mid.stride : d1.out_batch) + mid_tile;
    const int64_t out_base0 = (int64_t)c0 * d2.out_batch + i0;
    const int64_t out_base1 = (int64_t)c1 * d2.out_batch + i0;
    const unsigned in_i = (unsigned)p1.in_i, mid_k = blk16 ? 16u : (unsigned)p1.out_k, out_k = (unsigned)p2.out_k;
    const int wc = w < wvalid ? w : 0;

    auto kbase = [](int g) -> int {
        if constexpr (S == 2) {
            return g;
        } else if constexpr (S == 3) {
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
    // Natural-order offset of point (it, q) of this thread = lane part (it) + uniform part (q): the
    // uniform part goes into the scalar base, so a thread holds nitL offsets instead of nitL * RL.
    auto lane_off = [&](int it) -> unsigned {
        int g = rg + RG * it;
        if (rowsL % RG != 0) g = g < rowsL ? g : 0;
        return (unsigned)kbase(g) * mid_k + (unsigned)wc;
    };
    auto q_off = [&](int q) -> int64_t { return (int64_t)((L / RL) * q) * mid_k; };

    // ---- loads of the first transform (contiguous lines) ------------------------------------
    float2 v[nld];
#pragma unroll
    for (int it = 0; it < nld; ++it) {
        int e = tid + T * it;
        if ((L * W) % T != 0) e = e < L * W ? e : 0;
        const int wl = e / L, l = e - wl * L;
        const int wcl = wl < wvalid ? wl : 0;
        id.i = i0 + wcl;
        v[it] = load.fetch(id, l, in_base, (unsigned)wcl * in_i + (unsigned)l);
    }
    for (int e = tid; e < L; e += T) tw[e] = d1.stage_tw[e];
#pragma unroll
    for (int it = 0; it < nld; ++it) {
        const int e = tid + T * it;
        const int wl = e / L, l = e - wl * L;
        id.i = i0 + wl;
        const float2 x = load.post(id, l, v[it]);
        if ((L * W) % T == 0 || e < L * W) tile[lds_slot<true>(l, wl)] = x;
    }
    lds_barrier();
    stage_lds<L, R0, L, true, RG>(tile, tw, w, rg);
    lds_barrier();
    if constexpr (S >= 3) {
        stage_lds<L, R1, L / R0, true, RG>(tile, tw, w, rg);
        lds_barrier();
    }
    if constexpr (S >= 4) {
        stage_lds<L, R2, L / (R0 * R1), true, RG>(tile, tw, w, rg);
        lds_barrier();
    }

    // ---- point-wise inputs of member 0 (and of the split), then the last stage ----------------
    using In0 = decltype(mid.fetch0(mid_base0, mid_base1, 0u));
    In0 a0[nitL * RL];
#pragma unroll
    for (int it = 0; it < nitL; ++it)
#pragma unroll
        for (int q = 0; q < RL; ++q)
            a0[it * RL + q] = mid.fetch0(mid_base0 + q_off(q), mid_base1 + q_off(q), lane_off(it));

    // Last stage of the first transform and the split, one butterfly row at a time (the inputs a0
    // retire as the values u0 / keep appear: fewer registers than holding all of w first).
    id.i = i0 + w;
    using Keep = typename MidOp::Keep;
    float2 u0[nitL * RL];
    Keep keep[nitL * RL];
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
            float2 x[RL];
#pragma unroll
            for (int q = 0; q < RL; ++q) x[q] = tile[lds_slot<true>(g * RL + q, w)];
            dft_p<RL>(x);
#pragma unroll
            for (int q = 0; q < RL; ++q)
                u0[it * RL + q] = mid.first(x[dft_slot<RL>(q)], a0[it * RL + q], keep[it * RL + q]);
        }
        __builtin_amdgcn_sched_barrier(0);   // one row at a time: hoisted LDS reads of the next rows would spill
    }
    lds_barrier();   // every slot has been read
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
            const int kb = kbase(g);
#pragma unroll
            for (int q = 0; q < RL; ++q) tile[lds_slot<true>(kb + (L / RL) * q, w)] = u0[it * RL + q];
        }
    }
    lds_barrier();

    // ---- second transform (a strided pass of plan 2 whose input already sits in LDS), per member
    const unsigned f = (unsigned)((int64_t)(i0 + w) * p2.tw_i);
    const float2 D = big_twiddle(d2, f * (unsigned)(L / RL));
    const bool lane_ok = w < wvalid;
    auto second_transform = [&](int64_t out_base) {
        stage_lds<L, R0, L, true, RG>(tile, tw, w, rg);
        lds_barrier();
        if constexpr (S >= 3) {
            stage_lds<L, R1, L / R0, true, RG>(tile, tw, w, rg);
            lds_barrier();
        }
        if constexpr (S >= 4) {
            stage_lds<L, R2, L / (R0 * R1), true, RG>(tile, tw, w, rg);
            lds_barrier();
        }
#pragma unroll
        for (int it = 0; it < nitL; ++it) {
            const int g = rg + RG * it;
            if ((rowsL % RG == 0) || g < rowsL) {
                float2 x[RL];
#pragma unroll
                for (int q = 0; q < RL; ++q) x[q] = tile[lds_slot<true>(g * RL + q, w)];
                dft_p<RL>(x);
                const int kb = kbase(g);
                float2 Tw = big_twiddle(d2, f * (unsigned)kb);
                if (lane_ok) {
#pragma unroll
                    for (int q = 0; q < RL; ++q) {
                        const int k = kb + (L / RL) * q;
                        const float2 y = cmul(x[dft_slot<RL>(q)], Tw);
                        Tw = cmul(Tw, D);
                        store(id, k, out_base, (unsigned)k * out_k + (unsigned)w, y);
                    }
                }
            }
        }
    };

    float a1[nitL * RL];
#pragma unroll
    for (int it = 0; it < nitL; ++it)
#pragma unroll
        for (int q = 0; q < RL; ++q) a1[it * RL + q] = mid.fetch1(mid_base1 + q_off(q), lane_off(it));

    second_transform(out_base0);
    if (!has1) return;   // workgroup-uniform
    lds_barrier();     // member 0's last stage has read every slot
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
            const int kb = kbase(g);
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                const int k = kb + (L / RL) * q;
                tile[lds_slot<true>(k, w)] = mid.second(keep[it * RL + q], a1[it * RL + q]);
            }
        }
    }
    lds_barrier();
    second_transform(out_base1);
}

// ---- a long forward transform's last pass feeding a short inverse transform's first pass --------
//
// Spectral decimation B -> A (scipy.signal.resample) between two transforms on one tile.  Plan
// (n_1, L) of length B ends with rows tiles of 16 lines k_0 x L outputs k_1 (bin k = k_0 + n_1 k_1); when
// A = n_1 L2 (L2 even), the bins that survive the decimation, |k| <= A/2, are rows k_1 < L2/2 and
// k_1 > L - L2/2 (+ the two Nyquist rows) of the SAME lines, and bin kappa = k_0 + n_1 l of the short
// spectrum is row l of a strided first pass of plan (L2, n_1) of length A.  So: transform 1 (L points),
// keep L2 rows (the two Nyquist rows merge in line k_0 = 0), weight them (WinOp), transform 2 (L2
// points, inverse by the swap identity).  The long spectrum never reaches memory.
//
// WinOp contract: float weight(id, l, k0) (issued with the tile's loads), void dc_bin(id, float2 v0) receives
// the weighted bin kappa = 0.
template <int L, int R0, int R1, int R2, int R3, int L2, int Q0, int Q1, int T, class LoadOp, class WinOp, class StoreOp>
__global__ __launch_bounds__(T, (T * 16 / W >= 512 ? 4 : 1)) void k_fft_tile2_decim(FftPassDev d1, FftPassDev d2, LoadOp load,
                                                                           WinOp win, StoreOp store) {
    constexpr int S = (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1);
    static_assert(S >= 2 && R0 * R1 * R2 * R3 == L, "bad radix list");
    static_assert(Q0 * Q1 == L2 && L2 % 2 == 0 && L2 < L, "bad short transform");
    constexpr int RL = (S == 2) ? R1 : (S == 3) ? R2 : R3;
    constexpr int RG = T / W;
    constexpr int nld = (L * W + T - 1) / T;
    constexpr int rowsL = L / RL, nitL = (rowsL + RG - 1) / RG;
    constexpr int nwin = (L2 * W + T - 1) / T;          // weighting sweep: points per thread
    __shared__ __attribute__((aligned(16))) float2 tile[L * kRowsPitch];
    __shared__ __attribute__((aligned(16))) float2 tw[L];
    const FftPass& p1 = d1.p;
    const FftPass& p2 = d2.p;
    const int tid = threadIdx.x;
    const int w = tid & (W - 1), rg = tid >> kLogW;

    LineId id;
    const BlockPos bp = block_pos();
    id.batch = bp.batch;
    id.o1 = 0;
    id.o2 = 0;
    const int i0 = (int)bp.tile * W;
    const int left = (int)p1.n_inner - i0;
    const int wvalid = left < W ? left : W;
    const int64_t in_base = (int64_t)id.batch * d1.in_batch + (int64_t)i0 * p1.in_i;
    const int64_t out_base = (int64_t)id.batch * d2.out_batch + i0;
    const unsigned in_i = (unsigned)p1.in_i, out_k = (unsigned)p2.out_k;

    auto kbase = [](int g) -> int {
        if constexpr (S == 2) {
            return g;
        } else if constexpr (S == 3) {
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

    // ---- loads: the tile and the weights of the rows that survive -------------------------------
    float2 v[nld];
#pragma unroll
    for (int it = 0; it < nld; ++it) {
        int e = tid + T * it;
        if ((L * W) % T != 0) e = e < L * W ? e : 0;
        const int wl = e / L, l = e - wl * L;
        const int wcl = wl < wvalid ? wl : 0;
        id.i = i0 + wcl;
        v[it] = load.fetch(id, l, in_base, (unsigned)wcl * in_i + (unsigned)l);
    }
    float wgt[nwin];
#pragma unroll
    for (int it = 0; it < nwin; ++it) {
        int e = tid + T * it;
        if ((L2 * W) % T != 0) e = e < L2 * W ? e : 0;
        const int wl = e & (W - 1);
        wgt[it] = win.weight(id, e >> kLogW, i0 + (wl < wvalid ? wl : 0));
    }
    for (int e = tid; e < L; e += T) tw[e] = d1.stage_tw[e];
#pragma unroll
    for (int it = 0; it < nld; ++it) {
        const int e = tid + T * it;
        const int wl = e / L, l = e - wl * L;
        id.i = i0 + wl;
        const float2 x = load.post(id, l, v[it]);
        if ((L * W) % T == 0 || e < L * W) tile[lds_slot<true>(l, wl)] = x;
    }
    lds_barrier();
    stage_lds<L, R0, L, true, RG>(tile, tw, w, rg);
    lds_barrier();
    if constexpr (S >= 3) {
        stage_lds<L, R1, L / R0, true, RG>(tile, tw, w, rg);
        lds_barrier();
    }
    if constexpr (S >= 4) {
        stage_lds<L, R2, L / (R0 * R1), true, RG>(tile, tw, w, rg);
        lds_barrier();
    }

    // ---- last stage of transform 1; the surviving rows land in natural order -------------------
    float2 xr[nitL * RL];
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
#pragma unroll
            for (int q = 0; q < RL; ++q) xr[it * RL + q] = tile[lds_slot<true>(g * RL + q, w)];
            dft_p<RL>(&xr[it * RL]);
        }
    }
    lds_barrier();   // every slot has been read
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
            const int kb = kbase(g);
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                // kb < L / RL: the rows k = kb + (L / RL) q of this q lie in [(L/RL) q, (L/RL)(q + 1)); when that range
                // misses both kept bands and the Nyquist row, nothing of it survives the decimation -- decided at
                // compile time (q is unrolled), so the select, the LDS write and the butterfly's unused outputs go
                if ((L / RL) * q > L2 / 2 && (L / RL) * (q + 1) <= L - L2 / 2) continue;
                const int k = kb + (L / RL) * q;
                // row of the short spectrum: positive frequencies, negative frequencies, the negative
                // Nyquist row (kept by every line), the positive one (parked in row L2: only bin A/2 of
                // line k_0 = 0 needs it)
                int l = -1;
                if (k < L2 / 2) l = k;
                else if (k >= L - L2 / 2) l = k - (L - L2);
                else if (k == L2 / 2) l = L2;
                if (l >= 0) tile[lds_slot<true>(l, w)] = xr[it * RL + dft_slot<RL>(q)];
            }
        }
    }
    for (int e = tid; e < L2; e += T) tw[e] = d2.stage_tw[e];   // (every read of transform 1's table is done)
    lds_barrier();

    // ---- weights, Nyquist merge, swap for the inverse transform ----------------------------------
#pragma unroll
    for (int it = 0; it < nwin; ++it) {
        const int e = tid + T * it;
        if ((L2 * W) % T == 0 || e < L2 * W) {
            const int l = e >> kLogW, wl = e & (W - 1);
            float2 x = tile[lds_slot<true>(l, wl)];
            if (l == L2 / 2 && i0 + wl == 0) {   // Y[A/2] = X[A/2] + X[-A/2] (one point of one tile)
                const float2 y = tile[lds_slot<true>(L2, wl)];
                x = make_float2(x.x + y.x, x.y + y.y);
            }
            x = make_float2(x.x * wgt[it], x.y * wgt[it]);
            if (l == 0 && i0 + wl == 0) {
                id.i = 0;
                win.dc_bin(id, x);
            }
            tile[lds_slot<true>(l, wl)] = make_float2(x.y, x.x);
        }
    }
    lds_barrier();

    // ---- transform 2: a strided pass of the short plan whose input sits in LDS ---------------------
    stage_lds<L2, Q0, L2, true, RG>(tile, tw, w, rg);
    lds_barrier();
    constexpr int rows2 = L2 / Q1;
    static_assert(rows2 <= RG, "short transform: one sweep");
    id.i = i0 + w;
    const unsigned f = (unsigned)((int64_t)(i0 + w) * p2.tw_i);
    if (rg < rows2) {
        float2 x[Q1];
#pragma unroll
        for (int q = 0; q < Q1; ++q) x[q] = tile[lds_slot<true>(rg * Q1 + q, w)];
        dft_p<Q1>(x);
        const float2 D = big_twiddle(d2, f * (unsigned)(L2 / Q1));
        float2 Tw = big_twiddle(d2, f * (unsigned)rg);
        if (w < wvalid) {
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                const int k = rg + (L2 / Q1) * q;
                const float2 y = cmul(x[dft_slot<Q1>(q)], Tw);
                Tw = cmul(Tw, D);
                store(id, k, out_base, (unsigned)k * out_k + (unsigned)w, y);
            }
        }
    }
}

// ---- plain functors ------------------------------------------------------------
// SWAP = exchange re/im: the inverse transform by the swap identity ifft(x) = swap(fft(swap(x))).

// Streaming accesses of the tile passes, measured on MI355X (c2c 256 x 240000 / N = 2.4e8): non-temporal STORES +3 % /
// +2 %, non-temporal loads -7 % / -2 %: loads are plain, stores non-temporal -- for whole 128-byte segments only (the
// 16-line build; the 8-line build's 64-byte segments take plain stores: non-temporal partial lines stream at 2.4-3.6 TB/s).
__device__ __forceinline__ float2 stream_load(const float2* p) { return *p; }

__device__ __forceinline__ void stream_store(float2* p, float2 v) {
    if constexpr (W == 16) {
        using v2 = __attribute__((ext_vector_type(2))) float;
        v2 t;
        t.x = v.x;
        t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<v2*>(p));
    } else {
        *p = v;
    }
}

template <bool SWAP>
struct LoadPlainT {
    const float2* in;
    __device__ __forceinline__ float2 fetch(const LineId&, int, int64_t base, unsigned off) const {
        return stream_load(in + base + off);
    }
    __device__ __forceinline__ float2 post(const LineId&, int, float2 v) const {
        return SWAP ? make_float2(v.y, v.x) : v;
    }
};

template <bool SWAP>
struct StorePlainT {
    float2* out;
    float scale;
    __device__ __forceinline__ void operator()(const LineId&, int, int64_t base, unsigned off, float2 v) const {
        stream_store(out + base + off,
                     SWAP ? make_float2(v.y * scale, v.x * scale) : make_float2(v.x * scale, v.y * scale));
    }
};

// Last pass of a forward transform of which only a window of output rows is wanted (FftRowWindow): row =
// o1 + n_o1 (o2 + n_o2 k) in the last pass's line coordinates.
struct StoreRowWindow {
    float2* out;
    float scale;
    int n_o1, n_o2, lo, hi;
    // halo > 0 (single transforms only): the first `halo` outputs are repeated behind the end and the last `halo` in
    // front of the start (out[n + g] and out[g - n]): the tuner's haloed spectrum without two extra copy launches.
    int64_t n = 0;
    int halo = 0;
    int sc1_straddle = 0;   // 1: segments that straddle two 128-byte lines leave with sc1 (outputs beyond the Infinity Cache)
    __device__ __forceinline__ void operator()(const LineId& id, int k, int64_t base, unsigned off, float2 v) const {
        const int row = (int)id.o1 + n_o1 * ((int)id.o2 + n_o2 * k);
        const bool keep = lo <= hi ? (row >= lo && row <= hi) : (row >= lo || row <= hi);
        if (!keep) return;
        const float2 y = make_float2(v.x * scale, v.y * scale);
        const int64_t g = base + off;
        // Half of this pass's 128-byte segments straddle two lines when the output stride is an odd multiple of 64 bytes
        // (N = 2.4e8: 375 000 bins): those leave with sc1, the aligned ones non-temporal.  Same-address A/B: wideband FFT
        // 2.245 -> 2.221 ms (plain stores for the straddling ones: 2.285; sc1 or plain for ALL of them: slower than nt;
        // profiles/r05_b_kernel_ab.txt, r05_s_row_store_mix.txt).
        // Only for outputs that stream to HBM: a cache-resident transform (N = 1e7) measured 0.8 % slower with it.
        const bool straddle = sc1_straddle &&
                              ((reinterpret_cast<uintptr_t>(out + g) - ((uintptr_t)((int)id.i & (W - 1)) << 3)) & 127u) != 0;
        if (straddle) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(out + g), "v"(y) : "memory");
        else stream_store(out + g, y);
        if (halo > 0) {
            if (g < halo) out[n + g] = y;
            else if (g >= n - halo) out[g - n] = y;
        }
    }
};

}  // namespace fftk
RCFM_NS_CLOSE
}  // namespace rcfm
namespace rcfm {
RCFM_NS_OPEN
namespace fftk {

// Lengths with a compile-time specialisation (radices listed first stage first; they must
// match choose_radices() in fft_engine.hip).  Other lengths run the generic kernel.
// The 480- and 500-point tiles run two LDS stages of composite radices (dft_nat); the pair kernel keeps three.
#define RCFM_FFT_LONG_TABLE3(X) X(480, 10, 8, 6, 1) X(500, 10, 10, 5, 1)
#define RCFM_FFT_LONG_TABLE(X) X(480, 24, 20, 1, 1) X(500, 25, 20, 1, 1)
// Big tiles (fft_engine.h, kFftBigL): one 1024-thread workgroup per CU.  Two stages of radix 24..32 leave
// 60 % of the 1024 threads idle and measured slower here (3.1 vs 2.75 ms at N = 2.4e8).
#ifndef RCFM_R640
#define RCFM_R640 10, 8, 8, 1
#endif
#ifndef RCFM_R600
#define RCFM_R600 10, 10, 6, 1
#endif
#define RCFM_FFT_BIG_X_(X, LEN, ...) X(LEN, __VA_ARGS__)
#define RCFM_FFT_BIG_LENGTHS(X) RCFM_FFT_BIG_X_(X, 600, RCFM_R600) X(625, 5, 5, 5, 5) RCFM_FFT_BIG_X_(X, 640, RCFM_R640)
// Two stages wherever both keep at least ~60 % of the threads on a butterfly (L / R >= 10 rows of 16 lanes
// for 256 threads, >= 19 for 512); 125 as 25 x 5 (5 rows) measured 4 % slower on cfg5.
#define RCFM_FFT_LENGTHS_(X, LONGT) \
    X(75, 5, 5, 3, 1)            \
    X(80, 10, 8, 1, 1)           \
    X(100, 10, 10, 1, 1)         \
    X(120, 12, 10, 1, 1)         \
    X(125, 5, 5, 5, 1)           \
    X(128, 8, 8, 2, 1)           \
    X(150, 15, 10, 1, 1)         \
    X(160, 16, 10, 1, 1)         \
    X(192, 16, 12, 1, 1)         \
    X(200, 20, 10, 1, 1)         \
    X(240, 24, 10, 1, 1)         \
    X(250, 25, 10, 1, 1)         \
    X(256, 16, 16, 1, 1)         \
    X(300, 20, 15, 1, 1)         \
    X(320, 10, 8, 4, 1)          \
    X(375, 5, 5, 5, 3)           \
    X(384, 8, 8, 6, 1)           \
    X(400, 20, 20, 1, 1)         \
    LONGT(X)                     \
    X(512, 8, 8, 8, 1)
#define RCFM_FFT_FAST_LENGTHS(X) RCFM_FFT_LENGTHS_(X, RCFM_FFT_LONG_TABLE)
// k_fft_tile2_pair keeps RL points per last-stage butterfly row in registers next to the point-wise
// stage's inputs: with a last radix of 20 it spills, so it stays on three stages.
#define RCFM_FFT_PAIR_LENGTHS(X) RCFM_FFT_LENGTHS_(X, RCFM_FFT_LONG_TABLE3)


// Threads per tile.  Long tiles are LDS-limited to two workgroups per CU; 512 threads keep 16 waves per CU in flight
// there; the big tiles take 1024.
constexpr int tile_threads(int L) {   // (RG = T / W butterfly rows per sweep is the same for every tile width)
    return (big_tile_pair(L) ? 1024 : L >= 320 ? 512 : 256) * W / 16;
}

// Big tiles are instantiated for the plain functors only (the streaming passes of long transforms).
template <class T> struct is_plain_functor : std::false_type {};
template <bool S> struct is_plain_functor<LoadPlainT<S>> : std::true_type {};
template <bool S> struct is_plain_functor<StorePlainT<S>> : std::true_type {};
template <> struct is_plain_functor<StoreRowWindow> : std::true_type {};

// Which pass kinds a functor pair is ever used with (prunes template instantiations).
enum PassKinds : int { kAnyPass = 0, kStridedOnly = 1, kRowsOnly = 2 };

template <int LEN, int A, int B, int C, int D, bool ROWS, class LoadOp, class StoreOp>
inline void launch_fft_tile_one(const FftPassDev& d, dim3 grid, const LoadOp& ld, const StoreOp& st, hipStream_t s) {
    constexpr int T = tile_threads(LEN);
    hipLaunchKernelGGL((k_fft_tile<LEN, A, B, C, D, ROWS, T, LoadOp, StoreOp>), grid, dim3(T), 0, s, d, ld, st);
}

template <int KIND = kAnyPass, class LoadOp, class StoreOp>
inline void launch_fft_pass(const FftPassDev& d, int batch, const LoadOp& ld, const StoreOp& st, hipStream_t s) {
    bool done = false;
    const bool rows = d.p.load_along_l != 0;
    RC_REQUIRE(!(KIND == kStridedOnly && rows) && !(KIND == kRowsOnly && !rows), RCFM_ERR_RUNTIME,
               "FFT functor used with the wrong pass kind");
    const bool shape_ok = (rows || d.p.in_i == 1) && d.p.out_i == 1 && ((d.p.has_twiddle != 0) == !rows) &&
                          d.p.n_o1 * d.p.n_o2 <= 65535 && batch <= 65535;
    if (shape_ok) {
        const unsigned tiles_inner = (unsigned)((d.p.n_inner + W - 1) / W), outer = (unsigned)(d.p.n_o1 * d.p.n_o2);
        const dim3 grid = d.p.flat_outer ? dim3((tiles_inner * outer + 7u) / 8u * 8u, 1, (unsigned)batch)
                                         : dim3(tiles_inner, outer, (unsigned)batch);
        switch (d.p.L) {
#define RCFM_CASE(LEN, A, B, C, D)                                                                            \
    case LEN:                                                                                                 \
        if (rows) {                                                                                           \
            if constexpr (KIND != kStridedOnly)                                                               \
                launch_fft_tile_one<LEN, A, B, C, D, true, LoadOp, StoreOp>(d, grid, ld, st, s);               \
        } else {                                                                                              \
            if constexpr (KIND != kRowsOnly)                                                                  \
                launch_fft_tile_one<LEN, A, B, C, D, false, LoadOp, StoreOp>(d, grid, ld, st, s);              \
        }                                                                                                     \
        done = true;                                                                                          \
        break;
            RCFM_FFT_FAST_LENGTHS(RCFM_CASE)
            default: break;
        }
        if constexpr (W == 16 && is_plain_functor<LoadOp>::value && is_plain_functor<StoreOp>::value) {
            switch (d.p.L) {
                RCFM_FFT_BIG_LENGTHS(RCFM_CASE)
                default: break;
            }
        }
#undef RCFM_CASE
    }
    if (!done)
        hipLaunchKernelGGL((k_fft_pass<LoadOp, StoreOp>),
                           dim3((unsigned)(d.p.n_o1 * d.p.n_o2 * ((d.p.n_inner + W - 1) / W)), (unsigned)batch, 1),
                           dim3(kThreads), (size_t)d.p.L * (W * sizeof(float2) + sizeof(float2) + sizeof(uint16_t)), s, d,
                           ld, st);
    RC_HIP(hipGetLastError());
}

// d1: last pass of plan (n_2, n_1); d2: first pass of plan (n_1, n_2).  Returns false when the
// pair does not tile identically or the length has no specialisation (caller runs them apart).
inline bool fft_tile2_applies(const FftPassDev& d1, const FftPassDev& d2, int batch) {
    bool fast = false;
    switch (d1.p.L) {
#define RCFM_CASE(LEN, A, B, C, D) case LEN:
        RCFM_FFT_FAST_LENGTHS(RCFM_CASE)
#undef RCFM_CASE
        fast = true;
        break;
        default: break;
    }
    return fast && d1.p.load_along_l && !d2.p.load_along_l && d1.p.L == d2.p.L &&
           d1.p.n_inner == d2.p.n_inner && d1.p.n_o1 * d1.p.n_o2 == 1 && d2.p.n_o1 * d2.p.n_o2 == 1 &&
           d1.p.out_k == d2.p.in_l && d2.p.in_i == 1 && d2.p.out_i == 1 && d2.p.has_twiddle && batch <= 65535;
}

// Spectral decimation between two transforms (k_fft_tile2_decim): (long last-pass length, short
// first-pass length) pairs with an instantiation.
#define RCFM_FFT_DECIM_500(X) X(500, 25, 20, 1, 1, 100, 10, 10)
#define RCFM_FFT_DECIM_125(X) X(125, 5, 5, 5, 1, 80, 10, 8)
#define RCFM_FFT_DECIM_PAIRS(X)          \
    RCFM_FFT_DECIM_500(X)                \
    RCFM_FFT_DECIM_125(X)

inline bool fft_tile2_decim_applies(const FftPassDev& d1, const FftPassDev& d2, int batch) {
    bool fast = false;
#define RCFM_CASE(LEN, A, B, C, D, LEN2, E, F) fast = fast || (d1.p.L == LEN && d2.p.L == LEN2);
    RCFM_FFT_DECIM_PAIRS(RCFM_CASE)
#undef RCFM_CASE
    return fast && d1.p.load_along_l && !d2.p.load_along_l &&
           d1.p.n_inner == d2.p.n_inner && d1.p.n_o1 * d1.p.n_o2 == 1 && d2.p.n_o1 * d2.p.n_o2 == 1 &&
           d1.p.out_k == d2.p.in_l && d2.p.in_i == 1 && d2.p.out_i == 1 && d2.p.has_twiddle && batch <= 65535;
}

template <class LoadOp, class WinOp, class StoreOp>
inline bool launch_fft_tile2_decim(const FftPassDev& d1, const FftPassDev& d2, int batch, const LoadOp& ld,
                                   const WinOp& win, const StoreOp& st, hipStream_t s) {
    if (!fft_tile2_decim_applies(d1, d2, batch)) return false;
    const dim3 grid((unsigned)((d1.p.n_inner + W - 1) / W), 1, (unsigned)batch);
    bool done = false;
#define RCFM_CASE(LEN, A, B, C, D, LEN2, E, F)                                                                   \
    if (!done && d1.p.L == LEN && d2.p.L == LEN2) {                                                             \
        hipLaunchKernelGGL((k_fft_tile2_decim<LEN, A, B, C, D, LEN2, E, F, tile_threads(LEN), LoadOp, WinOp, StoreOp>), \
                           grid, dim3(tile_threads(LEN)), 0, s, d1, d2, ld, win, st);                           \
        done = true;                                                                                            \
    }
    RCFM_FFT_DECIM_PAIRS(RCFM_CASE)
#undef RCFM_CASE
    if (!done) return false;
    RC_HIP(hipGetLastError());
    return true;
}

// The pair form: `count` member signals, ceil(count / 2) first transforms.
template <class LoadOp, class MidOp, class StoreOp>
inline bool launch_fft_tile2_pair(const FftPassDev& d1, const FftPassDev& d2, int count, const LoadOp& ld,
                                  const MidOp& mid, const StoreOp& st, hipStream_t s) {
    const int pairs = (count + 1) / 2;
    if (!fft_tile2_applies(d1, d2, pairs)) return false;
    const dim3 grid((unsigned)((d1.p.n_inner + W - 1) / W), 1, (unsigned)pairs);
    switch (d1.p.L) {
#define RCFM_CASE(LEN, A, B, C, D)                                                                               \
    case LEN:                                                                                                    \
        hipLaunchKernelGGL((k_fft_tile2_pair<LEN, A, B, C, D, tile_threads(LEN), LoadOp, MidOp, StoreOp>), grid, \
                           dim3(tile_threads(LEN)), 0, s, d1, d2, ld, mid, st, count);                           \
        break;
        RCFM_FFT_PAIR_LENGTHS(RCFM_CASE)
#undef RCFM_CASE
        default: return false;
    }
    RC_HIP(hipGetLastError());
    return true;
}

template <class LoadOp, class MidOp, class StoreOp>
inline bool launch_fft_tile2(const FftPassDev& d1, const FftPassDev& d2, int batch, const LoadOp& ld,
                             const MidOp& mid, const StoreOp& st, hipStream_t s) {
    if (!fft_tile2_applies(d1, d2, batch)) return false;
    const dim3 grid((unsigned)((d1.p.n_inner + W - 1) / W), 1, (unsigned)batch);
    switch (d1.p.L) {
#define RCFM_CASE(LEN, A, B, C, D)                                                                           \
    case LEN:                                                                                                \
        hipLaunchKernelGGL((k_fft_tile2<LEN, A, B, C, D, tile_threads(LEN), LoadOp, MidOp, StoreOp>), grid,  \
                           dim3(tile_threads(LEN)), 0, s, d1, d2, ld, mid, st);                              \
        break;
        RCFM_FFT_FAST_LENGTHS(RCFM_CASE)
#undef RCFM_CASE
        default: return false;
    }
    RC_HIP(hipGetLastError());
    return true;
}

}  // namespace fftk
RCFM_NS_CLOSE
}  // namespace rcfm
