#include "fused_passes.h"
#include "device_math.h"

#include "fft_kernel.h"
#include "fft_decim_rt.h"
#include "kernels.h"

namespace rcfm {
RCFM_NS_OPEN

using fftk::LineId;
using fftk::kAnyPass;
using fftk::kRowsOnly;
using fftk::kStridedOnly;

namespace {

// ---- functors -------------------------------------------------------------------------
// LoadOp::operator() fetches only; LoadOp::post does the arithmetic when the tile is consumed.

// First pass of Tuner.run's inverse FFT: element k of the channel spectrum comes from
// bin (src - roll) mod N of the wideband spectrum, src = k (k < nyq) or N - (B - k).
// The hot form of the gather below: narrow channels (window argument < 0.25 rad: 4-term cosine
// series), B <= N (every bin has a source), haloed spectrum (no wrap-around), 32-bit indices.
// ~12 VALU instructions per element instead of ~30.  (A 2-term series -- enough below 0.029 rad, cfg4 has 0.003 -- measured
// the same to 0.03 % in a same-address A/B, profiles/r05_b_kernel_ab.txt: the window is not what this pass waits for.)
struct LoadTunerGatherFast {
    const float2* X;        // bin 0 of the haloed spectrum
    const int32_t* base;    // per channel: (N - roll) mod N
    float two_pi_over_n, delta;
    float c0, c1, c2, c3;   // a0 + (1-a0) cos(th) = c0 + t c1 + t^2 c2 + t^3 c3,  t = th^2
    int B, nyq;
    int merge;              // NYQ_DOWN: the bin (B/2) that also receives X[-B/2]; otherwise -1
    int line_stride;

    __device__ __forceinline__ int source_offset(int k) const { return k < nyq ? k : k - B; }
    __device__ __forceinline__ float window(int d) const {
        const float th = fmaf((float)d, two_pi_over_n, delta);
        const float t = th * th;
        return fmaf(t, fmaf(t, fmaf(t, c3, c2), c1), c0);
    }
    __device__ __forceinline__ float2 fetch(const LineId& id, int l, int64_t, unsigned) const {
        return X[base[id.batch] + source_offset(l * line_stride + (int)id.i)];
    }
    // NYQ_DOWN: Y[+B/2] += X[-B/2] w(-B/2): a workgroup-uniform value that one lane of one tile adds.
    static constexpr bool kCtx = true;
    using Ctx = float2;
    __device__ __forceinline__ Ctx prepare(const LineId& id) const {
        if (merge < 0) return make_float2(0.f, 0.f);
        const float2 x2 = X[base[id.batch] - merge];
        const float w2 = window(-merge);
        return make_float2(x2.x * w2, x2.y * w2);
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 v, const Ctx& m2) const {
        const int k = l * line_stride + (int)id.i;
        const float w = window(source_offset(k));
        const float sel = (k == merge) ? 1.f : 0.f;
        // inverse transform by the swap identity
        return make_float2(fmaf(sel, m2.y, v.y * w), fmaf(sel, m2.x, v.x * w));
    }
};

// The general form.  I = index type of a wideband bin: int32 when N < 2^30.
template <typename I, bool SERIES>
struct LoadTunerGatherT {
    const float2* X;
    const int64_t* roll;
    I N;
    float two_pi_over_n, delta;
    float a0;
    int B, nyq, nneg, nyq_mode;
    int line_stride;   // in_l of the pass: k = l * line_stride + i
    int64_t x_batch;   // spectrum of signal c starts at X + c * x_batch (0: one shared spectrum)

    // Signed offset d of the source bin of output bin k (source = d mod N); ok = false: zero fill.
    // Branch-free per lane; only the nyq_mode tests (kernel arguments) branch, uniformly.
    __device__ __forceinline__ int source_offset(int k, bool& ok) const {
        const int j = B - k;
        const bool pos = k < nyq;
        int d = pos ? k : -j;
        ok = pos | (j <= nneg);
        if (nyq_mode == NYQ_UP) {   // Y[-N/2] = Y[+N/2] / 2
            const bool up = !ok & (j == nyq - 1);
            d = up ? nyq - 1 : d;
            ok |= up;
        }
        return ok ? d : 0;
    }
    __device__ __forceinline__ I rolled(int d, I r) const {   // (d - r) mod N, |d| <= N/2, r in [0, N)
        I i = (I)d - r;
        if (d < 0) i += N;
        return i < 0 ? i + N : i;
    }
    // fftshift(get_window(...))[src] = a0 - (1 - a0) cos(2 pi ((src - N//2) mod N) / N)
    //                                = a0 + (1 - a0) cos(2 pi d / N + delta),  d the signed offset
    // (|d| <= N/2), delta = pi / N for odd N.  For |theta| < 0.06 a 4-term series is exact to 1e-14;
    // otherwise the library cosine.
    __device__ __forceinline__ float window(int d) const {
        const float th = (float)d * two_pi_over_n + delta;
        float c;
        if constexpr (SERIES) {
            const float t2 = th * th;
            c = 1.f - t2 * 0.5f * (1.f - t2 * (1.f / 12.f) * (1.f - t2 * (1.f / 30.f)));
        } else {
            c = cosf(th);
        }
        return a0 + (1.f - a0) * c;
    }
    __device__ __forceinline__ float2 fetch(const LineId& id, int l, int64_t, unsigned) const {
        bool ok;
        const int d = source_offset(l * line_stride + (int)id.i, ok);   // the same expression as in post()
        return (X + (int64_t)id.batch * x_batch)[rolled(d, (I)roll[id.batch])];
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 v) const {
        const int k = l * line_stride + (int)id.i;
        bool ok;
        const int d = source_offset(k, ok);
        float w = ok ? window(d) : 0.f;
        const int half = nyq - 1;
        if (nyq_mode == NYQ_UP) w = (k == half || (B - k) == half) ? 0.5f * w : w;
        float2 y = make_float2(v.x * w, v.y * w);
        if (nyq_mode == NYQ_DOWN) {   // Y[+N/2] += X[-N/2]: one element per channel
            // workgroup-uniform address and value (hoisted out of the per-element code); lanes select
            const float2 x2 = (X + (int64_t)id.batch * x_batch)[rolled(-half, (I)roll[id.batch])];
            const float w2 = (k == half) ? window(-half) : 0.f;
            y.x += x2.x * w2;
            y.y += x2.y * w2;
        }
        return make_float2(y.y, y.x);   // inverse transform by the swap identity
    }
};

struct LoadRealAsComplex {
    const float* x;
    __device__ __forceinline__ float2 fetch(const LineId&, int, int64_t base, unsigned off) const {
        return make_float2((x + base)[off], 0.f);
    }
    __device__ __forceinline__ float2 post(const LineId&, int, float2 v) const { return v; }
};

// Two real signals per complex FFT: u = x_even + j x_odd.
struct LoadRealPair {
    const float* x;
    int n, count;
    // blk16 != 0: x is tile-blocked (PilotBlocked, kernels.h): channel stride `stride`, line i of row l at
    // (i / 16) blk16 + 16 l + i % 16 -- the 16 (8) lines of a tile read one contiguous run
    int64_t stride = 0;
    int blk16 = 0;
    __device__ __forceinline__ float2 fetch(const LineId& id, int l, int64_t base, unsigned off) const {
        // in_batch = 0: base + off = sample index inside the signal, base = the tile's first line
        const int c0 = 2 * id.batch, c1 = (c0 + 1 < count) ? c0 + 1 : c0;
        if (blk16) {
            const int i = (int)id.i;
            const int64_t a = (int64_t)(i >> 4) * blk16 + l * 16 + (i & 15);
            return make_float2(x[(int64_t)c0 * stride + a], x[(int64_t)c1 * stride + a]);
        }
        const int64_t a = base + off;
        return make_float2(x[(int64_t)c0 * n + a], x[(int64_t)c1 * n + a]);
    }
    __device__ __forceinline__ float2 post(const LineId&, int, float2 v) const { return v; }
};

// The same pair with the FM discriminator (fm.py:60-65) on the load: theta = angle(x) / pi of the two
// channels; the real signals are the wrapped phase steps d[t] = wrap(theta[t] - theta[t-1]), d[0] = 0.
struct LoadPhaseStepPair {
    static constexpr int kFetches = 2;
    const float* theta;
    int n, count;
    int line_stride;
    // padded rows (fused_tuner_ifft's theta_pitch): sample t sits at t + (t / row) (pitch - row); magic = ceil(2^32 / row)
    int row = 0, pad = 0;
    unsigned magic = 0;
    int64_t stride = 0;               // values between consecutive channels
    __device__ __forceinline__ int where(int t) const {
        return row ? t + (int)__umulhi((unsigned)t, magic) * pad : t;
    }
    __device__ __forceinline__ float2 at(const LineId& id, int t) const {
        const int c0 = 2 * id.batch, c1 = (c0 + 1 < count) ? c0 + 1 : c0;
        const int i = where(t);
        return make_float2(theta[(int64_t)c0 * stride + i], theta[(int64_t)c1 * stride + i]);
    }
    __device__ __forceinline__ float2 fetch(const LineId& id, int l, int64_t, unsigned) const {
        return at(id, l * line_stride + (int)id.i);
    }
    __device__ __forceinline__ float2 fetch2(const LineId& id, int l, int64_t, unsigned) const {
        const int t = l * line_stride + (int)id.i;
        return at(id, t > 0 ? t - 1 : 0);
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 a, float2 b) const {
        const int t = l * line_stride + (int)id.i;
        if (t == 0) return make_float2(0.f, 0.f);
        return make_float2(phase_step_wrapped(a.x, b.x), phase_step_wrapped(a.y, b.y));
    }
};

LoadPhaseStepPair phase_pair_load(const float* x, int64_t n, int count, int line_stride, const PhaseRows& r) {
    LoadPhaseStepPair ld{x, (int)n, count, line_stride};
    ld.stride = r.channel_stride(n);
    if (r.row > 0 && r.pitch > r.row) {
        // floor(t / row) = umulhi(t, ceil(2^32 / row)) holds for t * row < 2^32 (t < n <= 2^20 here)
        RC_REQUIRE(n % r.row == 0 && n < (1 << 20) && r.row < (1 << 12), RCFM_ERR_RUNTIME, "phase rows outside the reader's range");
        ld.row = r.row;
        ld.pad = r.pitch - r.row;
        ld.magic = (unsigned)((((uint64_t)1 << 32) + (uint64_t)r.row - 1) / (uint64_t)r.row);
    }
    return ld;
}

// Hilbert mask applied to one member of such a pair: with U = FFT(x0 + j x1),
// X0[k] = (U[k] + conj U[-k]) / 2, X1[k] = (U[k] - conj U[-k]) / 2j; Z = h X, bins above n/2 are 0.
// HZ: the pass is the first of a two-pass plan of even length (point k = l * stride + i, n = L *
// stride), so the kernel may skip everything above the middle (fft_kernel.h, kHalfZones).
template <bool HZ>
struct LoadHilbertPairT {
    static constexpr int kFetches = 2;
    static constexpr bool kHalfZones = HZ;
    const float2* U;     // [ceil(count / 2)][n]
    int n;
    int line_stride;
    __device__ __forceinline__ const float2* pair(const LineId& id) const {
        return U + (int64_t)(id.batch >> 1) * n;
    }
    // zone 0: 0 <= k < n/2 is known; zone 1: anything
    __device__ __forceinline__ float2 fetch(const LineId& id, int l, int64_t, unsigned, int zone = 1) const {
        const int k = l * line_stride + (int)id.i;
        return pair(id)[(zone == 0 || k <= n / 2) ? k : 0];
    }
    __device__ __forceinline__ float2 fetch2(const LineId& id, int l, int64_t, unsigned, int zone = 1) const {
        const int k = l * line_stride + (int)id.i;
        return pair(id)[((zone == 0 || k <= n / 2) && k > 0) ? n - k : 0];
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 a, float2 b, int zone = 1) const {
        const int k = l * line_stride + (int)id.i;
        float h = (k == 0) ? 0.5f : 1.f;   // h / 2 of scipy.signal.hilbert's {1, 2, ..., 2, (1), 0, ...}
        if (zone != 0) {
            if (k >= (n + 1) / 2) h = 0.f;
            if ((n & 1) == 0 && k == n / 2) h = 0.5f;
        }
        float2 xk;
        if ((id.batch & 1) == 0) xk = make_float2(a.x + b.x, a.y - b.y);          // U[k] + conj U[-k]
        else xk = make_float2(a.y + b.y, -(a.x - b.x));                           // (U[k] - conj U[-k]) / j
        return make_float2(xk.y * h, xk.x * h);   // swapped: inverse transform
    }
};

// scipy.signal.hilbert's mask h = {1, 2, ..., 2, (1), 0, ...} times `scale` on a packed spectrum
// U2 [pairs][n] (no unpacking: the pair stays packed through the inverse transform).
template <bool HZ>
struct LoadHilbertPackedT {
    static constexpr bool kHalfZones = HZ;
    const float2* U2;
    int n;
    int line_stride;
    float scale;
    __device__ __forceinline__ float2 fetch(const LineId& id, int l, int64_t, unsigned, int zone = 1) const {
        const int k = l * line_stride + (int)id.i;
        return (U2 + (int64_t)id.batch * n)[(zone == 0 || k <= n / 2) ? k : 0];
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 v, int zone = 1) const {
        const int k = l * line_stride + (int)id.i;
        float h = (k == 0) ? scale : 2.f * scale;
        if (zone != 0) {
            if (k >= (n + 1) / 2) h = 0.f;
            if ((n & 1) == 0 && k == n / 2) h = scale;
        }
        return make_float2(v.y * h, v.x * h);   // swapped: inverse transform
    }
};

template <class Launch>
void with_hilbert_pair_load(const FftEngine& e, const float2* U, Launch&& launch) {
    const FftPass& p = e.desc().pass[0];
    const int64_t n = e.desc().n;
    if (e.npass() == 2 && !p.load_along_l && p.in_l * p.L == n && p.L % 2 == 0)
        launch(LoadHilbertPairT<true>{U, (int)n, (int)p.in_l});
    else
        launch(LoadHilbertPairT<false>{U, (int)n, (int)p.in_l});
}

// scipy.signal.hilbert's mask on the full spectrum U: h = {1, 2, ..., 2, (1), 0, ...}.
struct LoadHilbertMask {
    const float2* U;
    int n;
    int64_t line_stride;
    __device__ __forceinline__ float2 fetch(const LineId& id, int, int64_t base, unsigned off) const {
        // bins above n/2 are zeroed: read the channel's bin 0 again instead (cache hit, no HBM)
        const int64_t cbase = (int64_t)id.batch * n;
        const int64_t a = base + off;
        return U[(a - cbase) <= n / 2 ? a : cbase];
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 v) const {
        const int k = (int)(l * line_stride + id.i);
        float h = 0.f;
        if (k == 0) h = 1.f;
        else if (k < (n + 1) / 2) h = 2.f;
        else if ((n & 1) == 0 && k == n / 2) h = 1.f;
        return make_float2(v.y * h, v.x * h);   // swapped: inverse transform
    }
};

// wbfm.py:83,86-87 for one sample: z = (v.y, v.x) (swap identity), s2 = Im(z^2)/|z^2| = 2ab / (a^2 + b^2),
// lmr = s2 m 1.0175, packed (m + lmr, m - lmr).  One v_rcp_f32 (1 ulp); like pll.py:57-58 in complex64
// the quotient is NaN for z == 0 and meaningless once |z|^2 leaves the float32 range.
__device__ __forceinline__ float pilot_carrier(float a, float b) {   // Im(z^2)/|z^2|, z = a + j b
    return (2.f * a * b) * __builtin_amdgcn_rcpf(fmaf(a, a, b * b));
}
__device__ __forceinline__ float2 stereo_mix_point(float2 v, float mm) {
    const float s2 = pilot_carrier(v.y, v.x);
    const float lmr = (s2 * mm) * 1.0175f;
    return make_float2(mm + lmr, mm - lmr);
}

// Last pass of the analytic-signal IFFT: z (still swapped) -> stereo mix -> packed u.
struct StoreStereoMix {
    static constexpr bool kAux = true;   // m[n] is fetched with the tile's loads
    const float* m;
    float2* u;
    __device__ __forceinline__ float fetch_aux(const LineId&, int, int64_t base, unsigned off) const {
        return (m + base)[off];
    }
    __device__ __forceinline__ void operator()(const LineId&, int, int64_t base, unsigned off, float2 v,
                                               float mm) const {
        fftk::stream_store(u + base + off, stereo_mix_point(v, mm));
    }
};

// The same mix as the point-wise stage between two transforms (k_fft_tile2).
struct MidStereoMix {
    [[maybe_unused]] static constexpr bool kAux = true;
    const float* m;
    __device__ __forceinline__ float fetch_aux(const LineId&, int, int64_t base, unsigned off) const {
        return (m + base)[off];
    }
    __device__ __forceinline__ float2 operator()(const LineId&, int, float2 v, float mm) const {
        return stereo_mix_point(v, mm);
    }
};

// scipy.signal.hilbert's mask as the point-wise stage between the pair FFT's last pass and the masked
// inverse FFT's first pass (k_fft_tile2): bin = k * stride + i; the value leaves swapped and scaled for the
// inverse transform.  No auxiliary input.
template <bool UPPER_ZERO>
struct MidHilbertMaskT {
    [[maybe_unused]] static constexpr bool kAux = true;
    // UPPER_ZERO: n = L * stride with L even, so every row k > L / 2 holds bins above n / 2 only and the mask zeroes
    // it: the kernel then neither computes those outputs of the first transform nor calls the functor for them.
    static constexpr bool kUpperRowsZero = UPPER_ZERO;
    int n, stride;
    float scale;
    __device__ __forceinline__ float fetch_aux(const LineId&, int, int64_t, unsigned) const { return 0.f; }
    __device__ __forceinline__ float2 operator()(const LineId& id, int k, float2 v, float) const {
        const int bin = k * stride + (int)id.i;
        float h = (bin == 0) ? scale : 2.f * scale;
        if (bin >= (n + 1) / 2) h = 0.f;
        if ((n & 1) == 0 && bin == n / 2) h = scale;
        return make_float2(v.y * h, v.x * h);
    }
};

// The same mix for PAIRS of channels whose pilots went through one complex transform
// (k_fft_tile2_pair): U2 = FFT(p0 + j p1), w = IFFT(h U2) = z0 + j z1 with z_c = p_c + j H_c, so
// H1 = p0 - Re w and H0 = Im w - p1: one masked inverse FFT yields both analytic signals.
struct MidStereoMixPair {
    const float* p;   // [count][n] pilot bands
    const float* m;   // [count][n] mono signals
    // blk16 != 0: both are tile-blocked (PilotBlocked, kernels.h) with `stride` floats per channel; the kernel then
    // takes its row pitch (16) and tile base from here instead of the plan
    int64_t stride = 0;
    int blk16 = 0;
    struct In0 { float p0, p1, m0; };
    using Keep = float;   // member 1's carrier s2 = Im(z1^2)/|z1^2|
    // uniform 64-bit base + 32-bit lane byte offset: one VGPR of address per point, shared by the
    // three streams (54 loads per thread are in flight together)
    static __device__ __forceinline__ float at(const float* base, unsigned off) {
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(off * 4u));
    }
    __device__ __forceinline__ In0 fetch0(int64_t b0, int64_t b1, unsigned off) const {
        return In0{at(p + b0, off), at(p + b1, off), at(m + b0, off)};
    }
    __device__ __forceinline__ float fetch1(int64_t b1, unsigned off) const { return at(m + b1, off); }
    // v = w with re/im exchanged (inverse transform by the swap identity)
    __device__ __forceinline__ float2 first(float2 v, const In0& a, Keep& s2_1) const {
        s2_1 = pilot_carrier(a.p1, a.p0 - v.y);
        const float lmr = (pilot_carrier(a.p0, v.x - a.p1) * a.m0) * 1.0175f;
        return make_float2(a.m0 + lmr, a.m0 - lmr);
    }
    __device__ __forceinline__ float2 second(const Keep& s2_1, float m1) const {
        const float lmr = (s2_1 * m1) * 1.0175f;
        return make_float2(m1 + lmr, m1 - lmr);
    }
};

// wbfm.py:86-87 / decimate.py:48 as the load of IFFT_A's first pass: U = FFT_B of the packed signal
// (m+lmr) + j(m-lmr), pruned to |k| <= A/2; point k of the packed Hermitian pair V = HL + j HR
// (ifft(V) = l + j r), with the Hamming weight, truncation and Nyquist rule of scipy.signal.resample.
struct LoadStereoUnpack {
    static constexpr int kFetches = 3;
    const float2* U;      // [count][B]
    const float* wr;      // folded window, nyq entries
    float2* dc;           // [count] or null: receives V[c][0] = (sum l, sum r) / A
    int B, A, nyq, nmin;
    float nyq_factor, scale;
    int line_stride;
    __device__ __forceinline__ int folded(int k) const { return k > A / 2 ? A - k : k; }
    __device__ __forceinline__ float2 fetch(const LineId& id, int l, int64_t, unsigned) const {
        const int kk = folded(l * line_stride + (int)id.i);
        return (U + (int64_t)id.batch * B)[kk < nyq ? kk : 0];
    }
    __device__ __forceinline__ float2 fetch2(const LineId& id, int l, int64_t, unsigned) const {
        const int kk = folded(l * line_stride + (int)id.i);
        return (U + (int64_t)id.batch * B)[(kk < nyq && kk > 0) ? B - kk : 0];
    }
    __device__ __forceinline__ float fetch3(const LineId& id, int l, int64_t, unsigned) const {
        const int kk = folded(l * line_stride + (int)id.i);
        return wr[kk < nyq ? kk : 0];
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 a, float2 b, float w0) const {
        const int k = l * line_stride + (int)id.i;
        const bool mirrored = k > A / 2;
        const int kk = mirrored ? A - k : k;
        float2 hl = make_float2(0.f, 0.f), hr = make_float2(0.f, 0.f);
        if (kk < nyq) {
            // u = l + j r with l, r real  =>  L[k] = (U[k] + conj U[-k]) / 2,  R[k] = (U[k] - conj U[-k]) / 2j
            float w = w0 * scale;
            if ((nmin & 1) == 0 && kk == nmin / 2) w *= nyq_factor;
            hl = make_float2(0.5f * (a.x + b.x) * w, 0.5f * (a.y - b.y) * w);
            hr = make_float2(0.5f * (a.y + b.y) * w, -0.5f * (a.x - b.x) * w);
            if (kk == 0 || ((A & 1) == 0 && kk == A / 2)) {
                hl.y = 0.f;
                hr.y = 0.f;
            }
        }
        if (mirrored) {
            hl.y = -hl.y;
            hr.y = -hr.y;
        }
        const float2 out = make_float2(hl.x - hr.y, hl.y + hr.x);
        if (k == 0 && dc != nullptr) dc[id.batch] = out;   // one lane of one tile per channel
        return make_float2(out.y, out.x);                  // swapped: inverse transform
    }
};

// (WinAudioDecim, the weight of the decimation between two transforms, lives in fft_decim_rt.h: two translation units use it)

// Stores bins k <= lo and k >= hi only.
struct StorePruned {
    float2* out;
    int n, lo, hi;
    StorePruned(float2* o, int n_, int keep) : out(o), n(n_) {
        if (keep == kKeepLowerHalf) {   // one-sided users (Hilbert mask): bins 0 .. n/2
            lo = n_ / 2;
            hi = n_ + 1;
        } else if (keep < 0) {
            lo = n_;
            hi = 0;
        } else {                        // decimation: |k| <= keep
            lo = keep;
            hi = n_ - keep;
        }
    }
    __device__ __forceinline__ void operator()(const LineId& id, int, int64_t base, unsigned off, float2 v) const {
        const int k = (int)(base + off - (int64_t)id.batch * n);
        if (k <= lo || k >= hi) fftk::stream_store(out + base + off, v);
    }
};

struct StoreRealPart {
    float* y;
    float scale;
    __device__ __forceinline__ void operator()(const LineId&, int, int64_t base, unsigned off, float2 v) const {
        (y + base)[off] = v.y * scale;   // swap identity: the real part of the inverse transform is v.y
    }
};

// Runs passes [first, last] of a plan with plain functors between tmp buffers.
void middle_passes(const FftEngine& e, int from, int to, float2* tmp, int count, hipStream_t s) {
    const int64_t ts = e.tmp_stride();
    for (int t = from; t <= to; ++t) {
        fftk::LoadPlainT<false> ld{tmp};
        fftk::StorePlainT<false> st{tmp, 1.0f};
        fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(t, ts, ts), count, ld, st, s);
    }
}

}  // namespace

// Last pass of the tuner's inverse FFT when only the phase is wanted (the FM discriminator, fm.py:60-65):
// angle(x) / pi as float32, half the bytes of x.  v arrives with re/im exchanged (swap identity).
struct StorePhase {
    float* theta;
    __device__ __forceinline__ void operator()(const LineId&, int, int64_t base, unsigned off, float2 v) const {
        (theta + base)[off] = atan2_over_pi(v.x, v.y);
    }
};

void fused_tuner_ifft(const FftEngine& e, const TunerGather& g, float2* out, float2* tmp, int count,
                      hipStream_t s, float* theta, int theta_pitch) {
    if (count <= 0) return;
    const int64_t B = e.desc().n;
    const int np = e.npass();
    // in_batch = 0: the load functor addresses X itself; `a` is the bin index inside the channel
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    auto first_pass = [&](auto ld) {
        ld.X = g.X;
        ld.roll = g.roll;
        ld.N = (decltype(ld.N))g.N;
        ld.two_pi_over_n = (float)(6.28318530717958647692 / (double)g.N);
        ld.delta = (g.N % 2) ? (float)(3.14159265358979323846 / (double)g.N) : 0.f;
        ld.a0 = (float)g.a0;
        ld.B = (int)B;
        ld.nyq = g.nyq;
        ld.nneg = g.nneg;
        ld.nyq_mode = g.nyq_mode;
        ld.line_stride = (int)e.desc().pass[0].in_l;
        ld.x_batch = g.x_batch;
        fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), count, ld, st0, s);
    };
    // 4-term cosine series: truncation theta^8 / 40320 < 4e-10 for |theta| < 0.25 rad (channels up to 8 % of the
    // band), far below float32 rounding; wider channels take the library cosine of the general form
    const bool series = 6.28318530717958647692 * ((double)(B / 2 + 2) / (double)g.N) < 0.25;
    if (series && g.base32 && g.halo >= B / 2 + 1 && g.nyq_mode != NYQ_UP && B <= g.N && g.x_batch == 0) {
        LoadTunerGatherFast ld;
        const double a1 = 1.0 - g.a0;
        ld.X = g.X;
        ld.base = g.base32;
        ld.two_pi_over_n = (float)(6.28318530717958647692 / (double)g.N);
        ld.delta = (g.N % 2) ? (float)(3.14159265358979323846 / (double)g.N) : 0.f;
        ld.c0 = (float)(g.a0 + a1);
        ld.c1 = (float)(-a1 / 2.0);
        ld.c2 = (float)(a1 / 24.0);
        ld.c3 = (float)(-a1 / 720.0);
        ld.B = (int)B;
        ld.nyq = g.nyq;
        ld.merge = g.nyq_mode == NYQ_DOWN ? g.nyq - 1 : -1;
        ld.line_stride = (int)e.desc().pass[0].in_l;
        fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), count, ld, st0, s);
    } else if (g.N >= (int64_t)1 << 30) {
        first_pass(LoadTunerGatherT<int64_t, false>{});
    } else {
        first_pass(LoadTunerGatherT<int32_t, false>{});
    }
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
    if (theta != nullptr) {
        FftPassDev last = e.pass_dev(np - 1, e.tmp_stride(), B);
        if (theta_pitch > 0) {
            RC_REQUIRE(theta_pitch >= last.p.out_k && last.p.n_o1 * last.p.n_o2 == 1, RCFM_ERR_RUNTIME, "bad phase row pitch");
            last.out_batch = (B / last.p.out_k) * (int64_t)theta_pitch;
            last.p.out_k = theta_pitch;
        }
        fftk::launch_fft_pass<kRowsOnly>(last, count, ldl, StorePhase{theta}, s);
        return;
    }
    fftk::StorePlainT<true> stl{out, (float)(1.0 / (double)g.N)};   // ifft (1/B) * (B/N)
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), B), count, ldl, stl, s);
}

void fused_real_fft(const FftEngine& e, const float* x, float2* U, float2* tmp, int count, int keep,
                    hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    LoadRealAsComplex ld{x};
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, n, e.tmp_stride()), count, ld, st0, s);
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
    StorePruned stl{U, (int)n, keep};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), count, ldl, stl, s);
}

void fused_hilbert_ifft_mix(const FftEngine& e, const float2* U, const float* m, float2* u, float2* tmp,
                            int count, hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    LoadHilbertMask ld{U, (int)n, e.desc().pass[0].in_l};
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, n, e.tmp_stride()), count, ld, st0, s);
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
    StoreStereoMix stl{m, u};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), count, ldl, stl, s);
}

void fused_hilbert_ifft(const FftEngine& e, const float2* U, float2* z, float2* tmp, int count, hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    LoadHilbertMask ld{U, (int)n, e.desc().pass[0].in_l};
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, n, e.tmp_stride()), count, ld, st0, s);
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
    fftk::StorePlainT<true> stl{z, (float)(1.0 / (double)n)};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), count, ldl, stl, s);
}
void fused_real_pair_fft(const FftEngine& e, const float* x, float2* U, float2* tmp, int count, int keep,
                         hipStream_t s, bool from_phase, PhaseRows rows) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    const int pairs = (count + 1) / 2;
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    if (from_phase) {
        const LoadPhaseStepPair ld = phase_pair_load(x, n, count, (int)e.desc().pass[0].in_l, rows);
        fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), pairs, ld, st0, s);
    } else {
        LoadRealPair ld{x, (int)n, count};
        fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), pairs, ld, st0, s);
    }
    middle_passes(e, 1, np - 2, tmp, pairs, s);
    fftk::LoadPlainT<false> ldl{tmp};
    if (keep >= 0 || keep == kKeepLowerHalf) {
        StorePruned stl{U, (int)n, keep};
        fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), pairs, ldl, stl, s);
    } else {
        fftk::StorePlainT<false> stl{U, 1.0f};
        fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), pairs, ldl, stl, s);
    }
}

void fused_hilbert_pair_ifft_mix(const FftEngine& e, const float2* U, const float* m, float2* u, float2* tmp,
                                 int count, hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    with_hilbert_pair_load(e, U, [&](auto ld) {
        fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), count, ld, st0, s);
    });
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
    StoreStereoMix stl{m, u};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), count, ldl, stl, s);
}

bool fused_hilbert_pair_ifft_mix_fft(const FftEngine& ei, const FftEngine& ef, const float2* U, const float* m,
                                     float2* tmp_i, float2* tmp_f, int count, hipStream_t s) {
    if (count <= 0) return true;
    if (ei.npass() != 2 || ef.npass() != 2 || ei.desc().n != ef.desc().n) return false;
    const int64_t n = ei.desc().n;
    const FftPassDev d1 = ei.pass_dev(1, ei.tmp_stride(), n);
    const FftPassDev d2 = ef.pass_dev(0, n, ef.tmp_stride());
    if (!fftk::fft_tile2_applies(d1, d2, count)) return false;   // decided before anything is launched
    fftk::StorePlainT<false> st0{tmp_i, 1.0f};
    with_hilbert_pair_load(ei, U, [&](auto ld) {
        fftk::launch_fft_pass<kStridedOnly>(ei.pass_dev(0, 0, ei.tmp_stride()), count, ld, st0, s);
    });
    fftk::LoadPlainT<false> ldl{tmp_i};
    MidStereoMix mid{m};
    fftk::StorePlainT<false> st1{tmp_f, 1.0f};
    RC_REQUIRE(fftk::launch_fft_tile2(d1, d2, count, ldl, mid, st1, s), RCFM_ERR_RUNTIME,
               "two-transform tile kernel refused a pair it should accept");
    return true;
}

bool fused_hilbert_packed_applies(const FftEngine& ei, const FftEngine& ef, int count) {
    if (ei.npass() != 2 || ef.npass() != 2 || ei.desc().n != ef.desc().n) return false;
    const int64_t n = ei.desc().n;
    return fftk::fft_tile2_applies(ei.pass_dev(1, ei.tmp_stride(), n), ef.pass_dev(0, n, ef.tmp_stride()),
                                   (count + 1) / 2);
}

bool fused_hilbert_packed_ifft_mix_fft(const FftEngine& ei, const FftEngine& ef, const float2* U2, const float* p,
                                       const float* m, float2* tmp_i, float2* tmp_f, int count, hipStream_t s) {
    if (count <= 0) return true;
    if (ei.npass() != 2 || ef.npass() != 2 || ei.desc().n != ef.desc().n) return false;
    const int64_t n = ei.desc().n;
    const int pairs = (count + 1) / 2;
    const FftPassDev d1 = ei.pass_dev(1, ei.tmp_stride(), n);
    const FftPassDev d2 = ef.pass_dev(0, n, ef.tmp_stride());
    if (!fftk::fft_tile2_applies(d1, d2, pairs)) return false;   // decided before anything is launched
    fftk::StorePlainT<false> st0{tmp_i, 1.0f};
    const FftPass& p0 = ei.desc().pass[0];
    const float scale = (float)(1.0 / (double)n);
    if (!p0.load_along_l && p0.in_l * p0.L == n && p0.L % 2 == 0)
        fftk::launch_fft_pass<kStridedOnly>(ei.pass_dev(0, 0, ei.tmp_stride()), pairs,
                                            LoadHilbertPackedT<true>{U2, (int)n, (int)p0.in_l, scale}, st0, s);
    else
        fftk::launch_fft_pass<kStridedOnly>(ei.pass_dev(0, 0, ei.tmp_stride()), pairs,
                                            LoadHilbertPackedT<false>{U2, (int)n, (int)p0.in_l, scale}, st0, s);
    fftk::LoadPlainT<false> ldl{tmp_i};
    MidStereoMixPair mid{p, m};
    fftk::StorePlainT<false> st1{tmp_f, 1.0f};
    RC_REQUIRE(fftk::launch_fft_tile2_pair(d1, d2, count, ldl, mid, st1, s), RCFM_ERR_RUNTIME,
               "two-transform pair kernel refused a pair it should accept");
    return true;
}

bool fused_pilot_chain_applies(const FftEngine& ef, const FftEngine& ei, int count) {
    if (ei.npass() != 2 || ef.npass() != 2 || ei.desc().n != ef.desc().n) return false;
    const int64_t n = ef.desc().n;
    const int pairs = (count + 1) / 2;
    return fftk::fft_tile2_applies(ef.pass_dev(1, ef.tmp_stride(), n), ei.pass_dev(0, n, ei.tmp_stride()), pairs) &&
           fftk::fft_tile2_applies(ei.pass_dev(1, ei.tmp_stride(), n), ef.pass_dev(0, n, ef.tmp_stride()), pairs);
}

bool fused_pilot_chain_geometry(const FftEngine& ef, int64_t* rows, int64_t* row_length) {
    if (ef.npass() != 2) return false;
    const FftPass& p0 = ef.desc().pass[0];
    if (p0.load_along_l || p0.in_i != 1 || p0.n_o1 != 1 || p0.n_o2 != 1 || (int64_t)p0.L * p0.in_l != ef.desc().n ||
        p0.n_inner != p0.in_l)
        return false;
    *rows = p0.L;
    *row_length = p0.in_l;
    return true;
}

void fused_pilot_chain_fft_first(const FftEngine& ef, const float* p, float2* tmp_f, int count, hipStream_t s,
                                 int64_t blk_stride, int blk16) {
    if (count <= 0) return;
    const int64_t n = ef.desc().n;
    LoadRealPair ld{p, (int)n, count, blk_stride, blk16};
    fftk::StorePlainT<false> st0{tmp_f, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(ef.pass_dev(0, 0, ef.tmp_stride()), (count + 1) / 2, ld, st0, s);
}

void fused_pilot_chain_mask_mix(const FftEngine& ef, const FftEngine& ei, const float* p, const float* m,
                                float2* tmp_f, float2* tmp_i, int count, hipStream_t s, int64_t blk_stride, int blk16) {
    if (count <= 0) return;
    const int64_t n = ef.desc().n;
    const int pairs = (count + 1) / 2;
    {   // pair FFT last pass -> mask -> inverse FFT first pass: the pair spectrum never reaches memory
        const FftPassDev d1 = ef.pass_dev(1, ef.tmp_stride(), n);
        const FftPassDev d2 = ei.pass_dev(0, n, ei.tmp_stride());
        fftk::LoadPlainT<false> ld{tmp_f};
        fftk::StorePlainT<false> st{tmp_i, 1.0f};
        const float scale = (float)(1.0 / (double)n);
        bool ok;
        if (d1.p.L % 2 == 0 && (int64_t)d1.p.L * d1.p.out_k == n)
            ok = fftk::launch_fft_tile2(d1, d2, pairs, ld, MidHilbertMaskT<true>{(int)n, (int)d1.p.out_k, scale}, st, s);
        else
            ok = fftk::launch_fft_tile2(d1, d2, pairs, ld, MidHilbertMaskT<false>{(int)n, (int)d1.p.out_k, scale}, st, s);
        RC_REQUIRE(ok, RCFM_ERR_RUNTIME, "two-transform tile kernel refused a pair it should accept");
    }
    {   // inverse FFT last pass -> split + stereo mix -> packed L/R FFT first pass, per member of the pair
        const FftPassDev d1 = ei.pass_dev(1, ei.tmp_stride(), n);
        const FftPassDev d2 = ef.pass_dev(0, n, ef.tmp_stride());
        fftk::LoadPlainT<false> ld{tmp_i};
        MidStereoMixPair mid{p, m, blk_stride, blk16};
        RC_REQUIRE(!blk16 || (blk16 >= 16 * d1.p.L && (int64_t)d1.p.L * d1.p.out_k == n), RCFM_ERR_RUNTIME,
                   "tile-blocked pilot layout does not match the inverse transform's last pass");
        fftk::StorePlainT<false> st{tmp_f, 1.0f};
        RC_REQUIRE(fftk::launch_fft_tile2_pair(d1, d2, count, ld, mid, st, s), RCFM_ERR_RUNTIME,
                   "two-transform pair kernel refused a pair it should accept");
    }
}

void fused_fft_last_pruned(const FftEngine& e, const float2* tmp, float2* out, int count, int keep, hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    fftk::LoadPlainT<false> ldl{tmp};
    StorePruned stl{out, (int)n, keep};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), count, ldl, stl, s);
}

static bool decim_rt(const FftPassDev& d1, const FftPassDev& d2, int batch, const float2* tmp_f, const WinAudioDecim& win,
                     float2* tmp_a, hipStream_t s) {
    return launch_fft_tile2_decim_rt(d1, d2, batch, tmp_f, win, tmp_a, s);
}

bool fused_fft_decim_ifft_applies(const FftEngine& ef, const FftEngine& ea, int count) {
    if (ef.npass() != 2 || ea.npass() != 2) return false;
    const int64_t B = ef.desc().n, A = ea.desc().n;
    if (A >= B || (A & 1) || ea.desc().pass[1].L != ef.desc().pass[0].L) return false;
    const FftPassDev d1 = ef.pass_dev(1, ef.tmp_stride(), B), d2 = ea.pass_dev(0, A, ea.tmp_stride());
    return fftk::fft_tile2_decim_applies(d1, d2, count) || fftk::fft_tile2_decim_rt_applies(d1, d2, count);
}

void fused_fft_decim_ifft(const FftEngine& ef, const FftEngine& ea, const float2* tmp_f, float2* out, float2* tmp_a,
                          int count, const float* wr, float scale, float2* dc, hipStream_t s, int out_pitch) {
    if (count <= 0) return;
    const int64_t B = ef.desc().n, A = ea.desc().n;
    fftk::LoadPlainT<false> ld{tmp_f};
    WinAudioDecim win{wr, dc, scale, (int)A, ef.desc().pass[0].L, 0};
    fftk::StorePlainT<false> st{tmp_a, 1.0f};
    {   // the instantiated (L, L2) pairs first, then the kernel that takes L2 at run time
        const FftPassDev d1 = ef.pass_dev(1, ef.tmp_stride(), B), d2 = ea.pass_dev(0, A, ea.tmp_stride());
        RC_REQUIRE(fftk::launch_fft_tile2_decim(d1, d2, count, ld, win, st, s) || decim_rt(d1, d2, count, tmp_f, win, tmp_a, s),
                   RCFM_ERR_RUNTIME, "decimating two-transform kernel refused a pair it should accept");
    }
    fftk::LoadPlainT<false> ldl{tmp_a};
    fftk::StorePlainT<true> stl{out, 1.0f};
    FftPassDev last = ea.pass_dev(1, ea.tmp_stride(), A);
    if (out_pitch > 0) {   // rows of n_1 samples at a pitch of whole 128-byte lines: aligned 16-sample segments
        RC_REQUIRE(out_pitch >= last.p.out_k && last.p.n_o1 * last.p.n_o2 == 1, RCFM_ERR_RUNTIME, "bad audio row pitch");
        last.out_batch = (A / last.p.out_k) * (int64_t)out_pitch;
        last.p.out_k = out_pitch;
    }
    fftk::launch_fft_pass<kRowsOnly>(last, count, ldl, stl, s);
}

// Last pass of an inverse transform that carried two real signals: real part -> channel 2P, imaginary
// part -> channel 2P+1 of y [count][n] (v arrives with re/im exchanged).
struct StoreRealImagSplit {
    float* y;
    int n, count;
    __device__ __forceinline__ void operator()(const LineId& id, int, int64_t base, unsigned off, float2 v) const {
        const int64_t t = base + off - (int64_t)id.batch * n;          // sample index
        const int c0 = 2 * id.batch;
        y[(int64_t)c0 * n + t] = v.y;
        if (c0 + 1 < count) y[(int64_t)(c0 + 1) * n + t] = v.x;
    }
};

void fused_real_pair_fft_first(const FftEngine& e, const float* x, float2* tmp, int count, bool from_phase,
                               hipStream_t s, PhaseRows rows) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int pairs = (count + 1) / 2;
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    if (from_phase) {
        const LoadPhaseStepPair ld = phase_pair_load(x, n, count, (int)e.desc().pass[0].in_l, rows);
        fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), pairs, ld, st0, s);
    } else {
        LoadRealPair ld{x, (int)n, count};
        fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), pairs, ld, st0, s);
    }
}

void fused_fft_decim_ifft_pairs(const FftEngine& ef, const FftEngine& ea, const float2* tmp_f, float* y,
                                float2* tmp_a, int count, const float* wr, float scale, float2* dc, hipStream_t s) {
    if (count <= 0) return;
    const int64_t B = ef.desc().n, A = ea.desc().n;
    const int pairs = (count + 1) / 2;
    fftk::LoadPlainT<false> ld{tmp_f};
    WinAudioDecim win{wr, dc, scale, (int)A, ef.desc().pass[0].L, count};
    fftk::StorePlainT<false> st{tmp_a, 1.0f};
    {
        const FftPassDev d1 = ef.pass_dev(1, ef.tmp_stride(), B), d2 = ea.pass_dev(0, A, ea.tmp_stride());
        RC_REQUIRE(fftk::launch_fft_tile2_decim(d1, d2, pairs, ld, win, st, s) || decim_rt(d1, d2, pairs, tmp_f, win, tmp_a, s),
                   RCFM_ERR_RUNTIME, "decimating two-transform kernel refused a pair it should accept");
    }
    fftk::LoadPlainT<false> ldl{tmp_a};
    StoreRealImagSplit stl{y, (int)A, count};
    fftk::launch_fft_pass<kRowsOnly>(ea.pass_dev(1, ea.tmp_stride(), A), pairs, ldl, stl, s);
}

void fused_stereo_unpack_ifft(const FftEngine& e, const float2* U, int64_t B, float2* out, float2* tmp, int count,
                              const float* wr, int nyq, int nmin, float nyq_factor, float scale, float2* dc,
                              hipStream_t s) {
    if (count <= 0) return;
    const int64_t A = e.desc().n;
    const int np = e.npass();
    LoadStereoUnpack ld{U, wr, dc, (int)B, (int)A, nyq, nmin, nyq_factor, scale, (int)e.desc().pass[0].in_l};
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), count, ld, st0, s);
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
    fftk::StorePlainT<true> stl{out, 1.0f};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), A), count, ldl, stl, s);
}

void fused_fft_pruned(const FftEngine& e, const float2* in, float2* out, float2* tmp, int count, int keep,
                      hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    fftk::LoadPlainT<false> ld{in};
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, n, e.tmp_stride()), count, ld, st0, s);
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
    StorePruned stl{out, (int)n, keep};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), count, ldl, stl, s);
}

void fused_ifft_real_out(const FftEngine& e, const float2* Y, float* y, float2* tmp, int count, float scale,
                         hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    fftk::LoadPlainT<true> ld{Y};
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, n, e.tmp_stride()), count, ld, st0, s);
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
    StoreRealPart stl{y, scale};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), count, ldl, stl, s);
}

RCFM_NS_CLOSE
}  // namespace rcfm
