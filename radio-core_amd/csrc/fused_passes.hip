#include "fused_passes.h"

#include "fft_kernel.h"
#include "kernels.h"

namespace rcfm {

using fftk::LineId;
using fftk::kAnyPass;
using fftk::kRowsOnly;
using fftk::kStridedOnly;

namespace {

// ---- functors -------------------------------------------------------------------------
// LoadOp::operator() fetches only; LoadOp::post does the arithmetic when the tile is consumed.

// First pass of Tuner.run's inverse FFT: element k of the channel spectrum comes from
// bin (src - roll) mod N of the wideband spectrum, src = k (k < nyq) or N - (B - k).
struct LoadTunerGather {
    const float2* X;
    const int64_t* roll;
    int64_t N;
    float two_pi_over_n, delta;
    float a0;
    int B, nyq, nneg, nyq_mode;
    int64_t line_stride;   // in_l of the pass: k = l * line_stride + i

    __device__ __forceinline__ int64_t source_bin(int k) const {
        if (k < nyq) return k;
        const int j = B - k;
        if (j <= nneg) return N - j;
        if (nyq_mode == NYQ_UP && j == nyq - 1) return nyq - 1;   // Y[-N/2] = Y[+N/2] / 2
        return -1;
    }
    __device__ __forceinline__ int64_t rolled(int64_t src, int64_t r) const {
        int64_t i = src - r;
        return i < 0 ? i + N : i;
    }
    // fftshift(get_window(...))[src] = a0 - (1 - a0) cos(2 pi ((src - N//2) mod N) / N)
    //                                = a0 + (1 - a0) cos(2 pi d / N + delta),  d = src or src - N (signed,
    // |d| <= N/2), delta = pi / N for odd N.  In the hot case |theta| < 0.06: a 4-term series is exact
    // to 1e-14; otherwise the library cosine.
    __device__ __forceinline__ float window(int64_t src) const {
        const int64_t dsrc = (src <= N / 2) ? src : src - N;
        const float th = (float)dsrc * two_pi_over_n + delta;
        const float t2 = th * th;
        const float series = 1.f - t2 * 0.5f * (1.f - t2 * (1.f / 12.f) * (1.f - t2 * (1.f / 30.f)));
        const float c = (fabsf(th) < 0.06f) ? series : cosf(th);
        return a0 + (1.f - a0) * c;
    }
    __device__ __forceinline__ float2 fetch(const LineId& id, int, int64_t base, unsigned off) const {
        const int64_t src = source_bin((int)(base + off));   // in_batch = 0: base + off = bin inside the channel
        return X[rolled(src < 0 ? 0 : src, roll[id.batch])];
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 v) const {
        const int k = (int)(l * line_stride + id.i);
        const int64_t src = source_bin(k);
        float w = src < 0 ? 0.f : window(src);
        const int half = nyq - 1;
        if (k == half && nyq_mode == NYQ_UP) w *= 0.5f;
        if (k != half && (B - k) == half && nyq_mode == NYQ_UP) w *= 0.5f;
        float2 y = make_float2(v.x * w, v.y * w);
        if (k == half && nyq_mode == NYQ_DOWN) {   // Y[+N/2] += X[-N/2]: one element per channel
            const int64_t s2 = N - half;
            const float2 x2 = X[rolled(s2, roll[id.batch])];
            const float w2 = window(s2);
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
    __device__ __forceinline__ float2 fetch(const LineId& id, int, int64_t base, unsigned off) const {
        // in_batch = 0: base + off = sample index inside the signal
        const int c0 = 2 * id.batch, c1 = (c0 + 1 < count) ? c0 + 1 : c0;
        const int64_t a = base + off;
        return make_float2(x[(int64_t)c0 * n + a], x[(int64_t)c1 * n + a]);
    }
    __device__ __forceinline__ float2 post(const LineId&, int, float2 v) const { return v; }
};

// Hilbert mask applied to one member of such a pair: with U = FFT(x0 + j x1),
// X0[k] = (U[k] + conj U[-k]) / 2, X1[k] = (U[k] - conj U[-k]) / 2j; Z = h X, bins above n/2 are 0.
struct LoadHilbertPair {
    static constexpr int kFetches = 2;
    const float2* U;     // [ceil(count / 2)][n]
    int n;
    int64_t line_stride;
    __device__ __forceinline__ int bin(const LineId&, int64_t base, unsigned off) const {
        return (int)(base + off);   // in_batch = 0
    }
    __device__ __forceinline__ float2 fetch(const LineId& id, int, int64_t base, unsigned off) const {
        const int k = bin(id, base, off);
        return U[(int64_t)(id.batch >> 1) * n + (k <= n / 2 ? k : 0)];
    }
    __device__ __forceinline__ float2 fetch2(const LineId& id, int, int64_t base, unsigned off) const {
        const int k = bin(id, base, off);
        return U[(int64_t)(id.batch >> 1) * n + ((k <= n / 2 && k > 0) ? n - k : 0)];
    }
    __device__ __forceinline__ float2 post(const LineId& id, int l, float2 a, float2 b) const {
        const int k = (int)(l * line_stride + id.i);
        float h = 0.f;
        if (k == 0) h = 1.f;
        else if (k < (n + 1) / 2) h = 2.f;
        else if ((n & 1) == 0 && k == n / 2) h = 1.f;
        h *= 0.5f;
        float2 xk;
        if ((id.batch & 1) == 0) xk = make_float2(a.x + b.x, a.y - b.y);          // U[k] + conj U[-k]
        else xk = make_float2(a.y + b.y, -(a.x - b.x));                           // (U[k] - conj U[-k]) / j
        return make_float2(xk.y * h, xk.x * h);   // swapped: inverse transform
    }
};

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
        // z = (v.y, v.x); Im(z^2)/|z^2| = 2 ab / (a^2 + b^2), pre-scaled against underflow;
        // z == 0 gives NaN like pll.py:57-58
        const float inv = __frcp_rn(fmaxf(fabsf(v.x), fabsf(v.y)));
        const float za = v.y * inv, zb = v.x * inv;
        const float s2 = (2.f * za * zb) * __frcp_rn(za * za + zb * zb);
        const float lmr = (s2 * mm) * 1.0175f;
        fftk::stream_store(u + base + off, make_float2(mm + lmr, mm - lmr));
    }
};

// The same mix as the point-wise stage between two transforms (k_fft_tile2).
struct MidStereoMix {
    static constexpr bool kAux = true;
    const float* m;
    __device__ __forceinline__ float fetch_aux(const LineId&, int, int64_t base, unsigned off) const {
        return (m + base)[off];
    }
    __device__ __forceinline__ float2 operator()(const LineId&, int, float2 v, float mm) const {
        const float inv = __frcp_rn(fmaxf(fabsf(v.x), fabsf(v.y)));
        const float za = v.y * inv, zb = v.x * inv;           // z = (v.y, v.x): swap identity
        const float s2 = (2.f * za * zb) * __frcp_rn(za * za + zb * zb);
        const float lmr = (s2 * mm) * 1.0175f;
        return make_float2(mm + lmr, mm - lmr);
    }
};

struct StorePruned {
    float2* out;
    int n, keep;
    __device__ __forceinline__ void operator()(const LineId& id, int, int64_t base, unsigned off, float2 v) const {
        const int k = (int)(base + off - (int64_t)id.batch * n);
        if (keep < 0 || k <= keep || k >= n - keep) fftk::stream_store(out + base + off, v);
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

void fused_tuner_ifft(const FftEngine& e, const TunerGather& g, float2* out, float2* tmp, int count,
                      hipStream_t s) {
    if (count <= 0) return;
    const int64_t B = e.desc().n;
    const int np = e.npass();
    LoadTunerGather ld;
    ld.X = g.X;
    ld.roll = g.roll;
    ld.N = g.N;
    ld.two_pi_over_n = (float)(6.28318530717958647692 / (double)g.N);
    ld.delta = (g.N % 2) ? (float)(3.14159265358979323846 / (double)g.N) : 0.f;
    ld.a0 = (float)g.a0;
    ld.B = (int)B;
    ld.nyq = g.nyq;
    ld.nneg = g.nneg;
    ld.nyq_mode = g.nyq_mode;
    ld.line_stride = e.desc().pass[0].in_l;
    // in_batch = 0: the load functor addresses X itself; `a` is the bin index inside the channel
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), count, ld, st0, s);
    middle_passes(e, 1, np - 2, tmp, count, s);
    fftk::LoadPlainT<false> ldl{tmp};
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

void fused_real_pair_fft(const FftEngine& e, const float* x, float2* U, float2* tmp, int count, int keep,
                         hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    const int pairs = (count + 1) / 2;
    LoadRealPair ld{x, (int)n, count};
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), pairs, ld, st0, s);
    middle_passes(e, 1, np - 2, tmp, pairs, s);
    fftk::LoadPlainT<false> ldl{tmp};
    if (keep >= 0) {
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
    LoadHilbertPair ld{U, (int)n, e.desc().pass[0].in_l};
    fftk::StorePlainT<false> st0{tmp, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(e.pass_dev(0, 0, e.tmp_stride()), count, ld, st0, s);
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
    LoadHilbertPair ld{U, (int)n, ei.desc().pass[0].in_l};
    fftk::StorePlainT<false> st0{tmp_i, 1.0f};
    fftk::launch_fft_pass<kStridedOnly>(ei.pass_dev(0, 0, ei.tmp_stride()), count, ld, st0, s);
    fftk::LoadPlainT<false> ldl{tmp_i};
    MidStereoMix mid{m};
    fftk::StorePlainT<false> st1{tmp_f, 1.0f};
    RC_REQUIRE(fftk::launch_fft_tile2(d1, d2, count, ldl, mid, st1, s), RCFM_ERR_RUNTIME,
               "two-transform tile kernel refused a pair it should accept");
    return true;
}

void fused_fft_last_pruned(const FftEngine& e, const float2* tmp, float2* out, int count, int keep, hipStream_t s) {
    if (count <= 0) return;
    const int64_t n = e.desc().n;
    const int np = e.npass();
    fftk::LoadPlainT<false> ldl{tmp};
    StorePruned stl{out, (int)n, keep};
    fftk::launch_fft_pass<kRowsOnly>(e.pass_dev(np - 1, e.tmp_stride(), n), count, ldl, stl, s);
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

}  // namespace rcfm
