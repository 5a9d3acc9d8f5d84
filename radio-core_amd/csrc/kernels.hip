// Hand-written gfx950 kernels of the Tuner -> FM / MFM / WBFM path.
//
// Everything here is HBM/L2-streaming work (no dense contraction, so no MFMA):
// one thread per output element with coalesced 4/8-byte accesses, LDS tiles with
// halos for the two FIR stages, wave64 shuffle reductions for the DC sums.
// Reference call sites are cited per kernel; the algebra (closed forms of the
// scipy calls) is derived in DESIGN.md and pinned by tests/test_oracle_scipy.py.

#include "kernels.h"

#include <cmath>
#include "device_math.h"

namespace rcfm {

namespace {

constexpr int kThreads = 256;
constexpr float kInvPi = 0.31830988618379067154f;

__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

inline dim3 grid2(int64_t items, int per_block, int batch) {
    return dim3((unsigned)((items + per_block - 1) / per_block), (unsigned)batch, 1);
}

// ---------------------------------------------------------------------------
// Spectral resampling (scipy.signal.resample, spectrum side)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads) void k_spectrum_c2c(
    const float2* __restrict__ X, int64_t x_stride, int64_t n, const int64_t* __restrict__ roll,
    float2* __restrict__ Y, int64_t m, const float* __restrict__ wpos, const float* __restrict__ wneg,
    float w_merge, int nyq, int nneg, int nyq_mode, float scale) {
    const int c = blockIdx.y;
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= m) return;
    const float2* Xc = X + (int64_t)c * x_stride;
    const int64_t r = roll ? roll[c] : 0;
    // np.roll(X, r)[src] == X[(src - r) mod n]            (tuner.py:159)
    auto fetch = [&](int64_t src) {
        int64_t i = src - r;
        if (i < 0) i += n;
        return Xc[i];
    };
    const int64_t half = nyq - 1;  // min(n, m) / 2
    float2 y = make_float2(0.f, 0.f);
    if (k < nyq) {
        y = cscale(fetch(k), wpos[k]);
        if (k == half && nyq_mode == NYQ_DOWN) {  // Y[+N/2] += X[-N/2]
            float2 v = cscale(fetch(n - half), w_merge);
            y.x += v.x;
            y.y += v.y;
        } else if (k == half && nyq_mode == NYQ_UP) {
            y = cscale(y, 0.5f);
        }
    } else {
        const int64_t j = m - k;
        if (j <= nneg) {
            y = cscale(fetch(n - j), wneg[j]);
        } else if (nyq_mode == NYQ_UP && j == half) {  // Y[-N/2] = Y[+N/2]
            y = cscale(fetch(half), 0.5f * wpos[half]);
        }
    }
    Y[(int64_t)c * m + k] = cscale(y, scale);
}

__global__ __launch_bounds__(kThreads) void k_spectrum_r2c(const float2* __restrict__ X, int64_t n,
                                                           float2* __restrict__ Y, int64_t m,
                                                           const float* __restrict__ wr, int nyq, int nmin,
                                                           float nyq_factor, float scale) {
    const int c = blockIdx.y;
    const int64_t mh = m / 2 + 1;
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= mh) return;
    float2 y = make_float2(0.f, 0.f);
    if (k < nyq) {
        y = cscale(X[(int64_t)c * (n / 2 + 1) + k], wr[k]);
        if ((nmin & 1) == 0 && k == nmin / 2) y = cscale(y, nyq_factor);
    }
    // A real inverse transform never sees the imaginary part of DC / Nyquist
    // (pocketfft's c2r drops it); make that explicit for any backend.
    if (k == 0 || ((m & 1) == 0 && k == m / 2)) y.y = 0.f;
    Y[(int64_t)c * mh + k] = cscale(y, scale);
}

// paired != 0: X holds FFT(x[2p] + j x[2p+1]) per pair p (two real signals per transform); channel c
// takes X_c[k] = (U[k] + conj U[-k]) / 2 (even c) or (U[k] - conj U[-k]) / 2j (odd c).
__global__ __launch_bounds__(kThreads) void k_spectrum_real_full(const float2* __restrict__ X, int64_t n,
                                                                 float2* __restrict__ Y, int64_t m,
                                                                 const float* __restrict__ wr, int nyq, int nmin,
                                                                 float nyq_factor, float scale,
                                                                 float2* __restrict__ dc, int paired) {
    const int c = blockIdx.y;
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= m) return;
    const bool mirrored = k > m / 2;
    const int64_t kk = mirrored ? m - k : k;
    float2 y = make_float2(0.f, 0.f);
    if (kk < nyq) {
        float2 xk;
        if (paired) {
            const float2* U = X + (int64_t)(c >> 1) * n;
            const float2 a = U[kk], b = U[kk == 0 ? 0 : n - kk];
            xk = (c & 1) ? make_float2(0.5f * (a.y + b.y), -0.5f * (a.x - b.x))
                         : make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
        } else {
            xk = X[(int64_t)c * n + kk];
        }
        y = cscale(xk, wr[kk] * scale);
        if ((nmin & 1) == 0 && kk == nmin / 2) y = cscale(y, nyq_factor);
        if (kk == 0 || ((m & 1) == 0 && kk == m / 2)) y.y = 0.f;
    }
    if (mirrored) y.y = -y.y;
    Y[(int64_t)c * m + k] = y;
    if (k == 0 && dc != nullptr) dc[c] = y;
}

__global__ __launch_bounds__(kThreads) void k_hilbert_mask(const float2* __restrict__ P,
                                                           float2* __restrict__ Z, int64_t n, float scale) {
    const int c = blockIdx.y;
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= n) return;
    const float2* Pc = P + (int64_t)c * (n / 2 + 1);
    float2 v = make_float2(0.f, 0.f);
    if (k == 0) {
        v = Pc[0];
    } else if (k < (n + 1) / 2) {
        v = cscale(Pc[k], 2.f);
    } else if ((n & 1) == 0 && k == n / 2) {
        v = Pc[k];
    }
    Z[(int64_t)c * n + k] = cscale(v, scale);
}

__global__ __launch_bounds__(kThreads) void k_stereo_unpack(const float2* __restrict__ U, int64_t B,
                                                            float2* __restrict__ V, int64_t A,
                                                            const float* __restrict__ wr, int nyq, int nmin,
                                                            float nyq_factor, float scale,
                                                            float2* __restrict__ dc) {
    const int c = blockIdx.y;
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= A) return;
    const float2* Uc = U + (int64_t)c * B;
    const bool mirrored = k > A / 2;
    const int64_t kk = mirrored ? A - k : k;
    float2 hl = make_float2(0.f, 0.f), hr = make_float2(0.f, 0.f);
    if (kk < nyq) {
        // u = l + j r with l, r real  =>  L[k] = (U[k] + conj U[-k]) / 2,  R[k] = (U[k] - conj U[-k]) / 2j
        const float2 a = Uc[kk];
        const float2 b = Uc[kk == 0 ? 0 : B - kk];
        float w = wr[kk] * scale;
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
    // packed Hermitian pair: V = HL + j HR  ->  ifft(V) = l + j r
    const float2 out = make_float2(hl.x - hr.y, hl.y + hr.x);
    V[(int64_t)c * A + k] = out;
    if (k == 0 && dc != nullptr) dc[c] = out;   // sum_n l[n] = A Re V[0], sum_n r[n] = A Im V[0]
}

// ---------------------------------------------------------------------------
// Discriminator and the fused WBFM front end
// ---------------------------------------------------------------------------

__device__ __forceinline__ float phase_step(float2 a, float2 b) {
    // arg(a conj(b)) / pi == diff(unwrap(angle(x))) / pi without the float32 unwrap noise
    return atan2f(a.y * b.x - a.x * b.y, a.x * b.x + a.y * b.y) * kInvPi;
}

__global__ __launch_bounds__(kThreads) void k_discriminator(const float2* __restrict__ iq,
                                                            float* __restrict__ d, int64_t n) {
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float2* xc = iq + (int64_t)c * n;
    d[(int64_t)c * n + i] = (i == 0) ? 0.f : phase_step(xc[i], xc[i - 1]);
}

constexpr int kPilotTile = 1024;

__global__ __launch_bounds__(kThreads) void k_pilot_stage(const float2* __restrict__ iq,
                                                          const float* __restrict__ x,
                                                          float* __restrict__ m_out,
                                                          float* __restrict__ p_out, int64_t n,
                                                          const float* __restrict__ g, int H, float side_tap) {
    constexpr int T = kPilotTile;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* g_s = smem;                 // H + 1 taps, g_s[0] = centre
    float* m_s = g_s + (H + 1);        // T + 2H  : m[q0 + s]
    float* d_s = m_s + (T + 2 * H);    // T + 2H + 2 : d[q0 - 1 + s] (circular)
    const int c = blockIdx.y;
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * T;
    const int64_t q0 = t0 - H;
    for (int i = tid; i <= H; i += kThreads) g_s[i] = g[i];

    if (iq != nullptr) {
        const float2* xc = iq + (int64_t)c * n;
        for (int s = tid; s < T + 2 * H + 2; s += kThreads) {
            int64_t i = q0 - 1 + s;
            float v = 0.f;
            if (i >= -1 && i <= n) {
                if (i == -1) i = n - 1;   // the same-size Decimate is circular (decimate.py:48)
                if (i == n) i = 0;
                if (i > 0) v = phase_step(xc[i], xc[i - 1]);   // d[0] = 0 (fm.py:64)
            }
            d_s[s] = v;
        }
        __syncthreads();
        for (int s = tid; s < T + 2 * H; s += kThreads) {
            const int64_t q = q0 + s;
            float v = 0.f;
            if (q >= 0 && q < n) {
                v = 0.54f * d_s[s + 1] + side_tap * (d_s[s] + d_s[s + 2]);
                if (q >= t0 && q < t0 + T) m_out[(int64_t)c * n + q] = v;
            }
            m_s[s] = v;
        }
    } else {
        const float* xc = x + (int64_t)c * n;
        for (int s = tid; s < T + 2 * H; s += kThreads) {
            const int64_t q = q0 + s;
            m_s[s] = (q >= 0 && q < n) ? xc[q] : 0.f;
        }
    }
    __syncthreads();

    // filtfilt(b, 1, m) == sum_j g[|j|] e[i + j], e = m odd-extended at both ends.
    const int64_t last = n - 1;
    for (int o = tid; o < T; o += kThreads) {
        const int64_t i = t0 + o;
        if (i >= n) break;
        const int base = o + H;
        float acc = g_s[0] * m_s[base];
        if (i - H >= 0 && i + H <= last) {
            for (int j = 1; j <= H; ++j) acc = fmaf(g_s[j], m_s[base - j] + m_s[base + j], acc);
        } else {
            const float m_first = (q0 <= 0) ? m_s[0 - q0] : 0.f;
            const float m_last = (last - q0 < T + 2 * H) ? m_s[last - q0] : 0.f;
            for (int j = 1; j <= H; ++j) {
                const int64_t ql = i - j, qr = i + j;
                const float el = (ql < 0) ? 2.f * m_first - m_s[-ql - q0] : m_s[ql - q0];
                const float er = (qr > last) ? 2.f * m_last - m_s[2 * last - qr - q0] : m_s[qr - q0];
                acc = fmaf(g_s[j], el + er, acc);
            }
        }
        p_out[(int64_t)c * n + i] = acc;
    }
}

// Register-blocked version for the 41-tap pilot filter (H = 40), the WBFM hot path.
// Each thread produces 8 consecutive outputs from an 88-sample window held in registers
// (22 ds_read_b128 instead of 648 ds_read_b32); the 41 distinct taps arrive as kernel
// arguments (SGPRs).  Blocks that touch either end of the buffer take the reflecting path.
// The 81-tap zero-phase kernel h[t] = g[|t - 40|] as 41 pairs (h[2j], h[2j+1]), h[81] = 0 (kernel
// arguments: they reach the FIR as SGPR pairs).
typedef float v2f __attribute__((ext_vector_type(2)));
struct PilotTaps {
    v2f pair[41];
};

constexpr int kPilotPer = 8;
constexpr int kPilotFastTile = kThreads * kPilotPer;   // 2048 outputs per workgroup
constexpr int kPilotBlockedThreads = kThreads;         // tile-blocked output (PilotBlocked): whole rows per workgroup
constexpr int kPilotBlockedTile = kPilotBlockedThreads * kPilotPer;
// acc += taps (.) w and acc += reverse(taps) (.) w, taps in an SGPR pair: the halves are picked by
// op_sel, so the symmetric kernel serves even and odd outputs from one set of aligned pairs.
__device__ __forceinline__ void pk_fma_s(v2f& acc, v2f taps, v2f w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(taps), "v"(w));
}
__device__ __forceinline__ void pk_fma_s_rev(v2f& acc, v2f taps, v2f w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(taps), "v"(w));
}

__device__ __forceinline__ float phase_step_fast(float2 a, float2 b) {
    return atan2_over_pi(a.y * b.x - a.x * b.y, a.x * b.x + a.y * b.y);
}
// the same step when the tuner already left angle(x) / pi (fused_tuner_ifft's phase output)
__device__ __forceinline__ float phase_step_fast(float th, float th_prev) { return phase_step_wrapped(th, th_prev); }

// IN = float2: complex samples; IN = float: their phases in units of pi.
// BLK: m and p leave in the tile-blocked layout of PilotBlocked (kernels.h); a workgroup then owns rpt whole rows
// (Tv = rpt R <= T outputs; the threads beyond idle through the FIR) and tiles_x = ceil(L / rpt).  (THREADS = 1024,
// 16 rows of 500 per workgroup and 1 KB runs per block, was slower still than 256 threads with their 256-byte runs:
// 0.76 against 0.64 ms, 0.57 in natural order -- profiles/r05_z_pilot_blocked.md.)
template <class IN, bool BLK, int THREADS>
__global__ __launch_bounds__(THREADS) void k_pilot_stage_h40(const IN* __restrict__ iq,
                                                              float* __restrict__ m_out,
                                                              float* __restrict__ p_out, int64_t n,
                                                              PilotTaps taps, float side_tap,
                                                              unsigned tiles_x, unsigned batch, PilotBlocked blk) {
    constexpr int H = 40, PER = kPilotPer, T = THREADS * PER;
    const int Tv = BLK ? blk.rpt * blk.R : T;   // outputs this workgroup owns
    // m[q0 + e] lives at mpos(e): 8 values, 2 pad dwords.  The FIR reads each thread's window (thread tid
    // starts at e = 8 tid) with ds_read_b64: at a lane stride of 8 dwords those were 8-way bank conflicts
    // (SQ_LDS_BANK_CONFLICT = 69 % of the LDS cycles); at 10 dwords the 64 lanes x 2 dwords spread evenly.
    auto mpos = [](int e) -> int { return (e >> 3) * 10 + (e & 7); };
    __shared__ __attribute__((aligned(16))) float m_s[((T + 2 * H + 7) / 8) * 10];
    __shared__ __attribute__((aligned(16))) float d_s[T + 2 * H + 2];    // d[q0 - 1 + s] (circular)
    // XCD-aware order (workgroups go to the 8 XCDs round-robin, each XCD has its own L2): XCD k walks
    // a contiguous eighth of the (channel, tile) list, so the halo lines two neighbouring tiles share
    // are fetched once.  1-D grid of 8 * ceil(tiles * batch / 8) workgroups.
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned vid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (vid >= tiles_x * batch) return;
    const int c = (int)(vid / tiles_x);
    const int tid = threadIdx.x;
    const int n32 = (int)n;
    const int t0 = (int)(vid - (unsigned)c * tiles_x) * Tv;
    const int q0 = t0 - H;
    const IN* xc = iq + (int64_t)c * n;
    // workgroup-uniform: the tile and its halos lie strictly inside the channel (no wrap, no reflection)
    const bool interior = (q0 - 2 >= 0) && (q0 + T + 2 * H + 1 <= n32 - 1);
    // discriminator: both samples of every pair are fetched unconditionally (clamped index) and
    // up front, 18 loads in flight per thread; a load inside the loop's `if` would serialise
    // one HBM round trip per iteration
    constexpr int ND = (T + 2 * H + 2 + THREADS - 1) / THREADS;
    IN xa[ND], xb[ND];
    if (interior) {
        const IN* x0 = xc + (q0 - 1 + tid);
#pragma unroll
        for (int it = 0; it < ND; ++it) {
            // the last sweep is ragged: clamp instead of branching (the value is not stored)
            const int off = (it == ND - 1 && tid + THREADS * it >= T + 2 * H + 2) ? 0 : THREADS * it;
            xa[it] = x0[off];
            xb[it] = x0[off - 1];
        }
#pragma unroll
        for (int it = 0; it < ND; ++it) {
            const int s = tid + THREADS * it;
            const float v = phase_step_fast(xa[it], xb[it]);
            if (it < ND - 1 || s < T + 2 * H + 2) d_s[s] = v;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < (T + 2 * H + THREADS - 1) / THREADS; ++it) {
            const int s = tid + THREADS * it;
            if (s < T + 2 * H) {
                m_s[mpos(s)] = 0.54f * d_s[s + 1] + side_tap * (d_s[s] + d_s[s + 2]);
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < ND; ++it) {
            int i = q0 - 1 + tid + THREADS * it;
            if (i == -1) i = n32 - 1;   // the same-size Decimate is circular (decimate.py:48)
            if (i == n32) i = 0;
            i = i < 1 ? 1 : (i > n32 - 1 ? n32 - 1 : i);
            xa[it] = xc[i];
            xb[it] = xc[i - 1];
        }
#pragma unroll
        for (int it = 0; it < ND; ++it) {
            const int s = tid + THREADS * it;
            const int i = q0 - 1 + s;
            // d[0] = 0 (fm.py:64); i == n wraps to d[0]; outside [-1, n] is never read
            const bool live = (i == -1) || (i > 0 && i < n32);
            const float v = live ? phase_step_fast(xa[it], xb[it]) : 0.f;
            if (s < T + 2 * H + 2) d_s[s] = v;
        }
        __syncthreads();
        for (int s = tid; s < T + 2 * H; s += THREADS) {
            const int q = q0 + s;
            float v = 0.f;
            if (q >= 0 && q < n32) {
                v = 0.54f * d_s[s + 1] + side_tap * (d_s[s] + d_s[s + 2]);
                if (!BLK && q >= t0 && q < t0 + T) m_out[(int64_t)c * n + q] = v;
            }
            m_s[mpos(s)] = v;
        }
    }
    __syncthreads();

    const int last = n32 - 1;
    const int o = tid * PER;
    // Stores leave LANE-CONTIGUOUS: a wave owns 512 consecutive outputs and each of its two store instructions writes
    // 64 x 16 bytes = eight whole 128-byte lines (lane l: outputs 4 l .. 4 l + 3 of the wave's first / second half).
    // With each thread storing its own 8 outputs as two float4s, every line was completed by two instructions and the
    // L2 had to merge the halves.
    const int wbase = (tid >> 6) * (64 * PER), lane4 = (tid & 63) * 4;
    if (!BLK && interior) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pos = wbase + h * (32 * PER) + lane4;                    // H + pos is a multiple of 4: one 8-group
            const v2f* src = reinterpret_cast<const v2f*>(&m_s[mpos(H + pos)]);
            *reinterpret_cast<float4*>(m_out + (int64_t)c * n + t0 + pos) = make_float4(src[0].x, src[0].y, src[1].x, src[1].y);
        }
    }
    // Edge tiles (scipy filtfilt, padtype="odd", bandpass.py:72): the odd reflection of m about the first / last sample is
    // written INTO the LDS window -- m[q] = 2 m[0] - m[-q] for q < 0, 2 m[last] - m[2 last - q] for last < q <= last + H --
    // and the tile then runs the same packed FIR as every other tile.  (Until round 4 the two edge tiles of a channel
    // took a scalar loop with run-time tap indexing: 55 us, invisible under 1024 channels, 40 % of a single WBFM.run.)
    const bool inner_tile = (t0 - H >= 0 && t0 + T - 1 + H <= last);   // workgroup-uniform
    const bool reflect = !inner_tile && n32 > 2 * H + 1;
    if (reflect) {
        const float m_first = (q0 <= 0) ? m_s[mpos(0 - q0)] : 0.f;
        const float m_last = (last - q0 < T + 2 * H) ? m_s[mpos(last - q0)] : 0.f;
        for (int s = tid; s < T + 2 * H; s += THREADS) {
            const int q = q0 + s;
            if (q < 0) m_s[mpos(s)] = 2.f * m_first - m_s[mpos(-q - q0)];
            else if (q > last && q <= last + H) m_s[mpos(s)] = 2.f * m_last - m_s[mpos(2 * last - q - q0)];
        }
        __syncthreads();
    }
    if (inner_tile || reflect) {
        // out[r] = sum_t h[t] w[r + t], t = 0..80, w = m_s + o, two taps per packed FMA (lane 0: the
        // even t of the pair, lane 1: the odd one).  Even r: pairs (t, t+1) = (2j, 2j+1) sit on aligned
        // window pairs j + r/2.  Odd r: t = 0 alone, then pairs (t, t+1), t = 79 - 2j odd, whose taps
        // are pair j reversed (h is symmetric) and whose window pair is 40 - j + (r-1)/2.  No operand
        // shuffling: 41 instructions per output.  The two 4-pair windows slide in opposite directions
        // and are read from LDS as they are needed (16 live registers instead of the 88 of the whole
        // window: twice the waves per SIMD, which is what hides the HBM latency of the next tile).
        auto wpair = [&](int pi) -> v2f {
            v2f q = *reinterpret_cast<const v2f*>(&m_s[mpos(o + 2 * pi)]);
            asm volatile("" : "+v"(q));   // keep the 8-byte read (and its place in the sequence)
            return q;
        };
        v2f acc[PER];
#pragma unroll
        for (int r = 0; r < PER; ++r) acc[r] = v2f{0.f, 0.f};
        v2f we[PER / 2], wo[PER / 2];
#pragma unroll
        for (int i = 0; i < PER / 2; ++i) we[i] = wpair(i);
#pragma unroll
        for (int i = 0; i < PER / 2; ++i) wo[i] = wpair(H + i);
#pragma unroll
        for (int i = 0; i < PER / 2; ++i) acc[2 * i + 1].x = taps.pair[0].x * we[i].y;   // t = 0 of odd r
#pragma unroll
        for (int j = 0; j <= H; ++j) {
            const v2f tp = taps.pair[j];
#pragma unroll
            for (int i = 0; i < PER / 2; ++i) pk_fma_s(acc[2 * i], tp, we[i]);          // pairs j + i
            if (j < H) {
#pragma unroll
                for (int i = 0; i < PER / 2; ++i) pk_fma_s_rev(acc[2 * i + 1], tp, wo[i]);   // pairs 40 - j + i
#pragma unroll
                for (int i = 0; i + 1 < PER / 2; ++i) we[i] = we[i + 1];
                we[PER / 2 - 1] = wpair(j + PER / 2);
#pragma unroll
                for (int i = PER / 2 - 1; i > 0; --i) wo[i] = wo[i - 1];
                wo[0] = wpair(H - 1 - j);
            }
        }
        if constexpr (BLK) {
            // p through d_s (dead since m_s was built), m is still in m_s.  One 128-byte line = rows k, k + 1 of a
            // 16-column block = 8 lanes x float4: slot sl -> line sl / 8 = (block, row pair), a block's row pairs being
            // adjacent lines; part sl % 8.  The index arithmetic is 32-bit with a multiply-high for the one division:
            // this stage is VALU co-bound, and two 64-bit divisions per store cost it 0.07 ms.
            float4* stage = reinterpret_cast<float4*>(&d_s[o]);
            stage[0] = make_float4(acc[0].x + acc[0].y, acc[1].x + acc[1].y, acc[2].x + acc[2].y, acc[3].x + acc[3].y);
            stage[1] = make_float4(acc[4].x + acc[4].y, acc[5].x + acc[5].y, acc[6].x + acc[6].y, acc[7].x + acc[7].y);
            __syncthreads();
            const unsigned half = (unsigned)blk.rpt >> 1, k0 = (vid - (unsigned)c * tiles_x) * (unsigned)blk.rpt;
            const unsigned nslots = half * (unsigned)blk.nb * 8u;
            float* mc = m_out + (int64_t)c * blk.stride;
            float* pc = p_out + (int64_t)c * blk.stride;
#pragma unroll
            for (int it = 0; it < 3; ++it) {   // rpt R <= 2048 outputs + the padding of each row's last block: <= 3 sweeps
                const unsigned sl = (unsigned)tid + (unsigned)(THREADS * it);
                if (sl < nslots) {
                    const unsigned part = sl & 3u, lb = sl >> 3;
                    const unsigned b = half == 1u ? lb : __umulhi(lb, blk.half_magic), rp = lb - b * half;   // (2^32 / 1 has no 32-bit magic)
                    const unsigned r = 2u * rp + ((sl >> 2) & 1u), i = 16u * b + 4u * part;
                    if (i < (unsigned)blk.R && k0 + r < (unsigned)blk.L) {
                        const unsigned pos = r * (unsigned)blk.R + i;   // a multiple of 4: one 8-group of m_s
                        const unsigned off = b * (unsigned)blk.bs + (k0 + r) * 16u + 4u * part;
                        const v2f* src = reinterpret_cast<const v2f*>(&m_s[mpos(H + (int)pos)]);
                        *reinterpret_cast<float4*>(mc + off) = make_float4(src[0].x, src[0].y, src[1].x, src[1].y);
                        using v4 = __attribute__((ext_vector_type(4))) float;
                        // non-temporal: the pair FFT's first pass, which reads p next, measured 0.01-0.02 ms faster
                        __builtin_nontemporal_store(*reinterpret_cast<const v4*>(&d_s[pos]), reinterpret_cast<v4*>(pc + off));
                    }
                }
            }
        } else if (inner_tile || t0 + T <= n32) {   // (workgroup-uniform) the whole tile lies inside the channel
            // through d_s (dead since m_s was built): thread order in, lane-contiguous order out
            float4* stage = reinterpret_cast<float4*>(&d_s[o]);
            stage[0] = make_float4(acc[0].x + acc[0].y, acc[1].x + acc[1].y, acc[2].x + acc[2].y, acc[3].x + acc[3].y);
            stage[1] = make_float4(acc[4].x + acc[4].y, acc[5].x + acc[5].y, acc[6].x + acc[6].y, acc[7].x + acc[7].y);
            // a wave reads back only what its own lanes staged ([wbase, wbase + 512)), and the LDS serves one wave's
            // accesses in order: no workgroup barrier, the writes only have to be issued before the reads
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pos = wbase + h * (32 * PER) + lane4;
                *reinterpret_cast<float4*>(p_out + (int64_t)c * n + t0 + pos) = *reinterpret_cast<const float4*>(&d_s[pos]);
            }
        } else if (t0 + o + PER <= n32) {
            float4* dst = reinterpret_cast<float4*>(p_out + (int64_t)c * n + t0 + o);
            dst[0] = make_float4(acc[0].x + acc[0].y, acc[1].x + acc[1].y, acc[2].x + acc[2].y, acc[3].x + acc[3].y);
            dst[1] = make_float4(acc[4].x + acc[4].y, acc[5].x + acc[5].y, acc[6].x + acc[6].y, acc[7].x + acc[7].y);
        } else {   // the channel ends inside this thread's run
#pragma unroll
            for (int r = 0; r < PER; ++r)
                if (t0 + o + r < n32) p_out[(int64_t)c * n + t0 + o + r] = acc[r].x + acc[r].y;
        }
        return;
    }
    // channels shorter than the filter's reach (n <= 81): sample by sample
    auto tap = [&](int j) -> float {   // lag j >= 0
        const v2f pr = taps.pair[(H + j) >> 1];
        return ((H + j) & 1) ? pr.y : pr.x;
    };
    const float m_first = (q0 <= 0) ? m_s[mpos(0 - q0)] : 0.f;
    const float m_last = (last - q0 < T + 2 * H) ? m_s[mpos(last - q0)] : 0.f;
    for (int r = 0; r < PER; ++r) {
        const int i = t0 + o + r;
        if (i >= n32) break;
        float acc = tap(0) * m_s[mpos(o + r + H)];
        for (int j = 1; j <= H; ++j) {
            const int ql = i - j, qr = i + j;
            const float el = (ql < 0) ? 2.f * m_first - m_s[mpos(-ql - q0)] : m_s[mpos(ql - q0)];
            const float er = (qr > last) ? 2.f * m_last - m_s[mpos(2 * last - qr - q0)] : m_s[mpos(qr - q0)];
            acc = fmaf(tap(j), el + er, acc);
        }
        p_out[(int64_t)c * n + i] = acc;
    }
}

// z and u may be the same buffer (element i only depends on element i).
__global__ __launch_bounds__(kThreads) void k_stereo_mix(const float2* z, const float* __restrict__ m,
                                                         float2* u, size_t count) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= count) return;
    const float2 v = z[i];
    // Im(z^2) / |z^2| = 2ab / (a^2 + b^2); pre-scaled so tiny pilots do not underflow.
    // z == 0 gives 0/0 = NaN exactly like pll.py:57-58.
    const float s = fmaxf(fabsf(v.x), fabsf(v.y));
    const float a = v.x / s, b = v.y / s;
    const float s2 = (2.f * a * b) / (a * a + b * b);
    const float mm = m[i];
    const float lmr = (s2 * mm) * 1.0175f;
    u[i] = make_float2(mm + lmr, mm - lmr);
}

__global__ __launch_bounds__(kThreads) void k_pll_phase(const float2* __restrict__ z, size_t count,
                                                        double mult, int int_power, int want_imag,
                                                        float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= count) return;
    const float2 v = z[i];
    if (int_power >= 1) {
        // numpy's complex power multiplies out small integer exponents
        float2 w = v;
        for (int k = 1; k < int_power; ++k) w = make_float2(w.x * v.x - w.y * v.y, w.x * v.y + w.y * v.x);
        const float mag = hypotf(w.x, w.y);
        out[i] = (want_imag ? w.y : w.x) / mag;
    } else {
        const double th = mult * atan2((double)v.y, (double)v.x);
        const bool zero = (v.x == 0.f && v.y == 0.f);
        const double r = want_imag ? sin(th) : cos(th);
        out[i] = zero ? __builtin_nanf("") : (float)r;
    }
}

// ---------------------------------------------------------------------------
// De-emphasis FIR, DC removal, clip
// ---------------------------------------------------------------------------

constexpr int kFirTile = 1024;

__device__ __forceinline__ float block_sum(float v, float* scratch) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < kThreads / 64; ++w) t += scratch[w];
    return t;  // valid in thread 0
}

__global__ __launch_bounds__(kThreads) void k_fir(const float* __restrict__ x, float* __restrict__ y,
                                                  int64_t n, int ch, const float* __restrict__ taps, int nb,
                                                  const float* __restrict__ state,
                                                  float* __restrict__ partial) {
    constexpr int T = kFirTile;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* b_s = smem;            // nb
    float* x_s = b_s + nb;        // T + nb - 1 : x[t0 - (nb-1) + s]
    float* red = x_s + (T + nb - 1);
    const int tid = threadIdx.x;
    const int h = blockIdx.y, c = blockIdx.z;
    const int64_t t0 = (int64_t)blockIdx.x * T;
    const float* xc = x + (int64_t)c * n * ch + h;
    float* yc = y + (int64_t)c * n * ch + h;
    const float* zc = state + ((int64_t)c * ch + h) * (nb - 1);
    for (int i = tid; i < nb; i += kThreads) b_s[i] = taps[i];
    for (int s = tid; s < T + nb - 1; s += kThreads) {
        const int64_t i = t0 - (nb - 1) + s;
        x_s[s] = (i >= 0 && i < n) ? xc[i * ch] : 0.f;
    }
    __syncthreads();
    float local = 0.f;
    for (int o = tid; o < T; o += kThreads) {
        const int64_t i = t0 + o;
        if (i >= n) break;
        float acc = 0.f;
        const int base = o + nb - 1;
        for (int j = 0; j < nb; ++j) acc = fmaf(b_s[j], x_s[base - j], acc);
        if (i < nb - 1) acc += zc[i];   // lfilter's initial conditions
        yc[i * ch] = acc;
        local += acc;
    }
    if (partial != nullptr) {
        const float t = block_sum(local, red);
        if (tid == 0) partial[((int64_t)c * ch + h) * gridDim.x + blockIdx.x] = t;
    }
}

// De-emphasis FIR specialised for its 51 taps, on the interleaved stream z[i*CH + h]:
// y[e] = sum_j b[j] z[e - CH j].  Each thread produces 8 consecutive interleaved outputs (one
// 32-byte store) from a register window read with pinned ds_read_b128; taps arrive in SGPRs.
struct DeemphTaps {
    float b[51];
};

// Where interleaved input value e of a signal lives: plain (row_len = 0), or in rows of row_len values at a pitch of
// row_pitch (the inverse FFT's last pass writes 16-sample segments at a stride of n_1 samples; padding that stride to
// whole 128-byte lines keeps its stores aligned -- n_1 = 100 -> 112 -- and the reader skips the pad).
struct RowLayout {
    int row_len, row_pitch;
    int64_t signal_stride;     // values between consecutive signals
    __device__ __forceinline__ int64_t at(int64_t e) const {   // e in [0, 2^31): 32-bit division
        return row_len ? e + (int64_t)((unsigned)e / (unsigned)row_len) * (row_pitch - row_len) : e;
    }
};

constexpr int kFirPer = 8;
constexpr int kFirFastTile = kThreads * kFirPer;   // 2048 interleaved outputs per workgroup

// dc != nullptr: the DC removal and clip of mfm.py:64-65 / wbfm.py:97-100 happen here as well.
// The mean of the filter OUTPUT follows from sums the pipeline already has:
//   sum_n y_h[n] = sum_j b[j] (S_h - tail_h(j)) + sum_{n<50} z_h[n],
// S_h = sum of the leg's input = A * (DC bin of its spectrum, dc[c]), tail_h(j) = sum of its last j
// inputs, z = the carried filter state: 150 values per workgroup instead of a pass over the audio.
template <int CH>
__global__ __launch_bounds__(kThreads) void k_fir51(const float* __restrict__ x, float* __restrict__ y,
                                                    int64_t total, DeemphTaps taps,
                                                    const float* __restrict__ state,
                                                    float* __restrict__ partial,
                                                    const float2* __restrict__ dc, RowLayout lay) {
    constexpr int T = kFirFastTile, PER = kFirPer, HIST = 50 * CH;
    constexpr int HP = (HIST + 3) / 4 * 4;                          // history rounded to whole 16-byte loads
    constexpr int NV = (T + HP) / 4;                                // float4 loads per workgroup
    // z[t0 - HP + e] lives at xpos(e): 8 values, 2 pad dwords -- the window reads (thread tid starts at
    // e = 8 tid) are ds_read_b64 / b32 at a lane stride of 10 dwords instead of 8: the 64 lanes spread evenly
    // over the banks (at 8 they were 8-way conflicts, as in the pilot stage).
    auto xpos = [](int e) -> int { return (e >> 3) * 10 + (e & 7); };
    __shared__ __attribute__((aligned(16))) float x_s[((T + HP + 7) / 8) * 10];
    __shared__ float red[kThreads / 64];
    const int tid = threadIdx.x;
    const int c = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * T;
    const float* xc = x + (int64_t)c * lay.signal_stride;
    float* yc = y + (int64_t)c * total;
    constexpr int NL = (NV + kThreads - 1) / kThreads;
    float4 v[NL];
    float tail_v = 0.f, z_v = 0.f;
    if (dc != nullptr && tid < HIST) {                              // last 50 inputs of each leg + state
        tail_v = xc[lay.at(total - 1 - tid)];                       // interleaved: leg = (total-1-tid) % CH
        z_v = state[(int64_t)c * HIST + tid];
    }
#pragma unroll
    for (int it = 0; it < NL; ++it) {                              // unconditional, clamped 16-byte loads
        // the last sweep is ragged: its surplus threads re-read the tile's last quad (an L1 hit) -- reading on
        // into the next tile's samples made this kernel fetch 1.5x its input (profiles/r02_c_hbm_traffic.md)
        const int q = (tid + kThreads * it < NV) ? tid + kThreads * it : NV - 1;
        const int64_t e = t0 - HP + 4 * (int64_t)q;                 // total % 4 == 0: whole quads are in or out
        v[it] = *reinterpret_cast<const float4*>(xc + lay.at(e < 0 ? 0 : (e > total - 4 ? total - 4 : e)));
    }
#pragma unroll
    for (int it = 0; it < NL; ++it) {
        const int q = tid + kThreads * it;
        const int64_t e = t0 - HP + 4 * (int64_t)q;
        if (q < NV) {
            const float4 val = (e >= 0 && e < total) ? v[it] : make_float4(0.f, 0.f, 0.f, 0.f);
            v2f* dst = reinterpret_cast<v2f*>(&x_s[xpos(4 * q)]);       // 8-byte aligned (groups of 10 dwords)
            dst[0] = v2f{val.x, val.y};
            dst[1] = v2f{val.z, val.w};
        }
    }
    __syncthreads();
    float mean = 0.f;
    if (dc != nullptr) {
        // sum of the filter output = sum_legs (S_h B - sum_i tail_h[i] sfx[i + 1] + sum_i z_h[i]) with S_h the
        // leg's input sum (from the DC bin), B = sum b, tail_h[i] its i-th input from the end, sfx[k] =
        // sum_{j >= k} b[j], z_h the carried-in state.  One term per thread, one block reduction.
        float term = 0.f;
        if (tid < HIST) {
            const int i = tid / CH;   // xc[total - 1 - tid] is the i-th sample from the end of its leg
            float sf = 0.f;
#pragma unroll
            for (int j = 1; j <= 50; ++j) sf += (j > i) ? taps.b[j] : 0.f;
            term = z_v - tail_v * sf;
        }
        for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = term;
        __syncthreads();
        float bsum = 0.f;
#pragma unroll
        for (int j = 0; j <= 50; ++j) bsum += taps.b[j];
        const float2 d0 = dc[c];
        const float legs = (CH == 2) ? d0.x + d0.y : d0.x;
        float tot = (float)(total / CH) * legs * bsum;
        for (int wv = 0; wv < kThreads / 64; ++wv) tot += red[wv];
        mean = tot / (float)total;
    }
    const int o = tid * PER;
    const int64_t e0 = t0 + o;
    // y[e] = sum_j b[j] z[e - CH j], and the 51 taps are the first samples of a one-pole impulse response:
    // b[0] = 0, b[j] = c x^(j-1) (deemphasis.py:37-46).  A truncated geometric series obeys
    //     y[e] = x y[e - CH] + b[1] z[e - CH] - x b[50] z[e - 51 CH],
    // so only the first sample of each leg in a thread's run of 8 outputs takes the 51-tap sum (one 8-byte LDS read and
    // CH FMAs per tap); the other 8 - CH follow in three FMAs each: 15 instead of 51 FMAs per output, and the kernel
    // is no longer bound by the VALU (cfg4: 0.228 -> 0.18-0.20 ms).  Rounding: the recursion runs for
    // at most 8 / CH - 1 steps before the next full sum; against the float64 sum it is as close as the plain float32
    // FIR (3.9e-7 of peak either way).  Threads whose outputs still see the carried-in state (e0 < HIST) take the plain
    // sums: their recursion would need inputs of the previous buffer, of which only the state survives.
    auto zl = [&](int d) -> float { return x_s[xpos(o + HP + d)]; };   // z[e0 + d], d >= -HP (zero outside the signal)
    float acc[PER];
    if (e0 >= HIST) {
        // (the tap sum is split over independent partial sums: one accumulator would be a chain of 50 dependent FMAs)
        constexpr int NP = 4 / CH;
        float part[CH][NP];
#pragma unroll
        for (int r = 0; r < CH; ++r)
#pragma unroll
            for (int i = 0; i < NP; ++i) part[r][i] = 0.f;
#pragma unroll
        for (int j = 1; j <= 50; ++j) {
            if constexpr (CH == 2) {
                v2f q = *reinterpret_cast<const v2f*>(&x_s[xpos(o + HP - 2 * j)]);   // even offset: one aligned pair
                part[0][j % NP] = fmaf(taps.b[j], q.x, part[0][j % NP]);
                part[1][j % NP] = fmaf(taps.b[j], q.y, part[1][j % NP]);
            } else {
                part[0][j % NP] = fmaf(taps.b[j], zl(-j), part[0][j % NP]);
            }
        }
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            float a = part[r][0];
#pragma unroll
            for (int i = 1; i < NP; ++i) a += part[r][i];
            acc[r] = a;
        }
        const float xr = taps.b[2] / taps.b[1];                      // (uniform: scalar unit)
        const float tail = xr * taps.b[50];
#pragma unroll
        for (int r = CH; r < PER; ++r)
            acc[r] = fmaf(xr, acc[r - CH], fmaf(taps.b[1], zl(r - CH), -tail * zl(r - 51 * CH)));
    } else {
        // (eight independent sums, taps in the outer loop: this path is on the critical path of short launches --
        //  one wave per signal takes it, and a launch of 1.2 rounds of workgroups waits for exactly that wave)
#pragma unroll
        for (int r = 0; r < PER; ++r) acc[r] = 0.f;
#pragma unroll 5
        for (int j = 1; j <= 50; ++j) {
#pragma unroll
            for (int r = 0; r < PER; ++r) acc[r] = fmaf(taps.b[j], zl(r - CH * j), acc[r]);
        }
    }
    float local = 0.f;
    if (e0 < HIST) {                                               // lfilter's initial conditions
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int64_t e = e0 + r;
            if (e < HIST && e < total) acc[r] += state[(int64_t)c * HIST + (e % CH) * 50 + e / CH];
        }
    }
    if (dc != nullptr) {
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const float t = acc[r] - mean;
            acc[r] = (t < -0.999f) ? -0.999f : ((t > 0.999f) ? 0.999f : t);   // NaN stays NaN like np.clip
        }
    }
    if (e0 + PER <= total) {   // (lane-contiguous stores through LDS, as in the pilot stage, cost this small kernel 7 %: two more barriers)
        float4* dst = reinterpret_cast<float4*>(yc + e0);
        dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
#pragma unroll
        for (int r = 0; r < PER; ++r) local += acc[r];
    } else {
        for (int r = 0; r < PER; ++r)
            if (e0 + r < total) {
                yc[e0 + r] = acc[r];
                local += acc[r];
            }
    }
    if (partial != nullptr) {
        const float t = block_sum(local, red);
        if (tid == 0) partial[(int64_t)c * gridDim.x + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(128) void k_fir_state(const float* __restrict__ x, int64_t n, int ch,
                                                   const float* __restrict__ taps, int nb,
                                                   float* __restrict__ state, RowLayout lay) {
    // The last nb - 1 inputs and the taps go to LDS first (independent, coalesced loads); the triangular sums then run
    // from LDS in the same order as before.  (Round 3 read x and the taps inside the sum: fifty dependent-latency global
    // loads per thread, 5.9 us for a kernel that is the last launch of every MFM / WBFM call.)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* z_old = smem;             // nb - 1
    float* xt = smem + (nb - 1);     // xt[i] = x[n - 1 - i], i < nb - 1
    float* tp = xt + (nb - 1);       // nb taps
    const int h = blockIdx.x % ch, c = blockIdx.x / ch;
    const float* xc = x + (int64_t)c * lay.signal_stride;
    float* zc = state + ((int64_t)c * ch + h) * (nb - 1);
    for (int s = threadIdx.x; s < nb - 1; s += blockDim.x) {
        z_old[s] = zc[s];
        xt[s] = (s < n) ? xc[lay.at((n - 1 - s) * ch + h)] : 0.f;
    }
    for (int s = threadIdx.x; s < nb; s += blockDim.x) tp[s] = taps[s];
    __syncthreads();
    for (int s = threadIdx.x; s < nb - 1; s += blockDim.x) {
        // zf[s] = sum_i b[s+1+i] x[n-1-i]  (+ what is left of the old state when n < nb-1)
        float acc = 0.f;
        const int cnt = (int)((nb - 1 - s) < n ? (nb - 1 - s) : n);
        for (int i = 0; i < cnt; ++i) acc = fmaf(tp[s + 1 + i], xt[i], acc);
        if (s + n < nb - 1) acc += z_old[s + n];
        zc[s] = acc;
    }
}

__global__ __launch_bounds__(kThreads) void k_dc_clip(float* __restrict__ y, int64_t total,
                                                      const float* __restrict__ partial, int nparts) {
    __shared__ float mean_s;
    const int c = blockIdx.y;
    if (threadIdx.x < 64) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 64) acc += (double)partial[(int64_t)c * nparts + i];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (threadIdx.x == 0) mean_s = (float)(acc / (double)total);
    }
    __syncthreads();
    const float mean = mean_s;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    float v = y[(int64_t)c * total + i] - mean;
    v = (v < -0.999f) ? -0.999f : ((v > 0.999f) ? 0.999f : v);   // NaN stays NaN like np.clip
    y[(int64_t)c * total + i] = v;
}

}  // namespace

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------

#define RC_LAUNCH_CHECK() RC_HIP(hipGetLastError())

void launch_spectrum_c2c(const float2* X, int64_t x_stride, int64_t n, const int64_t* roll, float2* Y,
                         int64_t m, int batch, const float* wpos, const float* wneg, float w_merge,
                         int nyq, int nneg, int nyq_mode, float scale, hipStream_t stream) {
    if (batch <= 0 || m <= 0) return;
    hipLaunchKernelGGL(k_spectrum_c2c, grid2(m, kThreads, batch), dim3(kThreads), 0, stream, X, x_stride, n,
                       roll, Y, m, wpos, wneg, w_merge, nyq, nneg, nyq_mode, scale);
    RC_LAUNCH_CHECK();
}

void launch_spectrum_r2c(const float2* X, int64_t n, float2* Y, int64_t m, int batch, const float* wr,
                         int nyq, int nmin, float nyq_factor, float scale, hipStream_t stream) {
    if (batch <= 0) return;
    hipLaunchKernelGGL(k_spectrum_r2c, grid2(m / 2 + 1, kThreads, batch), dim3(kThreads), 0, stream, X, n, Y,
                       m, wr, nyq, nmin, nyq_factor, scale);
    RC_LAUNCH_CHECK();
}

void launch_spectrum_real_full(const float2* X, int64_t n, float2* Y, int64_t m, int batch, const float* wr,
                               int nyq, int nmin, float nyq_factor, float scale, float2* dc, bool paired,
                               hipStream_t stream) {
    if (batch <= 0) return;
    hipLaunchKernelGGL(k_spectrum_real_full, grid2(m, kThreads, batch), dim3(kThreads), 0, stream, X, n, Y, m,
                       wr, nyq, nmin, nyq_factor, scale, dc, paired ? 1 : 0);
    RC_LAUNCH_CHECK();
}

void launch_hilbert_mask(const float2* P, float2* Z, int64_t n, int batch, float scale,
                         hipStream_t stream) {
    if (batch <= 0) return;
    hipLaunchKernelGGL(k_hilbert_mask, grid2(n, kThreads, batch), dim3(kThreads), 0, stream, P, Z, n, scale);
    RC_LAUNCH_CHECK();
}

void launch_stereo_unpack(const float2* U, int64_t B, float2* V, int64_t A, int batch, const float* wr,
                          int nyq, int nmin, float nyq_factor, float scale, float2* dc, hipStream_t stream) {
    if (batch <= 0) return;
    hipLaunchKernelGGL(k_stereo_unpack, grid2(A, kThreads, batch), dim3(kThreads), 0, stream, U, B, V, A, wr,
                       nyq, nmin, nyq_factor, scale, dc);
    RC_LAUNCH_CHECK();
}

__global__ __launch_bounds__(kThreads) void k_discriminator_phase(const float* __restrict__ theta,
                                                                  float* __restrict__ d, int64_t n) {
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float* tc = theta + (int64_t)c * n;
    d[(int64_t)c * n + i] = (i == 0) ? 0.f : phase_step_wrapped(tc[i], tc[i - 1]);
}

void launch_discriminator_phase(const float* theta, float* d, int64_t n, int batch, hipStream_t stream) {
    if (batch <= 0) return;
    hipLaunchKernelGGL(k_discriminator_phase, grid2(n, kThreads, batch), dim3(kThreads), 0, stream, theta, d, n);
    RC_LAUNCH_CHECK();
}

void launch_discriminator(const float2* iq, float* d, int64_t n, int batch, hipStream_t stream) {
    if (batch <= 0) return;
    hipLaunchKernelGGL(k_discriminator, grid2(n, kThreads, batch), dim3(kThreads), 0, stream, iq, d, n);
    RC_LAUNCH_CHECK();
}

template <class IN>
static void launch_pilot_stage_h40_t(const IN* iq, float* m_out, float* p_out, int64_t n, int batch,
                                     const float* g_host, float side_tap, hipStream_t stream, const PilotBlocked* blk) {
    if (batch <= 0) return;
    PilotTaps taps;   // h[t] = g[|t - 40|], t = 0..80, h[81] = 0
    for (int j = 0; j <= 40; ++j) {
        const int t0 = 2 * j, t1 = 2 * j + 1;
        taps.pair[j].x = g_host[t0 >= 40 ? t0 - 40 : 40 - t0];
        taps.pair[j].y = t1 <= 80 ? g_host[t1 >= 40 ? t1 - 40 : 40 - t1] : 0.f;
    }
    if (blk && blk->valid()) {
        RC_REQUIRE((int64_t)blk->L * blk->R == n && blk->rpt * blk->R <= kPilotBlockedTile && blk->R % 4 == 0, RCFM_ERR_ARG,
                   "tile-blocked pilot layout does not match the signal");
        const unsigned tiles_x = (unsigned)((blk->L + blk->rpt - 1) / blk->rpt);
        const unsigned blocks = (tiles_x * (unsigned)batch + 7u) / 8u * 8u;
        hipLaunchKernelGGL((k_pilot_stage_h40<IN, true, kPilotBlockedThreads>), dim3(blocks), dim3(kPilotBlockedThreads), 0, stream,
                           iq, m_out, p_out, n, taps, side_tap, tiles_x, (unsigned)batch, *blk);
        RC_LAUNCH_CHECK();
        return;
    }
    const unsigned tiles_x = (unsigned)((n + kPilotFastTile - 1) / kPilotFastTile);
    const unsigned blocks = (tiles_x * (unsigned)batch + 7u) / 8u * 8u;
    hipLaunchKernelGGL((k_pilot_stage_h40<IN, false, kThreads>), dim3(blocks), dim3(kThreads), 0, stream, iq, m_out, p_out, n, taps,
                       side_tap, tiles_x, (unsigned)batch, PilotBlocked{});
    RC_LAUNCH_CHECK();
}

PilotBlocked PilotBlocked::plan(int64_t L, int64_t R) {
    PilotBlocked b;
    constexpr int T = kPilotBlockedTile;
    // R >= 32: at most three store sweeps per workgroup (the padding of every row's last block counts)
    if (R < 32 || L <= 0 || R % 4 != 0 || R > T / 2 || L * R > (int64_t)1 << 30) return b;
    const int rpt = (int)(T / R) & ~1;
    if (rpt < 2 || rpt * R < T - T / 8 || L * R < 2 * T) return b;
    b.R = (int)R;
    b.L = (int)L;
    b.rpt = rpt;
    b.nb = (int)((R + 15) / 16);
    b.bs = (int)(16 * L);   // (an odd number of lines per block measured the same)
    b.half_magic = (unsigned)((((uint64_t)1 << 32) + (rpt / 2) - 1) / (uint64_t)(rpt / 2));   // exact for lb < 2^16
    b.stride = (int64_t)b.nb * b.bs;
    return b;
}

void launch_pilot_stage_h40(const float2* iq, float* m_out, float* p_out, int64_t n, int batch,
                            const float* g_host, float side_tap, hipStream_t stream, const PilotBlocked* blk) {
    launch_pilot_stage_h40_t(iq, m_out, p_out, n, batch, g_host, side_tap, stream, blk);
}

void launch_pilot_stage_h40_phase(const float* theta, float* m_out, float* p_out, int64_t n, int batch,
                                  const float* g_host, float side_tap, hipStream_t stream, const PilotBlocked* blk) {
    launch_pilot_stage_h40_t(theta, m_out, p_out, n, batch, g_host, side_tap, stream, blk);
}

void launch_pilot_stage(const float2* iq, const float* x, float* m_out, float* p_out, int64_t n,
                        int batch, const float* g, int H, float side_tap, hipStream_t stream) {
    if (batch <= 0) return;
    const size_t lds = sizeof(float) * ((size_t)(H + 1) + (kPilotTile + 2 * H) + (kPilotTile + 2 * H + 2));
    hipLaunchKernelGGL(k_pilot_stage, grid2(n, kPilotTile, batch), dim3(kThreads), lds, stream, iq, x, m_out,
                       p_out, n, g, H, side_tap);
    RC_LAUNCH_CHECK();
}

void launch_stereo_mix(const float2* z, const float* m, float2* u, size_t count, hipStream_t stream) {
    if (count == 0) return;
    hipLaunchKernelGGL(k_stereo_mix, dim3((unsigned)((count + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       stream, z, m, u, count);
    RC_LAUNCH_CHECK();
}

void launch_pll_phase(const float2* z, size_t count, double mult, int want_imag, float* out,
                      hipStream_t stream) {
    if (count == 0) return;
    int int_power = 0;
    if (mult >= 1.0 && mult <= 64.0 && mult == (double)(int)mult) int_power = (int)mult;
    hipLaunchKernelGGL(k_pll_phase, dim3((unsigned)((count + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       stream, z, count, mult, int_power, want_imag, out);
    RC_LAUNCH_CHECK();
}

int fir_tiles(int64_t n) { return (int)((n + kFirTile - 1) / kFirTile); }

void launch_fir(const float* x, float* y, int64_t n, int ch, int batch, const float* taps, int nb,
                const float* state, float* partial, hipStream_t stream) {
    if (batch <= 0 || n <= 0) return;
    const size_t lds = sizeof(float) * ((size_t)nb + (kFirTile + nb - 1) + 8);
    hipLaunchKernelGGL(k_fir, dim3((unsigned)fir_tiles(n), (unsigned)ch, (unsigned)batch), dim3(kThreads), lds,
                       stream, x, y, n, ch, taps, nb, state, partial);
    RC_LAUNCH_CHECK();
}

int fir51_tiles(int64_t n, int ch) { return (int)((n * ch + kFirFastTile - 1) / kFirFastTile); }

static RowLayout row_layout(int64_t n, int ch, int row_samples, int row_pitch_samples) {
    if (row_samples <= 0 || row_pitch_samples == row_samples) return RowLayout{0, 0, n * ch};
    RC_REQUIRE(n % row_samples == 0 && (row_samples * ch) % 4 == 0 && (row_pitch_samples * ch) % 4 == 0,
               RCFM_ERR_RUNTIME, "padded audio rows must hold whole 16-byte quads");
    return RowLayout{row_samples * ch, row_pitch_samples * ch, (n / row_samples) * (int64_t)row_pitch_samples * ch};
}

void launch_fir51(const float* x, float* y, int64_t n, int ch, int batch, const float* taps_host,
                  const float* state, float* partial, const float2* dc, hipStream_t stream, int row_samples,
                  int row_pitch_samples) {
    if (batch <= 0 || n <= 0) return;
    DeemphTaps taps;
    for (int i = 0; i < 51; ++i) taps.b[i] = taps_host[i];
    // k_fir51 runs the one-pole recursion between full tap sums: the taps must be what deemphasis.py:37-46 designs,
    // b[0] = 0 and a geometric tail
    RC_REQUIRE(taps.b[0] == 0.f && taps.b[1] > 0.f, RCFM_ERR_RUNTIME, "de-emphasis taps are not a one-pole response");
    for (int i = 1; i < 50; ++i)   // (in double, with a floor: fast-decaying responses end in denormals and zeros)
        RC_REQUIRE(std::fabs((double)taps.b[i + 1] * taps.b[1] - (double)taps.b[i] * taps.b[2]) <=
                       4e-7 * (double)taps.b[i] * taps.b[1] + 1e-36,
                   RCFM_ERR_RUNTIME, "de-emphasis taps are not a one-pole response");
    const RowLayout lay = row_layout(n, ch, row_samples, row_pitch_samples);
    const dim3 grid((unsigned)fir51_tiles(n, ch), (unsigned)batch, 1);
    if (ch == 2)
        hipLaunchKernelGGL(k_fir51<2>, grid, dim3(kThreads), 0, stream, x, y, n * 2, taps, state, partial, dc, lay);
    else
        hipLaunchKernelGGL(k_fir51<1>, grid, dim3(kThreads), 0, stream, x, y, n, taps, state, partial, dc, lay);
    RC_LAUNCH_CHECK();
}

void launch_fir_state(const float* x, int64_t n, int ch, int batch, const float* taps, int nb,
                      float* state, hipStream_t stream, int row_samples, int row_pitch_samples) {
    if (batch <= 0 || nb < 2) return;
    hipLaunchKernelGGL(k_fir_state, dim3((unsigned)(batch * ch)), dim3(128), sizeof(float) * (3 * nb), stream,
                       x, n, ch, taps, nb, state, row_layout(n, ch, row_samples, row_pitch_samples));
    RC_LAUNCH_CHECK();
}

void launch_dc_clip(float* y, int64_t n, int ch, int batch, const float* partial, int nparts,
                    hipStream_t stream) {
    if (batch <= 0) return;
    const int64_t total = n * ch;
    hipLaunchKernelGGL(k_dc_clip, grid2(total, kThreads, batch), dim3(kThreads), 0, stream, y, total, partial,
                       nparts);
    RC_LAUNCH_CHECK();
}

}  // namespace rcfm
