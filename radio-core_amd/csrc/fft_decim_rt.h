// k_fft_tile2_decim with the SHORT transform's length known only at run time.
//
// k_fft_tile2_decim<L, ..., L2, Q0, Q1> (fft_kernel.h) is instantiated for the hot geometries (240 000 -> 48 000:
// 500 -> 100, 12 500 -> 8 000: 125 -> 80).  The reference accepts any sizes (wbfm.py:32-59, decimate.py:21-50) and
// benchmarks 256 000 -> 32 000 (tests/benchmark.py:85): here one kernel per LONG tile length L serves every even
// L2 = A / n_1 < L whose radices are small.  The first transform (L points, 16 lines, compile-time radices) and the
// selection of the surviving rows are those of k_fft_tile2_decim; the short transform runs all its stages in LDS with
// run-time radices (the short plan's own radix list and output-slot table, FftPassDev::p.radix / pos), then one
// twiddle per point.  It is a quarter of the tile's points or fewer, so the generic form costs a few percent of the
// kernel (measured in bench.py's other_configs).
#pragma once

#include "fft_kernel.h"

namespace rcfm {
RCFM_NS_OPEN

// decimate.py:48 for the packed stereo pair between FFT_B's last pass and IFFT_A's first pass
// (k_fft_tile2_decim): u = l + j r is one complex signal and the Hamming weight is real and even, so
// resampling u resamples both legs: V[kappa] = U[k] W[|kappa|] / B with scipy's Nyquist merge -- no
// unpacking into L and R at all.
struct WinAudioDecim {
    const float* wr;   // folded window, A/2 + 1 entries
    float2* dc;        // [count] or null: receives V[c][0] = (sum l, sum r) / A
    float scale;
    int A, n1;
    int split_count;   // > 0: the signal is a PAIR of real channels (2P, 2P+1): dc[2P] = (Re, 0), dc[2P+1] = (Im, 0)
    __device__ __forceinline__ float weight(const fftk::LineId&, int l, int k0) const {
        const int kappa = l * n1 + k0;
        int kk = kappa <= A / 2 ? kappa : A - kappa;
        kk = kk < 0 ? 0 : kk;   // rows past the tile's real extent (clamped lanes)
        return wr[kk] * scale;
    }
    __device__ __forceinline__ void dc_bin(const fftk::LineId& id, float2 v) const {
        if (dc == nullptr) return;
        if (split_count > 0) {
            dc[2 * id.batch] = make_float2(v.x, 0.f);
            if (2 * id.batch + 1 < split_count) dc[2 * id.batch + 1] = make_float2(v.y, 0.f);
        } else {
            dc[id.batch] = v;
        }
    }
};

namespace fftk {

// One in-place DIF stage of run-time geometry on the rows image (pitch 17): block length mt, radix R.
template <int R, int T>
__device__ __forceinline__ void stage_rt(float2* tile, const float2* tw, int L2, int mt, int tid) {
    const int m = mt / R, step = L2 / mt, nb = (L2 / R) * W;
    for (int e = tid; e < nb; e += T) {
        const int w = e & (W - 1), b = e >> kLogW;
        const int g = b / m, kp = b - g * m;
        const int base = g * mt + kp;
        float2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = tile[lds_slot<true>(base + q * m, w)];
        dft_p<R>(v);
        const int ts = kp * step;
#pragma unroll
        for (int q = 1; q < R; ++q) v[dft_slot<R>(q)] = cmul(v[dft_slot<R>(q)], tw[q * ts]);
#pragma unroll
        for (int q = 0; q < R; ++q) tile[lds_slot<true>(base + q * m, w)] = v[dft_slot<R>(q)];
    }
}

constexpr bool decim_rt_radix_ok(int r) {
    return r == 2 || r == 3 || r == 4 || r == 5 || r == 6 || r == 8 || r == 10 || r == 12 || r == 16;
}

template <int L, int R0, int R1, int R2, int R3, int T>
__global__ __launch_bounds__(T, (T * 16 / W >= 512 ? 4 : 1)) void k_fft_tile2_decim_rt(FftPassDev d1, FftPassDev d2,
                                                                              LoadPlainT<false> load, WinAudioDecim win,
                                                                              StorePlainT<false> store) {
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
    const int L2 = p2.L;
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

    // ---- the long transform: exactly k_fft_tile2_decim's ---------------------------------------------
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
    // rows of the short spectrum: positive frequencies, negative frequencies, the negative Nyquist row (kept by
    // every line), the positive one parked in row L2 (only bin A/2 of line k_0 = 0 needs it)
#pragma unroll
    for (int it = 0; it < nitL; ++it) {
        const int g = rg + RG * it;
        if ((rowsL % RG == 0) || g < rowsL) {
            const int kb = kbase(g);
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                const int k = kb + (L / RL) * q;
                int l = -1;
                if (k < L2 / 2) l = k;
                else if (k >= L - L2 / 2) l = k - (L - L2);
                else if (k == L2 / 2) l = L2;
                if (l >= 0) tile[lds_slot<true>(l, w)] = xr[it * RL + dft_slot<RL>(q)];
            }
        }
    }
    for (int e = tid; e < L2; e += T) tw[e] = d2.stage_tw[e];   // (every read of the long transform's table is done)
    lds_barrier();

    // ---- weights, Nyquist merge, swap for the inverse transform ----------------------------------------
    for (int e = tid; e < L2 * W; e += T) {
        const int l = e >> kLogW, wl = e & (W - 1);
        float2 x = tile[lds_slot<true>(l, wl)];
        if (l == L2 / 2 && i0 + wl == 0) {   // Y[A/2] = X[A/2] + X[-A/2] (one point of one tile)
            const float2 y = tile[lds_slot<true>(L2, wl)];
            x = make_float2(x.x + y.x, x.y + y.y);
        }
        const float wg = win.weight(id, l, i0 + (wl < wvalid ? wl : 0));
        x = make_float2(x.x * wg, x.y * wg);
        if (l == 0 && i0 + wl == 0) {
            id.i = 0;
            win.dc_bin(id, x);
        }
        tile[lds_slot<true>(l, wl)] = make_float2(x.y, x.x);
    }
    lds_barrier();

    // ---- the short transform: every stage in LDS, run-time radices ---------------------------------------
    int mt = L2;
    for (int s = 0; s < p2.nstages; ++s) {
        const int r = p2.radix[s];
        switch (r) {
            case 2: stage_rt<2, T>(tile, tw, L2, mt, tid); break;
            case 3: stage_rt<3, T>(tile, tw, L2, mt, tid); break;
            case 4: stage_rt<4, T>(tile, tw, L2, mt, tid); break;
            case 5: stage_rt<5, T>(tile, tw, L2, mt, tid); break;
            case 6: stage_rt<6, T>(tile, tw, L2, mt, tid); break;
            case 8: stage_rt<8, T>(tile, tw, L2, mt, tid); break;
            case 10: stage_rt<10, T>(tile, tw, L2, mt, tid); break;
            case 12: stage_rt<12, T>(tile, tw, L2, mt, tid); break;
            default: stage_rt<16, T>(tile, tw, L2, mt, tid); break;
        }
        mt /= r;
        lds_barrier();
    }
    id.i = i0 + w;
    const unsigned f = (unsigned)((int64_t)(i0 + w) * p2.tw_i);
    if (w < wvalid) {
        for (int e = tid; e < L2 * W; e += T) {
            const int k = e >> kLogW;
            const int row = d2.pos[k];
            const float2 y = cmul(tile[lds_slot<true>(row, w)], big_twiddle(d2, f * (unsigned)k));
            store(id, k, out_base, (unsigned)k * out_k + (unsigned)w, y);
        }
    }
}

// Long tile lengths with a run-time-L2 kernel (radices as in RCFM_FFT_FAST_LENGTHS).
#define RCFM_FFT_DECIM_RT_LENGTHS(X) \
    X(100, 10, 10, 1, 1)            \
    X(120, 12, 10, 1, 1)            \
    X(125, 5, 5, 5, 1)              \
    X(128, 8, 8, 2, 1)              \
    X(150, 15, 10, 1, 1)            \
    X(160, 16, 10, 1, 1)            \
    X(192, 16, 12, 1, 1)            \
    X(200, 20, 10, 1, 1)            \
    X(240, 24, 10, 1, 1)            \
    X(250, 25, 10, 1, 1)            \
    X(256, 16, 16, 1, 1)            \
    X(300, 20, 15, 1, 1)            \
    X(320, 10, 8, 4, 1)             \
    X(375, 5, 5, 5, 3)              \
    X(384, 8, 8, 6, 1)              \
    X(400, 20, 20, 1, 1)            \
    X(480, 24, 20, 1, 1)            \
    X(500, 25, 20, 1, 1)            \
    X(512, 8, 8, 8, 1)

inline bool fft_tile2_decim_rt_applies(const FftPassDev& d1, const FftPassDev& d2, int batch) {
    bool have = false;
#define RCFM_CASE(LEN, A, B, C, D) have = have || d1.p.L == LEN;
    RCFM_FFT_DECIM_RT_LENGTHS(RCFM_CASE)
#undef RCFM_CASE
    const int L2 = d2.p.L;
    if (!have || (L2 & 1) || L2 < 16 || L2 >= d1.p.L) return false;
    for (int s = 0; s < d2.p.nstages; ++s)
        if (!decim_rt_radix_ok(d2.p.radix[s])) return false;
    return d1.p.load_along_l && !d2.p.load_along_l && d1.p.n_inner == d2.p.n_inner &&
           d1.p.n_o1 * d1.p.n_o2 == 1 && d2.p.n_o1 * d2.p.n_o2 == 1 && d1.p.out_k == d2.p.in_l && d2.p.in_i == 1 &&
           d2.p.out_i == 1 && d2.p.has_twiddle && batch <= 65535;
}

}  // namespace fftk

// Defined in fused_decim.hip (its own translation unit: nineteen kernels per tile width).
bool launch_fft_tile2_decim_rt(const FftPassDev& d1, const FftPassDev& d2, int batch, const float2* tmp_f,
                               const WinAudioDecim& win, float2* tmp_a, hipStream_t s);

RCFM_NS_CLOSE
}  // namespace rcfm
