// Tuner.run + FM.run for a pair of narrow channels in ONE workgroup, every transform in LDS (lds_chain.h).
//
// LDS: xs = B complex (the signal being transformed), ds = B floats (the first member's phases): 12 B bytes, so
// B <= 13 653 fits the 160 KiB of a CU (cfg5: B = 12 500 -> 150 000 bytes, one workgroup per CU).  Per pair:
//   member 0: gather B bins (window, Nyquist merge: tuner.py:159-161) -> xs, IFFT_B in place, angle / pi -> ds
//   member 1: the same -> angle / pi -> upper half of xs
//   u[t] = d0[t] + j d1[t], d = wrapped phase step (fm.py:60-65, d[0] = 0)            -> xs
//   FFT_B(u); keep |k| <= A/2, Hamming weight, Nyquist merge (decimate.py:48)         -> xs[0 .. A)
//   IFFT_A: real part = member 0's audio, imaginary part = member 1's                  -> memory
// The transforms are in-place decimation-in-frequency with three stages of composite radices (dft_nat, fft_kernel.h);
// the last stage of each leaves its outputs in registers, from where they go to their natural-order place together
// with the point-wise work (arctangent; decimation), so no separate reordering pass exists.  Inverse transforms use
// the swap identity ifft(x) = swap(fft(swap(x))).  Stage twiddles: one entry of the global table W_L^e per butterfly
// (L2-resident: 100 KB for B = 12 500), powers by product tree (twiddle_powers).

#include "lds_chain.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"
#include "device_math.h"
#include "fft_kernel.h"

namespace rcfm {

namespace {

using fftk::cmul;
using fftk::dft_p;
using fftk::dft_slot;
using fftk::lds_barrier;
using fftk::twiddle_powers;

struct ChainDev {
    LdsChainArgs a;
    float two_pi_over_n, delta;   // window argument of source offset d: d * 2 pi / N + delta
    float c0, c1, c2, c3;         // a0 + (1 - a0) cos(th) as a series in th^2 (|th| < 0.25: fused_passes.hip)
    const float2* twB;            // W_B^e, e < B
    const float2* twA;            // W_A^e, e < A
};

// One in-place DIF stage on a single signal in LDS: block length MT entering the stage, radix R.
template <int L, int R, int MT, int T>
__device__ __forceinline__ void chain_stage(float2* x, const float2* __restrict__ tw, int tid) {
    constexpr int m = MT / R, step = L / MT, rows = L / R;
#pragma unroll 1
    for (int b = tid; b < rows; b += T) {          // one sweep where rows <= T (every stage but B's 625-row one at T = 512)
        const int g = b / m, kp = b - g * m;
        const int base = g * MT + kp;
        float2 v[R];
        const float2 w1 = tw[kp * step];          // W_MT^kp, issued ahead of the LDS reads
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = x[base + q * m];
        dft_p<R>(v);
        float2 pw[R];
        twiddle_powers<R>(w1, pw);
#pragma unroll
        for (int q = 1; q < R; ++q) v[dft_slot<R>(q)] = cmul(v[dft_slot<R>(q)], pw[q]);
#pragma unroll
        for (int q = 0; q < R; ++q) x[base + q * m] = v[dft_slot<R>(q)];
    }
}

// Last stage (block length R, no twiddle): butterfly g reads x[g R + q]; output q' is bin kb + (L / R) q' with
// kb = q1 + R0 q2 for g = q1 R1 + q2.  Results stay in v (slot dft_slot<R>(q')).
template <int L, int R0, int R1, int R, int T>
__device__ __forceinline__ int chain_last(const float2* x, int tid, float2* v) {
    static_assert(R0 * R1 * R == L, "three stages");
    constexpr int rows = L / R;
    static_assert(rows <= T, "one sweep per stage");
    const int g = tid < rows ? tid : 0;
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = x[g * R + q];
    dft_p<R>(v);
    const int q1 = g / R1, q2 = g - q1 * R1;
    return q1 + R0 * q2;
}

// DEEMPH: MFM -- the de-emphasis FIR with its carried state, mean removal and clip (mfm.py:62-66) run here as well, and the
// audio that leaves the chip is final.  A separate instantiation: the FM kernel (cfg5) keeps its registers.
template <int B, int R0, int R1, int R2, int A, int Q0, int Q1, int Q2, int T, bool DEEMPH>
__global__ __launch_bounds__(T) void k_fm_lds(ChainDev p) {
    static_assert(R0 * R1 * R2 == B && Q0 * Q1 * Q2 == A, "radix lists");
    static_assert(A % 2 == 0 && A < B && (size_t)B * 12 <= 160 * 1024, "geometry");
    __shared__ __attribute__((aligned(16))) float2 xs[B];
    __shared__ __attribute__((aligned(16))) float ds[B];
    float* const th1 = reinterpret_cast<float*>(xs) + B;      // member 1's phases: upper half of xs
    // the tail of ds, behind the decimation weights (A/2 + 1 <= B - 256): de-emphasis taps [0, 51), carried state of
    // member 0 [64, 114) and member 1 [128, 178), the two means [192, 194), the DC bins [200, 202), suffix sums of the
    // taps [204, 255)
    constexpr int kScratch = B - 256;
    // MFM: the audio in xs is read back in runs of 32 consecutive samples per lane (the de-emphasis below); one pad slot
    // every 32 samples puts adjacent lanes 33 slots = 66 dwords apart: conflict-free for every access of that phase.
    auto aswz = [](int n) -> int { return DEEMPH ? n + (n >> 5) : n; };
    static_assert(!DEEMPH || A + (A >> 5) + 1 <= B, "padded audio fits xs");
    static_assert(A / 2 + 1 <= kScratch && T <= 1024, "scratch behind the decimation weights");
    int tid = threadIdx.x;
    const int npairs = (p.a.count + 1) / 2;
    constexpr int NL = (B + T - 1) / T;

    auto window = [&](int d) -> float {
        const float th = fmaf((float)d, p.two_pi_over_n, p.delta);
        const float t = th * th;
        return fmaf(t, fmaf(t, fmaf(t, p.c3, p.c2), p.c1), p.c0);
    };
    // The B bins of one channel, raw, into registers: issued one phase AHEAD of their use (the member's loads fly
    // under the previous member's arctangents, the next pair's under this pair's IFFT_A) -- with one workgroup per CU
    // nothing else would hide the memory latency.
    // Fused gather: the bins arrive in the order the first stage of IFFT_B consumes them --
    // thread b holds points b + q (B / R0), q < R0, of one radix-R0 butterfly -- so window, Nyquist merge and that stage
    // run straight from the prefetched registers and xs is written once, already transformed: one LDS write + barrier +
    // LDS read less per member than "gather to xs, then stage 1" (round 3).  The loads stay coalesced (lanes = adjacent b).
    // (The 640-thread instantiations are capped at 168 VGPRs and spill another 200 dwords in this form: they keep round 3's.)
    constexpr int M0 = B / R0;                                   // butterflies of the first stage
    constexpr bool kFused = T <= 512 && M0 <= T;
    constexpr int NV = kFused ? R0 : NL;
    float2 v[NV], x2;
    auto issue_gather = [&](int c) {
        const float2* Xc = p.a.X + p.a.base[c];
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            int k;
            if constexpr (kFused) {
                k = (tid < M0 ? tid : M0 - 1) + M0 * it;         // (the surplus threads repeat the last butterfly's loads)
            } else {
                k = tid + T * it;
                k = k < B ? k : B - 1;                          // ragged last sweep: clamped, not stored
            }
            v[it] = Xc[k < p.a.nyq ? k : k - B];
        }
        x2 = Xc[p.a.merge >= 0 ? -p.a.merge : 0];               // Y[+B/2] += X[-B/2] w(-B/2) (NYQ_DOWN)
    };

    int pair = (int)blockIdx.x;
    if (pair < npairs) issue_gather(2 * pair);
#pragma unroll 1
    for (; pair < npairs; pair += (int)gridDim.x) {
    asm volatile("" : "+v"(tid));      // per-thread index tables stay inside the iteration (they would cost the registers
                                       // the butterflies need: fft_kernel.h, persistent form)
    const int c0 = 2 * pair;
    const bool has1 = c0 + 1 < p.a.count;
    const int c1 = has1 ? c0 + 1 : c0;

    // ---- the two channels: gather -> IFFT_B -> phases --------------------------------------------------------
    auto member = [&](auto MEM) {                               // (straight-line for both members: no conditional
        constexpr int mem = decltype(MEM)::value;               //  definition keeps the prefetched bins alive longer)
        float2 m2 = make_float2(0.f, 0.f);
        if (p.a.merge >= 0) {
            const float w2 = window(-p.a.merge);
            m2 = make_float2(x2.x * w2, x2.y * w2);
        }
        // (xs is free: member 0 starts behind the previous pair's closing barrier, member 1 behind the one below)
        if constexpr (kFused) {
            if (tid < M0) {
                const float2 w1 = p.twB[tid];                   // W_B^b: the first stage's twiddle of butterfly b
                float2 u[R0];
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    const int k = tid + M0 * q;
                    const float w = window(k < p.a.nyq ? k : k - B);
                    const float sel = (k == p.a.merge) ? 1.f : 0.f;
                    u[q] = make_float2(fmaf(sel, m2.y, v[q].y * w), fmaf(sel, m2.x, v[q].x * w));   // swapped
                }
                dft_p<R0>(u);
                float2 pw[R0];
                twiddle_powers<R0>(w1, pw);
#pragma unroll
                for (int q = 1; q < R0; ++q) u[dft_slot<R0>(q)] = cmul(u[dft_slot<R0>(q)], pw[q]);
#pragma unroll
                for (int q = 0; q < R0; ++q) xs[tid + M0 * q] = u[dft_slot<R0>(q)];
            }
            lds_barrier();
        } else {
#pragma unroll
            for (int it = 0; it < NV; ++it) {
                const int k = tid + T * it;
                if (k < B) {
                    const float w = window(k < p.a.nyq ? k : k - B);
                    const float sel = (k == p.a.merge) ? 1.f : 0.f;
                    xs[k] = make_float2(fmaf(sel, m2.y, v[it].y * w), fmaf(sel, m2.x, v[it].x * w));   // swapped
                }
            }
            lds_barrier();
            chain_stage<B, R0, B, T>(xs, p.twB, tid);
            lds_barrier();
        }
        chain_stage<B, R1, B / R0, T>(xs, p.twB, tid);
        lds_barrier();
        float2 y[R2];
        const int kb = chain_last<B, R0, R1, R2, T>(xs, tid, y);
        if constexpr (mem == 0) issue_gather(c1);               // member 1's bins fly under member 0's arctangents
        lds_barrier();                                          // every read of xs is done (th1 overlays it; member 1 refills it)
        if (tid < B / R2) {
            float* dst = mem ? th1 : ds;
#pragma unroll
            for (int q = 0; q < R2; ++q) {
                const float2 z = y[dft_slot<R2>(q)];           // swapped: re = z.y, im = z.x
                dst[kb + (B / R2) * q] = atan2_over_pi(z.x, z.y);
            }
        }
    };
    member(std::integral_constant<int, 0>{});
    member(std::integral_constant<int, 1>{});
    lds_barrier();

    // ---- u[t] = d0[t] + j d1[t]: the wrapped phase steps of both channels (fm.py:60-65) ---------------------------
    // two adjacent samples per thread: theta[t0 - 1] and the aligned pair (theta[t0], theta[t0 + 1]) of each member,
    // one 16-byte store of (u[t0], u[t0 + 1])
    {
        static_assert(B % 2 == 0, "pairs of samples");
        constexpr int NP = (B / 2 + T - 1) / T;
        float4 u[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            int t0 = 2 * (tid + T * it);
            t0 = t0 < B ? t0 : B - 2;
            const float2 a = *reinterpret_cast<const float2*>(&ds[t0]);
            const float2 b = *reinterpret_cast<const float2*>(&th1[t0]);
            const float ap = ds[t0 > 0 ? t0 - 1 : 0], bp = th1[t0 > 0 ? t0 - 1 : 0];
            u[it] = make_float4(t0 > 0 ? phase_step_wrapped(a.x, ap) : 0.f, t0 > 0 ? phase_step_wrapped(b.x, bp) : 0.f,
                                phase_step_wrapped(a.y, a.x), phase_step_wrapped(b.y, b.x));
        }
        lds_barrier();                                          // th1 is overwritten below, ds is free from here
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int t0 = 2 * (tid + T * it);
            if (t0 < B) *reinterpret_cast<float4*>(&xs[t0]) = u[it];
        }
    }
    // The decimation's weights (A/2 + 1 folded Hamming values times 1/B) go to ds now: their global loads fly under
    // FFT_B's first two stages instead of standing between its last stage and the stores.
    {
        constexpr int NW = (A / 2 + 1 + T - 1) / T;
        float wv[NW];
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const int k = tid + T * it;
            wv[it] = p.a.wr[k <= A / 2 ? k : A / 2] * p.a.scale;
        }
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const int k = tid + T * it;
            if (k <= A / 2) ds[k] = wv[it];
        }
        if constexpr (DEEMPH) {                                 // MFM: taps and both members' carried state ride along
            if (tid < 51) ds[kScratch + tid] = p.a.deemph_taps[tid];
            if (tid >= 192 && tid < 192 + 51) ds[kScratch + 12 + tid] = p.a.deemph_taps[51 + tid - 192];   // sfx -> [204, 255)
            if (tid >= 64 && tid < 64 + 50) ds[kScratch + tid] = p.a.deemph_state[(int64_t)c0 * 50 + (tid - 64)];
            if (tid >= 128 && tid < 128 + 50) ds[kScratch + tid] = p.a.deemph_state[(int64_t)c1 * 50 + (tid - 128)];
        }
    }
    lds_barrier();

    // ---- FFT_B(u), decimation to A (decimate.py:48 for the packed pair: the weight is real and even) ---------------
    chain_stage<B, R0, B, T>(xs, p.twB, tid);
    lds_barrier();
    chain_stage<B, R1, B / R0, T>(xs, p.twB, tid);
    lds_barrier();
    {
        float2 y[R2];
        const int kb = chain_last<B, R0, R1, R2, T>(xs, tid, y);
        // rows of the short spectrum: kappa = k (k < A/2), k - (B - A) (k > B - A/2), the two Nyquist bins meet in A/2
        // (k = B - A/2 is parked in slot A and added by one thread)
        lds_barrier();                                          // every read of xs is done
        if (tid < B / R2) {
#pragma unroll
            for (int q = 0; q < R2; ++q) {
                const int k = kb + (B / R2) * q;
                // kb < B / R2: the bins of this q lie in [(B/R2) q, (B/R2)(q + 1)); a range that misses both kept bands
                // is decided at compile time, and the butterfly outputs nobody uses are not computed at all
                if ((B / R2) * q > A / 2 && (B / R2) * (q + 1) <= B - A / 2) continue;
                int kap = -1;
                if (k <= A / 2) kap = k;
                else if (k > B - A / 2) kap = k - (B - A);
                else if (k == B - A / 2) kap = A;
                if (kap >= 0) {
                    const float w = ds[k <= A / 2 ? k : B - k];
                    const float2 z = y[dft_slot<R2>(q)];
                    const float2 val = make_float2(z.x * w, z.y * w);
                    xs[kap] = make_float2(val.y, val.x);        // swapped: inverse transform
                    if (k == 0 && p.a.dc != nullptr) {          // mean of each member's audio (the DC bin)
                        p.a.dc[c0] = make_float2(val.x, 0.f);
                        if (has1) p.a.dc[c1] = make_float2(val.y, 0.f);
                    }
                    if constexpr (DEEMPH) {
                        if (k == 0) {                            // ... kept on chip for the mean of the filter output
                            ds[kScratch + 200] = val.x;
                            ds[kScratch + 201] = val.y;
                        }
                    }
                }
            }
        }
    }
    lds_barrier();
    if (tid == 0) {
        const float2 a = xs[A / 2], b = xs[A];
        xs[A / 2] = make_float2(a.x + b.x, a.y + b.y);
    }
    // the next pair's first member: under IFFT_A (the last pair re-reads its own bins -- an unconditional definition
    // is what lets the registers of the previous gather go)
    issue_gather(2 * (pair + (int)gridDim.x < npairs ? pair + (int)gridDim.x : pair));
    lds_barrier();

    // ---- IFFT_A: real part -> member 0, imaginary part -> member 1 -----------------------------------------------
    chain_stage<A, Q0, A, T>(xs, p.twA, tid);
    lds_barrier();
    chain_stage<A, Q1, A / Q0, T>(xs, p.twA, tid);
    lds_barrier();
    {
        float2 y[Q2];
        const int kb = chain_last<A, Q0, Q1, Q2, T>(xs, tid, y);
        lds_barrier();
        if (tid < A / Q2) {
#pragma unroll
            for (int q = 0; q < Q2; ++q) xs[aswz(kb + (A / Q2) * q)] = y[dft_slot<Q2>(q)];    // natural order, swapped
        }
    }
    lds_barrier();
    float* out0 = p.a.audio + (int64_t)c0 * A;
    float* out1 = p.a.audio + (int64_t)c1 * A;
    if constexpr (DEEMPH) {
        // ---- MFM (mfm.py:62-66): y = lfilter(b, v, zi), y -= mean(y), clip -- both members at once (xs[n] = (v1, v0)) ------
        // b[0] = 0 and b[j] = c x^(j-1): a truncated geometric series obeys y[n] = x y[n-1] + b[1] v[n-1] - x b[50] v[n-51]
        // (kernels.hip, k_fir51), a stable recursion (|x| < 1): a lane's run of 32 outputs is one 50-tap sum and 31
        // three-term steps -- 5 LDS reads per output instead of 100.  The first 56 outputs still see the carried state
        // (zi[n], n < 50) or would need inputs of the previous buffer: one plain sum each (wave 1).
        // The mean of y follows from sums that are already here (k_fir51): sum_n y[n] = (sum b) A dc + sum_{i<50}
        // (zi[i] - v[A-1-i] sum_{j>i} b[j]), dc = the DC bin of v -- so every output is finished and stored as it appears.
        const float* tp = ds + kScratch;
        constexpr int PERD = 32, HEAD = 56;
        static_assert(A > HEAD + PERD, "audio shorter than the filter");
        constexpr int NRUN = (A - HEAD + PERD - 1) / PERD, NR = (NRUN + T - 1) / T;
        float2 term = make_float2(0.f, 0.f), hacc = make_float2(0.f, 0.f);
        if (tid < 50) {
            // wave 0: this thread's share of the mean, and element `tid` of the NEXT buffer's carried state (k_fir_state):
            // zf[s] = sum_i b[s + 1 + i] v[A - 1 - i].  (Fixed trip counts with a predicate: the LDS reads pipeline.)
            const float2 vl = xs[aswz(A - 1 - tid)];
            const float sf = ds[kScratch + 204 + tid];
            term = make_float2(ds[kScratch + 128 + tid] - vl.x * sf, ds[kScratch + 64 + tid] - vl.y * sf);
            float2 zf = make_float2(0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 50; ++i) {
                const float2 vv = xs[aswz(A - 1 - i)];
                const float bj = (i < 50 - tid) ? tp[tid + 1 + i] : 0.f;
                zf = make_float2(fmaf(bj, vv.x, zf.x), fmaf(bj, vv.y, zf.y));
            }
            p.a.deemph_state[(int64_t)c0 * 50 + tid] = zf.y;
            if (has1) p.a.deemph_state[(int64_t)c1 * 50 + tid] = zf.x;
        } else if (tid >= 64 && tid < 64 + HEAD) {
            // wave 1: the head -- outputs that still see the carried state, one plain sum each (finished after the barrier)
            const int n = tid - 64;
            hacc = n < 50 ? make_float2(ds[kScratch + 128 + n], ds[kScratch + 64 + n]) : make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 1; j <= 50; ++j) {
                const float2 vv = xs[aswz(j <= n ? n - j : 0)];
                const float bj = j <= n ? tp[j] : 0.f;
                hacc = make_float2(fmaf(bj, vv.x, hacc.x), fmaf(bj, vv.y, hacc.y));
            }
        }
        if (tid < 64) {
            for (int off = 32; off > 0; off >>= 1) {
                term.x += __shfl_down(term.x, off, 64);
                term.y += __shfl_down(term.y, off, 64);
            }
            if (tid == 0) {
                const float bsum = ds[kScratch + 204];           // sfx[0] = sum_{j >= 1} b[j] (b[0] = 0)
                const float inv = 1.0f / (float)A;
                ds[kScratch + 192] = fmaf(term.x, inv, bsum * ds[kScratch + 201]);   // member 1 (dc in .y of the swapped pair)
                ds[kScratch + 193] = fmaf(term.y, inv, bsum * ds[kScratch + 200]);   // member 0
            }
        }
        lds_barrier();
        const float2 mean = make_float2(ds[kScratch + 192], ds[kScratch + 193]);
        const float b1 = tp[1], bx = tp[2] / tp[1], xb50 = bx * tp[50];
        auto fin = [](float y, float m) -> float {
            const float t = y - m;
            return (t < -0.999f) ? -0.999f : ((t > 0.999f) ? 0.999f : t);   // NaN stays NaN like np.clip
        };
        if (tid >= 64 && tid < 64 + HEAD) {
            out0[tid - 64] = fin(hacc.y, mean.y);
            if (has1) out1[tid - 64] = fin(hacc.x, mean.x);
        }
#pragma unroll 1
        for (int r = 0; r < NR; ++r) {
            const int run = tid + T * r;
            const int n0 = HEAD + PERD * run;
            if (run < NRUN) {
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll 10
                for (int j = 1; j <= 50; ++j) {
                    const float2 vv = xs[aswz(n0 - j)];
                    const float bj = tp[j];
                    acc = make_float2(fmaf(bj, vv.x, acc.x), fmaf(bj, vv.y, acc.y));
                }
#pragma unroll
                for (int g4 = 0; g4 < PERD / 4; ++g4) {
                    float2 q4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (g4 > 0 || e > 0) {
                            int n = n0 + 4 * g4 + e;
                            n = n < A ? n : A - 1;                // (a ragged last run repeats its last sample: not stored)
                            const float2 va = xs[aswz(n - 1)], vb = xs[aswz(n - 51)];
                            acc = make_float2(fmaf(bx, acc.x, fmaf(b1, va.x, -xb50 * vb.x)),
                                              fmaf(bx, acc.y, fmaf(b1, va.y, -xb50 * vb.y)));
                        }
                        q4[e] = acc;
                    }
                    const int n4 = n0 + 4 * g4;
                    if ((A % 4 == 0) && n4 + 4 <= A) {
                        *reinterpret_cast<float4*>(out0 + n4) =
                            make_float4(fin(q4[0].y, mean.y), fin(q4[1].y, mean.y), fin(q4[2].y, mean.y), fin(q4[3].y, mean.y));
                        if (has1)
                            *reinterpret_cast<float4*>(out1 + n4) =
                                make_float4(fin(q4[0].x, mean.x), fin(q4[1].x, mean.x), fin(q4[2].x, mean.x), fin(q4[3].x, mean.x));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n4 + e < A) {
                                out0[n4 + e] = fin(q4[e].y, mean.y);
                                if (has1) out1[n4 + e] = fin(q4[e].x, mean.x);
                            }
                        }
                    }
                }
            }
        }
    } else if constexpr (A % 4 == 0) {
        for (int i = tid; i < A / 4; i += T) {
            const float2 s0 = xs[4 * i], s1 = xs[4 * i + 1], s2 = xs[4 * i + 2], s3 = xs[4 * i + 3];
            reinterpret_cast<float4*>(out0)[i] = make_float4(s0.y, s1.y, s2.y, s3.y);
            if (has1) reinterpret_cast<float4*>(out1)[i] = make_float4(s0.x, s1.x, s2.x, s3.x);
        }
    } else {
        for (int i = tid; i < A; i += T) {
            const float2 s0 = xs[i];
            out0[i] = s0.y;
            if (has1) out1[i] = s0.x;
        }
    }
    lds_barrier();                                              // the next pair refills xs
    }   // pairs of this workgroup
}

// (B; R0, R1, R2 | A; Q0, Q1, Q2 | threads): each stage has at most `threads` butterflies.
#define RCFM_LDS_CHAIN_T 512   // 640 threads (one sweep in every stage) spill 87 dwords at 168 VGPRs: 1.13 vs 0.81 ms on cfg5
#define RCFM_LDS_CHAINS(X)                          \
    X(12500, 25, 20, 25, 8000, 20, 20, 20, RCFM_LDS_CHAIN_T)     \
    X(12500, 25, 20, 25, 6250, 25, 10, 25, 640)     \
    X(10000, 20, 20, 25, 8000, 20, 20, 20, 512)     \
    X(12000, 24, 20, 25, 8000, 20, 20, 20, 640)     \
    X(12500, 25, 20, 25, 5000, 20, 10, 25, 512)     \
    X(10000, 20, 20, 25, 5000, 20, 10, 25, 512)     \
    X(12000, 24, 20, 25, 6000, 24, 10, 25, 640)     \
    X(8000, 20, 20, 20, 4000, 20, 10, 20, 512)

const float2* twiddle_table(int n) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::unique_ptr<DeviceBuffer>> tables;   // (device, n)
    int dev = 0;
    RC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = tables.find({dev, n});
    if (it == tables.end()) {
        std::vector<float2> tw((size_t)n);
        for (int e = 0; e < n; ++e) {
            const double a = -6.28318530717958647692 * (double)e / (double)n;
            tw[(size_t)e] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        auto buf = std::make_unique<DeviceBuffer>();
        buf->upload(tw.data(), tw.size() * sizeof(float2));
        it = tables.emplace(std::make_pair(dev, n), std::move(buf)).first;
    }
    return it->second->as<float2>();
}

}  // namespace

bool lds_chain_supported(int B, int A) {
#define RCFM_CASE(B_, R0, R1, R2, A_, Q0, Q1, Q2, T_) \
    if (B == B_ && A == A_) return true;
    RCFM_LDS_CHAINS(RCFM_CASE)
#undef RCFM_CASE
    return false;
}

// The 640-thread instantiations are capped at 168 VGPRs and spill ~190 dwords with the de-emphasis inside: those
// geometries keep the de-emphasis launches.
bool lds_chain_deemph_supported(int B, int A) {
#define RCFM_CASE(B_, R0, R1, R2, A_, Q0, Q1, Q2, T_) \
    if (B == B_ && A == A_) return (T_) <= 512;
    RCFM_LDS_CHAINS(RCFM_CASE)
#undef RCFM_CASE
    return false;
}

bool launch_lds_chain(int B, int A, const LdsChainArgs& args, hipStream_t stream) {
    if (args.count <= 0) return true;
    if (!lds_chain_supported(B, A)) return false;
    ChainDev p;
    p.a = args;
    const double a0 = 0.5, a1 = 1.0 - a0;                      // fftshifted periodic Hann (tuner.py:156-157)
    p.two_pi_over_n = (float)(6.28318530717958647692 / (double)args.N);
    p.delta = (args.N % 2) ? (float)(3.14159265358979323846 / (double)args.N) : 0.f;
    p.c0 = (float)(a0 + a1);
    p.c1 = (float)(-a1 / 2.0);
    p.c2 = (float)(a1 / 24.0);
    p.c3 = (float)(-a1 / 720.0);
    p.twB = twiddle_table(B);
    p.twA = twiddle_table(A);
    // one workgroup per CU is all the LDS allows: as many workgroups as CUs, each walking its pairs (the next pair's
    // bins are fetched under the current pair's last transform)
    const unsigned pairs = (unsigned)((args.count + 1) / 2);
    const dim3 grid(std::min<unsigned>(pairs, (unsigned)FftEngine::compute_units()));
    const bool deemph = args.deemph_taps != nullptr;
    RC_REQUIRE(!deemph || (args.deemph_state != nullptr && lds_chain_deemph_supported(B, A)), RCFM_ERR_ARG,
               "de-emphasis on chip: no carried state, or a geometry without that form");
#define RCFM_CASE(B_, R0, R1, R2, A_, Q0, Q1, Q2, T_)                                                              \
    if (B == B_ && A == A_) {                                                                                      \
        if constexpr ((T_) <= 512) {                                                                               \
            if (deemph)                                                                                            \
                hipLaunchKernelGGL((k_fm_lds<B_, R0, R1, R2, A_, Q0, Q1, Q2, T_, true>), grid, dim3(T_), 0, stream, p); \
        }                                                                                                          \
        if (!deemph)                                                                                               \
            hipLaunchKernelGGL((k_fm_lds<B_, R0, R1, R2, A_, Q0, Q1, Q2, T_, false>), grid, dim3(T_), 0, stream, p); \
        RC_HIP(hipGetLastError());                                                                                 \
        return true;                                                                                               \
    }
    RCFM_LDS_CHAINS(RCFM_CASE)
#undef RCFM_CASE
    return false;
}

}  // namespace rcfm
