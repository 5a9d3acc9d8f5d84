#include "fft_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <vector>

#include "fft_kernel.h"

namespace rcfm {

namespace {

constexpr double kTwoPi = 6.28318530717958647692;

// Largest-radix-first factorisation of an LDS transform length.
bool choose_radices(int L, int* radix, int* nstages) {
    // lengths with a specialised kernel: the radices that kernel was instantiated with
    switch (L) {
#define RCFM_CASE(LEN, A, B, C, D)                          \
    case LEN: {                                             \
        const int r[4] = {A, B, C, D};                      \
        int ns = 0;                                         \
        for (int i = 0; i < 4; ++i)                         \
            if (r[i] > 1) radix[ns++] = r[i];               \
        *nstages = ns;                                      \
        return true;                                        \
    }
        RCFM_FFT_FAST_LENGTHS(RCFM_CASE)
        RCFM_FFT_BIG_LENGTHS(RCFM_CASE)
#undef RCFM_CASE
        default: break;
    }
    static const int kRadix[] = {10, 8, 7, 6, 5, 4, 3, 2};
    int rem = L, ns = 0;
    while (rem > 1) {
        bool found = false;
        for (int r : kRadix) {
            if (rem % r == 0) {
                if (ns == kFftMaxStages) return false;
                radix[ns++] = r;
                rem /= r;
                found = true;
                break;
            }
        }
        if (!found) return false;
    }
    *nstages = ns;
    return ns > 0;
}

bool is_fast_length(int64_t L) {
    switch (L) {
#define RCFM_CASE(LEN, A, B, C, D) case LEN:
        RCFM_FFT_FAST_LENGTHS(RCFM_CASE)
        RCFM_FFT_BIG_LENGTHS(RCFM_CASE)
#undef RCFM_CASE
        return true;
        default: return false;
    }
}

bool smooth235(int64_t n) {   // (and 7 since round 6: the generic tile kernel has a radix-7 butterfly)
    for (int p : {2, 3, 5, 7})
        while (n % p == 0) n /= p;
    return n == 1;
}

// Split n into `np` factors <= kFftMaxL, each >= 16, as balanced as possible, preferring
// factorizations whose tiled dimensions (n_1 and m_1..m_{p-1}) are multiples of 16.
bool split(int64_t n, int np, int max_l, int64_t* f, double* best_cost = nullptr) {
    std::vector<int64_t> divs;
    for (int64_t d = 16; d <= max_l; ++d)
        if (n % d == 0) divs.push_back(d);
    double best = 1e300;
    bool ok = false;
    int64_t cur[kFftMaxPasses];
    // depth-first over divisor choices (np <= 4, |divs| small)
    std::function<void(int, int64_t)> rec = [&](int t, int64_t rem) {
        if (t == np - 1) {
            if (rem < 16 || rem > max_l) return;
            cur[t] = rem;
            int64_t mx = 0, mn = 1 << 30;
            for (int i = 0; i < np; ++i) {
                mx = std::max(mx, cur[i]);
                mn = std::min(mn, cur[i]);
            }
            double cost = (double)mx / (double)mn;
            for (int i = 0; i < np; ++i) {
                if (!is_fast_length(cur[i])) cost += 3.0;   // no compile-time specialised kernel
                // measured per-byte efficiency of the tile classes (plain passes, MI355X): tiles of <= 426 points and the
                // big tiles (two 1024-thread workgroups per CU) move 5.1-5.2 TB/s, the 427..512-point tiles (two
                // 512-thread workgroups, 100+ VGPRs) 4.1-4.9: N = 1e8 runs 8 % faster as 400 x 625 x 400 than as
                // 400 x 500 x 500.  (Forcing three workgroups per CU on the 400-point tiles -- 80 VGPRs, swizzled rows --
                // measured 20 % SLOWER: the weights are empirical, not an occupancy model.)
                if (cur[i] > kFftMaxL) cost += 0.15;
                else if (cur[i] > 426) cost += 0.35;
            }
            cost += 0.02 * (double)cur[0] / (double)mn;     // the first pass has the longest stride: keep it short
            // tail tiles waste lanes: penalise tiled extents that are not multiples of 16
            auto tail = [](int64_t extent) {
                const int64_t tiles = (extent + 15) / 16;
                return (double)(tiles * 16) / (double)extent - 1.0;
            };
            // 128-byte segments that straddle cache lines.  In-place passes: ~10 % slower (the lines they
            // write are the ones they just read, still in L2).  Last pass (fresh lines): 1.75x slower when
            // rows start at arbitrary 8-byte offsets (N = 240M with n_1 = 125), ~15 % when every row starts
            // on a 64-byte boundary (n_1 = 600: two half-line writes).
            auto straddle_write = [](int64_t extent) { return extent % 16 == 0 ? 0.0 : extent % 8 == 0 ? 0.5 : 2.0; };
            int64_t m = n;
            for (int i = 0; i < np - 1; ++i) {
                m /= cur[i];
                cost += 4.0 * tail(m);        // pass i tiles over j in [0, m_i)
                if (m % 16) cost += 0.5;
            }
            cost += 4.0 * tail(cur[0]);       // the last pass tiles over k_1
            cost += straddle_write(cur[0]);
            if (cost < best) {
                best = cost;
                ok = true;
                for (int i = 0; i < np; ++i) f[i] = cur[i];
                if (best_cost) *best_cost = cost;
            }
            return;
        }
        for (int64_t d : divs) {
            if (rem % d) continue;
            cur[t] = d;
            rec(t + 1, rem / d);
        }
    };
    rec(0, n);
    return ok;
}

constexpr bool big_tiles_enabled() { return true; }

// Three-pass plans that hand the first pass's output over tile by tile (fft_plan_describe).
bool hand_over_blocked(int64_t n, const int64_t* f, bool any_size) {
    return (any_size || (size_t)n * sizeof(float2) > ((size_t)256 << 20)) && f[2] % 16 == 0 && (f[1] * f[2]) % 16 == 0;
}

}  // namespace

bool fft_plan_aligned_order(const FftPlanDesc& plan, int64_t* lengths) {
    if (plan.npass != 3 || (size_t)plan.n * sizeof(float2) <= ((size_t)256 << 20)) return false;
    const int64_t f[3] = {plan.pass[0].L, plan.pass[1].L, plan.pass[2].L};
    if ((f[0] * f[1]) % 16 == 0) return false;   // the default order already stores aligned segments
    double best = 1e300;
    bool ok = false;
    const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    for (auto& pm : perm) {
        const int64_t a = f[pm[0]], b = f[pm[1]], c = f[pm[2]];
        if (a % 16 != 0 || (a * b) % 16 != 0) continue;       // whole tiles over k_1, aligned output stride
        const int64_t pitch = (c + 15) / 16 * 16;
        const double cost = (double)pitch / (double)c + 0.02 * (double)a / (double)std::min(a, std::min(b, c));
        if (cost < best) {
            best = cost;
            ok = true;
            lengths[0] = a;
            lengths[1] = b;
            lengths[2] = c;
        }
    }
    return ok;
}

bool fft_plan_describe(int64_t n, FftPlanDesc* out, int max_l, const int64_t* forced, int nforced, int layout) {
    const int blocked = layout == 2 ? 0 : layout;
    const bool default_cap = (max_l <= 0);
    if (max_l <= 0 || max_l > kFftMaxL) max_l = kFftMaxL;
    if (n < 256 || n >= (int64_t(1) << 32) || !smooth235(n)) return false;
    int np = 0;
    int64_t f[kFftMaxPasses];
    if (forced != nullptr) {
        if (nforced < 2 || nforced > kFftMaxPasses) return false;
        int64_t prod = 1;
        for (int i = 0; i < nforced; ++i) {
            if (forced[i] < 16 || forced[i] > kFftBigL) return false;
            f[i] = forced[i];
            prod *= forced[i];
        }
        if (prod != n) return false;
        np = nforced;
    } else {
        for (int cand = 2; cand <= kFftMaxPasses; ++cand) {
            // three passes over big tiles beat four over regular ones (one read + write of the data less)
            if (cand == 4 && default_cap && big_tiles_enabled() && split(n, 3, kFftBigL, f)) {
                bool all_fast = true;
                for (int i = 0; i < 3; ++i) all_fast = all_fast && is_fast_length(f[i]);
                if (all_fast) {
                    np = 3;
                    break;
                }
            }
            double c_reg = 0.0;
            if (split(n, cand, max_l, f, &c_reg)) {
                np = cand;
                // three passes: a big tile among them may beat the best all-regular split
                int64_t fb[kFftMaxPasses];
                double c_big = 0.0;
                if (cand == 3 && default_cap && big_tiles_enabled() && split(n, 3, kFftBigL, fb, &c_big) && c_big < c_reg) {
                    bool all_fast = true;
                    for (int i = 0; i < 3; ++i) all_fast = all_fast && is_fast_length(fb[i]);
                    if (all_fast)
                        for (int i = 0; i < 3; ++i) f[i] = fb[i];
                }
                break;
            }
        }
    }
    if (!np) return false;
    FftPlanDesc d{};
    d.n = n;
    d.npass = np;
    // fine angle below 0.03 rad: 2 pi F / n <= 0.03
    d.fine_bits = 0;
    while ((double)(int64_t(2) << d.fine_bits) * kTwoPi / (double)n <= 0.03) ++d.fine_bits;
    // m[t] = n_{t+1} * ... * n_p  (m[0] = n / n_1)
    int64_t m[kFftMaxPasses + 1];
    m[np - 1] = 1;
    for (int t = np - 2; t >= 0; --t) m[t] = m[t + 1] * f[t + 1];
    for (int t = 0; t < np; ++t) {
        FftPass& p = d.pass[t];
        p = FftPass{};
        p.L = (int)f[t];
        if (!choose_radices(p.L, p.radix, &p.nstages)) return false;
        p.n_o1 = p.n_o2 = 1;
        if (t < np - 1) {
            // lines (k_1 .. k_t-1 ; j in [0, m_t)): strided columns, written back in place
            p.n_inner = m[t];
            p.in_i = p.out_i = 1;
            p.in_l = p.out_k = m[t];
            if (t >= 1) {
                p.n_o1 = f[0];
                p.in_o1 = p.out_o1 = m[0];
            }
            if (t >= 2) {
                p.n_o2 = f[1];
                p.in_o2 = p.out_o2 = m[1];
            }
            p.has_twiddle = 1;
            p.tw_i = n / (m[t] * f[t]);   // n / m_{t-1}
            p.load_along_l = 0;
        } else {
            // last pass: lines (k_1 .. k_p-1) are contiguous runs of n_p points; tiles take 16
            // adjacent k_1; output k_p lands at k_1 + n_1 (k_2 + n_2 (k_3 + n_3 k_p))
            p.n_inner = f[0];
            p.in_i = m[0];
            p.in_l = 1;
            p.out_i = 1;
            int64_t stride = f[0];
            if (np >= 3) {
                p.n_o1 = f[1];
                p.in_o1 = m[1];
                p.out_o1 = stride;
                stride *= f[1];
            }
            if (np >= 4) {
                p.n_o2 = f[2];
                p.in_o2 = m[2];
                p.out_o2 = stride;
                stride *= f[2];
            }
            p.out_k = stride;
            p.has_twiddle = 0;
            p.load_along_l = 1;
        }
    }
    // Transforms of three passes that do not fit the 256 MiB Infinity Cache: the first pass hands its output to the second
    // tile by tile -- tile a of 16 lines j (j = l_2 n_3 + j_2, whole tiles inside one l_2 because n_3 is a multiple of 16) is
    // ONE contiguous run of 16 n_1 points, [a][k_1][16], instead of n_1 segments a row pitch of megabytes apart; the second
    // pass (block k_1, lines j_2, points l_2) finds point l_2 of its tile j_2 / 16 at ((l_2 n_3 / 16 + j_2 / 16) n_1 + k_1) 16.
    // Placement decides how fast a pass WRITES at a pitch of megabytes (profiles/r04_k_placement.md: 20 % between buffer
    // pairs); reads at such a pitch are the less sensitive side.  Same-address A/B at N = 2.4e8: 2.302 -> 2.260 ms
    // (profiles/r05_k_blocked_handover.txt; the non-temporal stores beat sc1 ones again once the runs are contiguous).
    if (np == 3 && blocked != 0 && hand_over_blocked(n, f, blocked > 0)) {
        d.pass[0].out_t = 16 * f[0];
        d.pass[0].out_k = 16;
        d.pass[1].in_t = 16 * f[0];
        d.pass[1].in_o1 = 16;
        d.pass[1].in_l = f[2] * f[0];
    }
    d.tmp_stride = n;
    if (layout == 2) {
        // Padded rows (fft_engine.h): scratch element (k_1 | l_1, k_2 | l_2, j_2) at (k_1 n_2 + k_2) pitch + j_2.  The first
        // pass reads the natural-order input -- line (l_2, j_2) starts at l_2 n_3 + j_2, so for n_3 = 8 mod 16 half of its
        // segments straddle lines (reads only; neighbouring tiles share those lines: flat_outer) -- every other access of
        // the three passes is a whole aligned 128-byte segment.
        if (np != 3) return false;
        const int64_t pitch = (f[2] + 15) / 16 * 16;
        FftPass& p0 = d.pass[0];
        p0.n_o1 = f[1];
        p0.n_inner = f[2];
        p0.in_o1 = f[2];
        p0.in_l = f[1] * f[2];
        p0.out_o1 = pitch;
        p0.out_k = f[1] * pitch;
        p0.tw_o1 = f[2] * p0.tw_i;
        p0.flat_outer = 1;
        FftPass& p1 = d.pass[1];
        p1.n_inner = f[2];
        p1.in_o1 = p1.out_o1 = f[1] * pitch;
        p1.in_l = p1.out_k = pitch;
        FftPass& p2 = d.pass[2];
        p2.in_i = f[1] * pitch;
        p2.in_o1 = pitch;
        d.tmp_stride = f[0] * f[1] * pitch;
    }
    if (np == 2 && (f[1] % 16) != 0) {
        const int64_t pitch = (f[1] + 15) / 16 * 16;
        d.pass[0].out_k = pitch;     // row k_1 of the scratch starts at k_1 * pitch
        d.pass[1].in_i = pitch;
        d.tmp_stride = f[0] * pitch;
    }
    *out = d;
    return true;
}

size_t FftEngine::lds_bytes(int L) { return (size_t)L * (kFftTileW * sizeof(float2) + sizeof(float2) + sizeof(uint16_t)); }

dim3 FftEngine::grid(const FftPass& p, int batch) {
    const int64_t tiles = p.n_o1 * p.n_o2 * ((p.n_inner + kFftTileW - 1) / kFftTileW);
    return dim3((unsigned)tiles, (unsigned)batch, 1);
}

int FftEngine::compute_units() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

FftEngine::FftEngine(int64_t n) {
    RC_REQUIRE(fft_plan_describe(n, &desc_blk_), RCFM_ERR_ARG, "length not supported by the FFT engine");
    split_layouts();
    build_tables();
}

FftEngine::FftEngine(int64_t n, const int64_t* factors, int nfactors, int layout) {
    RC_REQUIRE(fft_plan_describe(n, &desc_blk_, 0, factors, nfactors, layout), RCFM_ERR_ARG,
               "pass lengths not supported by the FFT engine");
    split_layouts();
    build_tables();
}

// desc_blk_ holds what the planner chose; desc_ is the same plan (same pass lengths) in the plain layout.
void FftEngine::split_layouts() {
    has_blk_ = desc_blk_.npass == 3 && desc_blk_.pass[0].out_t != 0;
    if (!has_blk_) {
        desc_ = desc_blk_;
        return;
    }
    int64_t f[kFftMaxPasses];
    for (int t = 0; t < desc_blk_.npass; ++t) f[t] = desc_blk_.pass[t].L;
    RC_REQUIRE(fft_plan_describe(desc_blk_.n, &desc_, 0, f, desc_blk_.npass, 0), RCFM_ERR_RUNTIME,
               "plain layout of a blocked plan");
}

void FftEngine::build_tables() {
    const int64_t n = desc_.n;
    for (int t = 0; t < desc_.npass; ++t) {
        const FftPass& p = desc_.pass[t];
        std::vector<float2> tw(p.L);
        for (int e = 0; e < p.L; ++e) {
            const double a = -kTwoPi * (double)e / (double)p.L;
            tw[e] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        stage_tw_[t].upload(tw.data(), tw.size() * sizeof(float2));
        // slot of output k after the in-place DIF stages: digits of k, least significant first,
        // weigh L/r_1, L/(r_1 r_2), ...
        std::vector<uint16_t> pos(p.L);
        for (int k = 0; k < p.L; ++k) {
            int rem = k, weight = p.L, slot = 0;
            for (int s = 0; s < p.nstages; ++s) {
                weight /= p.radix[s];
                slot += (rem % p.radix[s]) * weight;
                rem /= p.radix[s];
            }
            pos[k] = (uint16_t)slot;
        }
        pos_[t].upload(pos.data(), pos.size() * sizeof(uint16_t));
    }
    const int64_t F = int64_t(1) << desc_.fine_bits;
    const int64_t nc = (n + F - 1) / F;
    std::vector<float2> coarse(nc);
    for (int64_t c = 0; c < nc; ++c) {
        const double a = -kTwoPi * (double)(c * F) / (double)n;
        coarse[c] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    coarse_.upload(coarse.data(), coarse.size() * sizeof(float2));
}

FftPassDev FftEngine::pass_dev(int t, int64_t in_batch, int64_t out_batch, bool blocked) const {
    FftPassDev d;
    d.p = (blocked && has_blk_) ? desc_blk_.pass[t] : desc_.pass[t];
    d.stage_tw = stage_tw_[t].as<float2>();
    d.pos = pos_[t].as<uint16_t>();
    d.coarse = coarse_.as<float2>();
    d.fine_step = (float)(kTwoPi / (double)desc_.n);
    d.fine_bits = desc_.fine_bits;
    d.in_batch = in_batch;
    d.out_batch = out_batch;
    return d;
}

void FftEngine::c2c(const float2* in, float2* out, float2* tmp, int batch, bool inverse, float scale,
                    hipStream_t stream, const FftRowWindow* keep, float2* tmp2) const {
    if (batch <= 0) return;
    const int np = desc_.npass;
    const int64_t n = desc_.n;
    using namespace fftk;
    const int64_t ts = desc_.tmp_stride;
    // Plans of three and four passes on arrays that cannot stay in the 256 MiB Infinity Cache: no pass runs in place.  The
    // intermediate arrays alternate between `tmp` and the caller's `out` (a strided pass keeps the layout, and for np >= 3
    // the scratch rows are not padded: ts == n), ending in tmp -> out.  A tile copy that writes where it reads streams
    // 1.5 - 4 % slower than one whose output lies elsewhere (tools/microbench/dst_offset.hip, profiles/r04_h_dst_offset.md);
    // same-box alternation: cfg4 7.252 -> 7.224 ms (the middle pass of N = 2.4e8), cfg5 1.542 -> 1.534 ms.  Cache-resident
    // transforms keep the in-place middle passes: N = 1e7 (80 MB) lost 3.6 % with twice the footprint.  When out aliases
    // in or tmp the old routing stays.
    const bool ping_pong = np >= 3 && ts == n && in != out && out != tmp && in != tmp &&
                           (size_t)n * (size_t)batch * sizeof(float2) > ((size_t)256 << 20);
    // The tile-blocked hand-over (pass 1 reads another address set than it writes) only when no pass runs in place.
    const bool blk = ping_pong && has_blk_;
    // a second scratch array: in -> tmp2 -> tmp -> out (three passes; the padded-rows layout's way not to run in place)
    const bool two_scratch = np == 3 && tmp2 != nullptr && tmp2 != tmp && tmp2 != in && tmp2 != out && !ping_pong;
    auto mid = [&](int t) -> float2* {   // where pass t < np - 1 writes
        if (two_scratch) return t == 0 ? tmp2 : tmp;
        if (!ping_pong) return tmp;
        return ((np - 2 - t) & 1) ? out : tmp;
    };
    for (int t = 0; t < np; ++t) {
        const bool first = (t == 0), last = (t == np - 1);
        const float2* src = first ? in : mid(t - 1);
        const FftPassDev dev = pass_dev(t, first ? n : ts, last ? n : ts, blk);
        if (last) {
            LoadPlainT<false> ld{src};
            if (inverse)
                launch_fft_pass<kRowsOnly>(dev, batch, ld, StorePlainT<true>{out, scale}, stream);
            else if (keep != nullptr)
                launch_fft_pass<kRowsOnly>(dev, batch, ld,
                                           StoreRowWindow{out, scale, (int)dev.p.n_o1, (int)dev.p.n_o2, keep->lo, keep->hi,
                                                          n, batch == 1 ? keep->halo : 0,
                                                          (size_t)n * (size_t)batch * sizeof(float2) > ((size_t)256 << 20)},
                                           stream);
            else
                launch_fft_pass<kRowsOnly>(dev, batch, ld, StorePlainT<false>{out, scale}, stream);
        } else {
            StorePlainT<false> st{mid(t), 1.0f};
            if (first && inverse)
                launch_fft_pass<kStridedOnly>(dev, batch, LoadPlainT<true>{src}, st, stream);
            else
                launch_fft_pass<kStridedOnly>(dev, batch, LoadPlainT<false>{src}, st, stream);
        }
    }
}

}  // namespace rcfm
