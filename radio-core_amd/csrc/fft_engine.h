// Hand-written multi-pass FFT for gfx950 (fft_engine.hip / fft_kernel.h).
//
// A transform of length n = n1*n2*...*np (p = 2..4) runs as p "passes".  One
// workgroup of a pass owns a tile of W = 16 adjacent lines x L points: it loads the
// tile in 128-byte coalesced segments, runs a mixed-radix decimation-in-frequency
// FFT along L entirely in LDS, applies the inter-pass twiddle and writes the tile
// back, again in 128-byte segments.  Each pass reads and writes every element
// exactly once; there are no separate transpose or twiddle kernels (rocFFT spends
// 5-6 launches on the same lengths, profiles/r01_a_*).
//
// Index maps (decimation in time across passes, natural order in and out), with
// m_t = n_{t+1} * ... * n_p:
//   pass t < p : lines (k_1..k_{t-1}; j in [0, m_t)), point l at  sum k_s m_s + l m_t + j,
//                written back in place (l -> k_t) times W_n^((n / m_{t-1}) j k_t);
//   pass p     : lines (k_1..k_{p-1}), contiguous n_p points; output k_p lands at
//                k_1 + n_1 (k_2 + n_2 (... + n_{p-1} k_p)); tiles take 16 adjacent k_1.
#pragma once

#include <cstdint>

#include "common.h"

namespace rcfm {

constexpr int kFftTileW = 16;      // lines per tile: 16 x 8 B = one 128-byte segment
constexpr int kFftMaxStages = 8;
constexpr int kFftMaxPasses = 4;
constexpr int kFftMaxL = 512;      // longest in-LDS transform (tile + tables < 80 KiB: 2 workgroups per CU)
// "Big" tiles (one 1024-thread workgroup per CU, tile up to 87 KiB of LDS): only where they save a
// whole pass over the data -- lengths that need 4 passes of <= 512 but split into 3 of <= 640
// (N = 2.4e8 = 600 x 625 x 640).  Plain load / store functors only.
constexpr int kFftBigL = 640;

// One pass.  Lines are indexed by (o1, o2, i): a tile covers 16 adjacent i.
//   input  point l of a line: in [batch*in_batch  + o1*in_o1  + o2*in_o2  + i*in_i  + l*in_l ]
//   output point k of a line: out[batch*out_batch + o1*out_o1 + o2*out_o2 + i*out_i + k*out_k]
//   multiplied by W_n^((o1*tw_o1 + o2*tw_o2 + i*tw_i) * k) when has_twiddle (the product is < n).
struct FftPass {
    int L;
    int nstages;
    int radix[kFftMaxStages];
    int64_t n_o1, n_o2, n_inner;
    int64_t in_o1, in_o2, in_i, in_l;
    int64_t out_o1, out_o2, out_i, out_k;
    int64_t tw_o1, tw_o2, tw_i;
    int has_twiddle;
    int load_along_l;   // 1: a line is contiguous in memory (in_l == 1): lanes run along l when loading
    // Tile-blocked hand-over between two strided passes (0 = the plain layout, where tile t of 16 lines starts 16 in_i /
    // 16 elements after tile t - 1): tile t starts at t * in_t / t * out_t.  The first pass of a transform that does not
    // fit the Infinity Cache writes each of its tiles as ONE contiguous run (out_t = 16 L, out_k = 16) instead of L
    // segments a row pitch of megabytes apart, and the second pass reads that layout (fft_engine.hip).
    int64_t in_t, out_t;
    // 1: the launch walks (o1, o2, tile) as ONE flat tile index in XCD-aware order (each XCD owns a contiguous eighth of
    // all tiles) instead of grid.y = outer lines: for passes whose neighbouring tiles share 128-byte lines although
    // their rows are short (the padded-rows layout's first pass: rows of 37.5 lines, fft_plan_describe layout 2).
    int32_t flat_outer;
    int32_t reserved_;
};

struct FftPlanDesc {
    int64_t n;
    int npass;
    int fine_bits;      // W_n^e = coarse[e >> fine_bits] * cis(-2 pi (e & mask) / n)
    // Elements one signal occupies in the scratch buffer between passes.  Two-pass plans pad
    // the scratch rows to a multiple of 16 points so that the first pass writes, and the last
    // pass reads, whole 128-byte lines even when n_2 is not a multiple of 16 (B = 240 000 = 480 x 500).
    int64_t tmp_stride;
    FftPass pass[kFftMaxPasses];
};

// Factorises n (radices 2, 3, 4, 5, 6, 7, 8, 10) into passes.  Returns false when n has
// another prime factor, is too small to tile, or does not fit 4 passes; callers then
// fall back to rocFFT.
// max_l (<= kFftMaxL, 0 = default) caps the per-pass length; tests use it to force deep plans.
// `forced` (nforced factors whose product is n) overrides the planner's choice of pass lengths.
// `layout` of the intermediate arrays of a three-pass plan:
//   -1  the engine decides: tile-blocked hand-over between the first two passes (FftPass::out_t) when the transform does
//       not fit the Infinity Cache, else plain
//    0  plain: every strided pass writes where it read (safe in place)
//    1  tile-blocked whenever the lengths allow it (tests)
//    2  padded rows: the scratch rows of n_3 points get a pitch of whole 128-byte lines ([k_1][k_2][pitch]) -- a PRIVATE
//       layout (tmp_stride = n_1 n_2 pitch > n, never the caller's in / out) that lets an order of the pass lengths
//       whose LAST pass stores aligned segments (n_1 n_2 a multiple of 16) keep every other write and all reads but the
//       first pass's aligned too, even when n_3 is not a multiple of 16 (N = 2.4e8 as 640 x 625 x 600: pitch 608)
bool fft_plan_describe(int64_t n, FftPlanDesc* out, int max_l = 0, const int64_t* forced = nullptr,
                       int nforced = 0, int layout = -1);

// For a three-pass default plan whose last pass stores segments that straddle 128-byte lines (n_1 n_2 not a multiple of
// 16): an order of the same three lengths with aligned stores, to be run in layout 2.  false: none / not needed.
bool fft_plan_aligned_order(const FftPlanDesc& plan, int64_t* lengths);

// Device-side view of one pass, handed to the kernel by value.
struct FftPassDev {
    FftPass p;
    const float2* stage_tw;   // W_L^e, e in [0, L)
    const uint16_t* pos;      // LDS row that holds output k after the in-place DIF stages
    const float2* coarse;     // W_n^(c << fine_bits)
    float fine_step;          // 2 pi / n
    int fine_bits;
    int64_t in_batch, out_batch;
};

// Rows of the transform's output to keep (row r = output elements [r n_1, (r+1) n_1), n_1 = the plan's first
// pass length): lo <= hi keeps rows lo..hi, lo > hi keeps rows >= lo and rows <= hi (a window that wraps).
struct FftRowWindow {
    int lo, hi;
    // halo > 0 (batch 1, forward): outputs [0, halo) are also stored at out[n + g], outputs [n - halo, n) at
    // out[g - n] -- `out` must have `halo` elements of room on both sides.  lo = 0, hi = rows - 1 keeps every row.
    int halo = 0;
};

class FftEngine {
   public:
    explicit FftEngine(int64_t n);
    // Same transform with the pass lengths given (e.g. the planner's two factors swapped, so that
    // this plan's last pass tiles exactly like another plan's first pass: fused_passes.h).
    FftEngine(int64_t n, const int64_t* factors, int nfactors, int layout = -1);
    // The plan in the PLAIN layout (every strided pass keeps the layout it reads: safe in place, what every fused caller
    // of pass_dev() assumes).  desc_blocked() is the same plan with the tile-blocked hand-over between its first two
    // passes (FftPass::in_t / out_t) when the planner chose one -- c2c() runs it only when no pass runs in place.
    const FftPlanDesc& desc() const { return desc_; }
    bool has_blocked() const { return has_blk_; }
    const FftPlanDesc& desc_blocked() const { return has_blk_ ? desc_blk_ : desc_; }
    int npass() const { return desc_.npass; }
    int64_t tmp_stride() const { return desc_.tmp_stride; }   // scratch elements per signal (>= n)
    // Unnormalised c2c transform of `batch` contiguous length-n signals (distance n).
    // `tmp` holds batch * tmp_stride() complex values; in == out is allowed, tmp must be distinct.
    // A plan with a tile-blocked hand-over reads another address set in its second pass than it writes, so it runs only
    // when in, out and tmp are three distinct arrays (no pass in place); otherwise the plain layout of the same plan runs.
    // inverse = conjugate transform (no 1/n); every output is multiplied by `scale`.
    // keep (forward transforms only): the last pass stores only those rows of the output; the rest of
    // `out` is left untouched.
    // tmp2 (optional, tmp_stride() elements per signal, distinct from everything else): a second scratch array -- a
    // three-pass plan then runs in -> tmp2 -> tmp -> out with no pass in place (what the padded-rows layout, whose
    // intermediates do not fit `out`, needs to ping-pong).
    void c2c(const float2* in, float2* out, float2* tmp, int batch, bool inverse, float scale,
             hipStream_t stream, const FftRowWindow* keep = nullptr, float2* tmp2 = nullptr) const;
    int64_t row_length() const { return desc_.pass[0].L; }   // n_1
    FftPassDev pass_dev(int t, int64_t in_batch, int64_t out_batch, bool blocked = false) const;
    static size_t lds_bytes(int L);
    static dim3 grid(const FftPass& p, int batch);
    static int compute_units();   // CUs of the current device (256 on MI355X)

   private:
    void build_tables();
    void split_layouts();
    FftPlanDesc desc_;       // plain layout
    FftPlanDesc desc_blk_;   // tile-blocked hand-over (valid when has_blk_)
    bool has_blk_ = false;
    DeviceBuffer stage_tw_[kFftMaxPasses];
    DeviceBuffer pos_[kFftMaxPasses];
    DeviceBuffer coarse_;
};

}  // namespace rcfm
