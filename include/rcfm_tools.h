/*
 * rcfm_tools.h -- the part of librcfm.so's C ABI that exists for this repository's own tools, benches and tests:
 * placement arenas, FFT plan description and explicit plans, the rocFFT A/B partner, the per-handle switches between
 * kernel forms, per-stage profiling.  A host that replaces radio-core's device seam binds rcfm.h only
 * (INTEGRATION.md section 2); nothing declared here is needed to run the path, and results never depend on it.
 * Same conventions as rcfm.h (status codes, device pointers, streams).
 */
#ifndef RCFM_TOOLS_H
#define RCFM_TOOLS_H

#include "rcfm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rcfm_arena_s* rcfm_arena_t;

/* ---- placement of the library's workspaces (no reference counterpart: numpy / cupy allocate per call) ------------
 * Where hipMalloc puts a multi-GB workspace moves the kernels that stream through it by 1.5 - 4 % of a cfg4 buffer, and
 * the draw differs from handle to handle and from process to process (profiles/r04_k_placement.md).  A host that wants
 * ONE draw for a whole handle set, or wants to choose it (create several arenas, time its own workload in each, keep
 * the best), creates an arena: device memory taken in blocks of `block_bytes` (0: 1 GiB blocks, taken on demand; a
 * request larger than a block gets a block of its own) and handed out by a bump pointer on 2 MiB boundaries.
 *   bind      arena != NULL: tuner / demodulator handles CREATED by this thread from now on belong to the arena -- every
 *             workspace of 1 MiB or more they ever allocate (at creation and later, whichever thread calls) comes from
 *             it; NULL: back to hipMalloc per workspace (the default; existing handles keep their arena)
 *   stats     bytes reserved from the device, bytes handed out, pieces still owned by live handles
 *   destroy   RCFM_ERR_STATE while a handle created inside the arena is alive (pieces return with the arena, not one
 *             by one: a handle set that is rebuilt often should get a fresh arena)
 * Results do not depend on any of this.  bench.py --arena 1, tools/placement_sets.py. */
int rcfm_arena_create(size_t block_bytes, rcfm_arena_t* out);
/* The same over memory the HOST owns (a block of its own allocator -- a torch tensor, an rcfm_malloc block): the
 * library's workspaces then live where the host decided, and two handle sets built one after the other inside arenas
 * over the same block get the same addresses (tools/ab_libs.py compares two builds of the library that way, free of
 * placement noise).  The memory must outlive the arena; it is not freed by rcfm_arena_destroy.  What does not fit comes
 * from hipMalloc. */
int rcfm_arena_adopt(void* base, size_t bytes, rcfm_arena_t* out);
int rcfm_arena_bind(rcfm_arena_t arena);
int rcfm_arena_stats(rcfm_arena_t arena, size_t* reserved_bytes, size_t* used_bytes, size_t* live_pieces);
int rcfm_arena_destroy(rcfm_arena_t arena);

/* ---- kernel forms ---------------------------------------------------------------------------------------------- */

/* Which tile width rcfm_tuner_run uses (rcfm_pipeline_run passes the demodulator's RCFM_OPT_NARROW_TILES instead):
 * 0 = always 16 lines per tile, 1 (default) = 8 when a launch has fewer than two 16-line tiles per CU, 2 = always 8. */
enum { RCFM_TUNER_OPT_NARROW_TILES = 1,
       /* 1 (default): a wideband FFT whose default plan's last pass stores segments that straddle 128-byte lines
        * (N = 2.4e8 = 600 x 625 x 640: output stride 375 000 bins = 64 mod 128 bytes) runs the aligned order of the same
        * pass lengths in the padded-rows layout instead (640 x 625 x 600, rcfm_fft_describe_plan layout 2) -- while the
        * handle uses its own spectrum storage (an attached one has no room for the padded intermediate); 0: always the
        * default plan.  Takes effect with the next rcfm_tuner_load; same spectrum within float32 rounding. */
       RCFM_TUNER_OPT_ALIGNED_PLAN = 2 };
int rcfm_tuner_set_option(rcfm_tuner_t t, int option, int value);

/* Which forms of the chain a handle may use (no reference counterpart: the reference has one form of everything).
 * All default to 1.  The results do not depend on them beyond float32 rounding -- tests/test_hip_configs.py compares
 * every channel of a full-size buffer between the default handle and one with all three switched off, which shares no
 * kernel schedule with it.
 *   RCFM_OPT_LDS_CHAIN    narrow FM / MFM channels run tuner + demodulator of a channel pair in one workgroup (0: the
 *                         multi-pass launches)
 *   RCFM_OPT_FUSED_TILES  two transforms per tile: pilot chain, Hilbert mask, stereo mix, spectral decimation between
 *                         transforms (0: one transform per launch, the intermediate spectra go through memory); sets
 *                         the two switches below together
 *   RCFM_OPT_PILOT_CHAIN  ... only the tiles around the Hilbert mask (pilot pair FFT -> mask -> IFFT -> stereo matrix)
 *   RCFM_OPT_DECIM_TILE   ... only the spectral decimation between FFT_B's last pass and IFFT_A's first
 *   RCFM_OPT_PILOT_BLOCKED WBFM's mono signal and pilot band travel from the pilot stage to the pilot chain in a tile-blocked
 *                         layout (the chain's 16-line tiles read contiguous runs; 0: natural order, half-line reads)
 *   RCFM_OPT_LDS_DEEMPH   narrow MFM channels: de-emphasis, mean removal and clip inside the LDS chain (0: the
 *                         de-emphasis launches behind it)
 *   RCFM_OPT_PHASE_LINK   the tuner hands the demodulator angle(x) / pi as float32 (0: complex64 samples, as
 *                         tuner.py:161 returns them) */
enum { RCFM_OPT_LDS_CHAIN = 1, RCFM_OPT_FUSED_TILES = 2, RCFM_OPT_PHASE_LINK = 3, RCFM_OPT_PILOT_CHAIN = 6, RCFM_OPT_DECIM_TILE = 7,
       RCFM_OPT_LDS_DEEMPH = 8, RCFM_OPT_PILOT_BLOCKED = 9 };
/* (rcfm_demod_set_option itself is declared in rcfm.h.)
 * Reads an option back.  RCFM_OPT_PILOT_BLOCKED reads the EFFECTIVE value: 1 only when the switch is on AND this handle's
 * geometry has the layout and the three-launch pilot chain that reads it (what a test needs to know that it compared two
 * different forms).  RCFM_OPT_GRAPH (rcfm.h) reads 0 = off, 1 = on, 1 + k = on and k captured launch chains are being replayed.
 * No reference counterpart. */
int rcfm_demod_get_option(rcfm_demod_t d, int option, int* value);

/* ---- FFT engine: plans ---------------------------------------------------------------------------------------- */

/* Describes how librcfm runs a length-n complex FFT: fills a POD `rcfm_fft_plan`
 * (layout below) and returns 0, or RCFM_ERR_ARG when n is outside the engine
 * (prime factors other than 2, 3, 5, 7; n < 256; more than 4 passes): such lengths use rocFFT. */
typedef struct rcfm_fft_pass {
    int32_t L, nstages, radix[8];
    int64_t n_o1, n_o2, n_inner;
    int64_t in_o1, in_o2, in_i, in_l;
    int64_t out_o1, out_o2, out_i, out_k;
    int64_t tw_o1, tw_o2, tw_i;
    int32_t has_twiddle, load_along_l;
    int64_t in_t, out_t; /* tile-blocked hand-over between strided passes: element offset of tile t (16 lines) = t * in_t /
                            t * out_t; 0 = the plain layout (16 * in_i, 16) */
    int32_t flat_outer;  /* 1: the launch walks (o1, o2, tile) as one flat, XCD-aware tile index (layout 2's first pass) */
    int32_t reserved_;
} rcfm_fft_pass;
typedef struct rcfm_fft_plan {
    int64_t n;
    int32_t npass, fine_bits;
    int64_t tmp_stride; /* scratch elements per signal between passes (>= n) */
    rcfm_fft_pass pass[4];
} rcfm_fft_plan;
int rcfm_fft_describe(int64_t n, int max_l /* 0 = default cap on a pass length */, rcfm_fft_plan* plan);
/* The plan for given pass lengths (what rcfm_fft_c2c_plan runs).  layout of the intermediate arrays of a three-pass plan:
 *   -1 = as the engine decides (tile-blocked hand-over between the first two passes for transforms beyond the Infinity
 *        Cache, else plain), 0 = plain, 1 = tile-blocked whenever the lengths allow it (tests/test_fft_plan.py models it
 *        at small n), 2 = padded rows: scratch rows of n_3 points at a pitch of whole 128-byte lines -- tmp_stride > n, a
 *        private layout; with it an order of the pass lengths whose LAST pass stores aligned segments keeps every other
 *        write aligned as well (N = 2.4e8 as 640 x 625 x 600, what rcfm_tuner_load runs: RCFM_TUNER_OPT_ALIGNED_PLAN). */
int rcfm_fft_describe_plan(int64_t n, const int64_t* pass_lengths, int npass, int layout, rcfm_fft_plan* plan);
/* The same transform with the pass lengths given by the caller (their product is n, each within the engine's tile
 * lengths; RCFM_ERR_ARG otherwise): plan sweeps (tools/plan_sweep.py) and tests that put a tile length into a role the
 * planner does not use it in (tests/test_hip_fft.py). */
int rcfm_fft_c2c_plan(int64_t n, const int64_t* pass_lengths, int npass, int layout, int batch, int inverse,
                      const void* in, void* out, void* stream);
/* The same transform through rocFFT (any n): the A/B partner of rcfm_fft_c2c in
 * tools/bench_fft.py and the fallback for lengths the engine refuses. */
int rcfm_fft_c2c_rocfft(int64_t n, int batch, int inverse, const void* in, void* out, void* stream);

/* ---- measurement ------------------------------------------------------------ */

/* Per-stage timing with HIP events recorded on the stage's own stream (the reference
 * has only timeit around whole calls, tests/benchmark.py:22-24).  A stage is one
 * kernel launch or one FFT execute.  enable(mask): bit i switches stage i on;
 * read(): waits for the recorded events and returns the accumulated milliseconds
 * and the number of bracketed launches since the last reset. */
int rcfm_profile_stage_count(void);
const char* rcfm_profile_stage_name(int stage);
int rcfm_profile_enable(uint64_t stage_mask);
int rcfm_profile_reset(void);
int rcfm_profile_read(int stage, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* RCFM_TOOLS_H */
