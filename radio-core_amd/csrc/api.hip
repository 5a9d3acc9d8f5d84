// C ABI of librcfm.so (include/rcfm.h): handles, host-side filter / window design,
// and the per-chunk kernel chains of Tuner.run and FM / MFM / WBFM.run.

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <tuple>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the library is opened on the first rcfm_comm_* call (dlopen), never linked

#include "common.h"
#include "fft_engine.h"
#include "fft_plan.h"
#include "fused_passes.h"
#include "kernels.h"
#include "lds_chain.h"

// A fused_* function in the tile width the launch deserves (csrc/tile_ns.h): 16 lines per tile for batches, 8 for a
// handful of channels.
#define TILE_CALL(narrow_tiles, fn, ...) ((narrow_tiles) ? ::rcfm::narrow::fn(__VA_ARGS__) : ::rcfm::fn(__VA_ARGS__))

namespace rcfm {

namespace {
thread_local std::string g_last_error;
}

void set_last_error(const std::string& msg) { g_last_error = msg; }

// ---- arena (common.h; rcfm_arena_* below) ---------------------------------------------------------------------------
}  // namespace rcfm
struct rcfm_arena_s {
    std::mutex mu;
    size_t block_bytes = 0;                 // size of a block (a request larger than that gets a block of its own size)
    struct Block {
        char* base;
        size_t bytes, used;
        bool owned;                         // false: the caller's memory (rcfm_arena_adopt)
    };
    std::vector<Block> blocks;
    size_t live = 0;                        // pieces handed out and not yet dropped
    size_t handles = 0;                     // tuner / demodulator handles created inside (they keep the pointer for life,
                                            // whether or not they hold a piece at the moment)
    ~rcfm_arena_s() {
        for (auto& b : blocks)
            if (b.owned) (void)hipFree(b.base);
    }
};
namespace rcfm {

namespace {
// Two thread-local notions, on purpose apart: what the HOST bound (rcfm_arena_bind) is only ever read when a tuner or
// demodulator handle is created; what an allocation draws from is the arena of the handle whose entry point is running
// (ArenaScope).  Function-static scratch, plan caches and the resampler / feeder handles therefore never take a piece,
// whatever is bound when they happen to allocate.
thread_local Arena* g_bound = nullptr;
thread_local Arena* g_scope = nullptr;
constexpr size_t kArenaAlign = (size_t)2 << 20;   // pieces start on 2 MiB boundaries (the large-page size)
std::mutex g_arenas_mu;
std::set<Arena*> g_arenas;                        // arenas that exist (a binding left behind on another thread is checked)
}  // namespace

Arena* current_arena() { return g_scope; }

Arena* arena_enter_handle() {
    Arena* a = g_bound;
    if (!a) return nullptr;
    std::lock_guard<std::mutex> reg(g_arenas_mu);
    if (!g_arenas.count(a)) {   // destroyed on another thread while still bound here
        g_bound = nullptr;
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(a->mu);
    a->handles += 1;
    return a;
}

void arena_leave_handle(Arena* a) {
    if (!a) return;
    std::lock_guard<std::mutex> lock(a->mu);
    if (a->handles) a->handles -= 1;
}

void* arena_take(Arena* a, size_t bytes) {
    std::lock_guard<std::mutex> lock(a->mu);
    const size_t need = (bytes + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
    for (auto& b : a->blocks)
        if (b.bytes - b.used >= need) {
            void* p = b.base + b.used;
            b.used += need;
            a->live += 1;
            return p;
        }
    void* base = nullptr;
    const size_t sz = std::max(a->block_bytes, need);
    RC_HIP(hipMalloc(&base, sz));
    a->blocks.push_back(Arena::Block{static_cast<char*>(base), sz, need, true});
    a->live += 1;
    return base;
}

void arena_drop(Arena* a) {
    std::lock_guard<std::mutex> lock(a->mu);
    if (a->live) a->live -= 1;
}

ArenaScope::ArenaScope(Arena* a) : prev(g_scope) { g_scope = a; }
ArenaScope::~ArenaScope() { g_scope = prev; }

namespace {

constexpr double kPi = 3.14159265358979323846;

// fftshift(get_window(name, n))[k]  (tuner.py:156-157, decimate.py:32-33): the periodic
// general-cosine window a0 - (1 - a0) cos(2 pi i / n) read at i = (k - n//2) mod n.
double shifted_window(double a0, int64_t n, int64_t k) {
    if (n == 1) return 1.0;
    int64_t i = (k - n / 2) % n;
    if (i < 0) i += n;
    return a0 - (1.0 - a0) * std::cos(2.0 * kPi * (double)i / (double)n);
}

// Device tables + scalars of one scipy.signal.resample geometry n -> m.
struct ResampleGeom {
    int64_t n = 0, m = 0;
    int nmin = 0, nyq = 0, nneg = 0;
    int nyq_mode = NYQ_NONE;   // complex
    float w_merge = 0.f;       // complex, NYQ_DOWN
    float nyq_factor = 1.f;    // real
    float scale = 1.f;         // (m/n) * 1/m for an unnormalised inverse
    DeviceBuffer wpos, wneg;   // complex: float[nyq], float[nneg + 1]
    DeviceBuffer wr;           // real: float[nyq]

    void build(int64_t n_, int64_t m_, double a0, bool complex_input) {
        n = n_;
        m = m_;
        nmin = (int)std::min(n, m);
        nyq = nmin / 2 + 1;
        scale = (float)(1.0 / (double)n);
        const bool even = (nmin % 2) == 0;
        if (complex_input) {
            nneg = nmin > 2 ? nmin - nyq : 0;
            std::vector<float> p(nyq), q(nneg + 1, 0.f);
            for (int k = 0; k < nyq; ++k) p[k] = (float)shifted_window(a0, n, k);
            for (int j = 1; j <= nneg; ++j) q[j] = (float)shifted_window(a0, n, n - j);
            wpos.upload(p.data(), p.size() * sizeof(float));
            wneg.upload(q.data(), q.size() * sizeof(float));
            nyq_mode = NYQ_NONE;
            if (even && m < n && nmin > 2) {   // (for nmin == 2 scipy's slice is empty)
                nyq_mode = NYQ_DOWN;
                w_merge = (float)shifted_window(a0, n, n - nmin / 2);
            } else if (even && n < m) {
                nyq_mode = NYQ_UP;
            }
        } else {
            // "Fold the window back on itself to mimic complex behavior"
            std::vector<float> r(nyq);
            for (int k = 0; k < nyq; ++k) {
                double w = shifted_window(a0, n, k);
                if (k >= 1) w = 0.5 * (w + shifted_window(a0, n, n - k));
                r[k] = (float)w;
            }
            wr.upload(r.data(), r.size() * sizeof(float));
            nyq_factor = 1.f;
            if (even && m < n) nyq_factor = 2.f;
            if (even && n < m) nyq_factor = 0.5f;
        }
    }
};

// firwin(numtaps, [lo, hi], pass_zero=False, window="hamm"), bandpass.py:50-52.
std::vector<double> firwin_bandpass(int numtaps, double lo, double hi) {
    std::vector<double> h(numtaps);
    const double alpha = 0.5 * (numtaps - 1);
    auto sinc = [](double x) { return x == 0.0 ? 1.0 : std::sin(kPi * x) / (kPi * x); };
    const double fc = 0.5 * (lo + hi);
    double s = 0.0;
    for (int i = 0; i < numtaps; ++i) {
        const double mm = i - alpha;
        const double win = numtaps == 1 ? 1.0 : 0.54 - 0.46 * std::cos(2.0 * kPi * i / (numtaps - 1));
        h[i] = (hi * sinc(hi * mm) - lo * sinc(lo * mm)) * win;
        s += h[i] * std::cos(kPi * mm * fc);
    }
    for (auto& v : h) v /= s;
    return h;
}

// g = b (*) reverse(b): the zero-phase kernel of filtfilt for an FIR b; returns g[centre..edge].
std::vector<float> zero_phase_kernel(const float* b, int nb) {
    std::vector<float> g(nb);
    for (int lag = 0; lag < nb; ++lag) {
        double acc = 0.0;
        for (int i = 0; i + lag < nb; ++i) acc += (double)b[i] * (double)b[i + lag];
        g[lag] = (float)acc;
    }
    return g;
}

// deemphasis.py:37-49: first 51 impulse-response samples of (1-x)/(z-x), then lfilter_zi.
void deemphasis_design(int64_t fs, double tau, float* taps51, float* zi50) {
    const double x = std::exp(-1.0 / ((double)fs * tau));
    double st = 0.0, u = 1.0;
    for (int i = 0; i < 51; ++i) {
        taps51[i] = (float)((1.0 - x) * st);
        st = x * st + u;
        u = 0.0;
    }
    // lfilter_zi for an FIR: zi[k] = sum_{j>k} b[j]  (float32 cumulative sum from the tail)
    float acc = 0.f;
    for (int k = 49; k >= 0; --k) {
        acc += taps51[k + 1];
        zi50[k] = acc;
    }
}

// Plans are keyed by batch size: a full chunk and (at most) one remainder.
struct PlanCache {
    std::map<int, std::unique_ptr<FftPlan>> by_batch;
    FftPlan& get(FftKind kind, size_t n, int batch, bool in_place, size_t& work_need) {
        auto it = by_batch.find(batch);
        if (it == by_batch.end())
            it = by_batch.emplace(batch, std::make_unique<FftPlan>(kind, n, (size_t)batch, in_place)).first;
        work_need = std::max(work_need, it->second->work_bytes());
        return *it->second;
    }
};

// ---- per-stage HIP-event timing (rcfm_profile_*) ------------------------------
// bench.py reads these to price the dominant stage against the HBM roofline.
enum Stage : int {
    ST_TUNER_FFT = 0,   // T1  wideband forward FFT
    ST_TUNER_GATHER,    // T2a bin gather + weight
    ST_TUNER_IFFT,      // T2b per-channel inverse FFT
    ST_DISC,            // F1  discriminator (FM/MFM)
    ST_PILOT,           // W1  discriminator + 3-tap + pilot FIR
    ST_FFT_REAL_B,      // r2c of length B (pilot or discriminator)
    ST_HILBERT_MASK,    // W2b
    ST_IFFT_B,          // W2c analytic signal
    ST_STEREO_MIX,      // W3a
    ST_FFT_B,           // W3b packed L/R forward FFT
    ST_AUDIO_SPECTRUM,  // W4a unpack / real spectrum resample
    ST_IFFT_A,          // W4b inverse FFT of length A (c2c packed or c2r)
    ST_DEEMPH,          // W5a FIR51 + partial sums
    ST_DEEMPH_STATE,    // W5b
    ST_DC_CLIP,         // W5c
    ST_LDS_CHAIN,       // T2 + F1 + F2 of narrow FM / MFM channels in one kernel (lds_chain.h)
    ST_COUNT
};

const char* const kStageNames[ST_COUNT] = {
    "tuner_fft_N",   "tuner_gather",  "tuner_ifft_B", "discriminator", "pilot_stage",
    "rfft_B",        "hilbert_mask",  "ifft_B",       "stereo_mix",    "fft_B",
    "audio_spectrum", "ifft_A",       "deemphasis",   "deemph_state",  "dc_clip",
    "lds_chain"};

// One process-wide instance; handles of different threads may time stages concurrently, so every access goes
// through `mu` (uncontended in the single-DSP-thread use the reference has).
struct Profiler {
    std::mutex mu;
    std::atomic<uint64_t> mask{0};
    struct Pair {
        hipEvent_t a, b;
    };
    std::vector<Pair> pending[ST_COUNT];
    std::vector<Pair> pool;
    double total_ms[ST_COUNT] = {};
    int64_t count[ST_COUNT] = {};

    bool on(int st) const { return (mask.load(std::memory_order_relaxed) >> st) & 1u; }
    Pair take() {
        std::lock_guard<std::mutex> lock(mu);
        if (!pool.empty()) {
            Pair p = pool.back();
            pool.pop_back();
            return p;
        }
        Pair p;
        RC_HIP(hipEventCreate(&p.a));
        RC_HIP(hipEventCreate(&p.b));
        return p;
    }
    void push(int st, Pair p) {
        std::lock_guard<std::mutex> lock(mu);
        pending[st].push_back(p);
    }
    void collect() {
        std::lock_guard<std::mutex> lock(mu);
        for (int st = 0; st < ST_COUNT; ++st) {
            for (auto& p : pending[st]) {
                RC_HIP(hipEventSynchronize(p.b));
                float ms = 0.f;
                RC_HIP(hipEventElapsedTime(&ms, p.a, p.b));
                total_ms[st] += ms;
                count[st] += 1;
                pool.push_back(p);
            }
            pending[st].clear();
        }
    }
};

Profiler g_prof;

// Brackets one stage (one kernel, or one rocFFT execute) with events on its stream.
struct StageTimer {
    int st;
    hipStream_t s;
    Profiler::Pair p{};
    bool live;
    StageTimer(int stage, hipStream_t stream) : st(stage), s(stream), live(g_prof.on(stage)) {
        if (live) {
            p = g_prof.take();
            RC_HIP(hipEventRecord(p.a, s));
        }
    }
    ~StageTimer() {
        if (live) {
            (void)hipEventRecord(p.b, s);
            g_prof.push(st, p);
        }
    }
};

}  // namespace
}  // namespace rcfm

using namespace rcfm;

// ---------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------

// Tile width (RCFM_OPT_NARROW_TILES / RCFM_TUNER_OPT_NARROW_TILES, per handle): 0 = every tile kernel with 16 lines per
// tile, 1 (default) = 8 lines when a launch would leave most CUs without a tile, 2 = always 8.
constexpr int kNarrowDefault = 1;
// RCFM_TUNER_OPT_ALIGNED_PLAN (rcfm_tools.h): 1 = run the aligned order of a wideband plan whose last pass straddles lines.
constexpr int kAlignedPlanDefault = 1;
// One tile per CU and a half-empty chip: that is where a launch lasts one tile's latency and narrower tiles (twice as
// many, half the threads each) shorten it.  From two 16-line tiles per CU on, the wide ones stream better.
static bool narrow_launch(const FftEngine& e, int signals, int mode) {
    if (mode == 0) return false;
    if (mode >= 2) return true;
    const int64_t tiles = (e.desc().pass[0].n_inner + kFftTileW - 1) / kFftTileW;
    return (int64_t)signals * tiles < 2 * (int64_t)FftEngine::compute_units();
}

// RCFM_FFT=rocfft (environment, read once per process; include/rcfm.h): every transform through rocFFT -- the safety
// net for a host that suspects the engine, and the A/B partner of bench/reference_shapes.py.  The ONLY environment
// variable this library reads.
static bool use_engine() {
    static const bool v = [] {
        const char* e = std::getenv("RCFM_FFT");
        return !(e && std::string(e) == "rocfft");
    }();
    return v;
}

struct rcfm_tuner_s {
    Arena* arena = arena_enter_handle();   // rcfm_arena_bind at creation: every workspace of this handle, for its whole life
    ~rcfm_tuner_s() { arena_leave_handle(arena); }   // (the members drop their pieces after this body)
    int opt_narrow = kNarrowDefault;  // RCFM_TUNER_OPT_NARROW_TILES (rcfm_pipeline_run passes the demodulator's setting)
    int64_t n = 0;
    int nch = 0;
    std::vector<int64_t> roll;   // normalised to [0, n)
    std::vector<int32_t> bw;
    DeviceBuffer roll_dev;
    DeviceBuffer base_dev;   // int32 (n - roll) mod n per channel: start of the channel in the haloed spectrum
    DeviceBuffer X;          // [halo | n bins | halo]: the halos repeat the far ends, so a channel's bins
    int64_t halo = 0;        //   base + d, |d| <= B/2 + 1, need no wrap-around (fused_passes.h)
    float2* ext = nullptr;   // rcfm_tuner_attach_spectrum: caller-owned storage of the same layout instead of X
    // rcfm_tuner_attach_window: the caller's storage holds [halo | the window's bins | halo] only; `ext` is then the
    // address bin -halo WOULD have (never dereferenced outside the window), and only the window's channels may run
    bool ext_window = false;
    int ext_first = 0, ext_count = 0;
    float2* spectrum() { return (ext ? ext : X.as<float2>()) + halo; }
    DeviceBuffer work;
    DeviceBuffer forward_work;                 // rocFFT fallback of the FORWARD transform: its own workspace -- load() may run on
                                               // another stream than run() (sharding.SpectrumRing), and reserve() may reallocate
    std::unique_ptr<FftPlan> forward;          // rocFFT fallback for lengths outside the engine
    std::unique_ptr<FftEngine> forward_engine;
    DeviceBuffer forward_tmp;                  // engine: the last pass cannot run in place
    // RCFM_TUNER_OPT_ALIGNED_PLAN: the default plan's pass lengths in the order whose LAST pass stores aligned segments,
    // in the padded-rows layout (fft_engine.h, layout 2) -- N = 2.4e8 = 640 x 625 x 600 with 608-point scratch rows.  Its
    // first intermediate lives in the handle's own spectrum storage (sized for it), so an attached storage runs the
    // default plan.  bin_window() -- the protocol between ranks -- always speaks in rows of the default plan.
    std::unique_ptr<FftEngine> forward_aligned;
    int opt_aligned = kAlignedPlanDefault;
    FftRowWindow window_aligned{0, 0};
    bool windowed_aligned = false;
    size_t own_spectrum_bytes() const {
        int64_t elems = n + 2 * halo;
        if (forward_aligned) elems = std::max<int64_t>(elems, forward_aligned->tmp_stride());
        return sizeof(float2) * (size_t)elems;
    }
    bool use_aligned() const { return forward_aligned && opt_aligned && ext == nullptr && X.bytes() >= own_spectrum_bytes(); }
    DeviceBuffer band_tmp;
    bool loaded = false;
    // rcfm_tuner_shard: rows of the spectrum (FftRowWindow) the declared channel range reads
    bool windowed = false;
    FftRowWindow window{0, 0};
    int shard_first = 0, shard_count = 0;        // the declared range (valid while windowed)
    // what the spectrum in X was loaded for: a windowed load stores only the rows [loaded_first,
    // loaded_first + loaded_count) reads, so run() refuses channels outside it (the other bins are stale)
    bool loaded_windowed = false;
    int loaded_first = 0, loaded_count = 0;

    // Rows of the spectrum (row = n_1 consecutive bins, n_1 = the forward plan's first pass length) that channels
    // [first, first + count) read, as a circular window lo..hi: everything but the longest run of unused rows.
    // false: no engine, too few rows, or nothing worth skipping -- the range reads (nearly) the whole spectrum.
    bool row_window(int first, int count, FftRowWindow* w, const FftEngine* eng = nullptr) const {
        if (!eng) eng = forward_engine.get();
        if (!eng || count == 0) return false;
        const int64_t f0 = eng->row_length();
        const int64_t rows = n / f0;
        if (rows < 64) return false;
        std::vector<char> used((size_t)rows, 0);
        for (int c = first; c < first + count; ++c) {
            const int64_t centre = (n - roll[c]) % n, h = bw[c] / 2 + 2;
            const int64_t r0 = (centre - h) / f0 - ((centre - h) < 0 ? 1 : 0), r1 = (centre + h) / f0;
            for (int64_t r = r0; r <= r1; ++r) used[(size_t)(((r % rows) + rows) % rows)] = 1;
        }
        // the longest circular run of unused rows is dropped; everything else is kept
        int64_t best_len = 0, best_start = 0, run = 0;
        for (int64_t i = 0; i < 2 * rows; ++i) {
            if (!used[(size_t)(i % rows)]) {
                ++run;
                if (run > best_len && run <= rows) {
                    best_len = run;
                    best_start = i - run + 1;
                }
            } else {
                run = 0;
            }
        }
        if (best_len < rows / 64 || best_len >= rows) return false;   // nothing worth skipping (or nothing used)
        w->lo = (int)((best_start + best_len) % rows);
        w->hi = (int)(((best_start - 1) % rows + rows) % rows);
        return true;
    }

    void shard(int first, int count) {
        RC_REQUIRE(first >= 0 && count >= 0 && first + count <= nch, RCFM_ERR_INDEX, "channel index out of range");
        shard_first = first;
        shard_count = count;
        windowed = row_window(first, count, &window);
        windowed_aligned = forward_aligned && row_window(first, count, &window_aligned, forward_aligned.get());
    }

    // The same window in bins: [first_bin, first_bin + nbins) modulo n (nbins = n: everything).
    void bin_window(int first, int count, int64_t* first_bin, int64_t* nbins) const {
        FftRowWindow w{0, 0};
        if (count == 0) {            // a rank that owns no channels (C < G) reads nothing
            *first_bin = 0;
            *nbins = 0;
            return;
        }
        if (!row_window(first, count, &w)) {
            *first_bin = 0;
            *nbins = n;
            return;
        }
        const int64_t f0 = forward_engine->row_length(), rows = n / f0;
        *first_bin = (int64_t)w.lo * f0;
        *nbins = (((int64_t)w.hi - w.lo + rows) % rows + 1) * f0;
    }

    // The caller has written the bins channels [first, first + count) read (bin_window) into the spectrum storage,
    // e.g. received them from the GPU that ran the wideband FFT of this buffer: repeat the ends in the halos
    // (what the last FFT pass does for a local load) and accept exactly that channel range.
    void adopt(int first, int count, hipStream_t s) {
        RC_REQUIRE(first >= 0 && count >= 0 && first + count <= nch, RCFM_ERR_INDEX, "channel index out of range");
        int64_t fb = 0, nb = 0;
        bin_window(first, count, &fb, &nb);
        RC_REQUIRE(!ext_window || (first == ext_first && count == ext_count), RCFM_ERR_STATE,
                   "the attached storage holds the window of another channel range (rcfm_tuner_attach_window)");
        if (halo > 0 && !ext_window) {
            float2* Xs = spectrum();
            // segments of the window: [fb, min(fb + nb, n)) and, when it wraps, [0, fb + nb - n)
            const int64_t seg[2][2] = {{fb, std::min(fb + nb, n)}, {0, fb + nb > n ? fb + nb - n : 0}};
            for (auto& g : seg) {
                const int64_t a0 = std::max<int64_t>(g[0], 0), a1 = std::min<int64_t>(g[1], halo);   // bins [0, halo) -> behind the end
                if (a1 > a0)
                    RC_HIP(hipMemcpyAsync(Xs + n + a0, Xs + a0, sizeof(float2) * (size_t)(a1 - a0), hipMemcpyDeviceToDevice, s));
                const int64_t b0 = std::max<int64_t>(g[0], n - halo), b1 = std::min<int64_t>(g[1], n);     // bins [n - halo, n) -> in front
                if (b1 > b0)
                    RC_HIP(hipMemcpyAsync(Xs + (b0 - n), Xs + b0, sizeof(float2) * (size_t)(b1 - b0), hipMemcpyDeviceToDevice, s));
            }
        }
        loaded = true;
        loaded_windowed = nb < n;
        loaded_first = first;
        loaded_count = count;
    }

    // Can the bins of channels [first, first + count) live in a storage of their own, [halo | nbins | halo]?  Only a
    // window that neither wraps around bin 0 nor touches the far ends (whose halos repeat the other end of the spectrum).
    bool window_storage_ok(int first, int count, int64_t* fb, int64_t* nb) const {
        if (halo <= 0 || count <= 0) return false;
        bin_window(first, count, fb, nb);
        return *nb < n && *fb >= halo && *fb + *nb <= n - halo;
    }
    struct Band {
        ResampleGeom geom;
        PlanCache inverse;
        std::unique_ptr<FftEngine> engine;
    };
    std::map<int32_t, std::unique_ptr<Band>> bands;

    Band& band(int32_t b) {
        auto it = bands.find(b);
        if (it == bands.end()) {
            auto nb = std::make_unique<Band>();
            nb->geom.build(n, b, 0.5 /* hann */, true);
            FftPlanDesc probe;
            if (use_engine() && fft_plan_describe(b, &probe)) nb->engine = std::make_unique<FftEngine>(b);
            it = bands.emplace(b, std::move(nb)).first;
        }
        return *it->second;
    }

    // The fast gather's preconditions (fused_passes.hip, LoadTunerGatherFast) for the band of channel `first`: haloed
    // spectrum with 32-bit bases, window argument small enough for the series, no up-sampling Nyquist rule.
    bool fast_gather_ok(int first) {
        if (first < 0 || first >= nch || halo <= 0) return false;
        const int32_t B = bw[first];
        const ResampleGeom& g = band(B).geom;
        const bool series = 6.28318530717958647692 * ((double)(B / 2 + 2) / (double)n) < 0.25;
        return series && halo >= B / 2 + 1 && g.nyq_mode != NYQ_UP && B <= n;
    }

    // Can run() leave angle(x) / pi instead of x for this channel's band?  (engine path only)
    bool phase_capable(int first) { return first >= 0 && first < nch && band(bw[first]).engine != nullptr; }

    // theta != nullptr (phase_capable bands only): angle(x) / pi goes to theta [count][B] float32, out is unused.
    // First pass length n_1 of the band's inverse FFT (0 without an engine): its last pass stores rows of n_1 samples.
    int band_row_length(int first) { return phase_capable(first) ? (int)band(bw[first]).engine->row_length() : 0; }
    // Padded phase rows (fused_tuner_ifft's theta_pitch) need the last pass to write rows of n_1 samples with no
    // outer line index, i.e. a TWO-pass plan (B <= 262144); three-pass bands keep contiguous phases.
    bool band_two_pass(int first) { return phase_capable(first) && band(bw[first]).engine->npass() == 2; }

    void run(int first, int count, float2* out, hipStream_t s, float* theta = nullptr, int theta_pitch = 0,
             int narrow_mode = -1) {
        if (narrow_mode < 0) narrow_mode = opt_narrow;
        RC_REQUIRE(first >= 0 && count >= 0 && first + count <= nch, RCFM_ERR_INDEX, "channel index out of range");
        RC_REQUIRE(loaded, RCFM_ERR_STATE, "rcfm_tuner_run called before rcfm_tuner_load");
        RC_REQUIRE(!loaded_windowed || (first >= loaded_first && first + count <= loaded_first + loaded_count),
                   RCFM_ERR_STATE,
                   "channel outside the shard the spectrum was loaded for (rcfm_tuner_shard, then rcfm_tuner_load)");
        RC_REQUIRE(!ext_window || (first >= ext_first && first + count <= ext_first + ext_count), RCFM_ERR_STATE,
                   "channel outside the window the attached storage holds (rcfm_tuner_attach_window)");
        if (count == 0) return;
        const int32_t B = bw[first];
        for (int i = 0; i < count; ++i)
            RC_REQUIRE(bw[first + i] == B, RCFM_ERR_ARG, "channels of one rcfm_tuner_run range must share a bandwidth");
        Band& bd = band(B);
        const ResampleGeom& g = bd.geom;
        if (bd.engine) {
            // gather + window ride on the first pass of the inverse FFT (fused_passes.h)
            band_tmp.reserve((size_t)count * bd.engine->tmp_stride() * sizeof(float2));
            TunerGather tg{spectrum(), n, roll_dev.as<int64_t>() + first, 0.5, g.nyq, g.nneg, g.nyq_mode,
                           halo ? base_dev.as<int32_t>() + first : nullptr, halo};
            StageTimer tm(ST_TUNER_IFFT, s);
            TILE_CALL(narrow_launch(*bd.engine, count, narrow_mode), fused_tuner_ifft, *bd.engine, tg, out, band_tmp.as<float2>(), count, s,
                      theta, theta_pitch);
            return;
        }
        RC_REQUIRE(theta == nullptr, RCFM_ERR_STATE, "phase output needs the FFT engine");
        size_t need = 0;
        FftPlan& inv = bd.inverse.get(FftKind::C2C_INVERSE, (size_t)B, count, true, need);
        work.reserve(need);
        {
            StageTimer tm(ST_TUNER_GATHER, s);
            launch_spectrum_c2c(spectrum(), 0, n, roll_dev.as<int64_t>() + first, out, B, count,
                                g.wpos.as<float>(), g.wneg.as<float>(), g.w_merge, g.nyq, g.nneg, g.nyq_mode,
                                g.scale, s);
        }
        {
            StageTimer tm(ST_TUNER_IFFT, s);
            inv.exec(out, out, work.get(), s);
        }
    }
};

struct rcfm_demod_s {
    Arena* arena = arena_enter_handle();   // rcfm_arena_bind at creation
    ~rcfm_demod_s() {
        drop_graphs();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        arena_leave_handle(arena);
    }
    int kind = 0, C = 0, B = 0, A = 0, ch = 1, chunk = 1;
    double tau = 75e-6;
    float taps_h[51];
    float zi_h[50];
    float pilot_h[41];
    float pilot_g_h[41];   // zero-phase kernel g = b (*) reverse(b), centre first
    DeviceBuffer taps, pilot_g;
    // De-emphasis state, [C][ch][50].  Normally this handle's own buffer; after rcfm_demod_bind_state a
    // one-channel handle's state IS slot `index` of a batched handle's buffer (shared ownership), so the
    // per-channel caller and the batched caller carry ONE state per channel like the reference's
    // Deemphasis._state (deemphasis.py:48-49,64).
    // RCFM_OPT_STATE_FENCE: handles that share one state buffer may run consecutive buffers on DIFFERENT streams (one
    // handle set per stream, radiocore.tools.Lanes): every launch sequence that reads or writes the state then waits
    // for the event the previous one recorded.  The fence travels with the buffer (bind_state shares both).
    struct StateBuf : DeviceBuffer {
        hipEvent_t ev = nullptr;
        bool armed = false, recorded = false;
        ~StateBuf() {
            if (ev) (void)hipEventDestroy(ev);
        }
    };
    std::shared_ptr<StateBuf> state_buf = std::make_shared<StateBuf>();
    struct StateFence {   // scope of the launches that touch the state on stream s
        StateBuf& b;
        hipStream_t s;
        StateFence(rcfm_demod_s& d, hipStream_t st) : b(*d.state_buf), s(st) {
            if (b.armed && b.recorded) RC_HIP(hipStreamWaitEvent(s, b.ev, 0));
        }
        ~StateFence() {
            if (b.armed && hipEventRecord(b.ev, s) == hipSuccess) b.recorded = true;
        }
    };
    size_t state_off = 0;   // floats into state_buf
    float* state_ptr() const { return state_buf->as<float>() + state_off; }
    float side_tap = 0.23f;
    ResampleGeom geom;   // B -> A, real, Hamming
    PlanCache r2c_B, c2c_inv_B, c2c_fwd_B, c2c_inv_A, c2r_A;
    std::unique_ptr<FftEngine> eng_B, eng_A;   // both set: the engine path with fused passes
    std::unique_ptr<FftEngine> eng_Ad;         // length A as (A / n_1, n_1): its first pass tiles like eng_B's last
    std::unique_ptr<FftEngine> eng_Bi;         // eng_B's two pass lengths swapped (k_fft_tile2 pairing)
    DeviceBuffer buf_Ti;
    // rcfm_demod_set_option: which forms of the chain rcfm_pipeline_run / run_chunk may use (all on by default).  Parity
    // tests flip them per handle to get a second evaluation that shares no kernel schedule with the default one; the
    // A/B tools isolate one fusion at a time (RCFM_OPT_PILOT_CHAIN / RCFM_OPT_DECIM_TILE; RCFM_OPT_FUSED_TILES sets both).
    bool opt_lds_chain = true;
    bool opt_pilot_chain = true;     // pilot chain, Hilbert pair / packed tiles (two transforms per tile around the mask)
    bool opt_decim_tile = true;      // spectral decimation between FFT_B's last pass and IFFT_A's first
    bool opt_pilot_blocked = true;   // m, p between the pilot stage and the pilot chain in the tile-blocked layout
    bool opt_phase_link = true;
    bool opt_lds_deemph = true;      // MFM's de-emphasis inside the LDS chain
    int opt_narrow = kNarrowDefault;   // RCFM_OPT_NARROW_TILES
    // RCFM_OPT_GRAPH: one channel per call (the reference's per-channel demodulator.run, tests/benchmark.py:29-31) is ten
    // launches of a few microseconds each: launch-bound.  The second call with the same pointers captures the chain
    // into a hipGraph (the first one has warmed up every lazily built table and workspace, so the capture allocates
    // nothing); later calls replay it with ONE graph launch on the caller's stream.  Same kernels, same arguments:
    // bit-identical.  Not used while a stage timer or the state fence (events on the stream) is on.
    // OFF by default: measured on MI355X / ROCm 7 (profiles/r06_a_single_call.txt) a graph launch of the ten-kernel WBFM
    // chain costs what the ten stream launches cost (73.4 vs 74.8 us synchronous, 61.0 vs 57.8 us queued), and the
    // shorter MFM / FM chains lose 4 - 6 us per call: this runtime's graph launch is no cheaper than its kernel launches.
    bool opt_graph = false;
    struct GraphSlot {
        const void* iq;
        void* audio;
        const float* state;
        int first;
        hipGraphExec_t exec;
        uint64_t used;
    };
    std::vector<GraphSlot> graphs;
    GraphSlot last_call{nullptr, nullptr, nullptr, -1, nullptr, 0};
    hipStream_t cap_stream = nullptr;
    uint64_t graph_tick = 0;
    void drop_graphs() {   // whatever the captured chains depended on has changed (option, state binding)
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.exec);
        graphs.clear();
        last_call = GraphSlot{nullptr, nullptr, nullptr, -1, nullptr, 0};
    }
    bool narrow(int cnt) const { return eng_B && narrow_launch(*eng_B, cnt, opt_narrow); }
    DeviceBuffer work, buf_iq, buf_m, buf_p, buf_P, buf_Z, buf_V, buf_v, partial, buf_T, buf_TA, buf_U2, buf_dc;
    int tiles = 0;

    void alloc() {
        const size_t c = (size_t)chunk;
        tiles = fir_tiles(A);

        if (kind == RCFM_WBFM) {
            buf_P.reset(c * (B / 2 + 1) * sizeof(float2));
            buf_Z.reset(c * B * sizeof(float2));      // analytic pilot, then the packed L/R signal
            buf_V.reset(c * A * sizeof(float2));      // packed audio spectrum -> l + j r
        } else {
            buf_m.reset(c * B * sizeof(float));                    // discriminator output
            buf_P.reset(c * (B / 2 + 1) * sizeof(float2));          // its half spectrum
            buf_V.reset(c * (A / 2 + 1) * sizeof(float2));          // resampled half spectrum
            if (kind == RCFM_MFM) buf_v.reset(c * A * sizeof(float));
        }
        if (kind != RCFM_FM) partial.reset((size_t)chunk * ch * tiles * sizeof(float));
        buf_dc.reset((size_t)chunk * sizeof(float2));
        FftPlanDesc probe;
        if (use_engine() && fft_plan_describe(B, &probe) && fft_plan_describe(A, &probe)) {
            eng_B = std::make_unique<FftEngine>(B);
            eng_A = std::make_unique<FftEngine>(A);
            // Two-pass plans (n_1, L): the decimation B -> A rides between FFT_B's last pass and IFFT_A's first when
            // A = n_1 L2 with L2 even (k_fft_tile2_decim); if the planner's order of the two factors does not allow that
            // and the other order does (256 000 -> 32 000: 512 x 500 gives L2 = 62.5, 500 x 512 gives 64), take the other
            // one -- the fused chain uses both orders of the plan anyway (eng_Bi below).
            if (eng_B->npass() == 2 && A < B) {
                const FftPlanDesc& pd = eng_B->desc();
                auto decim_ok = [&](const FftEngine& eb) {
                    const int64_t n1 = eb.desc().pass[0].L;
                    const int64_t fa[2] = {A / n1, n1};
                    FftPlanDesc pa;
                    if (A % (2 * n1) != 0 || A / n1 < 16 || !fft_plan_describe(A, &pa, 0, fa, 2)) return false;
                    FftEngine ea(A, fa, 2);
                    return fused_fft_decim_ifft_applies(eb, ea, 1);
                };
                if (!decim_ok(*eng_B)) {
                    const int64_t swapped[2] = {pd.pass[1].L, pd.pass[0].L};
                    FftPlanDesc ps;
                    if (fft_plan_describe(B, &ps, 0, swapped, 2)) {
                        auto alt = std::make_unique<FftEngine>(B, swapped, 2);
                        if (decim_ok(*alt)) eng_B = std::move(alt);
                    }
                }
            }
            buf_T.reset(c * eng_B->tmp_stride() * sizeof(float2));
            buf_TA.reset(c * eng_A->tmp_stride() * sizeof(float2));
            if (kind == RCFM_WBFM) {
                buf_U2.reset(((c + 1) / 2) * B * sizeof(float2));
                const FftPlanDesc& pd = eng_B->desc();
                if (pd.npass == 2) {
                    const int64_t swapped[2] = {pd.pass[1].L, pd.pass[0].L};
                    eng_Bi = std::make_unique<FftEngine>(B, swapped, 2);
                    buf_Ti.reset(c * eng_Bi->tmp_stride() * sizeof(float2));
                }
            }
            {   // decimation between FFT_B's last pass and IFFT_A's first (fused_passes.h)
                const FftPlanDesc& pd = eng_B->desc();
                const int64_t n1 = pd.pass[0].L;
                const int64_t fa[2] = {A / n1, n1};
                FftPlanDesc pa;
                if (pd.npass == 2 && A < B && A % (2 * n1) == 0 && A / n1 >= 16 && fft_plan_describe(A, &pa, 0, fa, 2)) {
                    eng_Ad = std::make_unique<FftEngine>(A, fa, 2);
                    buf_TA.reserve(c * eng_Ad->tmp_stride() * sizeof(float2));
                    if (const int pitch = audio_pitch())   // padded rows of the packed audio
                        buf_V.reserve(c * (size_t)(A / eng_Ad->row_length()) * pitch * sizeof(float2));
                }
            }
            if (kind != RCFM_WBFM) {
                buf_Z.reset(c * B * sizeof(float2));   // full spectrum of the discriminator output
                buf_V.reset(c * A * sizeof(float2));   // Hermitian audio spectrum
            }
        }
        if (kind == RCFM_WBFM) {
            // mono signal and pilot band (after the engines: the tile-blocked layout of the pilot chain pads the last
            // 16-column block of every row, PilotBlocked::stride >= B)
            const size_t per_channel = std::max<size_t>((size_t)B, (size_t)pilot_blocked().stride);
            buf_m.reset(c * per_channel * sizeof(float));
            buf_p.reset(c * per_channel * sizeof(float));
        }
    }

    void reset_state(hipStream_t s) {
        if (kind == RCFM_FM) return;
        std::vector<float> all((size_t)C * ch * 50);
        for (size_t i = 0; i < all.size(); ++i) all[i] = zi_h[i % 50];
        RC_HIP(hipMemcpyAsync(state_ptr(), all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice, s));
        RC_HIP(hipStreamSynchronize(s));
    }

    // mfm.py:63-65 / wbfm.py:90-100: de-emphasis (per-leg state), joint DC removal, clip.
    // Can run_deemph take its input in padded rows (fused_fft_decim_ifft's out_pitch)?  Only the fused kernel does.
    bool deemph_fused() const { return ((int64_t)A * ch) % 4 == 0 && A >= 50; }
    // Row pitch (samples) of the packed audio between IFFT_A's last pass and the de-emphasis kernel: rows of n_1
    // samples padded to whole 128-byte lines; 0 = contiguous.
    int audio_pitch() const {
        if (!eng_Ad || kind != RCFM_WBFM || !deemph_fused()) return 0;
        const int64_t n1 = eng_Ad->row_length();
        return (n1 % 16 == 0 || (n1 * 2) % 4 != 0) ? 0 : (int)((n1 + 15) / 16 * 16);
    }

    // Are the 51 taps the first samples of a one-pole impulse response (b[0] = 0, geometric tail)?  The same test
    // launch_fir51 applies (kernels.hip): the recursive forms of the FIR rely on it.
    bool deemph_geometric() const {
        if (!(taps_h[0] == 0.f && taps_h[1] > 0.f)) return false;
        for (int i = 1; i < 50; ++i)
            if (std::fabs((double)taps_h[i + 1] * taps_h[1] - (double)taps_h[i] * taps_h[2]) >
                4e-7 * (double)taps_h[i] * taps_h[1] + 1e-36)
                return false;
        return true;
    }

    // The taps followed by their suffix sums sfx[i] = sum_{j > i} b[j] (what the on-chip de-emphasis of lds_chain.hip reads).
    DeviceBuffer taps_sfx;
    const float* taps_sfx_dev() {
        if (taps_sfx.bytes() == 0) {
            float h[102];
            for (int i = 0; i < 51; ++i) h[i] = taps_h[i];
            double acc = 0.0;
            for (int i = 50; i >= 0; --i) {
                h[51 + i] = (float)acc;          // sum of b[j], j > i
                acc += (double)taps_h[i];
            }
            taps_sfx.upload(h, sizeof(h));
        }
        return taps_sfx.as<float>();
    }

    void run_deemph(const float* v, float* audio, float* st, int cnt, hipStream_t s, bool have_dc = false,
                    int row = 0, int pitch = 0) {
        const bool fast = ((int64_t)A * ch) % 4 == 0;
        StateFence fence(*this, s);
        if (fast && have_dc && A >= 50) {
            // de-emphasis, DC removal and clip in one kernel: the mean comes from the DC bin (buf_dc)
            {
                StageTimer tm(ST_DEEMPH, s);
                launch_fir51(v, audio, A, ch, cnt, taps_h, st, nullptr, buf_dc.as<float2>(), s, row, pitch);
            }
            StageTimer tm(ST_DEEMPH_STATE, s);
            launch_fir_state(v, A, ch, cnt, taps.as<float>(), 51, st, s, row, pitch);
            return;
        }
        RC_REQUIRE(pitch == 0, RCFM_ERR_RUNTIME, "padded audio rows need the fused de-emphasis kernel");
        {
            StageTimer tm(ST_DEEMPH, s);
            if (fast)
                launch_fir51(v, audio, A, ch, cnt, taps_h, st, partial.as<float>(), nullptr, s);
            else
                launch_fir(v, audio, A, ch, cnt, taps.as<float>(), 51, st, partial.as<float>(), s);
        }
        {
            StageTimer tm(ST_DEEMPH_STATE, s);
            launch_fir_state(v, A, ch, cnt, taps.as<float>(), 51, st, s);
        }
        {
            StageTimer tm(ST_DC_CLIP, s);
            launch_dc_clip(audio, A, ch, cnt, partial.as<float>(), fast ? fir51_tiles(A, ch) : tiles * ch, s);
        }
    }

    // The tile-blocked layout of m and p (kernels.h) when the pilot chain's geometry allows it, else an invalid one.
    PilotBlocked pilot_blocked() const {
        int64_t rows = 0, row_length = 0;
        if (kind != RCFM_WBFM || !eng_B || !eng_Bi || B % 4 != 0 || !fused_pilot_chain_geometry(*eng_B, &rows, &row_length))
            return PilotBlocked{};
        return PilotBlocked::plan(rows, row_length);
    }

    // Does run_chunk take the samples' phases (theta = angle(x) / pi, float32 [cnt][B]) instead of iq?
    bool phase_capable() const { return eng_B != nullptr && (kind != RCFM_WBFM || B % 4 == 0); }

    void run_chunk(int first, int cnt, const float2* iq, float* audio, hipStream_t s, const float* theta = nullptr,
                   PhaseRows rows = PhaseRows{}) {
        size_t need = 0;
        const bool nw = narrow(cnt);   // 8-line tiles for a handful of channels (tile_ns.h)
        if (kind == RCFM_WBFM) {
            FftPlan* pf[4] = {nullptr, nullptr, nullptr, nullptr};
            if (!eng_B) {
                pf[0] = &r2c_B.get(FftKind::R2C, B, cnt, false, need);
                pf[1] = &c2c_inv_B.get(FftKind::C2C_INVERSE, B, cnt, true, need);
                pf[2] = &c2c_fwd_B.get(FftKind::C2C_FORWARD, B, cnt, true, need);
                pf[3] = &c2c_inv_A.get(FftKind::C2C_INVERSE, A, cnt, true, need);
                work.reserve(need);
            }
            float* m = buf_m.as<float>();
            float* p = buf_p.as<float>();
            float2* P = buf_P.as<float2>();
            float2* Z = buf_Z.as<float2>();
            float2* V = buf_V.as<float2>();
            // the three-launch pilot chain reads m and p as 16-line tiles: they leave the pilot stage tile-blocked then
            const bool chain = eng_B && eng_Bi && opt_pilot_chain && TILE_CALL(nw, fused_pilot_chain_applies, *eng_B, *eng_Bi, cnt);
            const PilotBlocked blk = (chain && opt_pilot_blocked) ? pilot_blocked() : PilotBlocked{};
            // wbfm.py:77-80  FM(B->B) and the pilot band-pass
            {
                StageTimer tm(ST_PILOT, s);
                if (theta != nullptr)
                    launch_pilot_stage_h40_phase(theta, m, p, B, cnt, pilot_g_h, side_tap, s, &blk);
                else if (B % 4 == 0)
                    launch_pilot_stage_h40(iq, m, p, B, cnt, pilot_g_h, side_tap, s, &blk);
                else
                    launch_pilot_stage(iq, nullptr, m, p, B, cnt, pilot_g.as<float>(), 40, side_tap, s);
            }
            if (eng_B) {
                float2* T = buf_T.as<float2>();
                float2* TA = buf_TA.as<float2>();
                float2* U2 = buf_U2.as<float2>();
                // RCFM_OPT_PILOT_CHAIN = 0: pair FFT -> U2 -> masked IFFT as separate transforms
                const bool packed = eng_Bi && opt_pilot_chain && TILE_CALL(nw, fused_hilbert_packed_applies, *eng_Bi, *eng_B, cnt);
                bool paired = false;
                if (chain) {
                    {   // wbfm.py:80 / pll.py:34: spectra of the pilot bands, two channels per complex FFT
                        StageTimer tm(ST_FFT_REAL_B, s);
                        TILE_CALL(nw, fused_pilot_chain_fft_first, *eng_B, p, T, cnt, s, blk.stride, blk.blk16());
                    }
                    {   // ... last pass, one-sided mask, inverse FFT, stereo matrix, first pass of the packed L/R FFT
                        StageTimer tm(ST_IFFT_B, s);
                        TILE_CALL(nw, fused_pilot_chain_mask_mix, *eng_B, *eng_Bi, p, m, T, buf_Ti.as<float2>(), cnt, s, blk.stride,
                                  blk.blk16());
                    }
                    paired = true;
                } else {
                    {   // wbfm.py:80 / pll.py:34: spectra of the pilot bands, two channels per complex FFT
                        StageTimer tm(ST_FFT_REAL_B, s);
                        TILE_CALL(nw, fused_real_pair_fft, *eng_B, p, U2, T, cnt, packed ? kKeepLowerHalf : -1, s);
                    }
                    if (eng_Bi && opt_pilot_chain) {
                        // one-sided mask -> inverse FFT -> stereo matrix -> first pass of the packed L/R FFT:
                        // the last IFFT pass and the first FFT pass share their tiles (fused_passes.h)
                        StageTimer tm(ST_IFFT_B, s);
                        paired = !packed ? TILE_CALL(nw, fused_hilbert_pair_ifft_mix_fft, *eng_Bi, *eng_B, U2, m, buf_Ti.as<float2>(), T, cnt, s)
                                         : TILE_CALL(nw, fused_hilbert_packed_ifft_mix_fft, *eng_Bi, *eng_B, U2, p, m,
                                                                             buf_Ti.as<float2>(), T, cnt, s);
                    }
                }
                const bool no_decim = !opt_decim_tile;
                if (paired && eng_Ad && !no_decim && TILE_CALL(nw, fused_fft_decim_ifft_applies, *eng_B, *eng_Ad, cnt)) {
                    const int pitch = audio_pitch();
                    {   // packed L/R FFT last pass -> decimation -> IFFT_A: the B-point spectrum stays on chip
                        StageTimer tm(ST_FFT_B, s);
                        TILE_CALL(nw, fused_fft_decim_ifft, *eng_B, *eng_Ad, T, V, TA, cnt, geom.wr.as<float>(), geom.scale,
                                             buf_dc.as<float2>(), s, pitch);
                    }
                    float* st = state_ptr() + (size_t)first * ch * 50;
                    run_deemph(reinterpret_cast<float*>(V), audio, st, cnt, s, true,
                               pitch ? (int)eng_Ad->row_length() : 0, pitch);
                    return;
                }
                if (paired) {
                    StageTimer tm(ST_FFT_B, s);
                    TILE_CALL(nw, fused_fft_last_pruned, *eng_B, T, Z, cnt, std::min(A, B) / 2, s);
                } else {
                    {   // one-sided mask -> inverse FFT -> 38 kHz carrier, L-R, stereo matrix (wbfm.py:83,86-87)
                        StageTimer tm(ST_IFFT_B, s);
                        TILE_CALL(nw, fused_hilbert_pair_ifft_mix, *eng_B, U2, m, Z, T, cnt, s);
                    }
                    {   // both stereo legs in one complex FFT; only |k| <= A/2 survives the decimation
                        StageTimer tm(ST_FFT_B, s);
                        TILE_CALL(nw, fused_fft_pruned, *eng_B, Z, Z, T, cnt, std::min(A, B) / 2, s);
                    }
                }
                {   // unpack + window + Nyquist rule ride on the first pass of IFFT_A
                    StageTimer tm(ST_IFFT_A, s);
                    TILE_CALL(nw, fused_stereo_unpack_ifft, *eng_A, Z, B, V, TA, cnt, geom.wr.as<float>(), geom.nyq, geom.nmin,
                                             geom.nyq_factor, geom.scale, buf_dc.as<float2>(), s);
                    // -> [cnt][A][2] float32, L/R interleaved
                }
                float* st = state_ptr() + (size_t)first * ch * 50;
                run_deemph(reinterpret_cast<float*>(V), audio, st, cnt, s, true);
                return;
            }
            // wbfm.py:80 / pll.py:34  analytic signal of the pilot
            {
                StageTimer tm(ST_FFT_REAL_B, s);
                pf[0]->exec(p, P, work.get(), s);
            }
            {
                StageTimer tm(ST_HILBERT_MASK, s);
                launch_hilbert_mask(P, Z, B, cnt, 1.0f / (float)B, s);
            }
            {
                StageTimer tm(ST_IFFT_B, s);
                pf[1]->exec(Z, Z, work.get(), s);
            }
            // wbfm.py:83,86-87  38 kHz carrier, L-R, stereo matrix; both legs packed in one complex signal
            {
                StageTimer tm(ST_STEREO_MIX, s);
                launch_stereo_mix(Z, m, Z, (size_t)cnt * B, s);
            }
            {
                StageTimer tm(ST_FFT_B, s);
                pf[2]->exec(Z, Z, work.get(), s);
            }
            {
                StageTimer tm(ST_AUDIO_SPECTRUM, s);
                launch_stereo_unpack(Z, B, V, A, cnt, geom.wr.as<float>(), geom.nyq, geom.nmin,
                                     geom.nyq_factor, geom.scale, nullptr, s);
            }
            {
                StageTimer tm(ST_IFFT_A, s);
                pf[3]->exec(V, V, work.get(), s);   // -> [cnt][A][2] float32, L/R interleaved
            }
            // wbfm.py:90-100  de-emphasis (separate L/R state), joint DC removal, clip
            float* st = state_ptr() + (size_t)first * ch * 50;
            StateFence fence(*this, s);
            {
                StageTimer tm(ST_DEEMPH, s);
                launch_fir(reinterpret_cast<float*>(V), audio, A, 2, cnt, taps.as<float>(), 51, st,
                           partial.as<float>(), s);
            }
            {
                StageTimer tm(ST_DEEMPH_STATE, s);
                launch_fir_state(reinterpret_cast<float*>(V), A, 2, cnt, taps.as<float>(), 51, st, s);
            }
            {
                StageTimer tm(ST_DC_CLIP, s);
                launch_dc_clip(audio, A, 2, cnt, partial.as<float>(), tiles * 2, s);
            }
            return;
        }
        float* d = buf_m.as<float>();
        // fm.py:60-66  discriminator, then Decimate(B -> A)
        if (theta == nullptr) {
            StageTimer tm(ST_DISC, s);
            launch_discriminator(iq, d, B, cnt, s);
        }
        const bool no_decim_pairs = !opt_decim_tile;
        if (eng_B && eng_Ad && !no_decim_pairs && ((int64_t)A % 4 == 0 || kind == RCFM_FM) &&
            TILE_CALL(nw, fused_fft_decim_ifft_applies, *eng_B, *eng_Ad, (cnt + 1) / 2)) {
            // two channels per complex signal from the pair FFT through the decimation to the inverse FFT:
            // 3 launches, no B-point spectrum in memory, half the inverse transforms
            {
                StageTimer tm(ST_FFT_REAL_B, s);
                TILE_CALL(nw, fused_real_pair_fft_first, *eng_B, theta != nullptr ? theta : d, buf_T.as<float2>(), cnt,
                                          theta != nullptr, s, rows);
            }
            float* dst = (kind == RCFM_FM) ? audio : buf_v.as<float>();
            {
                StageTimer tm(ST_IFFT_A, s);
                TILE_CALL(nw, fused_fft_decim_ifft_pairs, *eng_B, *eng_Ad, buf_T.as<float2>(), dst, buf_TA.as<float2>(), cnt,
                                           geom.wr.as<float>(), geom.scale, buf_dc.as<float2>(), s);
            }
            if (kind == RCFM_FM) return;
            float* st = state_ptr() + (size_t)first * 50;
            run_deemph(dst, audio, st, cnt, s, true);
            return;
        }
        if (eng_B) {
            float2* Dfull = buf_Z.as<float2>();
            float2* Yfull = buf_V.as<float2>();
            {
                StageTimer tm(ST_FFT_REAL_B, s);
                // two channels per complex FFT; only |k| <= A/2 is kept (and read back by the unpacking).
                // From the tuner's phases the discriminator is the load of the first pass.
                if (theta != nullptr)
                    TILE_CALL(nw, fused_real_pair_fft, *eng_B, theta, Dfull, buf_T.as<float2>(), cnt, std::min(A, B) / 2, s, true, rows);
                else
                    TILE_CALL(nw, fused_real_pair_fft, *eng_B, d, Dfull, buf_T.as<float2>(), cnt, std::min(A, B) / 2, s);
            }
            {
                StageTimer tm(ST_AUDIO_SPECTRUM, s);
                launch_spectrum_real_full(Dfull, B, Yfull, A, cnt, geom.wr.as<float>(), geom.nyq, geom.nmin,
                                          geom.nyq_factor, geom.scale, buf_dc.as<float2>(), true, s);
            }
            float* dst = (kind == RCFM_FM) ? audio : buf_v.as<float>();
            {
                StageTimer tm(ST_IFFT_A, s);
                TILE_CALL(nw, fused_ifft_real_out, *eng_A, Yfull, dst, buf_TA.as<float2>(), cnt, 1.0f, s);
            }
            if (kind == RCFM_FM) return;
            float* st = state_ptr() + (size_t)first * 50;
            run_deemph(dst, audio, st, cnt, s, true);
            return;
        }
        FftPlan& f1 = r2c_B.get(FftKind::R2C, B, cnt, false, need);
        FftPlan& f2 = c2r_A.get(FftKind::C2R, A, cnt, false, need);
        work.reserve(need);
        float2* D = buf_P.as<float2>();
        float2* Y = buf_V.as<float2>();
        {
            StageTimer tm(ST_FFT_REAL_B, s);
            f1.exec(d, D, work.get(), s);
        }
        {
            StageTimer tm(ST_AUDIO_SPECTRUM, s);
            launch_spectrum_r2c(D, B, Y, A, cnt, geom.wr.as<float>(), geom.nyq, geom.nmin, geom.nyq_factor,
                                geom.scale, s);
        }
        if (kind == RCFM_FM) {
            StageTimer tm(ST_IFFT_A, s);
            f2.exec(Y, audio, work.get(), s);
            return;
        }
        float* v = buf_v.as<float>();
        {
            StageTimer tm(ST_IFFT_A, s);
            f2.exec(Y, v, work.get(), s);
        }
        // mfm.py:63-65
        float* st = state_ptr() + (size_t)first * 50;
        StateFence fence(*this, s);
        {
            StageTimer tm(ST_DEEMPH, s);
            launch_fir(v, audio, A, 1, cnt, taps.as<float>(), 51, st, partial.as<float>(), s);
        }
        {
            StageTimer tm(ST_DEEMPH_STATE, s);
            launch_fir_state(v, A, 1, cnt, taps.as<float>(), 51, st, s);
        }
        {
            StageTimer tm(ST_DC_CLIP, s);
            launch_dc_clip(audio, A, 1, cnt, partial.as<float>(), tiles, s);
        }
    }
};

struct rcfm_resampler_s {
    int C = 0;
    int64_t n = 0, m = 0;
    bool cplx = false;
    ResampleGeom geom;
    std::unique_ptr<FftPlan> fwd, inv;
    DeviceBuffer spec_in, spec_out, work;
    // complex down-sampling on the FFT engine: forward transform, then the Tuner's fused gather + window +
    // inverse transform with roll = 0 and the Hamming weight (the same math, decimate.py:47-48 vs tuner.py:159-161)
    std::unique_ptr<FftEngine> eng_n, eng_m;
    DeviceBuffer tmp_n, tmp_m, zero_roll;
    // only bins |k| <= m/2 of the long spectrum are read: its last pass stores just the rows that hold them
    bool windowed = false;
    FftRowWindow window{0, 0};
};


// RCCL, bound at run time: a process that never gathers (one GPU, or a host that gathers with its own transport)
// does not load it; a process that already loaded an RCCL (PyTorch's ProcessGroupNCCL) gets that same copy.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy some other component of this process already mapped (PyTorch's ProcessGroupNCCL loads
        // torch/lib/librccl.so, SONAME librccl.so.1) is reused: two RCCL instances in one process would each
        // own a set of communicators and IPC handles
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (r.lib) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.Gather = reinterpret_cast<decltype(r.Gather)>(dlsym(r.lib, "ncclGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    });
    RC_REQUIRE(r.lib && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Gather && r.Send && r.Recv && r.GroupStart &&
                   r.GroupEnd,
               RCFM_ERR_RUNTIME,
               "RCCL (librccl.so) is not available in this process");
    return r;
}

#define RC_NCCL(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t rc_n_ = (expr);                                                               \
        if (rc_n_ != ncclSuccess)                                                                  \
            throw ::rcfm::Error{RCFM_ERR_RUNTIME, std::string(#expr) + ": " +                      \
                                                      (rccl().GetErrorString ? rccl().GetErrorString(rc_n_) : "RCCL error")}; \
    } while (0)

struct rcfm_comm_s {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    int group_depth = 0;   // rcfm_comm_group_start without its rcfm_comm_group_end
};

// Overlapped host -> device ingest (rcfm_feeder_*): `depth` device slots, one copy stream, an event pair per slot.
struct rcfm_feeder_s {
    size_t bytes = 0;
    int depth = 0;
    bool owns = false;
    std::vector<void*> slot;
    std::vector<hipEvent_t> ready, done;   // ready: the copy into the slot has landed; done: its consumer has finished
    hipStream_t copy = nullptr;
    uint64_t head = 0, tail = 0;           // submitted / released buffers
    uint64_t landed = 0;                   // buffers whose copy is known to have completed (rcfm_feeder_copied)
    ~rcfm_feeder_s() {
        if (copy) (void)hipStreamSynchronize(copy);
        for (auto e : ready) (void)hipEventDestroy(e);
        for (auto e : done) (void)hipEventDestroy(e);
        if (owns)
            for (auto p : slot) (void)hipFree(p);
        if (copy) (void)hipStreamDestroy(copy);
    }
};

// ---------------------------------------------------------------------------
// extern "C"
// ---------------------------------------------------------------------------

extern "C" {

int rcfm_version(void) { return RCFM_VERSION; }

const char* rcfm_last_error(void) { return g_last_error.c_str(); }

int rcfm_device_count(int* count) {
    return guarded([&] {
        RC_REQUIRE(count != nullptr, RCFM_ERR_ARG, "count is NULL");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        *count = (e == hipSuccess) ? n : 0;
        if (e != hipSuccess) (void)hipGetLastError();
    });
}

int rcfm_malloc(void** dptr, size_t bytes) {
    return guarded([&] {
        RC_REQUIRE(dptr != nullptr, RCFM_ERR_ARG, "dptr is NULL");
        RC_HIP(hipMalloc(dptr, bytes));
    });
}

int rcfm_free(void* dptr) {
    return guarded([&] { RC_HIP(hipFree(dptr)); });
}

int rcfm_memcpy_h2d(void* dst, const void* src_host, size_t bytes, void* stream) {
    return guarded([&] { RC_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream))); });
}

int rcfm_memcpy_d2h(void* dst_host, const void* src, size_t bytes, void* stream) {
    return guarded([&] { RC_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, as_stream(stream))); });
}

int rcfm_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    return guarded([&] { RC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream))); });
}

int rcfm_stream_sync(void* stream) {
    return guarded([&] { RC_HIP(hipStreamSynchronize(as_stream(stream))); });
}

int rcfm_stream_create(void** stream) {
    return guarded([&] {
        RC_REQUIRE(stream != nullptr, RCFM_ERR_ARG, "stream is NULL");
        hipStream_t s = nullptr;
        RC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        *stream = s;
    });
}

int rcfm_stream_destroy(void* stream) {
    return guarded([&] {
        if (stream) RC_HIP(hipStreamDestroy(as_stream(stream)));
    });
}

// ---- placement -----------------------------------------------------------------

int rcfm_arena_create(size_t block_bytes, rcfm_arena_t* out) {
    return guarded([&] {
        RC_REQUIRE(out != nullptr, RCFM_ERR_ARG, "out is NULL");
        auto a = std::make_unique<Arena>();
        a->block_bytes = block_bytes ? block_bytes : ((size_t)1 << 30);
        if (block_bytes) {   // the first block now: its placement is the draw the caller asked for
            void* base = nullptr;
            RC_HIP(hipMalloc(&base, a->block_bytes));
            a->blocks.push_back(Arena::Block{static_cast<char*>(base), a->block_bytes, 0, true});
        }
        {
            std::lock_guard<std::mutex> reg(g_arenas_mu);
            g_arenas.insert(a.get());
        }
        *out = a.release();
    });
}

int rcfm_arena_adopt(void* base, size_t bytes, rcfm_arena_t* out) {
    return guarded([&] {
        RC_REQUIRE(out && base && bytes >= kArenaAlign, RCFM_ERR_ARG, "bad arena memory");
        auto a = std::make_unique<Arena>();
        a->block_bytes = (size_t)1 << 30;   // what does not fit the caller's memory comes from hipMalloc in 1 GiB blocks
        char* p = static_cast<char*>(base);
        const size_t skew = (kArenaAlign - (reinterpret_cast<uintptr_t>(p) & (kArenaAlign - 1))) & (kArenaAlign - 1);
        RC_REQUIRE(bytes > skew + kArenaAlign, RCFM_ERR_ARG, "bad arena memory");
        a->blocks.push_back(Arena::Block{p + skew, (bytes - skew) / kArenaAlign * kArenaAlign, 0, false});
        {
            std::lock_guard<std::mutex> reg(g_arenas_mu);
            g_arenas.insert(a.get());
        }
        *out = a.release();
    });
}

int rcfm_arena_bind(rcfm_arena_t a) {
    return guarded([&] {
        if (a) {
            std::lock_guard<std::mutex> reg(g_arenas_mu);
            RC_REQUIRE(g_arenas.count(a) != 0, RCFM_ERR_ARG, "not a live arena");
        }
        g_bound = a;
    });
}

int rcfm_arena_stats(rcfm_arena_t a, size_t* reserved_bytes, size_t* used_bytes, size_t* live_pieces) {
    return guarded([&] {
        RC_REQUIRE(a != nullptr, RCFM_ERR_ARG, "NULL arena");
        std::lock_guard<std::mutex> lock(a->mu);
        size_t r = 0, u = 0;
        for (auto& b : a->blocks) {
            r += b.bytes;
            u += b.used;
        }
        if (reserved_bytes) *reserved_bytes = r;
        if (used_bytes) *used_bytes = u;
        if (live_pieces) *live_pieces = a->live;
    });
}

int rcfm_arena_destroy(rcfm_arena_t a) {
    return guarded([&] {
        if (!a) return;
        {
            std::lock_guard<std::mutex> reg(g_arenas_mu);
            RC_REQUIRE(g_arenas.count(a) != 0, RCFM_ERR_ARG, "not a live arena");
            {
                // handles, not pieces: a tuner whose spectrum was attached, or a handle of small buffers only, holds no
                // piece and still allocates from its arena on its next run
                std::lock_guard<std::mutex> lock(a->mu);
                RC_REQUIRE(a->handles == 0 && a->live == 0, RCFM_ERR_STATE,
                           "handles created inside this arena are still alive: destroy them first");
            }
            g_arenas.erase(a);   // a binding another thread still holds is dropped when that thread next creates a handle
        }
        if (g_bound == a) g_bound = nullptr;
        delete a;
    });
}

// ---- tuner -----------------------------------------------------------------

int rcfm_tuner_create(int64_t n, int nch, const int64_t* roll_host, const int32_t* bw_host, rcfm_tuner_t* out) {
    return guarded([&] {
        RC_REQUIRE(out != nullptr, RCFM_ERR_ARG, "out is NULL");
        RC_REQUIRE(n >= 1 && nch >= 0, RCFM_ERR_ARG, "bad tuner size");
        RC_REQUIRE(nch == 0 || (roll_host && bw_host), RCFM_ERR_ARG, "roll/bw is NULL");
        auto t = std::make_unique<rcfm_tuner_s>();
        ArenaScope scope(t->arena);
        t->n = n;
        t->nch = nch;
        t->roll.resize(nch);
        t->bw.assign(bw_host, bw_host + nch);
        for (int i = 0; i < nch; ++i) {
            RC_REQUIRE(bw_host[i] >= 1, RCFM_ERR_ARG, "channel bandwidth must be >= 1");
            // scipy.signal.resample(domain="freq") also up-samples; the Tuner never does
            RC_REQUIRE(bw_host[i] <= n, RCFM_ERR_ARG, "channel bandwidth exceeds the input bandwidth");
            int64_t r = roll_host[i] % n;
            if (r < 0) r += n;
            t->roll[i] = r;
        }
        if (nch) t->roll_dev.upload(t->roll.data(), sizeof(int64_t) * nch);
        // halo: whole 128-byte lines on both sides, wide enough for the widest channel
        int64_t h = 0;
        for (int i = 0; i < nch; ++i) h = std::max<int64_t>(h, bw_host[i] / 2 + 2);
        h = (h + 15) / 16 * 16;
        if (nch && h <= n && n + h < ((int64_t)1 << 31)) {
            t->halo = h;
            std::vector<int32_t> base(nch);
            for (int i = 0; i < nch; ++i) base[i] = (int32_t)((n - t->roll[i]) % n);
            t->base_dev.upload(base.data(), sizeof(int32_t) * nch);
        }
        FftPlanDesc probe;
        if (use_engine() && fft_plan_describe(n, &probe)) {
            t->forward_engine = std::make_unique<FftEngine>(n);
            int64_t tmp_elems = t->forward_engine->tmp_stride();
            int64_t order[3];
            if (fft_plan_aligned_order(t->forward_engine->desc(), order)) {
                t->forward_aligned = std::make_unique<FftEngine>(n, order, 3, 2);
                tmp_elems = std::max(tmp_elems, t->forward_aligned->tmp_stride());
            }
            t->forward_tmp.reset(sizeof(float2) * (size_t)tmp_elems);
        }
        t->X.reset(t->own_spectrum_bytes());
        if (!t->forward_engine) {
            t->forward = std::make_unique<FftPlan>(FftKind::C2C_FORWARD, (size_t)n, 1, false);
            t->forward_work.reserve(t->forward->work_bytes());
        }
        *out = t.release();
    });
}

int rcfm_tuner_load(rcfm_tuner_t t, const void* x, void* stream) {
    return guarded([&] {
        RC_REQUIRE(t && x, RCFM_ERR_ARG, "NULL argument");
        ArenaScope scope(t->arena);
        RC_REQUIRE(!t->ext_window, RCFM_ERR_STATE,
                   "rcfm_tuner_load needs storage for the whole spectrum: a window is attached (rcfm_tuner_attach_window)");
        {
            StageTimer tm(ST_TUNER_FFT, as_stream(stream));
            bool halo_done = false;
            if (t->forward_engine) {
                // the last pass writes the halos itself (bins near both ends are stored twice): no copy launches
                const bool al = t->use_aligned();
                const FftEngine& eng = al ? *t->forward_aligned : *t->forward_engine;
                FftRowWindow w = al ? (t->windowed_aligned ? t->window_aligned : FftRowWindow{0, (int)(t->n / eng.row_length()) - 1, 0})
                                    : (t->windowed ? t->window : FftRowWindow{0, (int)(t->n / eng.row_length()) - 1, 0});
                w.halo = (int)t->halo;
                halo_done = t->halo > 0;
                // (aligned plan: x -> the spectrum storage as scratch -> forward_tmp -> the spectrum, no pass in place)
                eng.c2c(static_cast<const float2*>(x), t->spectrum(), t->forward_tmp.as<float2>(), 1, false, 1.0f,
                        as_stream(stream), &w, al ? t->X.as<float2>() : nullptr);
            } else {
                t->forward_work.reserve(t->forward->work_bytes());
                t->forward->exec(const_cast<void*>(x), t->spectrum(), t->forward_work.get(), as_stream(stream));
            }
            if (t->halo && !halo_done) {
                float2* X = t->spectrum();
                const size_t hb = sizeof(float2) * (size_t)t->halo;
                RC_HIP(hipMemcpyAsync(X - t->halo, X + t->n - t->halo, hb, hipMemcpyDeviceToDevice, as_stream(stream)));
                RC_HIP(hipMemcpyAsync(X + t->n, X, hb, hipMemcpyDeviceToDevice, as_stream(stream)));
            }
        }
        t->loaded = true;
        t->loaded_windowed = t->forward_engine && (t->use_aligned() ? t->windowed_aligned : t->windowed);
        t->loaded_first = t->shard_first;
        t->loaded_count = t->shard_count;
    });
}

int rcfm_tuner_shard(rcfm_tuner_t t, int first, int count) {
    return guarded([&] {
        RC_REQUIRE(t != nullptr, RCFM_ERR_ARG, "NULL argument");
        t->shard(first, count);
    });
}

int rcfm_tuner_run(rcfm_tuner_t t, int first, int count, void* out, void* stream) {
    return guarded([&] {
        RC_REQUIRE(t && out, RCFM_ERR_ARG, "NULL argument");
        ArenaScope scope(t->arena);
        t->run(first, count, static_cast<float2*>(out), as_stream(stream));
    });
}

int rcfm_tuner_spectrum(rcfm_tuner_t t, void** X) {
    return guarded([&] {
        RC_REQUIRE(t && X, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(!t->ext_window, RCFM_ERR_STATE, "a window is attached: the whole spectrum is not on this device");
        *X = t->spectrum();
    });
}

int rcfm_tuner_spectrum_layout(rcfm_tuner_t t, int64_t* halo, int64_t* n) {
    return guarded([&] {
        RC_REQUIRE(t && halo && n, RCFM_ERR_ARG, "NULL argument");
        *halo = t->halo;
        *n = t->n;
    });
}

int rcfm_tuner_attach_spectrum(rcfm_tuner_t t, void* storage, int loaded_first, int loaded_count) {
    return guarded([&] {
        RC_REQUIRE(t != nullptr, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(loaded_count <= 0 || (loaded_first >= 0 && loaded_first + loaded_count <= t->nch), RCFM_ERR_INDEX,
                   "channel index out of range");
        ArenaScope scope(t->arena);
        t->ext = static_cast<float2*>(storage);
        t->ext_window = false;
        // the handle's own [halo | n | halo] buffer is not needed while the caller supplies the storage
        const size_t own_bytes = t->own_spectrum_bytes();
        if (storage != nullptr) t->X.reset(0);
        else if (t->X.bytes() < own_bytes) t->X.reset(own_bytes);
        // what the storage holds: the bins of channels [loaded_first, loaded_first + loaded_count), or nothing yet
        t->loaded = loaded_count > 0;
        int64_t fb = 0, nb = t->n;
        if (t->loaded) t->bin_window(loaded_first, loaded_count, &fb, &nb);
        t->loaded_windowed = t->loaded && nb < t->n;
        t->loaded_first = loaded_first;
        t->loaded_count = std::max(loaded_count, 0);
    });
}

int rcfm_tuner_window_layout(rcfm_tuner_t t, int first, int count, int64_t* halo, int64_t* nbins) {
    return guarded([&] {
        RC_REQUIRE(t && halo && nbins, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(first >= 0 && count >= 0 && first + count <= t->nch, RCFM_ERR_INDEX, "channel index out of range");
        int64_t fb = 0, nb = 0;
        RC_REQUIRE(t->window_storage_ok(first, count, &fb, &nb), RCFM_ERR_SIZE,
                   "these channels' bins wrap around the ends of the spectrum (or are all of it): no window storage");
        *halo = t->halo;
        *nbins = nb;
    });
}

int rcfm_tuner_attach_window(rcfm_tuner_t t, void* storage, int first, int count) {
    return guarded([&] {
        RC_REQUIRE(t && storage, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(first >= 0 && count >= 0 && first + count <= t->nch, RCFM_ERR_INDEX, "channel index out of range");
        int64_t fb = 0, nb = 0;
        RC_REQUIRE(t->window_storage_ok(first, count, &fb, &nb), RCFM_ERR_SIZE,
                   "these channels' bins wrap around the ends of the spectrum (or are all of it): no window storage");
        t->X.reset(0);
        // bin b of the window sits at storage[halo + b - fb]: `ext` is where bin -halo would be
        t->ext = static_cast<float2*>(storage) - fb;
        t->ext_window = true;
        t->ext_first = first;
        t->ext_count = count;
        t->loaded = false;
        t->loaded_windowed = false;
        t->loaded_first = first;
        t->loaded_count = 0;
    });
}

int rcfm_tuner_set_option(rcfm_tuner_t t, int option, int value) {
    return guarded([&] {
        RC_REQUIRE(t, RCFM_ERR_ARG, "NULL handle");
        switch (option) {
            case RCFM_TUNER_OPT_NARROW_TILES:
                RC_REQUIRE(value >= 0 && value <= 2, RCFM_ERR_ARG, "narrow tiles: 0 never, 1 automatic, 2 always");
                t->opt_narrow = value;
                break;
            case RCFM_TUNER_OPT_ALIGNED_PLAN: t->opt_aligned = value != 0; break;
            default: RC_REQUIRE(false, RCFM_ERR_ARG, "unknown tuner option");
        }
    });
}

int rcfm_tuner_window(rcfm_tuner_t t, int first, int count, int64_t* first_bin, int64_t* nbins) {
    return guarded([&] {
        RC_REQUIRE(t && first_bin && nbins, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(first >= 0 && count >= 0 && first + count <= t->nch, RCFM_ERR_INDEX, "channel index out of range");
        t->bin_window(first, count, first_bin, nbins);
    });
}

int rcfm_tuner_adopt(rcfm_tuner_t t, int first, int count, void* stream) {
    return guarded([&] {
        RC_REQUIRE(t != nullptr, RCFM_ERR_ARG, "NULL argument");
        t->adopt(first, count, as_stream(stream));
    });
}

int rcfm_tuner_destroy(rcfm_tuner_t t) {
    return guarded([&] { delete t; });
}

// ---- demodulators ------------------------------------------------------------

int rcfm_demod_create(int kind, int C, int B, int A, double tau, int chunk, rcfm_demod_t* out) {
    return guarded([&] {
        RC_REQUIRE(out != nullptr, RCFM_ERR_ARG, "out is NULL");
        RC_REQUIRE(kind >= RCFM_FM && kind <= RCFM_WBFM, RCFM_ERR_ARG, "unknown demodulator kind");
        RC_REQUIRE(C >= 1 && B >= 2 && A >= 1, RCFM_ERR_ARG, "bad demodulator size");
        auto d = std::make_unique<rcfm_demod_s>();
        ArenaScope scope(d->arena);
        d->kind = kind;
        d->C = C;
        d->B = B;
        d->A = A;
        d->tau = tau;
        d->ch = (kind == RCFM_WBFM) ? 2 : 1;
        if (chunk <= 0) {
            // Measured on MI355X (profiles/r01_c_chunk_sweep.txt): bigger chunks keep winning -- the kernels are
            // tile-latency bound below ~64 channels per launch, and every launch pays one partial last wave of
            // workgroups: 1024 channels of 240 kHz per launch (9.8 GB of workspace) are 1.3 % faster than 512.
            // Narrow channels take proportionally more per launch (the same workspace): cfg5 (B = 12 500) measured
            // 2.23 / 2.10 / 2.06 / 2.05 ms at 1024 / 2048 / 4096 / 8192 channels per launch.
            chunk = (int)std::min<int64_t>(8192, std::max<int64_t>(1024, (int64_t)1024 * 240000 / B));
        }
        d->chunk = std::min(chunk, C);
        d->geom.build(B, A, 0.54 /* hamm */, false);
        std::memset(d->pilot_h, 0, sizeof(d->pilot_h));
        if (kind == RCFM_WBFM) {
            // wbfm.py:45-46: Bandpass(B, 19e3-50, 19e3+50, num_taps=41); cut-offs relative to Nyquist
            const double nyq = 0.5 * (double)B;
            const double lo = (19e3 - 50) / nyq, hi = (19e3 + 50) / nyq;
            RC_REQUIRE(hi < 1.0, RCFM_ERR_ARG, "Invalid cutoff frequency: frequencies must be greater than 0 and less than fs/2.");
            RC_REQUIRE(B > 3 * 41, RCFM_ERR_ARG, "The length of the input vector x must be greater than padlen, which is 123.");
            auto h = firwin_bandpass(41, lo, hi);
            for (int i = 0; i < 41; ++i) d->pilot_h[i] = (float)h[i];
            auto g = zero_phase_kernel(d->pilot_h, 41);
            std::memcpy(d->pilot_g_h, g.data(), sizeof(d->pilot_g_h));
            d->pilot_g.upload(g.data(), g.size() * sizeof(float));
            d->side_tap = (B % 2) ? (float)(0.23 * std::cos(kPi / (double)B)) : 0.23f;
        }
        if (kind != RCFM_FM) {
            deemphasis_design(A, tau, d->taps_h, d->zi_h);
            d->taps.upload(d->taps_h, sizeof(d->taps_h));
            d->state_buf->reset((size_t)C * d->ch * 50 * sizeof(float));
            d->reset_state(nullptr);
        }
        d->alloc();
        *out = d.release();
    });
}

int rcfm_demod_run(rcfm_demod_t d, int first, int count, const void* iq, void* audio, void* stream) {
    return guarded([&] {
        RC_REQUIRE(d && iq && audio, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(first >= 0 && count >= 0 && first + count <= d->C, RCFM_ERR_INDEX, "channel index out of range");
        ArenaScope scope(d->arena);
        const float2* in = static_cast<const float2*>(iq);
        float* outp = static_cast<float*>(audio);
        if (count == 1 && d->opt_graph && d->eng_B && g_prof.mask == 0 && !d->state_buf->armed) {
            hipStream_t s = as_stream(stream);
            const float* st = d->kind == RCFM_FM ? nullptr : d->state_ptr();
            for (auto& g : d->graphs)
                if (g.iq == iq && g.audio == audio && g.first == first && g.state == st) {
                    g.used = ++d->graph_tick;
                    RC_HIP(hipGraphLaunch(g.exec, s));
                    return;
                }
            const auto& lc = d->last_call;
            if (lc.iq == iq && lc.audio == audio && lc.first == first && lc.state == st) {
                // second call in a row with these pointers: capture (on a stream of the handle's own) and replay
                if (!d->cap_stream) RC_HIP(hipStreamCreateWithFlags(&d->cap_stream, hipStreamNonBlocking));
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                bool ok = hipStreamBeginCapture(d->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    try {
                        d->run_chunk(first, 1, in, outp, d->cap_stream);
                    } catch (...) {
                        ok = false;
                    }
                    if (hipStreamEndCapture(d->cap_stream, &graph) != hipSuccess || graph == nullptr) ok = false;
                }
                if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) ok = false;
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                if (ok) {
                    if (d->graphs.size() >= 8) {   // a host that rotates more than eight buffer pairs: least recently used goes
                        size_t lru = 0;
                        for (size_t i = 1; i < d->graphs.size(); ++i)
                            if (d->graphs[i].used < d->graphs[lru].used) lru = i;
                        (void)hipGraphExecDestroy(d->graphs[lru].exec);
                        d->graphs.erase(d->graphs.begin() + (long)lru);
                    }
                    d->graphs.push_back(rcfm_demod_s::GraphSlot{iq, audio, st, first, exec, ++d->graph_tick});
                    RC_HIP(hipGraphLaunch(exec, s));
                    return;
                }
                d->opt_graph = false;   // this runtime cannot capture the chain: plain launches from now on
            }
            d->last_call = rcfm_demod_s::GraphSlot{iq, audio, st, first, nullptr, 0};
        }
        for (int off = 0; off < count; off += d->chunk) {
            const int cnt = std::min(d->chunk, count - off);
            d->run_chunk(first + off, cnt, in + (size_t)off * d->B, outp + (size_t)off * d->A * d->ch,
                         as_stream(stream));
        }
    });
}

int rcfm_demod_reset_state(rcfm_demod_t d, void* stream) {
    return guarded([&] {
        RC_REQUIRE(d, RCFM_ERR_ARG, "NULL handle");
        d->reset_state(as_stream(stream));
    });
}

int rcfm_demod_get_state(rcfm_demod_t d, float* state_host, void* stream) {
    return guarded([&] {
        RC_REQUIRE(d && state_host, RCFM_ERR_ARG, "NULL argument");
        if (d->kind == RCFM_FM) return;
        RC_HIP(hipMemcpyAsync(state_host, d->state_ptr(), (size_t)d->C * d->ch * 50 * sizeof(float),
                              hipMemcpyDeviceToHost, as_stream(stream)));
        RC_HIP(hipStreamSynchronize(as_stream(stream)));
    });
}

int rcfm_demod_set_state(rcfm_demod_t d, const float* state_host, void* stream) {
    return guarded([&] {
        RC_REQUIRE(d && state_host, RCFM_ERR_ARG, "NULL argument");
        if (d->kind == RCFM_FM) return;
        RC_HIP(hipMemcpyAsync(d->state_ptr(), state_host, (size_t)d->C * d->ch * 50 * sizeof(float),
                              hipMemcpyHostToDevice, as_stream(stream)));
        RC_HIP(hipStreamSynchronize(as_stream(stream)));
    });
}

int rcfm_demod_bind_state(rcfm_demod_t single, rcfm_demod_t batched, int index, int move_history, void* stream) {
    return guarded([&] {
        RC_REQUIRE(single && batched, RCFM_ERR_ARG, "NULL handle");
        RC_REQUIRE(single != batched && single->kind == batched->kind && single->A == batched->A &&
                       single->tau == batched->tau,
                   RCFM_ERR_ARG, "bind_state needs demodulators of one class, audio rate and time constant");
        RC_REQUIRE(index >= 0 && index + single->C <= batched->C, RCFM_ERR_INDEX, "channel index out of range");
        if (single->kind == RCFM_FM) return;   // fm.py carries no state
        const size_t per = (size_t)single->C * single->ch * 50;
        const size_t slot = batched->state_off + (size_t)index * single->ch * 50;
        float* dst = batched->state_buf->as<float>() + slot;
        if (single->state_buf == batched->state_buf && single->state_off == slot) return;   // already bound to this slot
        if (move_history) {
            // the history this demodulator has carried so far moves into the slot (stream-ordered)
            RC_HIP(hipMemcpyAsync(dst, single->state_ptr(), per * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
            RC_HIP(hipStreamSynchronize(as_stream(stream)));   // the old buffer may be freed right below
        }
        single->drop_graphs();
        single->state_buf = batched->state_buf;
        single->state_off = slot;
    });
}

int rcfm_demod_set_option(rcfm_demod_t d, int option, int value) {
    return guarded([&] {
        RC_REQUIRE(d, RCFM_ERR_ARG, "NULL handle");
        d->drop_graphs();
        switch (option) {
            case RCFM_OPT_GRAPH: d->opt_graph = value != 0; break;
            case RCFM_OPT_LDS_CHAIN: d->opt_lds_chain = value != 0; break;
            case RCFM_OPT_FUSED_TILES: d->opt_pilot_chain = d->opt_decim_tile = value != 0; break;
            case RCFM_OPT_PILOT_CHAIN: d->opt_pilot_chain = value != 0; break;
            case RCFM_OPT_DECIM_TILE: d->opt_decim_tile = value != 0; break;
            case RCFM_OPT_PILOT_BLOCKED: d->opt_pilot_blocked = value != 0; break;
            case RCFM_OPT_LDS_DEEMPH: d->opt_lds_deemph = value != 0; break;
            case RCFM_OPT_PHASE_LINK: d->opt_phase_link = value != 0; break;
            case RCFM_OPT_NARROW_TILES:
                RC_REQUIRE(value >= 0 && value <= 2, RCFM_ERR_ARG, "narrow tiles: 0 never, 1 automatic, 2 always");
                d->opt_narrow = value;
                break;
            case RCFM_OPT_STATE_FENCE: {
                auto& b = *d->state_buf;
                if (value != 0 && b.ev == nullptr) RC_HIP(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
                b.armed = value != 0;
                if (!b.armed) b.recorded = false;
                break;
            }
            default: RC_REQUIRE(false, RCFM_ERR_ARG, "unknown demodulator option");
        }
    });
}

int rcfm_demod_get_option(rcfm_demod_t d, int option, int* value) {
    return guarded([&] {
        RC_REQUIRE(d && value, RCFM_ERR_ARG, "NULL handle or output");
        switch (option) {
            case RCFM_OPT_LDS_CHAIN: *value = d->opt_lds_chain; break;
            case RCFM_OPT_FUSED_TILES: *value = d->opt_pilot_chain && d->opt_decim_tile; break;
            case RCFM_OPT_PILOT_CHAIN: *value = d->opt_pilot_chain; break;
            case RCFM_OPT_DECIM_TILE: *value = d->opt_decim_tile; break;
            // the EFFECTIVE value: the switch is on and this handle's geometry has the layout (and the chain that reads it)
            case RCFM_OPT_PILOT_BLOCKED:
                *value = d->opt_pilot_blocked && d->opt_pilot_chain && d->pilot_blocked().valid() &&
                         fused_pilot_chain_applies(*d->eng_B, *d->eng_Bi, 2);
                break;
            case RCFM_OPT_LDS_DEEMPH: *value = d->opt_lds_deemph; break;
            case RCFM_OPT_PHASE_LINK: *value = d->opt_phase_link; break;
            case RCFM_OPT_NARROW_TILES: *value = d->opt_narrow; break;
            case RCFM_OPT_STATE_FENCE: *value = d->state_buf->armed; break;
            // 0 = off (or this runtime refused the capture), 1 = on, 1 + k = on and k captured chains are being replayed
            case RCFM_OPT_GRAPH: *value = d->opt_graph ? 1 + (int)d->graphs.size() : 0; break;
            default: RC_REQUIRE(false, RCFM_ERR_ARG, "unknown demodulator option");
        }
    });
}

int rcfm_demod_get_taps(rcfm_demod_t d, float* deemph51_host, float* pilot41_host) {
    return guarded([&] {
        RC_REQUIRE(d, RCFM_ERR_ARG, "NULL handle");
        if (deemph51_host) std::memcpy(deemph51_host, d->taps_h, sizeof(d->taps_h));
        if (pilot41_host) std::memcpy(pilot41_host, d->pilot_h, sizeof(d->pilot_h));
    });
}

int rcfm_demod_destroy(rcfm_demod_t d) {
    return guarded([&] { delete d; });
}

int rcfm_pipeline_run(rcfm_tuner_t t, rcfm_demod_t d, int first, int count, void* audio, void* stream) {
    return guarded([&] {
        RC_REQUIRE(t && d && audio, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(first >= 0 && count >= 0 && first + count <= t->nch && first + count <= d->C, RCFM_ERR_INDEX,
                   "channel index out of range");
        ArenaScope scope(d->arena);
        d->buf_iq.reserve((size_t)d->chunk * d->B * sizeof(float2));
        float* outp = static_cast<float*>(audio);
        for (int off = 0; off < count; off += d->chunk) {
            const int cnt = std::min(d->chunk, count - off);
            RC_REQUIRE(t->bw[first + off] == d->B, RCFM_ERR_SIZE, "input_sig size and input_size mismatch");
            // Narrow FM / MFM channels whose whole chain fits the LDS of a CU: gather, IFFT_B, discriminator, FFT_B,
            // decimation and IFFT_A of a channel pair in ONE kernel (lds_chain.h); only the audio reaches memory.
            // RCFM_OPT_LDS_CHAIN = 0: the multi-pass launches.
            const bool no_lds = !d->opt_lds_chain;
            if (!no_lds && d->kind != RCFM_WBFM && lds_chain_supported(d->B, d->A) && t->fast_gather_ok(first + off)) {
                RC_REQUIRE(t->loaded, RCFM_ERR_STATE, "rcfm_pipeline_run called before rcfm_tuner_load");
                RC_REQUIRE(!t->loaded_windowed || (first + off >= t->loaded_first &&
                                                   first + off + cnt <= t->loaded_first + t->loaded_count),
                           RCFM_ERR_STATE,
                           "channel outside the shard the spectrum was loaded for (rcfm_tuner_shard, then rcfm_tuner_load)");
                for (int i = 0; i < cnt; ++i)
                    RC_REQUIRE(t->bw[first + off + i] == d->B, RCFM_ERR_SIZE, "input_sig size and input_size mismatch");
                const ResampleGeom& tg = t->band(d->B).geom;
                float* out_c = outp + (size_t)off * d->A;
                // MFM: de-emphasis, mean removal and clip (mfm.py:62-66) run inside the same kernel when the taps are the
                // one-pole response deemphasis.py:37-46 designs (always, unless a caller replaced them): only the audio
                // leaves the chip.  Otherwise the chain stops at the decimated signal and the de-emphasis launches follow.
                const bool deemph_on_chip = d->kind == RCFM_MFM && d->opt_lds_deemph && d->deemph_geometric() &&
                                            lds_chain_deemph_supported(d->B, d->A);
                float* dst = (d->kind == RCFM_FM || deemph_on_chip) ? out_c : d->buf_v.as<float>();
                LdsChainArgs a{t->spectrum(), t->base_dev.as<int32_t>() + first + off, t->n, tg.nyq,
                               tg.nyq_mode == NYQ_DOWN ? tg.nyq - 1 : -1, d->geom.wr.as<float>(), d->geom.scale, dst,
                               d->buf_dc.as<float2>(), cnt};
                if (deemph_on_chip) {
                    a.deemph_taps = d->taps_sfx_dev();
                    a.deemph_state = d->state_ptr() + (size_t)(first + off) * 50;
                    a.dc = nullptr;
                }
                {
                    std::unique_ptr<rcfm_demod_s::StateFence> fence;   // the on-chip de-emphasis reads and writes the state
                    if (deemph_on_chip) fence = std::make_unique<rcfm_demod_s::StateFence>(*d, as_stream(stream));
                    StageTimer tm(ST_LDS_CHAIN, as_stream(stream));
                    RC_REQUIRE(launch_lds_chain(d->B, d->A, a, as_stream(stream)), RCFM_ERR_RUNTIME,
                               "LDS chain refused a geometry it lists");
                }
                if (d->kind == RCFM_MFM && !deemph_on_chip)
                    d->run_deemph(dst, out_c, d->state_ptr() + (size_t)(first + off) * 50, cnt, as_stream(stream), true);
                continue;
            }
            // Every demodulator starts with the FM discriminator, which only needs the samples' phases:
            // the tuner's last pass leaves angle(x) / pi (float32) instead of x (complex64) -- half the
            // bytes written here and read back by the first demod kernel.  RCFM_OPT_PHASE_LINK = 0: complex hand-over.
            const bool no_phase = !d->opt_phase_link;
            if (!no_phase && d->phase_capable() && t->phase_capable(first + off)) {
                // FM / MFM read the phases through LoadPhaseStepPair, which understands padded rows: when the tuner's
                // last pass would store rows of n_1 phases that are not whole 64-byte segments apart (cfg5: n_1 = 100),
                // the rows go to a pitch of whole 128-byte lines.  (WBFM's pilot stage reads contiguous phases.)
                PhaseRows rows;
                const int n1 = t->band_row_length(first + off);
                if (d->kind != RCFM_WBFM && d->eng_B && t->band_two_pass(first + off) && n1 > 0 &&
                    n1 % 16 != 0 && d->B % n1 == 0 && d->B < (1 << 20) && n1 < (1 << 12)) {
                    rows.row = n1;
                    rows.pitch = (n1 + 15) / 16 * 16;     // floats: a 64-byte store segment never straddles a line
                    d->buf_iq.reserve((size_t)d->chunk * rows.channel_stride(d->B) * sizeof(float));
                }
                float* theta = d->buf_iq.as<float>();
                {
                    ArenaScope ts(t->arena);
                    t->run(first + off, cnt, nullptr, as_stream(stream), theta, rows.pitch, d->opt_narrow);
                }
                d->run_chunk(first + off, cnt, nullptr, outp + (size_t)off * d->A * d->ch, as_stream(stream), theta, rows);
                continue;
            }
            {
                ArenaScope ts(t->arena);
                t->run(first + off, cnt, d->buf_iq.as<float2>(), as_stream(stream), nullptr, 0, d->opt_narrow);
            }
            d->run_chunk(first + off, cnt, d->buf_iq.as<float2>(), outp + (size_t)off * d->A * d->ch,
                         as_stream(stream));
        }
    });
}

// ---- host ingest -----------------------------------------------------------------

int rcfm_host_register(void* host, size_t bytes) {
    return guarded([&] {
        RC_REQUIRE(host != nullptr && bytes > 0, RCFM_ERR_ARG, "bad host range");
        RC_HIP(hipHostRegister(host, bytes, hipHostRegisterDefault));
    });
}

int rcfm_host_unregister(void* host) {
    return guarded([&] { RC_HIP(hipHostUnregister(host)); });
}

int rcfm_feeder_create(size_t bytes, int depth, void* const* device_slots, rcfm_feeder_t* out) {
    return guarded([&] {
        RC_REQUIRE(out != nullptr && bytes > 0 && depth >= 1 && depth <= 16, RCFM_ERR_ARG, "bad feeder geometry");
        auto f = std::make_unique<rcfm_feeder_s>();
        f->bytes = bytes;
        f->depth = depth;
        f->owns = device_slots == nullptr;
        RC_HIP(hipStreamCreateWithFlags(&f->copy, hipStreamNonBlocking));
        for (int i = 0; i < depth; ++i) {
            void* p = device_slots ? device_slots[i] : nullptr;
            if (!device_slots) RC_HIP(hipMalloc(&p, bytes));
            RC_REQUIRE(p != nullptr, RCFM_ERR_ARG, "NULL device slot");
            f->slot.push_back(p);
            hipEvent_t a, b;
            RC_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            f->ready.push_back(a);
            RC_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            f->done.push_back(b);
        }
        *out = f.release();
    });
}

int rcfm_feeder_submit(rcfm_feeder_t f, const void* src_host) {
    return guarded([&] {
        RC_REQUIRE(f && src_host, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(f->head - f->tail < (uint64_t)f->depth, RCFM_ERR_STATE,
                   "every feeder slot is in flight: release one before submitting another buffer");
        const int i = (int)(f->head % (uint64_t)f->depth);
        RC_HIP(hipStreamWaitEvent(f->copy, f->done[i], 0));   // the kernels that read this slot last time are finished
        RC_HIP(hipMemcpyAsync(f->slot[i], src_host, f->bytes, hipMemcpyHostToDevice, f->copy));
        RC_HIP(hipEventRecord(f->ready[i], f->copy));
        f->head += 1;
    });
}

int rcfm_feeder_acquire(rcfm_feeder_t f, void* stream, void** dptr) {
    return guarded([&] {
        RC_REQUIRE(f && dptr, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(f->tail < f->head, RCFM_ERR_STATE, "rcfm_feeder_acquire without a submitted buffer");
        const int i = (int)(f->tail % (uint64_t)f->depth);
        RC_HIP(hipStreamWaitEvent(as_stream(stream), f->ready[i], 0));
        *dptr = f->slot[i];
    });
}

int rcfm_feeder_release(rcfm_feeder_t f, void* stream) {
    return guarded([&] {
        RC_REQUIRE(f, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(f->tail < f->head, RCFM_ERR_STATE, "rcfm_feeder_release without an acquired buffer");
        const int i = (int)(f->tail % (uint64_t)f->depth);
        RC_HIP(hipEventRecord(f->done[i], as_stream(stream)));
        f->tail += 1;
    });
}

int rcfm_feeder_copied(rcfm_feeder_t f, uint64_t* count) {
    return guarded([&] {
        RC_REQUIRE(f && count, RCFM_ERR_ARG, "NULL argument");
        // Copies complete in submission order (one copy stream).  ready[i] always refers to the LATEST copy into
        // slot i; when buffer k's slot has been re-submitted since, that newer copy ran behind k on the same stream,
        // so "the newer copy is complete" still implies "k has landed", and "not ready" merely answers conservatively.
        while (f->landed < f->head) {
            const hipError_t e = hipEventQuery(f->ready[(size_t)(f->landed % (uint64_t)f->depth)]);
            if (e == hipErrorNotReady) break;
            RC_HIP(e);
            f->landed += 1;
        }
        *count = f->landed;
    });
}

int rcfm_feeder_destroy(rcfm_feeder_t f) {
    return guarded([&] { delete f; });
}

// ---- multi-GPU: the audio gather ------------------------------------------------------

int rcfm_comm_unique_id(void* id128_host) {
    return guarded([&] {
        RC_REQUIRE(id128_host != nullptr, RCFM_ERR_ARG, "NULL argument");
        static_assert(sizeof(ncclUniqueId) == RCFM_UNIQUE_ID_BYTES, "RCCL unique id size changed");
        ncclUniqueId id;
        RC_NCCL(rccl().GetUniqueId(&id));
        std::memcpy(id128_host, &id, sizeof(id));
    });
}

int rcfm_comm_init_rank(int world, int rank, const void* id128_host, rcfm_comm_t* out) {
    return guarded([&] {
        RC_REQUIRE(out && id128_host && world >= 1 && rank >= 0 && rank < world, RCFM_ERR_ARG, "bad communicator geometry");
        auto c = std::make_unique<rcfm_comm_s>();
        c->world = world;
        c->rank = rank;
        ncclUniqueId id;
        std::memcpy(&id, id128_host, sizeof(id));
        RC_NCCL(rccl().CommInitRank(&c->comm, world, id, rank));
        *out = c.release();
    });
}

int rcfm_gather_audio(rcfm_comm_t c, int root, const void* send, size_t floats_per_rank, void* recv, void* stream) {
    return guarded([&] {
        RC_REQUIRE(c && send, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(root >= 0 && root < c->world, RCFM_ERR_INDEX, "root rank outside the communicator");
        RC_REQUIRE(c->rank != root || recv != nullptr, RCFM_ERR_ARG, "the root rank needs a receive buffer");
        RC_NCCL(rccl().Gather(send, recv, floats_per_rank, ncclFloat32, root, c->comm, as_stream(stream)));
    });
}

// ---- multi-GPU: the spectrum hand-over of the rotating FFT owner -------------------------

int rcfm_comm_group_start(rcfm_comm_t c) {
    return guarded([&] {
        RC_REQUIRE(c, RCFM_ERR_ARG, "NULL communicator");
        RC_NCCL(rccl().GroupStart());
        ++c->group_depth;
    });
}

int rcfm_comm_group_end(rcfm_comm_t c) {
    return guarded([&] {
        RC_REQUIRE(c, RCFM_ERR_ARG, "NULL communicator");
        RC_REQUIRE(c->group_depth > 0, RCFM_ERR_STATE, "rcfm_comm_group_end without rcfm_comm_group_start");
        --c->group_depth;
        RC_NCCL(rccl().GroupEnd());
    });
}

int rcfm_send_bins(rcfm_comm_t c, int peer, const void* bins, size_t nbins, void* stream) {
    return guarded([&] {
        RC_REQUIRE(c && (bins || nbins == 0), RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(peer >= 0 && peer < c->world, RCFM_ERR_INDEX, "peer rank outside the communicator");
        RC_REQUIRE(peer != c->rank || c->group_depth > 0, RCFM_ERR_STATE,
                   "a transfer to this rank itself needs its receive in the same group (rcfm_comm_group_start)");
        if (nbins == 0) return;   // a rank that owns no channels reads no bins
        RC_NCCL(rccl().Send(bins, 2 * nbins, ncclFloat32, peer, c->comm, as_stream(stream)));
    });
}

int rcfm_recv_bins(rcfm_comm_t c, int peer, void* bins, size_t nbins, void* stream) {
    return guarded([&] {
        RC_REQUIRE(c && (bins || nbins == 0), RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(peer >= 0 && peer < c->world, RCFM_ERR_INDEX, "peer rank outside the communicator");
        RC_REQUIRE(peer != c->rank || c->group_depth > 0, RCFM_ERR_STATE,
                   "a transfer from this rank itself needs its send in the same group (rcfm_comm_group_start)");
        if (nbins == 0) return;
        RC_NCCL(rccl().Recv(bins, 2 * nbins, ncclFloat32, peer, c->comm, as_stream(stream)));
    });
}

int rcfm_comm_destroy(rcfm_comm_t c) {
    return guarded([&] {
        if (c && c->comm) (void)rccl().CommDestroy(c->comm);
        delete c;
    });
}

// ---- primitives --------------------------------------------------------------

int rcfm_resampler_create(int C, int n, int m, int is_complex, rcfm_resampler_t* out) {
    return guarded([&] {
        RC_REQUIRE(out != nullptr, RCFM_ERR_ARG, "out is NULL");
        RC_REQUIRE(C >= 1 && n >= 1 && m >= 1, RCFM_ERR_ARG, "bad resampler size");
        auto r = std::make_unique<rcfm_resampler_s>();
        r->C = C;
        r->n = n;
        r->m = m;
        r->cplx = is_complex != 0;
        r->geom.build(n, m, 0.54, r->cplx);
        FftPlanDesc probe;
        if (r->cplx && use_engine() && m <= n && fft_plan_describe(n, &probe) && fft_plan_describe(m, &probe)) {
            r->eng_n = std::make_unique<FftEngine>(n);
            r->eng_m = std::make_unique<FftEngine>(m);
            r->spec_in.reset((size_t)C * n * sizeof(float2));
            r->tmp_n.reset((size_t)C * r->eng_n->tmp_stride() * sizeof(float2));
            r->tmp_m.reset((size_t)C * r->eng_m->tmp_stride() * sizeof(float2));
            std::vector<int64_t> zeros((size_t)C, 0);
            r->zero_roll.upload(zeros.data(), zeros.size() * sizeof(int64_t));
            const int64_t n1 = r->eng_n->row_length(), rows = n / n1;
            const int64_t hi = (r->geom.nyq + 1) / n1;                        // last row of the positive bins
            const int64_t lo = (n - r->geom.nneg - 2) / n1;                   // first row of the negative bins
            if (C == 1 && lo > hi + 1 && lo < rows) {                         // (rows are counted per signal)
                r->windowed = true;
                r->window = FftRowWindow{(int)lo, (int)hi};
            }
            *out = r.release();
            return;
        }
        if (r->cplx) {
            r->fwd = std::make_unique<FftPlan>(FftKind::C2C_FORWARD, (size_t)n, (size_t)C, false);
            r->inv = std::make_unique<FftPlan>(FftKind::C2C_INVERSE, (size_t)m, (size_t)C, true);
            r->spec_in.reset((size_t)C * n * sizeof(float2));
        } else {
            r->fwd = std::make_unique<FftPlan>(FftKind::R2C, (size_t)n, (size_t)C, false);
            r->inv = std::make_unique<FftPlan>(FftKind::C2R, (size_t)m, (size_t)C, false);
            r->spec_in.reset((size_t)C * (n / 2 + 1) * sizeof(float2));
            r->spec_out.reset((size_t)C * (m / 2 + 1) * sizeof(float2));
        }
        r->work.reserve(std::max(r->fwd->work_bytes(), r->inv->work_bytes()));
        *out = r.release();
    });
}

int rcfm_resampler_run(rcfm_resampler_t r, const void* in, void* out, void* stream) {
    return guarded([&] {
        RC_REQUIRE(r && in && out, RCFM_ERR_ARG, "NULL argument");
        hipStream_t s = as_stream(stream);
        const ResampleGeom& g = r->geom;
        if (r->eng_n) {
            r->eng_n->c2c(static_cast<const float2*>(in), r->spec_in.as<float2>(), r->tmp_n.as<float2>(), r->C, false,
                          1.0f, s, r->windowed ? &r->window : nullptr);
            TunerGather tg{r->spec_in.as<float2>(), r->n, r->zero_roll.as<int64_t>(), 0.54, g.nyq, g.nneg, g.nyq_mode,
                           nullptr, 0, r->n};
            fused_tuner_ifft(*r->eng_m, tg, static_cast<float2*>(out), r->tmp_m.as<float2>(), r->C, s);
            return;
        }
        if (r->cplx) {
            r->fwd->exec(const_cast<void*>(in), r->spec_in.get(), r->work.get(), s);
            launch_spectrum_c2c(r->spec_in.as<float2>(), r->n, r->n, nullptr, static_cast<float2*>(out), r->m,
                                r->C, g.wpos.as<float>(), g.wneg.as<float>(), g.w_merge, g.nyq, g.nneg,
                                g.nyq_mode, g.scale, s);
            r->inv->exec(out, out, r->work.get(), s);
        } else {
            r->fwd->exec(const_cast<void*>(in), r->spec_in.get(), r->work.get(), s);
            launch_spectrum_r2c(r->spec_in.as<float2>(), r->n, r->spec_out.as<float2>(), r->m, r->C,
                                g.wr.as<float>(), g.nyq, g.nmin, g.nyq_factor, g.scale, s);
            r->inv->exec(r->spec_out.get(), out, r->work.get(), s);
        }
    });
}

int rcfm_resampler_destroy(rcfm_resampler_t r) {
    return guarded([&] { delete r; });
}

extern "C++" {
namespace {
// Device copies of filter taps, keyed by their values: Bandpass / Deemphasis hand the same host array to every call
// (bandpass.py:72, deemphasis.py:64), so after the first call nothing is allocated, uploaded or waited for.
// The caller holds a shared_ptr across its launch: an eviction on another thread cannot free taps a kernel is about
// to be launched with (the buffer dies when the last holder lets go; hipFree then orders itself behind the launch).
// At most 64 sets are kept, least recently used out first, one per miss.
std::shared_ptr<DeviceBuffer> cached_taps(const std::vector<float>& taps) {
    struct Entry {
        std::shared_ptr<DeviceBuffer> buf;
        uint64_t used;
    };
    static std::mutex mu;
    static std::map<std::vector<float>, Entry> cache;
    static uint64_t tick = 0;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(taps);
    if (it == cache.end()) {
        if (cache.size() >= 64) {
            auto oldest = cache.begin();
            for (auto e = cache.begin(); e != cache.end(); ++e)
                if (e->second.used < oldest->second.used) oldest = e;
            cache.erase(oldest);
        }
        auto buf = std::make_shared<DeviceBuffer>();
        buf->upload(taps.data(), taps.size() * sizeof(float));
        it = cache.emplace(taps, Entry{std::move(buf), 0}).first;
    }
    it->second.used = ++tick;
    return it->second.buf;
}
}  // namespace
}  // extern "C++"

int rcfm_filtfilt(int C, int n, const float* taps_host, int ntaps, const void* x, void* y, void* stream) {
    return guarded([&] {
        RC_REQUIRE(taps_host && x && y, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(C >= 1 && ntaps >= 1, RCFM_ERR_ARG, "bad filtfilt size");
        RC_REQUIRE(n > 3 * ntaps, RCFM_ERR_ARG,
                   "The length of the input vector x must be greater than padlen, which is " +
                       std::to_string(3 * ntaps) + ".");
        hipStream_t s = as_stream(stream);
        const std::shared_ptr<DeviceBuffer> gd = cached_taps(zero_phase_kernel(taps_host, ntaps));
        launch_pilot_stage(nullptr, static_cast<const float*>(x), nullptr, static_cast<float*>(y), n, C,
                           gd->as<float>(), ntaps - 1, 0.f, s);
    });
}

int rcfm_lfilter_fir(int C, int n, const float* taps_host, int ntaps, void* state, const void* x, void* y,
                     void* stream) {
    return guarded([&] {
        RC_REQUIRE(taps_host && x && y && (state || ntaps < 2), RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(C >= 1 && n >= 1 && ntaps >= 1, RCFM_ERR_ARG, "bad lfilter size");
        hipStream_t s = as_stream(stream);
        const std::shared_ptr<DeviceBuffer> td = cached_taps(std::vector<float>(taps_host, taps_host + ntaps));
        launch_fir(static_cast<const float*>(x), static_cast<float*>(y), n, 1, C, td->as<float>(), ntaps,
                   static_cast<const float*>(state), nullptr, s);
        launch_fir_state(static_cast<const float*>(x), n, 1, C, td->as<float>(), ntaps, static_cast<float*>(state), s);
    });
}

extern "C++" {
namespace {
// Plans and workspaces of rcfm_hilbert: PLL.step (pll.py:25-34) is called once per buffer with the same geometry,
// so nothing is planned, allocated or synchronised after the first call.  The key includes the device and the
// STREAM: calls on one stream are ordered by that stream and may share spec / tmp; two PLLs on different streams
// (or threads) get separate workspaces instead of racing on one.  At most 16 entries are kept, least recently
// used out first (hipFree waits for the device, so an evicted workspace is never freed under a running kernel).
struct HilbertPlan {
    std::unique_ptr<FftEngine> eng;            // engine lengths: real -> full spectrum -> masked inverse transform
    std::unique_ptr<FftPlan> fwd, inv;         // otherwise rocFFT (r2c, mask kernel, c2c inverse)
    DeviceBuffer spec, tmp, work;
    uint64_t used = 0;
};
using HilbertKey = std::tuple<int, hipStream_t, int, int>;   // device, stream, n, C
HilbertPlan& hilbert_plan(int n, int C, hipStream_t s) {
    static std::map<HilbertKey, std::unique_ptr<HilbertPlan>> plans;
    static uint64_t tick = 0;
    int dev = 0;
    RC_HIP(hipGetDevice(&dev));
    const HilbertKey key{dev, s, n, C};
    auto it = plans.find(key);
    if (it == plans.end()) {
        if (plans.size() >= 16) {
            auto oldest = plans.begin();
            for (auto e = plans.begin(); e != plans.end(); ++e)
                if (e->second->used < oldest->second->used) oldest = e;
            plans.erase(oldest);
        }
        auto p = std::make_unique<HilbertPlan>();
        FftPlanDesc probe;
        if (use_engine() && fft_plan_describe(n, &probe)) {
            p->eng = std::make_unique<FftEngine>(n);
            p->spec.reset((size_t)C * n * sizeof(float2));
            p->tmp.reset((size_t)C * p->eng->tmp_stride() * sizeof(float2));
        } else {
            p->fwd = std::make_unique<FftPlan>(FftKind::R2C, (size_t)n, (size_t)C, false);
            p->inv = std::make_unique<FftPlan>(FftKind::C2C_INVERSE, (size_t)n, (size_t)C, true);
            p->spec.reset((size_t)C * (n / 2 + 1) * sizeof(float2));
            p->work.reserve(std::max(p->fwd->work_bytes(), p->inv->work_bytes()));
        }
        it = plans.emplace(key, std::move(p)).first;
    }
    it->second->used = ++tick;
    return *it->second;
}
std::mutex g_hilbert_mu;
}  // namespace
}  // extern "C++"

int rcfm_hilbert(int C, int n, const void* x, void* z, void* stream) {
    return guarded([&] {
        RC_REQUIRE(x && z, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(C >= 1 && n >= 1, RCFM_ERR_ARG, "bad hilbert size");
        hipStream_t s = as_stream(stream);
        std::lock_guard<std::mutex> lock(g_hilbert_mu);   // the cache itself; the kernels are ordered by their stream
        HilbertPlan& p = hilbert_plan(n, C, s);
        if (p.eng) {
            fused_real_fft(*p.eng, static_cast<const float*>(x), p.spec.as<float2>(), p.tmp.as<float2>(), C,
                           kKeepLowerHalf /* bins above n/2 are never read */, s);
            fused_hilbert_ifft(*p.eng, p.spec.as<float2>(), static_cast<float2*>(z), p.tmp.as<float2>(), C, s);
            return;
        }
        p.fwd->exec(const_cast<void*>(x), p.spec.get(), p.work.get(), s);
        launch_hilbert_mask(p.spec.as<float2>(), static_cast<float2*>(z), n, C, 1.0f / (float)n, s);
        p.inv->exec(z, z, p.work.get(), s);
    });
}

int rcfm_pll_phase(const void* z, size_t count, double mult, int want_imag, void* out, void* stream) {
    return guarded([&] {
        RC_REQUIRE(z && out, RCFM_ERR_ARG, "NULL argument");
        launch_pll_phase(static_cast<const float2*>(z), count, mult, want_imag, static_cast<float*>(out),
                         as_stream(stream));
    });
}

int rcfm_discriminator(int C, int n, const void* iq, void* d, void* stream) {
    return guarded([&] {
        RC_REQUIRE(iq && d, RCFM_ERR_ARG, "NULL argument");
        launch_discriminator(static_cast<const float2*>(iq), static_cast<float*>(d), n, C, as_stream(stream));
    });
}

// ---- FFT engine --------------------------------------------------------------------

static_assert(sizeof(rcfm_fft_pass) == sizeof(FftPass), "ABI mirror of FftPass out of date");
static_assert(sizeof(rcfm_fft_plan) == sizeof(FftPlanDesc), "ABI mirror of FftPlanDesc out of date");

int rcfm_fft_describe(int64_t n, int max_l, rcfm_fft_plan* plan) {
    return guarded([&] {
        RC_REQUIRE(plan != nullptr, RCFM_ERR_ARG, "plan is NULL");
        FftPlanDesc d;
        RC_REQUIRE(fft_plan_describe(n, &d, max_l), RCFM_ERR_ARG, "length not supported by the FFT engine");
        std::memcpy(plan, &d, sizeof(d));
    });
}

int rcfm_fft_describe_plan(int64_t n, const int64_t* pass_lengths, int npass, int layout, rcfm_fft_plan* plan) {
    return guarded([&] {
        RC_REQUIRE(plan != nullptr && pass_lengths != nullptr, RCFM_ERR_ARG, "NULL argument");
        RC_REQUIRE(layout >= -1 && layout <= 2, RCFM_ERR_ARG, "layout: -1 automatic, 0 plain, 1 tile-blocked, 2 padded rows");
        FftPlanDesc d;
        RC_REQUIRE(fft_plan_describe(n, &d, 0, pass_lengths, npass, layout), RCFM_ERR_ARG,
                   "pass lengths not supported by the FFT engine");
        std::memcpy(plan, &d, sizeof(d));
    });
}

int rcfm_fft_c2c(int64_t n, int batch, int inverse, const void* in, void* out, void* stream) {
    return guarded([&] {
        RC_REQUIRE(in && out && batch >= 1, RCFM_ERR_ARG, "bad argument");
        static std::mutex mu;
        static std::map<int64_t, std::unique_ptr<FftEngine>> engines;
        static DeviceBuffer tmp;
        std::lock_guard<std::mutex> lock(mu);
        auto it = engines.find(n);
        if (it == engines.end()) it = engines.emplace(n, std::make_unique<FftEngine>(n)).first;
        tmp.reserve((size_t)batch * it->second->tmp_stride() * sizeof(float2));
        it->second->c2c(static_cast<const float2*>(in), static_cast<float2*>(out), tmp.as<float2>(), batch,
                        inverse != 0, 1.0f, as_stream(stream));
    });
}

int rcfm_fft_c2c_plan(int64_t n, const int64_t* pass_lengths, int npass, int layout, int batch, int inverse, const void* in,
                      void* out, void* stream) {
    return guarded([&] {
        RC_REQUIRE(in && out && batch >= 1 && pass_lengths && npass >= 1 && npass <= kFftMaxPasses, RCFM_ERR_ARG, "bad argument");
        RC_REQUIRE(layout >= -1 && layout <= 2, RCFM_ERR_ARG, "layout: -1 automatic, 0 plain, 1 tile-blocked, 2 padded rows");
        static std::mutex mu;
        static std::map<std::vector<int64_t>, std::unique_ptr<FftEngine>> engines;
        static DeviceBuffer tmp, tmp2;
        std::lock_guard<std::mutex> lock(mu);
        std::vector<int64_t> key(pass_lengths, pass_lengths + npass);
        key.push_back(layout);
        auto it = engines.find(key);
        if (it == engines.end()) it = engines.emplace(key, std::make_unique<FftEngine>(n, pass_lengths, npass, layout)).first;
        const FftEngine& e = *it->second;
        RC_REQUIRE(e.desc().n == n, RCFM_ERR_ARG, "the pass lengths do not multiply to n");
        tmp.reserve((size_t)batch * e.tmp_stride() * sizeof(float2));
        // the padded-rows layout's intermediates do not fit `out`: a second scratch array, so that (like the default plan
        // of a transform beyond the Infinity Cache) no pass runs in place
        const bool second = layout == 2 && (size_t)n * (size_t)batch * sizeof(float2) > ((size_t)256 << 20);
        if (second) tmp2.reserve((size_t)batch * e.tmp_stride() * sizeof(float2));
        e.c2c(static_cast<const float2*>(in), static_cast<float2*>(out), tmp.as<float2>(), batch, inverse != 0, 1.0f,
              as_stream(stream), nullptr, second ? tmp2.as<float2>() : nullptr);
    });
}

int rcfm_fft_c2c_rocfft(int64_t n, int batch, int inverse, const void* in, void* out, void* stream) {
    return guarded([&] {
        RC_REQUIRE(in && out && batch >= 1 && n >= 1, RCFM_ERR_ARG, "bad argument");
        static std::mutex mu;
        static std::map<std::tuple<int64_t, int, int, int>, std::unique_ptr<FftPlan>> plans;
        static DeviceBuffer work;
        std::lock_guard<std::mutex> lock(mu);
        const bool in_place = (in == out);
        auto key = std::make_tuple(n, batch, inverse ? 1 : 0, in_place ? 1 : 0);
        auto it = plans.find(key);
        if (it == plans.end())
            it = plans.emplace(key, std::make_unique<FftPlan>(inverse ? FftKind::C2C_INVERSE : FftKind::C2C_FORWARD,
                                                               (size_t)n, (size_t)batch, in_place)).first;
        work.reserve(it->second->work_bytes());
        it->second->exec(const_cast<void*>(in), out, work.get(), as_stream(stream));
    });
}

// ---- profiling ------------------------------------------------------------------

int rcfm_profile_stage_count(void) { return ST_COUNT; }

const char* rcfm_profile_stage_name(int stage) {
    return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : "";
}

int rcfm_profile_enable(uint64_t stage_mask) {
    return guarded([&] {
        g_prof.collect();
        g_prof.mask = stage_mask;
    });
}

int rcfm_profile_reset(void) {
    return guarded([&] {
        g_prof.collect();
        std::lock_guard<std::mutex> lock(g_prof.mu);
        for (int i = 0; i < ST_COUNT; ++i) {
            g_prof.total_ms[i] = 0.0;
            g_prof.count[i] = 0;
        }
    });
}

int rcfm_profile_read(int stage, double* total_ms, int64_t* launches) {
    return guarded([&] {
        RC_REQUIRE(stage >= 0 && stage < ST_COUNT && total_ms && launches, RCFM_ERR_ARG, "bad stage");
        g_prof.collect();
        std::lock_guard<std::mutex> lock(g_prof.mu);
        *total_ms = g_prof.total_ms[stage];
        *launches = g_prof.count[stage];
    });
}

}  // extern "C"
