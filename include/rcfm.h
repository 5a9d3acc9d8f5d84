/*
 * rcfm.h -- C ABI of librcfm.so: the MI355X (gfx950) implementation of
 * radio-core's per-buffer DSP hot path (Tuner -> FM / MFM / WBFM).
 *
 * The reference (luigifcruz/radio-core v1.0.0) has no FFI: its device seam is
 * the module swap in radiocore/_internal/injector.py:16-29 (numpy+scipy vs
 * cupy+cusignal).  This header is what a third Injector branch binds instead
 * (INTEGRATION.md shows the ctypes stub).  Every entry point names the
 * reference call site(s) it replaces, relative to the reference checkout.
 *
 * Conventions
 *   - plain C types only; all data pointers are DEVICE pointers unless the
 *     parameter name ends in _host;
 *   - complex data is interleaved float (re, im) = numpy complex64;
 *   - every function returns 0 on success or a negative rcfm_status;
 *     rcfm_last_error() gives the message of the calling thread's last failure;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     are asynchronous on that stream; handles are not thread-safe (the
 *     reference is driven by one DSP thread, examples/multi_fm_server.py:86-106);
 *   - inputs are never modified; outputs are caller-owned; state and
 *     workspaces are owned by the handle and released by *_destroy.
 *   - the library reads ONE environment variable, once per process: RCFM_FFT=rocfft
 *     routes every transform through rocFFT (safety net and A/B partner; the
 *     default is the hand-written engine, rocFFT only for lengths outside it).
 *     Every other choice is an argument or a *_set_option of a handle;
 *   - this header is what a HOST binds (tuner, demodulators, pipeline, ingest, gather, primitives, plain FFT).  The
 *     entry points that exist for the repo's own tools and tests -- placement arenas, explicit FFT plans, the
 *     kernel-form switches, per-stage profiling -- are declared in rcfm_tools.h (same library, same conventions);
 *   - batched arrays are channel-major and contiguous: iq [C][B] complex64,
 *     audio [C][A][ch] float32 (ch = 1 for FM/MFM, 2 = interleaved L,R for WBFM,
 *     the byte layout of the reference's (1, A, 2) array, wbfm.py:94).
 */
#ifndef RCFM_H
#define RCFM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCFM_VERSION 102 /* 0.1.2: tooling entry points moved to rcfm_tools.h (same symbols), RCFM_OPT_GRAPH */

typedef enum rcfm_status {
    RCFM_OK = 0,
    RCFM_ERR_SIZE = -1,    /* -> ValueError("input_sig size and input_size mismatch") fm.py:57-58 */
    RCFM_ERR_INDEX = -2,   /* -> IndexError (bad channel index, tuner.py:151) */
    RCFM_ERR_RUNTIME = -3, /* HIP / rocFFT failure */
    RCFM_ERR_ARG = -4,     /* invalid argument (NULL handle, unsupported size, ...) */
    RCFM_ERR_STATE = -5    /* call order (tuner_run before tuner_load, tuner.py:140-161) */
} rcfm_status;

typedef enum rcfm_demod_kind {
    RCFM_FM = 0,  /* radiocore/analog/fm.py:26-72   */
    RCFM_MFM = 1, /* radiocore/analog/mfm.py:29-71  */
    RCFM_WBFM = 2 /* radiocore/analog/wbfm.py:32-105 */
} rcfm_demod_kind;

typedef struct rcfm_tuner_s* rcfm_tuner_t;
typedef struct rcfm_demod_s* rcfm_demod_t;
typedef struct rcfm_resampler_s* rcfm_resampler_t;
typedef struct rcfm_feeder_s* rcfm_feeder_t;
typedef struct rcfm_comm_s* rcfm_comm_t;

/* ---- library / device ---------------------------------------------------- */

int rcfm_version(void);
const char* rcfm_last_error(void);
/* replaces radiocore.HasCuda() (radiocore/__init__.py:6-26): number of HIP devices */
int rcfm_device_count(int* count);
/* Plain device-memory helpers for hosts that do not bring their own allocator
 * (replace cupy.asarray / cupy.asnumpy at tuner.py:137, fm.py:60,70). */
int rcfm_malloc(void** dptr, size_t bytes);
int rcfm_free(void* dptr);
int rcfm_memcpy_h2d(void* dst, const void* src_host, size_t bytes, void* stream);
int rcfm_memcpy_d2h(void* dst_host, const void* src, size_t bytes, void* stream);
int rcfm_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int rcfm_stream_sync(void* stream);
/* A stream of the host's own for hosts without a HIP header (examples/c_host.c runs two lanes, RCFM_OPT_STATE_FENCE below):
 * hipStreamCreateWithFlags(hipStreamNonBlocking) / hipStreamDestroy.  Every `void* stream` argument of this header takes
 * one (or any hipStream_t, or NULL for the default stream). */
int rcfm_stream_create(void** stream);
int rcfm_stream_destroy(void* stream);

/* ---- Tuner (radiocore/tools/tuner.py) ------------------------------------ */

/* Tuner geometry is host-side Python in both trees (tuner.py:77-124,163-174);
 * the device handle receives the result: n = int(input_bandwidth) wideband
 * samples per buffer, and per channel roll[c] = int(f_in - f_c) (tuner.py:152)
 * and bw[c] = int(bandwidth) (tuner.py:153). */
int rcfm_tuner_create(int64_t n, int nch, const int64_t* roll_host, const int32_t* bw_host,
                      rcfm_tuner_t* out);
/* Tuner.load, tuner.py:126-138: X = FFT_n(x), kept by the handle. x: [n] complex64. */
int rcfm_tuner_load(rcfm_tuner_t t, const void* x, void* stream);
/* Multi-GPU sharding (no reference counterpart: the reference has one device; its loop over all channels is
 * examples/multi_fm_server.py:100-106).  A rank that will only run channels [first, first + count) declares
 * it before rcfm_tuner_load: the last pass of the wideband FFT then stores only the part of the spectrum
 * those channels read (the other bins of the kept spectrum are undefined).  Default: every channel.
 * rcfm_tuner_run / rcfm_pipeline_run remember the range that was in force at load time and fail with
 * RCFM_ERR_STATE for a channel outside it; a new range takes effect with the next rcfm_tuner_load. */
int rcfm_tuner_shard(rcfm_tuner_t t, int first, int count);
/* Tuner.run for channels [first, first+count), tuner.py:140-161: circular shift
 * by roll, fftshifted-Hann weight, brick-wall truncation to bw bins, inverse
 * FFT, x bw/n.  All channels of the range must share one bandwidth B;
 * out: [count][B] complex64. */
int rcfm_tuner_run(rcfm_tuner_t t, int first, int count, void* out, void* stream);
/* Device pointer of the stored spectrum X [n] complex64 (Tuner._buffer). */
int rcfm_tuner_spectrum(rcfm_tuner_t t, void** X);
/* ---- the spectrum as an object that can travel (multi-GPU: rotating FFT owner, radiocore/tools/sharding.py) ------
 * Tuner.load (tuner.py:126-138) keeps the spectrum in `self._buffer`; with channels sharded over G GPUs only ONE GPU
 * needs to run the wideband FFT of a given buffer, if it then hands every peer the bins that peer's channels read.
 *   spectrum_layout   storage = [halo | n bins | halo] complex64 (the halos repeat the far ends)
 *   attach_spectrum   use caller-owned storage of that layout instead of the handle's own (NULL: back to its own);
 *                     loaded_count > 0 declares that it already holds the bins of channels [loaded_first, +count)
 *   window            the bins channels [first, first + count) read: [first_bin, first_bin + nbins) modulo n, whole
 *                     rows of the forward plan (nbins = n: no window, everything)
 *   adopt             the caller has written those bins into the storage (received them over xGMI): refresh the
 *                     halos and accept exactly that channel range in rcfm_tuner_run / rcfm_pipeline_run
 *   window_layout     a rank that never OWNS a given buffer needs room for its window only: storage =
 *                     [halo | nbins | halo] complex64, the window's first bin at element `halo` (at G = 8 an eighth of
 *                     a spectrum per slot instead of a whole one).  RCFM_ERR_SIZE when the window wraps around bin 0
 *                     or touches the far ends (their halos repeat the other end): such a rank keeps whole slots
 *   attach_window     use such a storage for channels [first, first + count): receive the window's bins into
 *                     elements [halo, halo + nbins), then adopt(first, count).  rcfm_tuner_load and
 *                     rcfm_tuner_spectrum fail with RCFM_ERR_STATE while a window is attached; channels outside
 *                     the range are refused.  attach_spectrum (storage or NULL) ends it.                          */
int rcfm_tuner_spectrum_layout(rcfm_tuner_t t, int64_t* halo, int64_t* n);
int rcfm_tuner_attach_spectrum(rcfm_tuner_t t, void* storage, int loaded_first, int loaded_count);
int rcfm_tuner_window(rcfm_tuner_t t, int first, int count, int64_t* first_bin, int64_t* nbins);
int rcfm_tuner_window_layout(rcfm_tuner_t t, int first, int count, int64_t* halo, int64_t* nbins);
int rcfm_tuner_attach_window(rcfm_tuner_t t, void* storage, int first, int count);
int rcfm_tuner_adopt(rcfm_tuner_t t, int first, int count, void* stream);
int rcfm_tuner_destroy(rcfm_tuner_t t);

/* ---- demodulators (radiocore/analog/{fm,mfm,wbfm}.py) --------------------- */

/* C independent channels (or C consecutive buffers of distinct channels) of
 * identical geometry B -> A.  tau = deemphasis rate (mfm.py:32, wbfm.py:35).
 * chunk = channels processed per pass through the kernel chain (0 = default: 1024 at B = 240 000, proportionally
 * more for narrower channels up to 8192, i.e. the same workspace): bounds the workspace (about 9.6 MB per channel of
 * a 240 kHz WBFM chunk). */
int rcfm_demod_create(int kind, int C, int B, int A, double tau, int chunk, rcfm_demod_t* out);
/* FM.run / MFM.run / WBFM.run on channels [first, first+count):
 * iq [count][B] complex64 -> audio [count][A][ch] float32. */
int rcfm_demod_run(rcfm_demod_t d, int first, int count, const void* iq, void* audio,
                   void* stream);
/* De-emphasis filter state, the reference's Deemphasis._state (deemphasis.py:48-49,64):
 * [C][ch][50] float32, host memory.  reset = lfilter_zi(taps) for every channel. */
int rcfm_demod_reset_state(rcfm_demod_t d, void* stream);
int rcfm_demod_get_state(rcfm_demod_t d, float* state_host, void* stream);
int rcfm_demod_set_state(rcfm_demod_t d, const float* state_host, void* stream);
/* Design outputs for parity checks: 51 de-emphasis taps, 41 pilot band-pass taps
 * (host, float32).  Either pointer may be NULL. */
int rcfm_demod_get_taps(rcfm_demod_t d, float* deemph51_host, float* pilot41_host);
/* One de-emphasis state per channel, whoever runs it: makes the one-channel handle `single` (the demodulator object
 * a Channel carries, tuner.py:24) keep its state in slot `index` of `batched` (the handle rcfm_pipeline_run uses for
 * all channels) from now on; move_history != 0: what `single` has carried so far is moved into the slot (it has run
 * before), 0: the slot's content stands (`single` is new, the batched handle has the history).  Afterwards a caller may mix
 * `demodulator.run(tuner.run(i))` (multi_fm_server.py:101-102) and the batched call across buffers and get the
 * reference's results: there, the state lives in the demodulator object (deemphasis.py:48-49,64) and nowhere else.
 * `single` may itself hold several channels (slots [index, index + its C) are taken: two batched handles of one
 * geometry then share one state).  Same class, audio rate and time constant required; FM has no state (no-op). */
int rcfm_demod_bind_state(rcfm_demod_t single, rcfm_demod_t batched, int index, int move_history, void* stream);
/* Per-handle options a HOST may need (the kernel-form switches that tools and tests use live in rcfm_tools.h).
 *   RCFM_OPT_NARROW_TILES the tile kernels exist with 16 and with 8 lines per tile: 0 = always 16, 1 (default) = 8 when a
 *                         launch has fewer than two 16-line tiles per CU (one WBFM.run per call, the reference's harness
 *                         shape tests/benchmark.py:29-31: 60 short tiles instead of 30 long ones), 2 = always 8; applies
 *                         to the demodulator's kernels and, in rcfm_pipeline_run, to the tuner's inverse FFT of the chunk
 *   RCFM_OPT_STATE_FENCE  (default 0) consecutive buffers on DIFFERENT streams: the reference's loop
 *                         (examples/multi_fm_server.py:98-106) handles one buffer at a time; a host that keeps two handle
 *                         sets (tuner + demodulator, the second demodulator bound to the first one's state with
 *                         rcfm_demod_bind_state) and alternates them, each on its own stream, lets the kernels of buffer
 *                         i + 1 fill the gaps of buffer i.  The de-emphasis state is the one thing buffer i + 1 needs
 *                         from buffer i (deemphasis.py:64): with the fence on, every launch sequence that touches the
 *                         shared state waits for the event the previous one recorded, on whichever stream that was.
 *                         Set it on any ONE handle of the sharing group, after the binding
 *   RCFM_OPT_GRAPH        (default 0) a handle of ONE channel (the reference's per-channel call, fm.py:46 / mfm.py:51 /
 *                         wbfm.py:66; tests/benchmark.py:29-31 times exactly these) replays its launch chain from a
 *                         captured hipGraph when the call's pointers and stream kind repeat: one graph launch instead of
 *                         ten kernel launches.  0: plain launches.  Results are bit-identical either way.  Off by default: on ROCm 7 the
 *                         graph launch costs as much as the launches it replaces (profiles/r06_a_single_call.txt) */
enum { RCFM_OPT_NARROW_TILES = 4, RCFM_OPT_STATE_FENCE = 5, RCFM_OPT_GRAPH = 10 };
int rcfm_demod_set_option(rcfm_demod_t d, int option, int value);
int rcfm_demod_destroy(rcfm_demod_t d);

/* Whole hot path for one wideband buffer already loaded with rcfm_tuner_load:
 * the loop of examples/multi_fm_server.py:100-106 (run -> demodulator.run) for
 * channels [first, first+count), chunk by chunk.  audio: [count][A][ch]. */
int rcfm_pipeline_run(rcfm_tuner_t t, rcfm_demod_t d, int first, int count, void* audio,
                      void* stream);

/* ---- host ingest (the step before Tuner.load) -------------------------------- */

/* The reference hands its DSP thread a buffer in mapped / shared memory (cusignal.get_shared_mem at
 * radiocore/tools/buffer.py:43 and ringbuffer.py:51; consumer loop examples/multi_fm_server.py:95-98).  Here the
 * host side is page-locked memory and the wideband buffer crosses PCIe once: a feeder owns `depth` device slots of
 * `bytes` each, a copy stream and one event pair per slot, so the copy of buffer i+1 runs under the kernels of
 * buffer i.  device_slots = NULL: the library allocates the slots; otherwise `depth` caller-owned device pointers.
 *   submit(src_host)      queue the H2D copy of the next buffer (page-locked source: rcfm_host_register, or any
 *                         pinned allocation) into the next free slot; RCFM_ERR_STATE when all slots are in flight
 *   acquire(stream, &p)   make `stream` wait for the oldest submitted copy; p = its device slot (pass it to
 *                         rcfm_tuner_load on the same stream)
 *   release(stream)       the work queued on `stream` so far is the last reader of that slot
 *   copied(&count)        how many submitted buffers have LANDED in their slot (host-side query, no wait): source
 *                         buffers [0, count) may be reused or freed by the producer (the RingBuffer's read pointer
 *                         may pass them, ringbuffer.py:129-160)                                              */
int rcfm_host_register(void* host, size_t bytes);
int rcfm_host_unregister(void* host);
int rcfm_feeder_create(size_t bytes, int depth, void* const* device_slots, rcfm_feeder_t* out);
int rcfm_feeder_submit(rcfm_feeder_t f, const void* src_host);
int rcfm_feeder_acquire(rcfm_feeder_t f, void* stream, void** dptr);
int rcfm_feeder_release(rcfm_feeder_t f, void* stream);
int rcfm_feeder_copied(rcfm_feeder_t f, uint64_t* count);
int rcfm_feeder_destroy(rcfm_feeder_t f);

/* ---- multi-GPU: the audio gather (the publish step, examples/multi_fm_server.py:103-106) ---------- */

/* One process per GPU; channels shard by contiguous index range (radiocore/tools/sharding.py) and the only
 * collective of the path brings every rank's [C/G][A][ch] float32 block to the publishing rank over xGMI.  RCCL is
 * opened at run time by the first of these calls (a single-GPU process never loads it).
 *   unique_id(id)                 rank 0 creates the 128-byte rendezvous token; the host distributes it to the other
 *                                 ranks by whatever control channel it has (file, socket, MPI, torch.distributed store)
 *   comm_init_rank(G, r, id, &c)  collective: every rank calls it once, with its GPU current (hipSetDevice)
 *   gather_audio(c, root, send, floats, recv, stream)
 *                                 `floats` float32 values from every rank land at recv + rank * floats on `root`
 *                                 (recv may be NULL elsewhere), asynchronously on `stream`: rank blocks are equal,
 *                                 pad the shorter ones when C is not divisible by G (sharding.gather_audio does).  */
#define RCFM_UNIQUE_ID_BYTES 128
int rcfm_comm_unique_id(void* id128_host);
int rcfm_comm_init_rank(int world, int rank, const void* id128_host, rcfm_comm_t* out);
int rcfm_gather_audio(rcfm_comm_t c, int root, const void* send, size_t floats_per_rank, void* recv, void* stream);
/* The rotating FFT owner's hand-over without torch.distributed (Python: radiocore.tools.sharding.SpectrumRing; no
 * reference counterpart -- its one process keeps Tuner._buffer, tuner.py:57,138, to itself).  Rank i mod G runs the
 * wideband FFT of buffer i into a spectrum slot (rcfm_tuner_attach_spectrum + rcfm_tuner_load) and sends every peer the
 * bins that peer's channels read (rcfm_tuner_window: at most two contiguous pieces of the circular spectrum); the peer
 * receives them into the same positions of its own slot and calls rcfm_tuner_adopt.  Point-to-point over xGMI:
 *   group_start / group_end   bracket all sends and receives of one buffer (ncclGroupStart / ncclGroupEnd): they are
 *                             posted together, so no ordering between peers can deadlock
 *   send_bins(c, peer, p, n)  n complex64 bins from device pointer p to rank `peer`, asynchronously on `stream`
 *   recv_bins(c, peer, p, n)  the matching receive.  peer = this rank is allowed inside a group (send + receive to
 *                             itself: a device copy) -- a one-GPU host can run the whole protocol.
 * examples/c_host.c runs it on a one-rank communicator. */
int rcfm_comm_group_start(rcfm_comm_t c);
int rcfm_send_bins(rcfm_comm_t c, int peer, const void* bins, size_t nbins, void* stream);
int rcfm_recv_bins(rcfm_comm_t c, int peer, void* bins, size_t nbins, void* stream);
int rcfm_comm_group_end(rcfm_comm_t c);
int rcfm_comm_destroy(rcfm_comm_t c);

/* ---- primitives (class parity with radiocore/analog) ---------------------- */

/* Decimate, decimate.py:21-50 = scipy.signal.resample with the fftshifted
 * periodic Hamming window: C signals of n samples -> m samples.
 * is_complex = 0: float32 in/out; 1: complex64 in/out (receive_fm.py:80). */
int rcfm_resampler_create(int C, int n, int m, int is_complex, rcfm_resampler_t* out);
int rcfm_resampler_run(rcfm_resampler_t r, const void* in, void* out, void* stream);
int rcfm_resampler_destroy(rcfm_resampler_t r);

/* Bandpass.run, bandpass.py:59-74 = filtfilt(taps, [1], x), padtype odd:
 * x [C][n] float32 -> y [C][n] float32; taps_host: ntaps float32 (firwin output,
 * designed on the host in both trees, bandpass.py:50-54).  n > 3*ntaps. */
int rcfm_filtfilt(int C, int n, const float* taps_host, int ntaps, const void* x, void* y,
                  void* stream);
/* Deemphasis.run, deemphasis.py:51-66 = lfilter(taps, 1, x, zi=state):
 * x [C][n] -> y [C][n] float32; state [C][ntaps-1] float32 DEVICE, updated in place. */
int rcfm_lfilter_fir(int C, int n, const float* taps_host, int ntaps, void* state, const void* x,
                     void* y, void* stream);
/* PLL.step, pll.py:25-34 = scipy.signal.hilbert: x [C][n] float32 -> z [C][n] complex64. */
int rcfm_hilbert(int C, int n, const void* x, void* z, void* stream);
/* PLL.real / PLL.image, pll.py:36-58: out = Re or Im of z^mult / |z^mult|;
 * z [count] complex64 -> out [count] float32.  Integer mult in [1, 64] is multiplied
 * out like numpy's complex power; any other mult uses the principal branch. */
int rcfm_pll_phase(const void* z, size_t count, double mult, int want_imag, void* out,
                   void* stream);
/* FM discriminator, fm.py:60-65: iq [C][n] complex64 -> d [C][n] float32 (d[0] = 0). */
int rcfm_discriminator(int C, int n, const void* iq, void* d, void* stream);

/* ---- FFT engine (the hand-written replacement of cupy.fft / scipy.fft calls) -- */

/* Unnormalised forward (inverse = 0) or conjugate (inverse = 1) transform of `batch`
 * contiguous length-n complex64 signals; in == out allowed.  (scipy.fft.fft / ifft*n) */
int rcfm_fft_c2c(int64_t n, int batch, int inverse, const void* in, void* out, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* RCFM_H */
