// Launch wrappers of the hand-written gfx950 kernels (kernels.hip).
// All pointers are device pointers; all launches are asynchronous on `stream`.
#pragma once

#include "common.h"

namespace rcfm {

// How scipy.signal.resample treats the Nyquist bin when min(n, m) is even
// (scipy 1.15.3 _signaltools.py, "Split/join Nyquist component(s)").
enum NyquistMode : int { NYQ_NONE = 0, NYQ_DOWN = 1, NYQ_UP = 2 };

// Y[c][k] (m bins) <- roll + window + truncate/zero-pad of X[c][.] (n bins), complex
// spectra.  Shared by Tuner.run (tuner.py:159-161, x_stride = 0: every channel reads
// the same wideband spectrum) and complex Decimate (decimate.py:48).
//   wpos[k], k in [0, nyq): weight of source bin k; wneg[j], j in [1, nneg]: weight
//   of source bin n-j (wneg[0] unused); w_merge: weight of bin n - nmin/2 (NYQ_DOWN).
//   roll[c] in [0, n): Xrolled[k] = X[(k - roll) mod n]; may be null (no roll).
void launch_spectrum_c2c(const float2* X, int64_t x_stride, int64_t n, const int64_t* roll, float2* Y,
                         int64_t m, int batch, const float* wpos, const float* wneg, float w_merge,
                         int nyq, int nneg, int nyq_mode, float scale, hipStream_t stream);

// Half-spectrum version for real signals (FM/MFM audio decimation, real Decimate):
// X [batch][n/2+1] -> Y [batch][m/2+1]; wr = folded window, nyq_factor = 2, 0.5 or 1.
void launch_spectrum_r2c(const float2* X, int64_t n, float2* Y, int64_t m, int batch, const float* wr,
                         int nyq, int nmin, float nyq_factor, float scale, hipStream_t stream);

// Same resampling rule, but from the FULL complex spectrum X [batch][n] of a real signal to
// the full Hermitian spectrum Y [batch][m] (feeds a complex inverse FFT whose real part is y).
// dc (optional, [batch]): receives Y[c][0], the DC bin (sum of the resampled signal / m).
// paired: X is [ceil(batch/2)][n], one complex FFT per pair of real signals (x[2p] + j x[2p+1]).
void launch_spectrum_real_full(const float2* X, int64_t n, float2* Y, int64_t m, int batch, const float* wr,
                               int nyq, int nmin, float nyq_factor, float scale, float2* dc, bool paired,
                               hipStream_t stream);

// scipy.signal.hilbert's mask (pll.py:34): P [batch][n/2+1] half spectrum of the real
// input -> Z [batch][n] one-sided spectrum {1, 2, ..., 2, 1, 0, ...} * scale.
void launch_hilbert_mask(const float2* P, float2* Z, int64_t n, int batch, float scale,
                         hipStream_t stream);

// WBFM audio decimation of both stereo legs at once (wbfm.py:86-87): U = FFT_B of the
// packed signal (m+lmr) + j(m-lmr); V [batch][A] = packed spectrum whose inverse FFT is
// l + j r, with the Hamming weight, truncation and Nyquist rule of decimate.py:48.
// dc (optional, [batch]): receives V[c][0] = (sum l, sum r) / A.
void launch_stereo_unpack(const float2* U, int64_t B, float2* V, int64_t A, int batch, const float* wr,
                          int nyq, int nmin, float nyq_factor, float scale, float2* dc, hipStream_t stream);

// fm.py:60-65: d[0] = 0, d[i] = arg(x[i] conj(x[i-1])) / pi.
void launch_discriminator(const float2* iq, float* d, int64_t n, int batch, hipStream_t stream);

// Fused front end of WBFM.run (wbfm.py:77-80): discriminator -> same-size Decimate
// (3-tap circular Hamming, side tap = 0.23 or 0.23 cos(pi/n)) -> m; zero-phase pilot
// band-pass (filtfilt as a (2H+1)-tap symmetric FIR over an odd extension) -> p.
// g: H+1 taps, g[0] = centre.  iq != null: full chain.  iq == null: x is the FIR input
// (Bandpass.run primitive) and m_out is not written.
void launch_pilot_stage(const float2* iq, const float* x, float* m_out, float* p_out, int64_t n,
                        int batch, const float* g, int H, float side_tap, hipStream_t stream);

// Tile-blocked layout of the real signals m and p for readers that walk them as 16-line tiles of an L x R
// matrix (sample t = k R + i: the first pass of the pair FFT and the point-wise stage of k_fft_tile2_pair):
// element (k, i) of channel c sits at  c stride + (i / 16) bs + 16 k + i % 16  (bs >= 16 L),  so the 16 L values of a tile
// are one contiguous run instead of L 64-byte half lines R floats apart (each fetched as a whole line, half of
// them twice).  valid() = the pilot stage can write it: whole row PAIRS per workgroup (rpt rows, even), so that
// every 128-byte line (rows k, k + 1 of one 16-column block) leaves in one store instruction.
struct PilotBlocked {
    int R = 0, L = 0;       // row length (columns), rows: n = L R
    int rpt = 0, nb = 0;    // rows per workgroup (even, rpt R <= 2048), 16-column blocks per row
    int bs = 0;             // floats from one 16-column block to the next (>= 16 L)
    unsigned half_magic = 0;   // ceil(2^32 / (rpt / 2)): the stage's one division as a multiply-high
    int64_t stride = 0;     // floats per channel: nb * bs
    bool valid() const { return rpt > 0; }
    int blk16() const { return valid() ? bs : 0; }
    // the layout for an n = L R signal, or an invalid one when the stage would waste more than an eighth of its
    // threads (rpt R < 1792 of 2048), R is not a multiple of 4 (16-byte stores) or the signal is shorter than a workgroup
    static PilotBlocked plan(int64_t L, int64_t R);
};

// The same chain specialised for WBFM's 41-tap pilot filter (H = 40); g_host: 41 taps on the host.
// Needs n % 4 == 0 (16-byte stores) and n > 123.  blk (optional, valid()): m and p leave tile-blocked.
void launch_pilot_stage_h40(const float2* iq, float* m_out, float* p_out, int64_t n, int batch,
                            const float* g_host, float side_tap, hipStream_t stream, const PilotBlocked* blk = nullptr);
// The same chain from the samples' phases theta = angle(x) / pi (what the pipeline's tuner stage leaves,
// fused_tuner_ifft): d[i] = the phase step wrapped into [-1, 1], exactly diff(unwrap(angle(x))) / pi.
void launch_pilot_stage_h40_phase(const float* theta, float* m_out, float* p_out, int64_t n, int batch,
                                  const float* g_host, float side_tap, hipStream_t stream, const PilotBlocked* blk = nullptr);
void launch_discriminator_phase(const float* theta, float* d, int64_t n, int batch, hipStream_t stream);

// wbfm.py:83,86-87: s2 = Im(z^2)/|z^2|; lmr = s2 m 1.0175; u = (m + lmr) + j (m - lmr).
void launch_stereo_mix(const float2* z, const float* m, float2* u, size_t count, hipStream_t stream);

// pll.py:36-58 for arbitrary multiplier.
void launch_pll_phase(const float2* z, size_t count, double mult, int want_imag, float* out,
                      hipStream_t stream);

// lfilter(taps, 1, x, zi=state) for an FIR (deemphasis.py:64).  x, y: [batch][n][ch]
// interleaved; state: [batch][ch][nb-1] transposed-direct-form-II state (read only).
// partial (optional): [batch][ch][fir_tiles(n)] per-tile sums of y for the DC removal.
int fir_tiles(int64_t n);
void launch_fir(const float* x, float* y, int64_t n, int ch, int batch, const float* taps, int nb,
                const float* state, float* partial, hipStream_t stream);
// The 51-tap de-emphasis case, register-blocked (n*ch % 4 == 0 for its 16-byte stores);
// partial: [batch][fir51_tiles(n, ch)] (both stereo legs share a tile).  taps_host: 51 floats.
int fir51_tiles(int64_t n, int ch);
// dc != nullptr ([batch], from launch_stereo_unpack / launch_spectrum_real_full): the kernel also
// removes the mean and clips (mfm.py:64-65, wbfm.py:97-100); `partial` is then unused.  Needs n >= 50.
// row_samples / row_pitch_samples (fir51 and fir_state): the input signal is stored in rows of row_samples samples at a
// pitch of row_pitch_samples (0 = contiguous): the padded output of fused_fft_decim_ifft.
void launch_fir51(const float* x, float* y, int64_t n, int ch, int batch, const float* taps_host,
                  const float* state, float* partial, const float2* dc, hipStream_t stream, int row_samples = 0,
                  int row_pitch_samples = 0);
// Advance `state` to the end of the buffer (must run after launch_fir on the stream).
void launch_fir_state(const float* x, int64_t n, int ch, int batch, const float* taps, int nb,
                      float* state, hipStream_t stream, int row_samples = 0,
                      int row_pitch_samples = 0);
// mfm.py:64-65 / wbfm.py:97-100: y -= mean(y) over the channel's n*ch samples; clip +-0.999.
// partial: [batch][nparts] partial sums of y.
void launch_dc_clip(float* y, int64_t n, int ch, int batch, const float* partial, int nparts,
                    hipStream_t stream);

}  // namespace rcfm
