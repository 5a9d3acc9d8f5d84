// Declarations of fused_passes.hip, included by fused_passes.h once per tile width (no include guard on purpose).

namespace rcfm {
RCFM_DECL_NS_OPEN


// theta != nullptr: instead of out, angle(ifft) / pi goes to theta [count][B] float32 -- all an FM
// discriminator needs (fm.py:60-65), and half the bytes.
// theta_pitch > 0: phase k_1 + n_1 k_2 of a channel goes to theta[k_2 theta_pitch + k_1] (n_1 = e's first pass length; a
// channel then occupies (B / n_1) theta_pitch values): rows of n_1 = 100 phases start every 400 bytes and their 64-byte
// store segments straddle lines; at a pitch of 112 they are aligned (PhaseRows describes the layout to the reader).
void fused_tuner_ifft(const FftEngine& e, const TunerGather& g, float2* out, float2* tmp, int count,
                      hipStream_t s, float* theta = nullptr, int theta_pitch = 0);


// Forward FFT of real signals x [count][n] -> full complex spectrum U [count][n].
// keep >= 0: only bins |k| <= keep are written (the rest of U is left untouched).
void fused_real_fft(const FftEngine& e, const float* x, float2* U, float2* tmp, int count, int keep,
                    hipStream_t s);

// pll.py:34 + wbfm.py:83,86-87: z = ifft(h U) (scipy.signal.hilbert's one-sided mask as the
// load of the first pass), then s2 = Im(z^2)/|z^2|, lmr = s2 m 1.0175 and the packed stereo
// signal u = (m + lmr) + j (m - lmr) as the store of the last pass.  U and u may alias.
void fused_hilbert_ifft_mix(const FftEngine& e, const float2* U, const float* m, float2* u, float2* tmp,
                            int count, hipStream_t s);

// scipy.signal.hilbert (pll.py:34): z = ifft(h U) / n for full spectra U [count][n] of real signals; U and z may alias.
void fused_hilbert_ifft(const FftEngine& e, const float2* U, float2* z, float2* tmp, int count, hipStream_t s);

// The same two stages for real signals transformed two at a time: U2 [ceil(count/2)][n] =
// FFT(x[2c] + j x[2c+1]); the Hilbert load of channel c unpacks its own spectrum from U2.
// U2 must not alias u (u is written while other channels still read their pair).
// keep >= 0: only bins |k| <= keep of U2 are written; kKeepLowerHalf: only bins 0 .. n/2 (all the packed
// Hilbert chain below reads).
// from_phase: x holds angle / pi of complex samples and the real signals are its wrapped steps -- the FM
// discriminator (fm.py:60-65) computed on the load, d[0] = 0.
void fused_real_pair_fft(const FftEngine& e, const float* x, float2* U2, float2* tmp, int count, int keep,
                         hipStream_t s, bool from_phase = false, PhaseRows rows = PhaseRows{});
void fused_hilbert_pair_ifft_mix(const FftEngine& e, const float2* U2, const float* m, float2* u, float2* tmp,
                                 int count, hipStream_t s);

// Analytic-signal IFFT, stereo mix and the FIRST pass of the packed L/R FFT in two launches:
// `ei` is the plan of length n with its two pass lengths swapped relative to `ef`, so ei's last
// pass and ef's first pass own the same tiles and run as one kernel (k_fft_tile2); the mixed
// signal u never reaches memory.  Leaves ef's scratch (tmp_f) ready for fused_fft_last_pruned.
// Returns false (nothing launched) when the two plans do not pair up.
bool fused_hilbert_pair_ifft_mix_fft(const FftEngine& ei, const FftEngine& ef, const float2* U2, const float* m,
                                     float2* tmp_i, float2* tmp_f, int count, hipStream_t s);
// The same chain with the PAIR kept packed through the inverse transform: w = IFFT(h U2) = z0 + j z1
// (z_c = p_c + j Hilbert(p_c)), so Hilbert(p1) = p0 - Re w and Hilbert(p0) = Im w - p1: one inverse
// transform per two channels; the mix reads p as well as m (k_fft_tile2_pair).
bool fused_hilbert_packed_applies(const FftEngine& ei, const FftEngine& ef, int count);
bool fused_hilbert_packed_ifft_mix_fft(const FftEngine& ei, const FftEngine& ef, const float2* U2, const float* p,
                                       const float* m, float2* tmp_i, float2* tmp_f, int count, hipStream_t s);
// The whole pilot chain of WBFM (wbfm.py:80-87) in three launches when ef = (n_1, n_2) and ei = (n_2, n_1)
// are the two two-pass plans of one length:  (1) first pass of the pair FFT of p (two channels per complex
// signal);  (2) its last pass + scipy.signal.hilbert's mask + the first pass of the inverse FFT on one
// tile (k_fft_tile2): the pair spectrum never reaches memory;  (3) the inverse FFT's last pass + split
// into the two analytic signals + stereo mix + first pass of each member's packed L/R FFT
// (k_fft_tile2_pair).  Leaves ef's scratch tmp_f ready for fused_fft_last_pruned.
// blk16 != 0: p and m are in the tile-blocked layout of PilotBlocked (kernels.h) for the L x R view of the signal
// that fused_pilot_chain_geometry reports (rows = points of the pair FFT's first pass, row_length = their distance):
// both readers then fetch whole contiguous tiles instead of half lines (blk_stride floats per channel, blk16 = 16 L).
bool fused_pilot_chain_applies(const FftEngine& ef, const FftEngine& ei, int count);
bool fused_pilot_chain_geometry(const FftEngine& ef, int64_t* rows, int64_t* row_length);
void fused_pilot_chain_fft_first(const FftEngine& ef, const float* p, float2* tmp_f, int count, hipStream_t s,
                                 int64_t blk_stride = 0, int blk16 = 0);
void fused_pilot_chain_mask_mix(const FftEngine& ef, const FftEngine& ei, const float* p, const float* m,
                                float2* tmp_f, float2* tmp_i, int count, hipStream_t s, int64_t blk_stride = 0,
                                int blk16 = 0);

void fused_fft_last_pruned(const FftEngine& ef, const float2* tmp_f, float2* out, int count, int keep,
                           hipStream_t s);

// wbfm.py:86-87 without the long spectrum: the packed L/R FFT's last pass (plan ef = (n_1, L) of length B,
// scratch tmp_f from the pilot chain), the decimation to A = n_1 L2 (window, truncation, Nyquist merge,
// decimate.py:48) and the first pass of IFFT_A (plan ea = (L2, n_1)) on one tile (k_fft_tile2_decim), then
// ea's last pass.  out [count][A] = l + j r (= float32 [count][A][2]); dc as below.
bool fused_fft_decim_ifft_applies(const FftEngine& ef, const FftEngine& ea, int count);
// out_pitch > 0: sample k_1 + n_1 k_2 of a signal goes to out[k_2 out_pitch + k_1] (n_1 = ea's first pass length; a signal
// then occupies (A / n_1) out_pitch values): with n_1 = 100 the 16-sample store segments of the last pass start every
// 800 bytes and straddle 128-byte lines; at a pitch of 112 they are aligned, and the de-emphasis kernel skips the pad.
void fused_fft_decim_ifft(const FftEngine& ef, const FftEngine& ea, const float2* tmp_f, float2* out, float2* tmp_a,
                          int count, const float* wr, float scale, float2* dc, hipStream_t s, int out_pitch = 0);

// The same decimation for PAIRS of real channels (FM / MFM, fm.py:66): tmp_f holds the first pass of the
// pair FFT (fused_real_pair_fft_first: x real signals, or their phases with the discriminator on the load);
// the packed pair is decimated like one complex signal and one inverse transform returns channel 2P in the
// real part, 2P+1 in the imaginary part: y [count][A] float32; dc [count] receives (sum y_c / A, 0).
void fused_real_pair_fft_first(const FftEngine& e, const float* x, float2* tmp, int count, bool from_phase,
                               hipStream_t s, PhaseRows rows = PhaseRows{});
void fused_fft_decim_ifft_pairs(const FftEngine& ef, const FftEngine& ea, const float2* tmp_f, float* y,
                                float2* tmp_a, int count, const float* wr, float scale, float2* dc, hipStream_t s);

// wbfm.py:86-87: audio decimation of both stereo legs.  U [count][B] = FFT_B of the packed signal (only
// |k| <= A/2 is read); the unpacking into the packed Hermitian spectrum of l + j r, the Hamming weight and
// the Nyquist rule of decimate.py:48 are the load of IFFT_A's first pass; out [count][A] = l + j r
// (= float32 [count][A][2]).  dc (optional, [count]) receives (sum l, sum r) / A.  e is the length-A plan.
void fused_stereo_unpack_ifft(const FftEngine& e, const float2* U, int64_t B, float2* out, float2* tmp, int count,
                              const float* wr, int nyq, int nmin, float nyq_factor, float scale, float2* dc,
                              hipStream_t s);

// Forward FFT whose last pass stores only the bins |k| <= keep (decimation to A needs no more).
void fused_fft_pruned(const FftEngine& e, const float2* in, float2* out, float2* tmp, int count, int keep,
                      hipStream_t s);

// Inverse FFT of a Hermitian spectrum Y [count][n] whose real result goes to y [count][n] floats.
void fused_ifft_real_out(const FftEngine& e, const float2* Y, float* y, float2* tmp, int count, float scale,
                         hipStream_t s);


RCFM_DECL_NS_CLOSE
}  // namespace rcfm
