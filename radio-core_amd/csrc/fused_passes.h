// FFT passes with the radio path's element-wise stages riding on their loads / stores
// (fused_passes.hip).  Each function is one complete transform of `count` signals.
#pragma once

#include "fft_engine.h"

namespace rcfm {

// Tuner.run (tuner.py:159-161) for `count` channels of bandwidth B = e.desc().n:
// bin gather from the wideband spectrum X (circular roll, fftshifted periodic window
// a0 - (1-a0) cos(2 pi i / N), truncation, Nyquist merge) is the load of the first pass of
// the inverse FFT; out = ifft(Y) * B / N.
struct TunerGather {
    const float2* X;
    int64_t N;
    const int64_t* roll;   // device, one entry per channel of the range, already in [0, N)
    double a0;
    int nyq, nneg, nyq_mode;
    // Optional fast addressing: X carries `halo` repeated bins on both sides and base32[c] =
    // (N - roll[c]) mod N, so source bin d (|d| <= halo) of channel c is X[base32[c] + d].
    const int32_t* base32 = nullptr;
    int64_t halo = 0;
    // Distance between the spectra of consecutive signals (general path only): 0 = every channel reads the one
    // wideband spectrum (the Tuner); n = one spectrum per signal (complex Decimate, decimate.py:47-48).
    int64_t x_batch = 0;
};

// Layout of a phase array written with theta_pitch: sample t of a channel sits at (t / row) pitch + t % row.
struct PhaseRows {
    int row = 0, pitch = 0;            // row = 0: contiguous
    int64_t channel_stride(int64_t n) const { return row ? (n / row) * (int64_t)pitch : n; }
};

// fused_real_pair_fft / StorePruned: keep only bins 0 .. n/2 (all the packed Hilbert chain reads)
constexpr int kKeepLowerHalf = -2;

}  // namespace rcfm

// The functions themselves, once per tile width (tile_ns.h): rcfm::fused_* (W = 16) and rcfm::narrow::fused_* (W = 8).
#define RCFM_DECL_NS_OPEN
#define RCFM_DECL_NS_CLOSE
#include "fused_passes_decl.h"
#undef RCFM_DECL_NS_OPEN
#undef RCFM_DECL_NS_CLOSE
#define RCFM_DECL_NS_OPEN namespace narrow {
#define RCFM_DECL_NS_CLOSE }
#include "fused_passes_decl.h"
#undef RCFM_DECL_NS_OPEN
#undef RCFM_DECL_NS_CLOSE
