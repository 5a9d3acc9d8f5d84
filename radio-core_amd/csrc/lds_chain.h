// Narrow channels entirely on chip (lds_chain.hip): Tuner.run + FM.run of a PAIR of channels in one workgroup.
//
// Reference path (per channel): radiocore/tools/tuner.py:151-161 (roll, window, truncation, IFFT_B), then
// radiocore/analog/fm.py:60-67 (angle, unwrap, diff, /pi, Decimate B -> A = scipy.signal.resample,
// decimate.py:47-48).  For B <= ~13 000 complex samples the whole chain of two channels fits the 160 KiB LDS
// of one CU: the channel's B bins are gathered from the wideband spectrum, every transform runs in LDS, and only
// the audio goes back to memory -- 8B bytes read and 4A written per channel instead of the ~40B the multi-pass
// launches move.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace rcfm {

struct LdsChainArgs {
    // tuner side (fused_passes.h, TunerGather's fast form): haloed wideband spectrum, per-channel base bin
    const float2* X;
    const int32_t* base;          // [count], already offset to the first channel of the range
    int64_t N;
    int nyq, merge;               // B/2 + 1;  NYQ_DOWN: B/2 (the bin that also receives X[-B/2]), else -1
    // demodulator side: folded Hamming weights of Decimate(B -> A) (A/2 + 1 entries) and 1/B
    const float* wr;
    float scale;
    float* audio;                 // [count][A] float32 (FM: the result; MFM without deemph_taps: the de-emphasis kernel's input)
    float2* dc;                   // [count] or null: (mean of the channel's audio, 0) from the DC bin
    int count;
    // MFM entirely on chip (mfm.py:62-66): the 51 de-emphasis taps (device; a one-pole response, deemphasis.py:37-46 --
    // the caller checks) and the carried FIR state [count][50] (deemphasis.py:48-49,64: read, then replaced): `audio` then
    // receives lfilter(taps, v, zi) - mean, clipped to +-0.999.  Null: the kernel stops at the decimated signal v.
    const float* deemph_taps = nullptr;   // 102 floats: b[0..50], then sfx[i] = sum_{j > i} b[j], i = 0..50
    float* deemph_state = nullptr;
};

// Is there an instantiation for B -> A?  (lengths listed in lds_chain.hip)
bool lds_chain_supported(int B, int A);
// Does the instantiation for B -> A also exist with MFM's de-emphasis inside (LdsChainArgs::deemph_taps)?
bool lds_chain_deemph_supported(int B, int A);
// Launches ceil(count / 2) workgroups; returns false (nothing launched) when (B, A) has no instantiation.
bool launch_lds_chain(int B, int A, const LdsChainArgs& args, hipStream_t stream);

}  // namespace rcfm
