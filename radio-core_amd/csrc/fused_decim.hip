// The run-time-L2 decimating tile kernels (fft_decim_rt.h), one per long tile length.
#include "fft_decim_rt.h"

namespace rcfm {
RCFM_NS_OPEN

bool launch_fft_tile2_decim_rt(const FftPassDev& d1, const FftPassDev& d2, int batch, const float2* tmp_f,
                               const WinAudioDecim& win, float2* tmp_a, hipStream_t s) {
    using namespace fftk;
    if (!fft_tile2_decim_rt_applies(d1, d2, batch)) return false;
    const dim3 grid((unsigned)((d1.p.n_inner + W - 1) / W), 1, (unsigned)batch);
    LoadPlainT<false> ld{tmp_f};
    StorePlainT<false> st{tmp_a, 1.0f};
    switch (d1.p.L) {
#define RCFM_CASE(LEN, A, B, C, D)                                                                               \
    case LEN:                                                                                                    \
        hipLaunchKernelGGL((k_fft_tile2_decim_rt<LEN, A, B, C, D, tile_threads(LEN)>), grid,                     \
                           dim3(tile_threads(LEN)), 0, s, d1, d2, ld, win, st);                                  \
        break;
        RCFM_FFT_DECIM_RT_LENGTHS(RCFM_CASE)
#undef RCFM_CASE
        default: return false;
    }
    RC_HIP(hipGetLastError());
    return true;
}

RCFM_NS_CLOSE
}  // namespace rcfm
