// The 4-line build of the tile kernels (compiled with -DRCFM_TILE_W=4, namespace rcfm::quad; tile_ns.h): the two
// passes of a plan whose pass lengths are ~3200 points (RCFM_FFT_QUAD_LENGTHS).  Only transforms that stay in the
// Infinity Cache are planned this way (fft_engine.hip): their 32-byte row segments hit the cache, and one pass of the three
// a 16-line plan needs is gone.  Reference: the Tuner benchmark of tests/benchmark.py:105 is N = 1e7 = 3125 x 3200.
#include "fft_engine.h"
#include "fft_kernel.h"

namespace rcfm {

void quad_pass(const FftPassDev& dev, int batch, const float2* src, float2* dst, bool swap_in, bool swap_out, float scale,
               const FftRowWindow* keep, int64_t n, hipStream_t stream) {
    using namespace quad::fftk;
    static_assert(W == 4, "fft_quad.hip is the 4-line build");
    if (!dev.p.load_along_l) {   // first pass: strided columns
        StorePlainT<false> st{dst, scale};
        if (swap_in) launch_fft_pass<kStridedOnly>(dev, batch, LoadPlainT<true>{src}, st, stream);
        else launch_fft_pass<kStridedOnly>(dev, batch, LoadPlainT<false>{src}, st, stream);
        return;
    }
    LoadPlainT<false> ld{src};
    if (swap_out)
        launch_fft_pass<kRowsOnly>(dev, batch, ld, StorePlainT<true>{dst, scale}, stream);
    else if (keep != nullptr)
        launch_fft_pass<kRowsOnly>(dev, batch, ld,
                                   StoreRowWindow{dst, scale, (int)dev.p.n_o1, (int)dev.p.n_o2, keep->lo, keep->hi, n,
                                                  batch == 1 ? keep->halo : 0},
                                   stream);
    else
        launch_fft_pass<kRowsOnly>(dev, batch, ld, StorePlainT<false>{dst, scale}, stream);
}

}  // namespace rcfm
