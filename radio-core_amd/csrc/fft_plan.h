// Batched 1-D single-precision FFT plans.  The north star assigns the
// channelizer FFT to rocFFT; this wrapper is the only place that talks to it.
#pragma once

#include <cstddef>

#include "common.h"

struct rocfft_plan_t;
struct rocfft_execution_info_t;

namespace rcfm {

enum class FftKind { C2C_FORWARD, C2C_INVERSE, R2C, C2R };

class FftPlan {
   public:
    // n = transform length (real length for R2C / C2R), batch = number of
    // contiguous transforms.  Batch distances are the natural ones: n complex
    // for C2C, n real / (n/2+1) complex for the real transforms.
    FftPlan(FftKind kind, size_t n, size_t batch, bool in_place);
    ~FftPlan();
    FftPlan(const FftPlan&) = delete;
    FftPlan& operator=(const FftPlan&) = delete;

    size_t work_bytes() const { return work_bytes_; }
    // Unnormalised transform.  `work` must hold work_bytes() bytes.
    void exec(void* in, void* out, void* work, hipStream_t stream);

   private:
    rocfft_plan_t* plan_ = nullptr;
    rocfft_execution_info_t* info_ = nullptr;
    size_t work_bytes_ = 0;
    bool in_place_ = false;
};

}  // namespace rcfm
