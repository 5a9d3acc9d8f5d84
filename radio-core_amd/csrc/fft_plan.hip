#include "fft_plan.h"

#include <rocfft/rocfft.h>

#include <mutex>

namespace rcfm {

namespace {

const char* status_name(rocfft_status s) {
    switch (s) {
        case rocfft_status_success: return "success";
        case rocfft_status_failure: return "failure";
        case rocfft_status_invalid_arg_value: return "invalid_arg_value";
        case rocfft_status_invalid_dimensions: return "invalid_dimensions";
        case rocfft_status_invalid_array_type: return "invalid_array_type";
        case rocfft_status_invalid_strides: return "invalid_strides";
        case rocfft_status_invalid_distance: return "invalid_distance";
        case rocfft_status_invalid_offset: return "invalid_offset";
        case rocfft_status_invalid_work_buffer: return "invalid_work_buffer";
    }
    return "?";
}

#define RC_FFT(expr)                                                                        \
    do {                                                                                    \
        rocfft_status rc_s_ = (expr);                                                       \
        if (rc_s_ != rocfft_status_success)                                                 \
            throw Error{RCFM_ERR_RUNTIME, std::string(#expr) + ": rocfft " + status_name(rc_s_)}; \
    } while (0)

void ensure_setup() {
    static std::once_flag once;
    std::call_once(once, [] { (void)rocfft_setup(); });
}

}  // namespace

FftPlan::FftPlan(FftKind kind, size_t n, size_t batch, bool in_place) : in_place_(in_place) {
    ensure_setup();
    rocfft_transform_type type = rocfft_transform_type_complex_forward;
    switch (kind) {
        case FftKind::C2C_FORWARD: type = rocfft_transform_type_complex_forward; break;
        case FftKind::C2C_INVERSE: type = rocfft_transform_type_complex_inverse; break;
        case FftKind::R2C: type = rocfft_transform_type_real_forward; break;
        case FftKind::C2R: type = rocfft_transform_type_real_inverse; break;
    }
    const size_t lengths[1] = {n};
    RC_FFT(rocfft_plan_create(&plan_, in_place ? rocfft_placement_inplace : rocfft_placement_notinplace,
                              type, rocfft_precision_single, 1, lengths, batch, nullptr));
    RC_FFT(rocfft_plan_get_work_buffer_size(plan_, &work_bytes_));
    RC_FFT(rocfft_execution_info_create(&info_));
}

FftPlan::~FftPlan() {
    if (info_) (void)rocfft_execution_info_destroy(info_);
    if (plan_) (void)rocfft_plan_destroy(plan_);
}

void FftPlan::exec(void* in, void* out, void* work, hipStream_t stream) {
    RC_FFT(rocfft_execution_info_set_stream(info_, stream));
    if (work_bytes_) {
        RC_REQUIRE(work != nullptr, RCFM_ERR_RUNTIME, "FFT work buffer missing");
        RC_FFT(rocfft_execution_info_set_work_buffer(info_, work, work_bytes_));
    }
    void* ins[1] = {in};
    void* outs[1] = {out};
    RC_REQUIRE(in_place_ == (out == in), RCFM_ERR_RUNTIME, "FFT placement does not match its plan");
    RC_FFT(rocfft_execute(plan_, ins, in_place_ ? nullptr : outs, info_));
}

}  // namespace rcfm
