// Device helpers shared by the stencil kernels and the fused FFT passes.
#pragma once

#include <hip/hip_runtime.h>

namespace rcfm {

// atan2(y, x) / pi: odd minimax polynomial of min/max (degree 17, 1e-7 rad), octant folding by
// selects; (0, 0) -> 0 like numpy.angle.
__device__ __forceinline__ float atan2_over_pi(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1e-37f), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(mx);
    const float t = a * a;
    // the polynomial's coefficients carry the 1 / pi (max error 3.7e-8 in units of pi against 4.2e-8 with a final multiply)
    float p = 0.0007893929141573608f;
    p = fmaf(p, t, -0.0046154772862792015f);
    p = fmaf(p, t, 0.012717602774500847f);
    p = fmaf(p, t, -0.023080114275217056f);
    p = fmaf(p, t, 0.03344602882862091f);
    p = fmaf(p, t, -0.045084625482559204f);
    p = fmaf(p, t, 0.06361839175224304f);
    p = fmaf(p, t, -0.10610104352235794f);
    p = fmaf(p, t, 0.31830984354019165f);
    float r = p * a;                                // [0, 1/4]
    r = (ay > ax) ? 0.5f - r : r;                  // [0, 1/2]
    r = (x < 0.f) ? 1.f - r : r;                   // [0, 1]
    return copysignf(r, y);
}

// fm.py:60-65 from stored phases (units of pi): diff(unwrap(angle(x))) / pi = the phase step wrapped
// into [-1, 1].
__device__ __forceinline__ float phase_step_wrapped(float th, float th_prev) {
    const float d = th - th_prev;          // (-2, 2)
    return d - 2.f * rintf(0.5f * d);
}

}  // namespace rcfm
