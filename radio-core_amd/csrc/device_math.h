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
    float p = 0.002479950897395611f;
    p = fmaf(p, t, -0.014499950222671032f);
    p = fmaf(p, t, 0.039953526109457016f);
    p = fmaf(p, t, -0.0725083202123642f);
    p = fmaf(p, t, 0.10507379472255707f);
    p = fmaf(p, t, -0.14163753390312195f);
    p = fmaf(p, t, 0.19986307621002197f);
    p = fmaf(p, t, -0.3333262503147125f);
    p = fmaf(p, t, 0.9999998807907104f);
    float r = (p * a) * 0.31830988618379067154f;   // [0, 1/4]
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
