// VALU issue-rate microbenchmark on gfx950: plain v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul/v_pk_add, 8 independent
// accumulators per lane, N waves per SIMD.  Prints cycles per wave-instruction per SIMD and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    v2f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v2f{(float)threadIdx.x + i, (float)i};
    v2f A{a, a * 0.5f}, B{b, b * 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(A.x), "v"(B.x));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(A));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(A));
                if (MODE == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(A.x));
                if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc[i]) : "v"(A), "v"(B));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int flops_per_lane_inst, int blocks_per_cu) {
    float* out;
    const int cus = 256, blocks = cus * blocks_per_cu, iters = 4000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 32 * blocks_per_cu;      // 4 waves per block = 1 per SIMD
    const double tf = (double)blocks * 256 * iters * 32 * flops_per_lane_inst / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d: %7.3f ms  %6.2f ns per wave-inst per SIMD (%.2f cycles @2.4GHz)  %7.1f TFLOP/s\n", name,
           blocks_per_cu, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4, tf);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", 2, w);
        run<1>("v_pk_fma_f32", 4, w);
        run<5>("v_pk_fma_f32 op_sel/neg", 4, w);
        run<2>("v_pk_mul_f32", 2, w);
        run<3>("v_pk_add_f32", 2, w);
        run<4>("v_mul_f32", 1, w);
    }
    return 0;
}
