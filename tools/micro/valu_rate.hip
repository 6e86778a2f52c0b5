// micro-benchmark: issue cost of v_fma_f32 / v_pk_fma_f32 / v_med3_f32 (wave64), 1 and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float x[16]; for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    f32x2 y[8]; for (int i = 0; i < 8; ++i) y[i] = f32x2{(float)threadIdx.x, (float)i};
    const f32x2 m2 = {1.0001f, 0.9999f}, a2 = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = __builtin_elementwise_fma(y[i], m2, a2);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_fmed3f(x[i], 0.0f, 6.0f + u);
            }
        }
    }
    float s = 0.f; for (int i = 0; i < 16; ++i) s += x[i]; for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> static void run(float* d, int wgs, const char* name, int per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    k<MODE><<<256 * wgs, 256>>>(d, iters);
    hipEventRecord(e0);
    k<MODE><<<256 * wgs, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * per_iter * wgs;        // per SIMD
    printf("%-14s %d wave/SIMD: %.0f us, %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, wgs, ms * 1e3, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    for (int w = 1; w <= 2; ++w) { run<0>(d, w, "v_fma_f32", 64); run<1>(d, w, "v_pk_fma_f32", 32); run<2>(d, w, "v_med3_f32", 64); }
    return 0;
}
