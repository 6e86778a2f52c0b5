// micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 in one dependent chain vs 2/3/4 interleaved accumulators,
// 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wgs_per_cu, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000 / NACC;
    const int grid = 256 * wgs_per_cu;
    k<NACC><<<grid, 256>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * 8 * NACC;
    const double waves_per_simd = wgs_per_cu;      // 4 waves per WG, 4 SIMDs
    const double cyc = ms * 1e-3 * 2.4e9 / (mfma_per_wave * waves_per_simd);
    printf("NACC %d, %d WG/CU: %.3f ms, %.1f cycles per MFMA per SIMD (64 = peak), %.1f TF\n", NACC, wgs_per_cu, ms, cyc,
           mfma_per_wave * 4 * grid * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 2; ++w) { run<1>(w, d); run<2>(w, d); run<3>(w, d); run<4>(w, d); }
    return 0;
}
