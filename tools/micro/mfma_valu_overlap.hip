// micro-benchmark: do an MFMA-only wave and a VALU-only wave on the same SIMD overlap?
// 512-thread workgroups, one per CU: waves 0-3 issue a dependent MFMA chain, waves 4-7 an fma loop / LDS loop / global loads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, const float* in, int n_mfma, int n_valu, int mode) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    float res = 0.f;
    if (wave < 4) {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float a = threadIdx.x, b = 2.f;
        for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) res += acc[i];
    } else {
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
        if (mode == 0) {
            for (int it = 0; it < n_valu; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f); }
            }
        } else if (mode == 1) {
            for (int it = 0; it < n_valu; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { lds[(threadIdx.x + u * 64) & 8191] = x0; x0 += lds[(threadIdx.x * 4 + u) & 8191]; }
            }
        } else {
            for (int it = 0; it < n_valu; ++it) { x0 += in[((size_t)it * 512 + threadIdx.x) & 0xfffff]; }
        }
        res = x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}
static float run(float* d, float* in, int nm, int nv, int mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512>>>(d, in, nm, nv, mode);
    hipEventRecord(e0);
    k<<<256, 512>>>(d, in, nm, nv, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f;
}
int main() {
    float *d, *in; hipMalloc(&d, 256 * 512 * 4); hipMalloc(&in, (1 << 20) * 4 + 4096); hipMemset(in, 0, (1 << 20) * 4);
    const int nm = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        const int nv = mode == 0 ? 4000 : mode == 1 ? 4000 : 300;
        printf("mode %d (0 VALU fma, 1 LDS, 2 global loads): MFMA only %.0f us, other only %.0f us, both %.0f us\n", mode, run(d, in, nm, 0, mode),
               run(d, in, 0, nv, mode), run(d, in, nm, nv, mode));
    }
    return 0;
}
