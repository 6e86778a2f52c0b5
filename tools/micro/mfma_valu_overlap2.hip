// micro-benchmark 2: (a) MFMA + independent VALU in the SAME wave, (b) MFMA wave with NACC independent accumulators next to a VALU wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int VPM>
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int same_wave) {
    const int wave = threadIdx.x >> 6;
    float res = 0.f;
    if (wave < 4) {
        f32x16 acc[NACC];
        for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
        float a = threadIdx.x, b = 2.f;
        float x[8]; for (int i = 0; i < 8; ++i) x[i] = i + threadIdx.x;
        for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int n = 0; n < NACC; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
                    if (same_wave) {
#pragma unroll
                        for (int v = 0; v < VPM; ++v) x[v & 7] = fmaf(x[v & 7], 1.0001f, 0.5f);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    }
                }
            }
        }
        for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) res += acc[n][i];
        for (int i = 0; i < 8; ++i) res += x[i];
    } else {
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
        for (int it = 0; it < n_valu; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f); }
        }
        res = x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int NACC, int VPM> static float run(float* d, int nm, int nv, int same) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, VPM><<<256, 512>>>(d, nm, nv, same);
    hipEventRecord(e0);
    k<NACC, VPM><<<256, 512>>>(d, nm, nv, same);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    printf("NACC 1: MFMA only %.0f, +VALU wave %.0f (VALU alone %.0f)\n", run<1, 4>(d, 2000, 0, 0), run<1, 4>(d, 2000, 4000, 0), run<1, 4>(d, 0, 4000, 0));
    printf("NACC 2: MFMA only %.0f, +VALU wave %.0f\n", run<2, 4>(d, 1000, 0, 0), run<2, 4>(d, 1000, 4000, 0));
    printf("NACC 4: MFMA only %.0f, +VALU wave %.0f\n", run<4, 4>(d, 500, 0, 0), run<4, 4>(d, 500, 4000, 0));
    printf("same wave, NACC 1: MFMA only %.0f, MFMA + 4 VALU each %.0f, + 8 VALU each %.0f, + 12 VALU each %.0f\n", run<1, 4>(d, 2000, 0, 0),
           run<1, 4>(d, 2000, 0, 1), run<1, 8>(d, 2000, 0, 1), run<1, 12>(d, 2000, 0, 1));
    printf("same wave, NACC 2: MFMA only %.0f, + 4 VALU each %.0f, + 8 VALU each %.0f\n", run<2, 4>(d, 1000, 0, 0), run<2, 4>(d, 1000, 0, 1), run<2, 8>(d, 1000, 0, 1));
    return 0;
}
