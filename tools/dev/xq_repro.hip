// xq_repro.hip -- stand-alone attempt at NOTEBOOK.md R4.8, without the engine: stream A runs producer -> filler -> consumer every iteration (the
// consumer checks that the rows it reads carry THIS iteration's stamp; it read the same rows in the previous iteration, from the same
// workgroups), stream B runs a burst of short register-heavy one-wave-workgroup kernels (bf16 MFMA + a little LDS, as k_block_fused8's split-bf16
// forms) beside it.   build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dev/xq_repro.hip -o tools/dev/xq_repro
//   run (GPU box): tools/dev/xq_repro [iterations] [burst kernels] [consumer delay kernels]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_produce(float* rows, int* index, int n, int it) {       // 4 rows per workgroup (a wave each), 256 floats per row
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    const float stamp = (float)(it * 4096 + (r & 4095));
    *(f32x4*)(rows + (size_t)r * 256 + lane * 4) = f32x4{stamp, stamp, stamp, stamp};
    if (lane == 0) index[r] = (r * 7 + it) % n;                                                  // (an index table rewritten every iteration, like cell_row)
}
__global__ __launch_bounds__(256) void k_filler(const float* rows, float* out, int n) {          // something between the two, reading other memory
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = rows[(size_t)i * 256] * 0.5f;
}
__global__ __launch_bounds__(256) void k_consume(const float* rows, const int* index, int n, int it, unsigned* bad) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;                  // a wave per "keypoint": four rows through the index table
    if (k >= n) return;
    for (int t = 0; t < 4; ++t) {
        const int slot = (k + t * 97) % n, r = index[slot];
        const int want = (slot * 7 + it) % n;
        bool wrong = r != want;
        if (r >= 0 && r < n) {
            const f32x4 v = *(const f32x4*)(rows + (size_t)r * 256 + lane * 4);
            const float stamp = (float)(it * 4096 + (r & 4095));
            wrong = wrong || v[0] != stamp || v[3] != stamp;
        }
        if (__ballot(wrong) && lane == 0) atomicAdd(bad, 1u);
    }
}
// the sampler's wave sums (kernels_detect.hip tree256_wave_x4: ds_bpermute shuffles + v_readlane), evaluated twice on the same registers
__device__ __forceinline__ void tree_x4(const f32x4 (&p)[4], int lane, float (&out)[4]) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    float v8[8];
    for (int k = 0; k < 8; ++k) { const float lo = p[k >> 2][k & 3], hi = p[2 + (k >> 2)][k & 3]; v8[k] = (b5 ? hi : lo) + __shfl_xor(b5 ? lo : hi, 32, 64); }
    float v4[4];
    for (int k = 0; k < 4; ++k) v4[k] = (b4 ? v8[4 + k] : v8[k]) + __shfl_xor(b4 ? v8[k] : v8[4 + k], 16, 64);
    float v2[2];
    for (int k = 0; k < 2; ++k) v2[k] = (b3 ? v4[2 + k] : v4[k]) + __shfl_xor(b3 ? v4[k] : v4[2 + k], 8, 64);
    float v = (b2 ? v2[1] : v2[0]) + __shfl_xor(b2 ? v2[0] : v2[1], 4, 64);
    v = v + __shfl_xor(v, 2, 64); v = v + __shfl_xor(v, 1, 64); v = v + __shfl_xor(v, 8, 64); v = v + __shfl_xor(v, 4, 64);
    for (int t = 0; t < 4; ++t) out[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * t));
}
__global__ __launch_bounds__(256) void k_sums(const float* rows, int n, unsigned* bad) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n) return;
    f32x4 sq[4];
    for (int t = 0; t < 4; ++t) { const f32x4 v = *(const f32x4*)(rows + (size_t)((k + t * 131) % n) * 256 + lane * 4); for (int j = 0; j < 4; ++j) sq[t][j] = v[j] * 1e-3f + (float)lane; }
    float s1[4], s2[4];
    tree_x4(sq, lane, s1);
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(sq[t][j]));
    tree_x4(sq, lane, s2);
    bool wrong = false;
    for (int t = 0; t < 4; ++t) wrong = wrong || __float_as_int(s1[t]) != __float_as_int(s2[t]);
    // (and against the value every lane can compute alone: the four sums are equal by construction only for equal rows -- so just the repeat test)
    if (__ballot(wrong) && lane == 0) atomicAdd(bad + 1, 1u);
}
// the neighbour: one-wave workgroups, ~200 accumulator registers, bf16 MFMA back to back, a 4.6 KB LDS tile written and read per step
__global__ __launch_bounds__(64, 2) void k_busy(const float* src, float* dst, int steps) {
    __shared__ float tile[32 * 36];
    const int lane = threadIdx.x;
    f32x16 acc[12];
    for (int m = 0; m < 12; ++m) for (int i = 0; i < 16; ++i) acc[m][i] = (float)(m + i);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(src[(blockIdx.x * 64 + lane) * 8 + e]); b[e] = (__bf16)(0.001f * (float)(e + 1)); }
    for (int s = 0; s < steps; ++s) {
        for (int m = 0; m < 12; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
        tile[(s & 31) * 36 + (lane & 31)] = acc[s % 12][0];
        asm volatile("" ::: "memory");
        a[0] = (__bf16)tile[((s + 1) & 31) * 36 + (lane & 31)];
    }
    float sum = 0.f;
    for (int m = 0; m < 12; ++m) for (int i = 0; i < 16; ++i) sum += acc[m][i];
    dst[blockIdx.x * 64 + lane] = sum;
}
__global__ void k_delay(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, burst = argc > 2 ? atoi(argv[2]) : 20, delay_us = argc > 3 ? atoi(argv[3]) : 0;
    const int n = 16000;
    float *rows, *tmp, *bsrc, *bdst; int* index; unsigned* bad;
    CK(hipMalloc(&rows, (size_t)n * 256 * 4)); CK(hipMalloc(&tmp, n * 4)); CK(hipMalloc(&index, n * 4)); CK(hipMalloc(&bad, 8));
    CK(hipMalloc(&bsrc, 4096 * 64 * 8 * 4)); CK(hipMalloc(&bdst, 4096 * 64 * 4));
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(bsrc, 0, 4096 * 64 * 8 * 4)); CK(hipMemset(rows, 0, (size_t)n * 256 * 4)); CK(hipMemset(index, 0, n * 4));
    CK(hipDeviceSynchronize());
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    unsigned total_bad = 0, sums_bad = 0; int bad_iters = 0;
    for (int it = 1; it <= iters; ++it) {
        CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0));
        hipLaunchKernelGGL(k_produce, dim3(n / 4), dim3(256), 0, sa, rows, index, n, it);
        for (int b = 0; b < burst; ++b) {
            hipLaunchKernelGGL(k_busy, dim3(48 + 48 * (b % 3)), dim3(64), 0, sb, bsrc, bdst, 40 + 10 * (b % 5));
            if (b == burst / 2) hipLaunchKernelGGL(k_filler, dim3((n + 255) / 256), dim3(256), 0, sa, rows, tmp, n);
        }
        if (delay_us) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, sa, (long long)delay_us * 100);
        hipLaunchKernelGGL(k_consume, dim3(n / 4), dim3(256), 0, sa, rows, index, n, it, bad);
        hipLaunchKernelGGL(k_sums, dim3(n / 4), dim3(256), 0, sa, rows, n, bad);
        CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0));
        unsigned hb[2] = {0, 0};
        CK(hipMemcpyAsync(hb, bad, 8, hipMemcpyDeviceToHost, sa)); CK(hipStreamSynchronize(sa));
        if (hb[0] != total_bad) { ++bad_iters; total_bad = hb[0]; }
        sums_bad = hb[1];
    }
    printf("iterations %d, burst %d kernels, consumer delay %d us: %d iterations with stale reads, %u wave-rows in all; wave sums that differ when repeated: %u\n", iters, burst, delay_us, bad_iters, total_bad, sums_bad);
    return 0;
}
