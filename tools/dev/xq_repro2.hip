// xq_repro2.hip -- as xq_repro.hip, but the neighbour on the second stream is the REAL split-bf16 fused block (kernels_block.hip, layer given on the
// command line, 4 frames), launched back to back.   build: BB_MAIN=tools/dev/xq_repro2.hip BB_OUT=xq_repro2 BB_DEFS=hfnet_slam_amd/csrc/kernels_conv.hip bash tools/dev/build_block_bench.sh
//   run (GPU box): tools/dev/xq_repro2 <layer 9..14> <iterations> <burst> <bf16x3 0|1> <variant>
#include "../../hfnet_slam_amd/csrc/kernels.hpp"
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace hfnet;
namespace hfnet { void set_error(const char*, ...) {} const char* get_error() { return ""; } }
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
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
    // ... and the sampler's double-precision tail (cv::normalize: squares and tree in f64, sqrt, reciprocal), twice
    auto tail = [&](const f32x4 (&q)[4]) -> float {
        f32x4 o;
        const float i0 = 1.0f / sqrtf(fmaxf(s1[0], 1e-12f)), i1 = 1.0f / sqrtf(fmaxf(s1[1], 1e-12f)), i2 = 1.0f / sqrtf(fmaxf(s1[2], 1e-12f)), i3 = 1.0f / sqrtf(fmaxf(s1[3], 1e-12f));
        for (int j = 0; j < 4; ++j) o[j] = 0.3f * q[0][j] * i0 + 0.2f * q[1][j] * i1 + 0.4f * q[2][j] * i2 + 0.1f * q[3][j] * i3;
        double pd[4];
        for (int j = 0; j < 4; ++j) pd[j] = (double)o[j] * (double)o[j];
        const bool b5 = lane & 32, b4 = lane & 16;
        double v2[2];
        for (int kk = 0; kk < 2; ++kk) v2[kk] = (b5 ? pd[2 + kk] : pd[kk]) + __shfl_xor(b5 ? pd[kk] : pd[2 + kk], 32, 64);
        double v = (b4 ? v2[1] : v2[0]) + __shfl_xor(b4 ? v2[0] : v2[1], 16, 64);
        for (int off = 8; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
        v = v + __shfl_xor(v, 32, 64);
        v = v + __shfl_xor(v, 16, 64);
        const double nrm = sqrt(v);
        const float sc = (float)(nrm > 2.2e-16 ? 1.0 / nrm : 0.0);
        return o[0] * sc + o[1] * sc + o[2] * sc + o[3] * sc;
    };
    const float t1 = tail(sq);
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(sq[t][j]));
    const float t2 = tail(sq);
    if (__ballot(__float_as_int(t1) != __float_as_int(t2)) && lane == 0) atomicAdd(bad + 1, 1000000u);
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


static float* dev_rand(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 13, iters = argc > 2 ? atoi(argv[2]) : 2000, burst = argc > 3 ? atoi(argv[3]) : 12, bf = argc > 4 ? atoi(argv[4]) : 1, variant = argc > 5 ? atoi(argv[5]) : 4;
    const int frames = 4;
    const int st[19] = {0, 2, 1, 2, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1};
    const int co[19] = {0, 24, 16, 24, 24, 24, 48, 96, 48, 48, 48, 48, 72, 72, 72, 120, 120, 120, 240};
    BlockPack b{};
    b.cin = co[L - 1]; b.expand = b.cin * 6; b.stride = st[L]; b.cout = co[L]; b.residual = b.stride == 1 && b.cin == b.cout; b.has_expand = 1;
    b.ex.taps = 1; b.ex.cin = b.cin; b.ex.n = b.expand; b.ex.nt_total = (b.expand + 31) / 32;
    b.ex.w = dev_rand((size_t)b.cin / 8 * b.ex.nt_total * 256, 0.2f); b.ex.bias = dev_rand(b.ex.nt_total * 32, 0.2f);
    b.dw.c = b.expand; b.dw.w = dev_rand(9 * b.expand, 0.3f); b.dw.bias = dev_rand(b.expand, 0.2f);
    b.pr.taps = 1; b.pr.cin = b.expand; b.pr.n = b.cout; b.pr.nt_total = (b.cout + 31) / 32;
    b.pr.w = dev_rand((size_t)b.expand / 8 * b.pr.nt_total * 256, 0.1f); b.pr.bias = dev_rand(b.pr.nt_total * 32, 0.2f);
    b.ex16.cin = b.cin; b.ex16.n = b.expand; b.ex16.n16 = (b.expand + 15) / 16;
    b.ex16.w = dev_rand((size_t)((b.cin + 15) / 16) * b.ex16.n16 * 256, 0.2f);
    b.pr16.cin = b.expand; b.pr16.n = b.cout; b.pr16.n16 = (b.cout + 15) / 16;
    b.pr16.w = dev_rand((size_t)((b.expand + 15) / 16) * b.pr16.n16 * 256, 0.1f);
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    void *exbf, *prbf; CK(hipMalloc(&exbf, bf16x3_pack_bytes(b.ex))); CK(hipMalloc(&prbf, bf16x3_pack_bytes(b.pr)));
    CK(launch_repack_bf16x3(b.ex, exbf, sb)); CK(launch_repack_bf16x3(b.pr, prbf, sb)); CK(hipStreamSynchronize(sb));
    b.ex_bf = exbf; b.pr_bf = prbf;
    Geom g{};
    g.n_levels = 1; g.batch = frames;
    int h = 480, w = 752;
    for (int k = 1; k < L; ++k) { h = same_out(h, st[k]); w = same_out(w, st[k]); }
    LevelGeom& v = g.lv[0];
    v.H = h; v.W = w; v.Ho = same_out(h, b.stride); v.Wo = same_out(w, b.stride);
    v.pt = same_pad_before(h, 3, b.stride); v.pl = same_pad_before(w, 3, b.stride); v.in_off = 0; v.out_off = 0;
    float* X = dev_rand((size_t)frames * h * w * b.cin, 2.0f);
    float* Y; CK(hipMalloc(&Y, (size_t)frames * v.Ho * v.Wo * b.cout * 4));
    const int n = 16000;
    float *rows, *tmp; int* index; unsigned* bad;
    CK(hipMalloc(&rows, (size_t)n * 256 * 4)); CK(hipMalloc(&tmp, n * 4)); CK(hipMalloc(&index, n * 4)); CK(hipMalloc(&bad, 8));
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(rows, 0, (size_t)n * 256 * 4)); CK(hipMemset(index, 0, n * 4)); CK(hipDeviceSynchronize());
    hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    unsigned total_bad = 0, sums_bad = 0; int bad_iters = 0;
    for (int it = 1; it <= iters; ++it) {
        CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0));
        hipLaunchKernelGGL(k_produce, dim3(n / 4), dim3(256), 0, sa, rows, index, n, it);
        for (int q = 0; q < burst; ++q) {
            CK(launch_block_fused(X, b, Y, g, variant, sb, bf));
            if (q == burst / 2) hipLaunchKernelGGL(k_filler, dim3((n + 255) / 256), dim3(256), 0, sa, rows, tmp, n);
        }
        hipLaunchKernelGGL(k_consume, dim3(n / 4), dim3(256), 0, sa, rows, index, n, it, bad);
        hipLaunchKernelGGL(k_sums, dim3(n / 4), dim3(256), 0, sa, rows, n, bad);
        CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0));
        unsigned hb[2] = {0, 0};
        CK(hipMemcpyAsync(hb, bad, 8, hipMemcpyDeviceToHost, sa)); CK(hipStreamSynchronize(sa));
        if (hb[0] != total_bad) { ++bad_iters; total_bad = hb[0]; }
        sums_bad = hb[1];
    }
    printf("layer %d bf16x3 %d variant %d, %d iterations, burst %d: %d iterations with stale reads (%u wave-rows); wave sums that differ when repeated: %u\n", L, bf, variant, iters, burst, bad_iters, total_bad, sums_bad);
    return 0;
}
