// kernels_global.hip -- NetVLAD global-descriptor head (hfnet/models/utils/layers.py:57-109).
// The memberships 1x1 conv runs on the MFMA pointwise kernel; everything after it is here.
#include "kernels.hpp"

#include <algorithm>
#include <type_traits>
#include <utility>

namespace hfnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float hf_expf_g(float x) {   // == oracle hfo_expf
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = fmaf(p, r2, r) + 1.0f;
    return ldexpf(y, (int)n);
}

// softmax over the n (<= 64) leading entries of every row, left-to-right sum (layers.py:75).  The row lives in registers:
// one round trip to memory (three passes over global memory made this 20 us for the 360 rows of a single frame).
__global__ __launch_bounds__(256) void k_softmax_rows(float* __restrict__ x, long long rows, int n, int ld) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float* r = x + row * ld;
    float v[64];
    if ((n & 3) == 0 && (ld & 3) == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q * 4 < n) {
                const f32x4 t = *(const f32x4*)(r + q * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[q * 4 + j] = t[j];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 64; ++k) if (k < n) v[k] = r[k];
    }
    float mx = v[0];
#pragma unroll
    for (int k = 1; k < 64; ++k) if (k < n) mx = fmaxf(mx, v[k]);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 64; ++k) if (k < n) { v[k] = hf_expf_g(v[k] - mx); sum = sum + v[k]; }
    if ((n & 3) == 0 && (ld & 3) == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q * 4 < n) {
                f32x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = v[q * 4 + j] / sum;
                *(f32x4*)(r + q * 4) = t;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 64; ++k) if (k < n) r[k] = v[k] / sum;
    }
}

hipError_t launch_softmax_rows(float* x, long long rows, int n, int ld, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (n < 1 || n > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, x, rows, n, ld);
    return hipGetLastError();
}

// desc[k][d] = sum_p (c[k][d] - f[p][d]) * m[p][k] (layers.py:82-87).  Canonical order (oracle/hfnet_oracle.c global_head): the
// pixels are split into VLAD_PARTS contiguous ranges of ceil(P / VLAD_PARTS), every range is one chain in pixel order from 0
// (r = c - f; t = r * m; acc = acc + t), the partial sums are added as a balanced binary tree (k_vlad_norm).  Round 3's single
// chain over all 360 pixels per output -- 7680 outputs per frame, each thread pulling its own f and m values through L1 -- took
// 106 us per 128 frames.  Here a thread owns one channel d of one range and SIXTEEN clusters at a time: f[p][d] is loaded once
// per 16 clusters (coalesced), the memberships m[p][k0 .. k0 + 15] are wave-uniform scalar loads.
// feat is in the device channel layout; thread handles physical slot pd == logical channel d.
#define VLAD_PARTS 8
template <int KC>                                             // clusters per pass (K is a multiple of KC)
__global__ __launch_bounds__(256) void k_vlad_aggregate(const float* __restrict__ feat, const float* __restrict__ memb,
                                                        const float* __restrict__ clusters, float* __restrict__ out /* [frames][VLAD_PARTS][K * D] */,
                                                        int P, int D, int K) {
    typedef float mvec_t __attribute__((ext_vector_type(KC)));
    const int frame = blockIdx.z, part = blockIdx.y;
    const int pd = blockIdx.x * 256 + threadIdx.x;
    const int pp = (P + VLAD_PARTS - 1) / VLAD_PARTS;
    const int p0 = min(part * pp, P), p1 = min(p0 + pp, P);
    const bool live = pd < D;
    const int pdc = live ? pd : D - 1;
    const int rr = pdc & 7;
    const int d = (pdc & ~7) | (rr < 4 ? 2 * rr : 2 * (rr - 4) + 1);
    const float* __restrict__ f = feat + (long long)frame * P * D + pdc;
    const float* __restrict__ m = memb + (long long)frame * P * K;         // uniform: the memberships come through the scalar cache
    float* __restrict__ o = out + ((long long)frame * VLAD_PARTS + part) * K * D;
    for (int k0 = 0; k0 < K; k0 += KC) {
        float c[KC], acc[KC];
#pragma unroll
        for (int j = 0; j < KC; ++j) { c[j] = clusters[(k0 + j) * D + d]; acc[j] = 0.0f; }
        int p = p0;
        for (; p + 4 <= p1; p += 4) {                              // four pixels of loads in flight
            float fv[4];
            mvec_t mv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { fv[u] = f[(long long)(p + u) * D]; mv[u] = *(const mvec_t*)(m + (long long)(p + u) * K + k0); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < KC; ++j) { const float r = c[j] - fv[u]; const float tt = r * mv[u][j]; acc[j] = acc[j] + tt; }
        }
        for (; p < p1; ++p) {
            const float fv = f[(long long)p * D];
            const mvec_t mv = *(const mvec_t*)(m + (long long)p * K + k0);
#pragma unroll
            for (int j = 0; j < KC; ++j) { const float r = c[j] - fv; const float tt = r * mv[j]; acc[j] = acc[j] + tt; }
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < KC; ++j) o[(k0 + j) * D + d] = acc[j];
        }
    }
}

// block-wide tree256 sum of squares of v[0..n): partial tid accumulates elements tid + 256 j
// (workgroups may be larger than 256 threads: the 256 partials and their tree are the definition, the extra threads only wait)
__device__ __forceinline__ float block_sumsq_tree256(const float* v, int n, float* red) {
    if (threadIdx.x < 256) {
        float p = 0.0f;
        for (int i = threadIdx.x; i < n; i += 256) p = fmaf(v[i], v[i], p);
        red[threadIdx.x] = p;
    }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + off];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// intra-normalisation over clusters (layers.py:89), flatten (K-major), L2, [tap], L2 (layers.py:92,97)
__global__ __launch_bounds__(1024) void k_vlad_norm(const float* __restrict__ raw, float* __restrict__ vlad_tap, float* __restrict__ out,
                                                    int D, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* v = smem;            // K*D
    float* red = smem + K * D;  // 256
    const int frame = blockIdx.x, N = K * D, T = blockDim.x;
    const float* src = raw + (long long)frame * VLAD_PARTS * N;   // the VLAD_PARTS partial sums of every output: balanced binary tree
    for (int i = threadIdx.x; i < N; i += T) {
        float pt[VLAD_PARTS];
#pragma unroll
        for (int w = 0; w < VLAD_PARTS; ++w) pt[w] = src[(long long)w * N + i];
#pragma unroll
        for (int n = VLAD_PARTS; n > 1; n >>= 1)
#pragma unroll
            for (int w = 0; w < n / 2; ++w) pt[w] = pt[2 * w] + pt[2 * w + 1];
        v[i] = pt[0];
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += T) {
        float ss = 0.0f;
        for (int k = 0; k < K; ++k) ss = fmaf(v[k * D + d], v[k * D + d], ss);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        for (int k = 0; k < K; ++k) v[k * D + d] = v[k * D + d] * inv;
    }
    __syncthreads();
    float ss = block_sumsq_tree256(v, N, red);
    float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int i = threadIdx.x; i < N; i += T) v[i] = v[i] * inv;
    __syncthreads();
    if (vlad_tap) for (int i = threadIdx.x; i < N; i += T) vlad_tap[(long long)frame * N + i] = v[i];
    ss = block_sumsq_tree256(v, N, red);
    inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    // handed to the FC kernel in its slot order within every group of 16 inputs (FcPack, common.hpp)
    for (int i = threadIdx.x; i < N; i += T) {
        const int rr = i & 15;
        out[(long long)frame * N + ((i & ~15) | ((rr & 3) << 2) | (rr >> 2))] = v[i] * inv;
    }
}

int vlad_scratch_parts() { return VLAD_PARTS; }
hipError_t launch_vlad_aggregate(const float* feat, const float* memb, const float* clusters, float* scratch, int frames, int P, int D, int K, hipStream_t s) {
    if (frames <= 0) return hipSuccess;
    const dim3 grid((D + 255) / 256, VLAD_PARTS, frames);
    if (K % 16 == 0) hipLaunchKernelGGL((k_vlad_aggregate<16>), grid, dim3(256), 0, s, feat, memb, clusters, scratch, P, D, K);
    else if (K % 8 == 0) hipLaunchKernelGGL((k_vlad_aggregate<8>), grid, dim3(256), 0, s, feat, memb, clusters, scratch, P, D, K);
    else if (K % 4 == 0) hipLaunchKernelGGL((k_vlad_aggregate<4>), grid, dim3(256), 0, s, feat, memb, clusters, scratch, P, D, K);
    else hipLaunchKernelGGL((k_vlad_aggregate<1>), grid, dim3(256), 0, s, feat, memb, clusters, scratch, P, D, K);
    return hipGetLastError();
}
hipError_t launch_vlad_norm(const float* scratch, float* vlad_tap, float* out, int frames, int D, int K, hipStream_t s) {
    if (frames <= 0) return hipSuccess;
    const size_t lds = (size_t)(K * D + 256) * sizeof(float);
    hipLaunchKernelGGL(k_vlad_norm, dim3(frames), dim3(1024), lds, s, scratch, vlad_tap, out, D, K);
    return hipGetLastError();
}

// y[f][j] = b[j] + sum_i x[f][i] * W[i][j]: slim.fully_connected (layers.py:99-107) as ONE GEMM over the frames of the
// batch on v_mfma_f32_16x16x4_f32, split FC_PARTS ways along the inputs: a wave owns 16 frames x 16 outputs x one input
// range and walks its k-ascending fma chain from 0 (n_in / 64 dependent MFMAs); k_fc_combine_l2 adds the partial sums as
// a balanced binary tree, then the bias -- the oracle's order (oracle/hfnet_oracle.c global_head), chosen for this kernel:
// the single chain of round 2 was 1920 dependent MFMAs per output tile (33 us however many frames) and left one 480 KB
// weight panel per workgroup in flight (256 workgroups: ~2 TB/s); 4096 workgroups of 30 KB each stream the 126 MB at the
// rate HBM delivers them, for one frame as for 64.  Operands of four consecutive MFMAs are one 16-byte load per lane
// (FcPack, common.hpp); NBUF groups are in flight per lane.  A workgroup is up to four 16-frame row tiles of ONE column
// tile and input range (waves in step: the later waves' weight loads hit L1).
#define FC_PARTS 16
template <class F, int... I>
__device__ __forceinline__ void fc_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void fc_static_for(F&& f) { fc_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int NBUF>
__global__ __launch_bounds__(256) void k_fc_mfma(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y /* [FC_PARTS][frames][n_out] */,
                                                 int frames, int n_in, int n_out, int parts_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ctiles = n_out >> 4, ct = blockIdx.x, rt = blockIdx.y * 4 + wave;   // up to four row tiles share a weight panel through L1
    if (rt * 16 >= frames) return;
    const int row = min(rt * 16 + (lane & 15), frames - 1);
    const f32x4* __restrict__ ap = (const f32x4*)(x + (long long)row * n_in) + (lane >> 4);
    const f32x4* __restrict__ wp = (const f32x4*)w + (size_t)ct * 64 + lane;
    const size_t wstep = (size_t)ctiles * 64;
    const int KG_all = n_in >> 4, gp = (KG_all + FC_PARTS - 1) / FC_PARTS;
    const int col = ct * 16 + (lane & 15);
    f32x4 av[NBUF], bv[NBUF];
    // (many frames per call: a workgroup walks several input ranges one after the other -- fewer, longer-lived workgroups
    //  stream the weights better than 4096 short ones next to the other stream's kernels; one accumulator chain per range)
    for (int part = blockIdx.z * parts_per_wg; part < (int)(blockIdx.z + 1) * parts_per_wg; ++part) {
    const int kg0 = min(part * gp, KG_all), KG = min((part + 1) * gp, KG_all);      // this part's groups of 16 inputs: [kg0, KG)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](int kg, auto buf_tag) {
        constexpr int buf = decltype(buf_tag)::value;
        kg = min(kg, KG_all - 1);                      // unconditional (see k_pointwise_deep)
        av[buf] = ap[kg * 4];
        bv[buf] = wp[(size_t)kg * wstep];
    };
    auto compute = [&](int kg, auto buf_tag) {
        constexpr int buf = decltype(buf_tag)::value;
        if (kg < KG) {                                 // uniform
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][t], bv[buf][t], acc, 0, 0, 0);
        }
    };
    fc_static_for<NBUF - 1>([&](auto i) { load(kg0 + decltype(i)::value, i); });
    for (int kg = kg0; kg < KG; kg += NBUF) {
        fc_static_for<NBUF>([&](auto i) {
            constexpr int I = decltype(i)::value;
            load(kg + I + NBUF - 1, std::integral_constant<int, (I + NBUF - 1) % NBUF>{});
            __builtin_amdgcn_sched_barrier(0);
            compute(kg + I, i);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int rr = rt * 16 + (lane >> 4) * 4 + reg;
        if (rr < frames) y[((long long)part * frames + rr) * n_out + col] = acc[reg];
    }
    }
}

// Many frames per call (> 16): one workgroup per column tile, one wave per 16-frame row tile, and the 16 input ranges walked as ONE
// stream of weight pieces (the ring of NBUF loads never drains at a range boundary: restarting it 16 times -- or writing and
// re-reading 17 MB of partial sums -- is what made the split form slower than round 2's single chain at 64 frames).  At a range
// boundary the accumulator goes to LDS and restarts from zero; at the end every lane adds its 16 partial sums as
// fc_combine_one does (balanced binary tree, then the bias) and writes y_raw.  Same chains, same tree, same bits.
template <int NBUF>
__global__ __launch_bounds__(256) void k_fc_mfma_seq(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ y_raw, int frames, int n_in, int n_out) {
    __shared__ f32x4 part_acc[4][FC_PARTS][64];                   // 64 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ctiles = n_out >> 4, ct = blockIdx.x, rt = blockIdx.y * 4 + wave;
    if (rt * 16 >= frames) return;
    const int row = min(rt * 16 + (lane & 15), frames - 1);
    const f32x4* __restrict__ ap = (const f32x4*)(x + (long long)row * n_in) + (lane >> 4);
    const f32x4* __restrict__ wp = (const f32x4*)w + (size_t)ct * 64 + lane;
    const size_t wstep = (size_t)ctiles * 64;
    const int KG_all = n_in >> 4, gp = (KG_all + FC_PARTS - 1) / FC_PARTS;
    const int col = ct * 16 + (lane & 15);
    f32x4 av[NBUF], bv[NBUF];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int part = 0, boundary = gp;                                  // the current range is [part * gp, boundary)
    auto load = [&](int kg, auto buf_tag) {
        constexpr int buf = decltype(buf_tag)::value;
        kg = min(kg, KG_all - 1);                      // unconditional (see k_pointwise_deep)
        av[buf] = ap[kg * 4];
        bv[buf] = wp[(size_t)kg * wstep];
    };
    auto compute = [&](int kg, auto buf_tag) {
        constexpr int buf = decltype(buf_tag)::value;
        if (kg < KG_all) {                             // uniform
            if (kg == boundary) {                      // uniform: the range is complete
                part_acc[wave][part][lane] = acc;
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                ++part; boundary += gp;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][t], bv[buf][t], acc, 0, 0, 0);
        }
    };
    fc_static_for<NBUF - 1>([&](auto i) { load(decltype(i)::value, i); });
    for (int kg = 0; kg < KG_all; kg += NBUF) {
        fc_static_for<NBUF>([&](auto i) {
            constexpr int I = decltype(i)::value;
            load(kg + I + NBUF - 1, std::integral_constant<int, (I + NBUF - 1) % NBUF>{});
            __builtin_amdgcn_sched_barrier(0);
            compute(kg + I, i);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    part_acc[wave][part][lane] = acc;
    for (int q = part + 1; q < FC_PARTS; ++q) part_acc[wave][q][lane] = f32x4{0.f, 0.f, 0.f, 0.f};   // (ranges past the inputs: empty chains)
    // (a lane reads back only what it wrote: no barrier)
    f32x4 p[FC_PARTS];
#pragma unroll
    for (int q = 0; q < FC_PARTS; ++q) p[q] = part_acc[wave][q][lane];
#pragma unroll
    for (int m = FC_PARTS; m > 1; m >>= 1)
#pragma unroll
        for (int q = 0; q < m / 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) p[q][e] = p[2 * q][e] + p[2 * q + 1][e];
    const float b = bias[col];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int rr = rt * 16 + (lane >> 4) * 4 + reg;
        if (rr < frames) y_raw[(long long)rr * n_out + col] = p[0][reg] + b;
    }
}

// Many frames per call, blocked (calls that fill the chip with 64-frame x 16 NCT-column workgroups).  k_fc_mfma_seq runs at the OPERAND
// rate: a wave fetches 1 KB of activations and 1 KB of weights per four MFMAs, the four waves of a workgroup fetch the SAME weights,
// and two workgroups per CU ask the load path for 128 bytes per clock (0.34 of the f32 MFMA peak at 256 frames).  Here a wave owns 16
// frames x NCT column tiles (its activations serve NCT tiles), the workgroup's weight pieces -- the same for its four row tiles -- are
// fetched ONCE, one 16-byte piece per thread and k-group, and shared through a double-buffered LDS block of FC_S k-groups (one barrier
// per round: a buffer is rewritten two rounds after it was read), and the 64 KB of range partials are gone: a completed range is merged
// as a binary counter (range 1 joins range 0 when it completes, (2, 3) joins (0, 1) when 3 completes, ...), which adds exactly the
// pairs of fc_combine_one's balanced tree in its order -- four pending levels in registers instead of sixteen LDS slots.  Per
// (frame, output) the chain is k_fc_mfma_seq's: k-groups ascending, four MFMAs each, a fresh accumulator per input range.
#define FC_S 6
template <int NCT>
__global__ __launch_bounds__(256) void k_fc_mfma_tile(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ y_raw, int frames, int n_in, int n_out) {
    constexpr int PCS = NCT * 64;                                 // weight pieces of the workgroup per k-group
    constexpr int PER_T = FC_S * PCS / 256;                       // pieces a thread stages per round
    static_assert((FC_S * PCS) % 256 == 0, "whole pieces per thread");
    __shared__ f32x4 wb[2][FC_S * PCS];                           // 2 x 6 x NCT KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctiles = n_out >> 4, ct0 = blockIdx.x * NCT, rt = blockIdx.y * 4 + wave;
    const bool live = rt * 16 < frames;                           // (a wave without frames still stages weights and meets the barriers)
    const int row = min(rt * 16 + (lane & 15), frames - 1);
    const f32x4* __restrict__ ap = (const f32x4*)(x + (long long)row * n_in) + (lane >> 4);
    const f32x4* __restrict__ wp = (const f32x4*)w + (size_t)ct0 * 64;
    const size_t wstep = (size_t)ctiles * 64;
    const int KG_all = n_in >> 4, gp = (KG_all + FC_PARTS - 1) / FC_PARTS;
    f32x4 acc[NCT], lvl[4][NCT], total[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) { acc[c] = f32x4{0.f, 0.f, 0.f, 0.f}; total[c] = acc[c]; }
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int c = 0; c < NCT; ++c) lvl[l][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // rounds never straddle a range: a range is rpp rounds of FC_S k-groups (7680 inputs: 30 k-groups = 5 rounds of 6); the slots of a
    // short last round get ZERO activations -- fma(0, w, acc) == acc, and an accumulator that started at +0 is never -0 -- so the MFMAs
    // of a round are straight-line code (a branch per k-group makes the compiler park the accumulators in vector registers around it:
    // 32 moves and a drained matrix pipe per k-group) and a range completes at ONE place in the code
    const int rpp = (gp + FC_S - 1) / FC_S;
    f32x4 an[FC_S], bs[PER_T];                                    // the next round's activations / this thread's weight pieces
    int nv_next = 0;
    auto fetch = [&](int q, int rr) {                             // round rr of range q -> registers
        const int k0 = q * gp + rr * FC_S, kl = max(KG_all - 1, 0);
        nv_next = min(min((q + 1) * gp, KG_all), k0 + FC_S) - k0;                  // slots of the round that exist (the others are zeroed when USED:
#pragma unroll                                                                      //  a select right here would wait for the loads it is meant to hide)
        for (int s = 0; s < FC_S; ++s) an[s] = ap[min(k0 + s, kl) * 4];
#pragma unroll
        for (int j = 0; j < PER_T; ++j) {
            const int pc = j * 256 + tid, sl = pc / PCS, o = pc - sl * PCS;       // slot (uniform per wave: PCS is 128 or 256) and piece
            const int kg = __builtin_amdgcn_readfirstlane(min(k0 + sl, kl));
            bs[j] = wp[(size_t)kg * wstep + o];
        }
    };
    fetch(0, 0);
    int buf = 0;
    for (int q = 0; q < FC_PARTS; ++q) {
        for (int rr = 0; rr < rpp; ++rr) {
            f32x4 av[FC_S];
            const int nv = nv_next;
#pragma unroll
            for (int s = 0; s < FC_S; ++s) av[s] = s < nv ? an[s] : f32x4{0.f, 0.f, 0.f, 0.f};          // (uniform select)
#pragma unroll
            for (int j = 0; j < PER_T; ++j) wb[buf][j * 256 + tid] = bs[j];
            __syncthreads();
            {   // the next round, unconditionally (the last one re-reads itself; see k_pointwise_deep): in flight during this round's MFMAs
                const bool last_rr = rr + 1 == rpp, last = last_rr && q + 1 == FC_PARTS;
                fetch(last ? q : last_rr ? q + 1 : q, last ? rr : last_rr ? 0 : rr + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4* __restrict__ bp = wb[buf] + lane;
#pragma unroll
            for (int s = 0; s < FC_S; ++s) {
                f32x4 bv[NCT];
#pragma unroll
                for (int c = 0; c < NCT; ++c) bv[c] = bp[s * PCS + c * 64];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], bv[c][t], acc[c], 0, 0, 0);
            }
            buf ^= 1;
        }
        // range q is complete: merge it as a binary counter -- the pairs and the order of fc_combine_one's balanced tree (an empty range
        // adds zeros, as there)
        f32x4 v[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) { v[c] = acc[c]; acc[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        bool placed = false;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            if (!placed) {
                if ((q >> l) & 1) {
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[c][e] = lvl[l][c][e] + v[c][e];
                } else {
#pragma unroll
                    for (int c = 0; c < NCT; ++c) lvl[l][c] = v[c];
                    placed = true;
                }
            }
        }
        if (!placed) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) total[c] = v[c];
        }
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int col = (ct0 + c) * 16 + (lane & 15);
        const float b = bias[col];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rr = rt * 16 + (lane >> 4) * 4 + reg;
            if (rr < frames) y_raw[(long long)rr * n_out + col] = total[c][reg] + b;
        }
    }
}

__device__ __forceinline__ float fc_combine_one(const float* __restrict__ partial, const float* __restrict__ bias, int frames, int n, int f, int i) {
    float p[FC_PARTS];
#pragma unroll
    for (int q = 0; q < FC_PARTS; ++q) p[q] = partial[((long long)q * frames + f) * n + i];
#pragma unroll
    for (int m = FC_PARTS; m > 1; m >>= 1)
#pragma unroll
        for (int q = 0; q < m / 2; ++q) p[q] = p[2 * q] + p[2 * q + 1];
    return p[0] + bias[i];
}

// partial sums [FC_PARTS][frames][n] -> y = tree + bias (kept in y_raw), then the L2 normalisation of layers.py:108.
// One launch for the single-frame path (a workgroup per frame) ...
__global__ __launch_bounds__(1024) void k_fc_combine_l2(const float* __restrict__ partial, const float* __restrict__ bias, float* __restrict__ y_raw,
                                                        float* __restrict__ out, int frames, int n, FcHostOut host) {
    __shared__ float red[256];
    const int f = blockIdx.x;
    float* v = y_raw + (long long)f * n;
    for (int i = threadIdx.x; i < n; i += 1024) v[i] = fc_combine_one(partial, bias, frames, n, f, i);
    __syncthreads();
    const float ss = block_sumsq_tree256(v, n, red);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int i = threadIdx.x; i < n; i += 1024) out[(long long)f * n + i] = v[i] * inv;
    if (host.out) {
        // the caller of a single-frame call waits for exactly these 16 KB: they go straight into its pinned block, followed
        // by the call's number (written by the workgroup that finishes last), instead of through a copy after the join
        for (int i = threadIdx.x; i < n; i += 1024) host.out[(long long)f * n + i] = v[i] * inv;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(host.seq + 1, 1) == frames - 1) {
            host.seq[1] = 0;
            const int call = host.seq[0] + 1;
            host.seq[0] = call;
            __threadfence_system();
            __hip_atomic_store(host.flag, call, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// ... two for many frames (the 17 MB of partial sums of a 64-frame call want more than 64 workgroups)
__global__ __launch_bounds__(256) void k_fc_combine(const float* __restrict__ partial, const float* __restrict__ bias, float* __restrict__ y_raw, int frames, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i < n) y_raw[(long long)f * n + i] = fc_combine_one(partial, bias, frames, n, f, i);
}
__global__ __launch_bounds__(256) void k_l2norm_vec(const float* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ float red[256];
    const float* v = in + (long long)blockIdx.x * n;
    const float ss = block_sumsq_tree256(v, n, red);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int i = threadIdx.x; i < n; i += 256) out[(long long)blockIdx.x * n + i] = v[i] * inv;
}

size_t fc_scratch_floats(const FcPack& fc, int frames) { return (size_t)FC_PARTS * (size_t)frames * (size_t)fc.n_out; }

bool fc_host_out_supported(int frames) { return frames >= 1 && frames <= 4; }

hipError_t launch_fc_l2(const float* x, const FcPack& fc, float* partial, float* y_raw, float* out, int frames, hipStream_t s, FcHostOut host, int fc_tile) {
    if (frames <= 0) return hipSuccess;
    if (fc.n_in % 16 || fc.n_out % 16) return hipErrorInvalidValue;
    const int waves = std::min(4, (frames + 15) / 16);
    if (host.out && !fc_host_out_supported(frames)) return hipErrorInvalidValue;
    if (frames > 16) {
        // blocked form when its workgroups fill the chip (256 CUs): 64 frames x 64 columns each, 64 x 32 for fewer frames
        const int rg = (frames + 63) / 64;
        const int nct = fc_tile <= 0 ? 0 : fc_tile == 2 || fc_tile == 4 ? fc_tile : (fc.n_out / 64) * rg >= 256 ? 4 : (fc.n_out / 32) * rg >= 256 ? 2 : 0;
        if (nct == 4 && fc.n_out % 64 == 0) hipLaunchKernelGGL(k_fc_mfma_tile<4>, dim3(fc.n_out / 64, rg), dim3(256), 0, s, x, fc.w, fc.bias, y_raw, frames, fc.n_in, fc.n_out);
        else if (nct == 2 && fc.n_out % 32 == 0) hipLaunchKernelGGL(k_fc_mfma_tile<2>, dim3(fc.n_out / 32, rg), dim3(256), 0, s, x, fc.w, fc.bias, y_raw, frames, fc.n_in, fc.n_out);
        else
        hipLaunchKernelGGL(k_fc_mfma_seq<16>, dim3(fc.n_out / 16, (frames + 63) / 64), dim3(64 * waves), 0, s, x, fc.w, fc.bias, y_raw, frames, fc.n_in, fc.n_out);
        hipLaunchKernelGGL(k_l2norm_vec, dim3(frames), dim3(256), 0, s, y_raw, out, fc.n_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_fc_mfma<16>, dim3(fc.n_out / 16, (frames + 63) / 64, FC_PARTS), dim3(64 * waves), 0, s, x, fc.w, partial, frames, fc.n_in, fc.n_out, 1);
    if (frames <= 4) {
        hipLaunchKernelGGL(k_fc_combine_l2, dim3(frames), dim3(1024), 0, s, partial, fc.bias, y_raw, out, frames, fc.n_out, host);
    } else {
        hipLaunchKernelGGL(k_fc_combine, dim3((fc.n_out + 255) / 256, frames), dim3(256), 0, s, partial, fc.bias, y_raw, frames, fc.n_out);
        hipLaunchKernelGGL(k_l2norm_vec, dim3(frames), dim3(256), 0, s, y_raw, out, fc.n_out);
    }
    return hipGetLastError();
}

// ---- the FC on split-bf16 operands: partial GEMMs (kernels_conv.hip) -> slabs summed in part order + bias -> L2 normalisation
__global__ __launch_bounds__(256) void k_fc_combine_parts(const float* __restrict__ partial, int parts, const float* __restrict__ bias, float* __restrict__ y_raw, int frames, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= n) return;
    float v = bias[i];
    for (int q = 0; q < parts; ++q) v += partial[((long long)q * frames + f) * n + i];
    y_raw[(long long)f * n + i] = v;
}
hipError_t launch_fc_l2_bf16x3(const float* x, const FcPack& fc, const void* Wb, float* partial, float* y_raw, float* out, int frames, hipStream_t s) {
    const hipError_t e = launch_fc_partials_bf16x3(x, fc, Wb, partial, frames, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fc_combine_parts, dim3((fc.n_out + 255) / 256, frames), dim3(256), 0, s, partial, FC_BF_PARTS, fc.bias, y_raw, frames, fc.n_out);
    hipLaunchKernelGGL(k_l2norm_vec, dim3(frames), dim3(256), 0, s, y_raw, out, fc.n_out);
    return hipGetLastError();
}

}  // namespace hfnet
