// kernels_detect.hip -- detector tail and keypoint / descriptor post-processing on the GPU.
//
// The reference does all of this on the CPU after copying the dense maps back
// (src/Extractors/HFNetRTModel.cc:134,139-196 == HFNetTFModelV2.cc:111-168); here only the selected
// keypoints and their descriptors ever leave HBM.
#include "kernels.hpp"

#include <cfloat>

namespace hfnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// exp() of the softmaxes: identical operation sequence to oracle/hfnet_oracle.c hfo_expf
__device__ __forceinline__ float hf_expf(float x) {
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

// =========================================================================== softmax + depth_to_space
// One thread per 8x8 cell: softmax over 65 logits (left-to-right sum), drop the dustbin, scatter the
// 64 probabilities to pixel (8*cy + k/8, 8*cx + k%8)  (hf_net.py:88-93).
__global__ __launch_bounds__(256) void k_softmax_d2s(const float* __restrict__ logits, int ld, float* __restrict__ dense, Geom g) {
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];       // H, W: cell grid; Ho, Wo: dense map (8H, 8W)
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= lv.H * lv.W) return;
    const int cy = cell / lv.W, cx = cell - cy * lv.W;
    const float* r = logits + (lv.in_off + (long long)frame * lv.H * lv.W + cell) * ld;
    float e[65];
    float mx = r[0];
#pragma unroll
    for (int k = 0; k < 65; ++k) { e[k] = r[k]; mx = fmaxf(mx, e[k]); }
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 65; ++k) { e[k] = hf_expf(e[k] - mx); sum = sum + e[k]; }
    float* d = dense + lv.out_off + (long long)frame * lv.Ho * lv.Wo + (long long)(cy * 8) * lv.Wo + cx * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = e[i * 8 + j] / sum; b[j] = e[i * 8 + 4 + j] / sum; }
        *(f32x4*)(d + (long long)i * lv.Wo) = a;
        *(f32x4*)(d + (long long)i * lv.Wo + 4) = b;
    }
}

hipError_t launch_softmax_d2s(const float* logits, int ld, float* dense, const Geom& g, hipStream_t s) {
    int maxcells = 0;
    for (int l = 0; l < g.n_levels; ++l) maxcells = max(maxcells, g.lv[l].H * g.lv[l].W);
    dim3 grid((maxcells + 255) / 256, g.n_levels * g.batch);
    hipLaunchKernelGGL(k_softmax_d2s, grid, dim3(256), 0, s, logits, ld, dense, g);
    return hipGetLastError();
}

// =========================================================================== simple_nms + candidates
// layers.py:10-32 with radius 4, iterations 2 (export_model.py:35,37): three dependent 9x9 max-pools.
// A workgroup produces a 32x32 output tile from a 56x56 LDS tile (halo 3*4), each pool done
// separably (row max, column max).  Out-of-image cells are -inf (max_pool 'SAME' ignores them).
// Survivors with score >= threshold are appended to the image's candidate list as 64-bit keys
//   (~score_bits << 32) | (col * H + row)
// so ascending key order == (response descending, column-major index ascending).
#define NMS_T 32
#define NMS_R 4
#define NMS_S 56   // NMS_T + 6 * NMS_R

// sliding 9-max over 16 consecutive values -> 8 outputs: out[i] = max(v[i..i+8]) as
// max(suffix-max of v[i..7], prefix-max of v[8..i+8]): 22 max operations instead of 64
__device__ __forceinline__ void max9_strip(const float (&v)[16], float (&o)[8]) {
    float sfx[8], pfx[8];
    sfx[7] = v[7];
#pragma unroll
    for (int i = 6; i >= 0; --i) sfx[i] = fmaxf(v[i], sfx[i + 1]);
    pfx[0] = v[8];
#pragma unroll
    for (int i = 1; i < 8; ++i) pfx[i] = fmaxf(pfx[i - 1], v[8 + i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaxf(sfx[i], pfx[i]);
}
// row pass: in [ROWS][CIN] -> out [ROWS][CIN-8]; one work item = (row, strip of 8 outputs), 16-byte LDS accesses
template <int ROWS, int CIN>
__device__ __forceinline__ void nms_row_pass(const float* __restrict__ in, float* __restrict__ out) {
    constexpr int COUT = CIN - 8, STRIPS = COUT / 8;
    for (int w = threadIdx.x; w < ROWS * STRIPS; w += 256) {
        const int row = w / STRIPS, st = w - row * STRIPS;
        float v[16], o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *(const f32x4*)(in + row * CIN + st * 8 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[q * 4 + j] = t[j];
        }
        max9_strip(v, o);
        *(f32x4*)(out + row * COUT + st * 8) = (f32x4){o[0], o[1], o[2], o[3]};
        *(f32x4*)(out + row * COUT + st * 8 + 4) = (f32x4){o[4], o[5], o[6], o[7]};
    }
}
// column pass: in [RIN][COLS] -> 8 pooled values per work item (col, strip of 8 output rows), handed to `emit`
template <int RIN, int COLS, class F>
__device__ __forceinline__ void nms_col_pass(const float* __restrict__ in, F emit) {
    constexpr int STRIPS = (RIN - 8) / 8;
    for (int w = threadIdx.x; w < COLS * STRIPS; w += 256) {
        const int st = w / COLS, col = w - st * COLS;
        float v[16], o[8];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = in[(st * 8 + i) * COLS + col];
        max9_strip(v, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) emit(st * 8 + i, col, o[i]);
    }
}

__global__ __launch_bounds__(256) void k_nms(const float* __restrict__ dense, float* __restrict__ nms, unsigned long long* __restrict__ cand,
                                             unsigned int* __restrict__ counters, long long cand_stride, float threshold, Geom g) {
    __shared__ __attribute__((aligned(16))) float s[NMS_S * NMS_S];
    __shared__ __attribute__((aligned(16))) float tmp[NMS_S * 48];
    __shared__ __attribute__((aligned(16))) float m0[48 * 48];
    __shared__ __attribute__((aligned(16))) float supp[40 * 40];
    __shared__ __attribute__((aligned(16))) float ss[40 * 40];
    const int image = blockIdx.z, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int H = lv.H, W = lv.W;
    const int x0 = blockIdx.x * NMS_T, y0 = blockIdx.y * NMS_T;
    if (x0 >= W || y0 >= H) return;
    const float* src = dense + lv.in_off + (long long)frame * H * W;
    const float NEG = -INFINITY;
    for (int i = threadIdx.x; i < NMS_S * NMS_S; i += 256) {
        const int ty = i / NMS_S, tx = i - ty * NMS_S;
        const int gy = y0 - 3 * NMS_R + ty, gx = x0 - 3 * NMS_R + tx;
        s[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(long long)gy * W + gx] : NEG;
    }
    __syncthreads();
    // pool 1: 56x56 -> 48x48; m0 = (score == pooled) inside the image
    nms_row_pass<NMS_S, NMS_S>(s, tmp);
    __syncthreads();
    nms_col_pass<NMS_S, 48>(tmp, [&](int ty, int tx, float m) {
        const float c = s[(ty + NMS_R) * NMS_S + tx + NMS_R];
        m0[ty * 48 + tx] = (c != NEG && c == m) ? 1.0f : 0.0f;
    });
    __syncthreads();
    // pool 2 (of the mask): 48x48 -> 40x40; suppressed scores
    nms_row_pass<48, 48>(m0, tmp);
    __syncthreads();
    nms_col_pass<48, 40>(tmp, [&](int ty, int tx, float m) {
        const float c = s[(ty + 2 * NMS_R) * NMS_S + tx + 2 * NMS_R];
        supp[ty * 40 + tx] = m;
        ss[ty * 40 + tx] = (c == NEG) ? NEG : (m != 0.0f ? 0.0f : c);
    });
    __syncthreads();
    // pool 3: 40x40 -> 32x32 (row pass into tmp, column pass below)
    nms_row_pass<40, 40>(ss, tmp);
    __syncthreads();
    float* dst = nms + lv.out_off + (long long)frame * H * W;
    unsigned long long* cl = cand + (long long)image * cand_stride;
    // candidates are collected in LDS first: one global atomic per workgroup instead of one per
    // candidate (all candidates of an image hit the same counter)
    unsigned long long* lkeys = (unsigned long long*)tmp;     // 1024 keys = 8 KB <= sizeof(tmp) (tmp is dead after pool 3)
    __shared__ unsigned int lcount, lbase;
    float ovals[4];
    // pool 3's column pass is folded into the final per-pixel stage (9 LDS reads per output)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 256;
        const int ty = i / NMS_T, tx = i - ty * NMS_T;
        float m = tmp[ty * 32 + tx];
#pragma unroll
        for (int d = 1; d <= 2 * NMS_R; ++d) m = fmaxf(m, tmp[(ty + d) * 32 + tx]);
        const float sv = s[(ty + 3 * NMS_R) * NMS_S + tx + 3 * NMS_R];
        const bool is_max0 = m0[(ty + 2 * NMS_R) * 48 + tx + 2 * NMS_R] != 0.0f;
        const bool is_supp = supp[(ty + NMS_R) * 40 + tx + NMS_R] != 0.0f;
        const bool new_max = ss[(ty + NMS_R) * 40 + tx + NMS_R] == m;
        ovals[it] = (is_max0 || (new_max && !is_supp)) ? sv : 0.0f;
    }
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();                                           // every read of tmp is done
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 256;
        const int ty = i / NMS_T, tx = i - ty * NMS_T;
        const int gy = y0 + ty, gx = x0 + tx;
        if (gy >= H || gx >= W) continue;
        const float o = ovals[it];
        dst[(long long)gy * W + gx] = o;
        if (o >= threshold) {
            const unsigned int slot = atomicAdd(&lcount, 1u);
            lkeys[slot] = ((unsigned long long)(~__float_as_uint(o)) << 32) | (unsigned int)(gx * H + gy);
        }
    }
    __syncthreads();
    const unsigned int cnt = lcount;
    if (cnt == 0) return;
    if (threadIdx.x == 0) lbase = atomicAdd(&counters[image], cnt);
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < cnt; i += 256) cl[lbase + i] = lkeys[i];
}

hipError_t launch_nms(const float* dense, float* nms, unsigned long long* cand, unsigned int* counters, long long cand_stride,
                      float threshold, const Geom& g, hipStream_t s) {
    int maxw = 0, maxh = 0;
    for (int l = 0; l < g.n_levels; ++l) { maxw = max(maxw, g.lv[l].W); maxh = max(maxh, g.lv[l].H); }
    dim3 grid((maxw + NMS_T - 1) / NMS_T, (maxh + NMS_T - 1) / NMS_T, g.n_levels * g.batch);
    hipLaunchKernelGGL(k_nms, grid, dim3(256), 0, s, dense, nms, cand, counters, cand_stride, threshold, g);
    return hipGetLastError();
}

// =========================================================================== top-K
// HFNetTFModelV2.cc:144-151.  One 1024-thread workgroup per image.  n <= K: keep everything in scan
// (column-major) order.  n > K: exact radix select of the K smallest 64-bit keys (8 passes of 8 bits),
// then a bitonic sort in LDS -> (response desc, column-major index asc).
#define TOPK_CAP 8192
__global__ __launch_bounds__(1024) void k_topk(const unsigned long long* __restrict__ cand, const unsigned int* __restrict__ counters,
                                               long long cand_stride, TopkBudget kmax_per_level,
                                               hfnet_keypoint* __restrict__ kps, long long kps_stride, int* __restrict__ n_out, Geom g) {
    __shared__ unsigned long long buf[TOPK_CAP];
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sh_prefix;
    __shared__ unsigned int sh_remaining, sh_count;
    const int image = blockIdx.x, level = image / g.batch;
    const int H = g.lv[level].H;
    const unsigned long long* cl = cand + (long long)image * cand_stride;
    const unsigned int n = counters[image];
    int K = kmax_per_level.k[level];
    if (K > TOPK_CAP) K = TOPK_CAP;
    const int tid = threadIdx.x;
    unsigned int m;        // number of selected keys
    bool by_index;
    if (K <= 0) {
        m = 0; by_index = true;
    } else if (n <= (unsigned)K) {
        m = n; by_index = true;
        for (unsigned int i = tid; i < n; i += 1024) { const unsigned long long k = cl[i]; buf[i] = (k << 32) | (k >> 32); }
    } else {
        m = (unsigned)K; by_index = false;
        if (tid == 0) { sh_prefix = 0ull; sh_remaining = (unsigned)K; }
        __syncthreads();
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned long long prefix = sh_prefix;
            const unsigned long long mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
            for (unsigned int i = tid; i < n; i += 1024) {
                const unsigned long long k = cl[i];
                if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 64) {     // wave 0: find the bucket where the cumulative count reaches `remaining`
                const unsigned int rem = sh_remaining;
                const unsigned int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                const unsigned int mine = h0 + h1 + h2 + h3;
                unsigned int incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned int v = __shfl_up(incl, off, 64);
                    if (tid >= off) incl += v;
                }
                const unsigned int excl = incl - mine;
                if (excl < rem && rem <= incl) {          // exactly one lane
                    unsigned int before = excl, b = 4 * tid;
                    if (before + h0 < rem) { before += h0; ++b; if (before + h1 < rem) { before += h1; ++b; if (before + h2 < rem) { before += h2; ++b; } } }
                    sh_remaining = rem - before;
                    sh_prefix = prefix | ((unsigned long long)b << shift);
                }
            }
            __syncthreads();
        }
        const unsigned long long kth = sh_prefix;   // the K-th smallest key (keys are unique)
        if (tid == 0) sh_count = 0;
        __syncthreads();
        for (unsigned int i = tid; i < n; i += 1024) {
            const unsigned long long k = cl[i];
            if (k <= kth) { const unsigned int slot = atomicAdd(&sh_count, 1u); if (slot < TOPK_CAP) buf[slot] = k; }
        }
    }
    __syncthreads();
    unsigned int np2 = 1;
    while (np2 < m) np2 <<= 1;
    for (unsigned int i = m + tid; i < np2; i += 1024) buf[i] = ~0ull;
    __syncthreads();
    for (unsigned int k = 2; k <= np2; k <<= 1)
        for (unsigned int j = k >> 1; j > 0; j >>= 1) {
            for (unsigned int i = tid; i < np2; i += 1024) {
                const unsigned int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = buf[i], b = buf[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { buf[i] = b; buf[ixj] = a; }
                }
            }
            __syncthreads();
        }
    hfnet_keypoint* out = kps + (long long)image * kps_stride;
    for (unsigned int i = tid; i < m; i += 1024) {
        unsigned long long k = buf[i];
        if (by_index) k = (k << 32) | (k >> 32);
        const unsigned int idx = (unsigned int)k;
        hfnet_keypoint kp;
        kp.x = (float)(idx / (unsigned)H);
        kp.y = (float)(idx % (unsigned)H);
        kp.response = __uint_as_float(~(unsigned int)(k >> 32));
        kp.octave = 0;
        out[i] = kp;
    }
    if (tid == 0) n_out[image] = (int)m;
}

hipError_t launch_topk(const unsigned long long* cand, const unsigned int* counters, long long cand_stride,
                       const TopkBudget& kmax_per_level, hfnet_keypoint* kps, long long kps_stride, int* n_out, const Geom& g,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_topk, dim3(g.n_levels * g.batch), dim3(1024), 0, s, cand, counters, cand_stride, kmax_per_level, kps,
                       kps_stride, n_out, g);
    return hipGetLastError();
}

// =========================================================================== per-pixel L2 normalise
// tf.nn.l2_normalize over 256 channels: x * (1 / sqrt(max(sum x^2, 1e-12))), sum in tree256 order.
// One wave per pixel, lane l holds channels 4l..4l+3 (== tree256 partials 4l..4l+3).
__device__ __forceinline__ float tree256_wave(f32x4 p) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = p[j] + __shfl_xor(p[j], off, 64);
    }
    const float a = p[0] + p[2], b = p[1] + p[3];
    return a + b;
}

__global__ __launch_bounds__(256) void k_l2norm256(const float* __restrict__ in, float* __restrict__ out, long long P) {
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= P) return;
    const int lane = threadIdx.x & 63;
    const f32x4 v = *(const f32x4*)(in + pix * 256 + lane * 4);
    f32x4 sq;
#pragma unroll
    for (int j = 0; j < 4; ++j) sq[j] = v[j] * v[j];
    const float ss = tree256_wave(sq);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] * inv;
    *(f32x4*)(out + pix * 256 + lane * 4) = o;
}

hipError_t launch_l2norm256(const float* in, float* out, long long P, hipStream_t s) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_l2norm256, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, s, in, out, P);
    return hipGetLastError();
}

// =========================================================================== sample + normalise
// One wave per keypoint.  warp = (x*(Wd-1)/(W-1), y*(Hd-1)/(H-1)) (HFNetTFModelV2.cc:119-120,156-160);
// bilinear Resampler with the reference's expression order (BaseModel.cc:491-562); cv::normalize:
// norm accumulated in double (tree256 order), row *= (float)(1/norm).
__global__ __launch_bounds__(256) void k_sample(SampleArgs a, Geom g) {
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];           // H, W: score map; Ho, Wo: descriptor map
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    int base = 0, total = 0;
    for (int l = 0; l < g.n_levels; ++l) {
        const int nl = a.n_in[l * g.batch + frame];
        if (l < level) base += nl;
        total += nl;
    }
    const int n = a.n_in[image];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (level == 0 && a.n_out_frame) a.n_out_frame[frame] = total;
        if (a.n_out_level) a.n_out_level[frame * g.n_levels + level] = n;
    }
    if (i >= n) return;
    const hfnet_keypoint kp = a.kps_in[(long long)image * a.kps_stride + i];
    const int dw = lv.Wo, dh = lv.Ho;
    const float sw = ((float)dw - 1.f) / (float)((float)lv.W - 1.f);
    const float sh = ((float)dh - 1.f) / (float)((float)lv.H - 1.f);
    const float x = sw * kp.x, y = sh * kp.y;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (x > -1.0f && y > -1.0f && x < (float)dw && y < (float)dh) {
        const int fx = (int)floorf(x), fy = (int)floorf(y), cx = fx + 1, cy = fy + 1;
        const float dx = (float)cx - x, dy = (float)cy - y;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const bool fxin = fx >= 0 && fx <= dw - 1, cxin = cx >= 0 && cx <= dw - 1;
        const bool fyin = fy >= 0 && fy <= dh - 1, cyin = cy >= 0 && cy <= dh - 1;
        f32x4 vff, vcc, vfc, vcf;
        if (a.sparse) {   // rows 4i..4i+3 of the image slot hold the taps (fx,fy) (cx,cy) (fx,cy) (cx,fy)
            const float* d = a.desc_map + (((long long)image * a.kps_stride + i) * 4) * 256 + lane * 4;
            vff = (fxin && fyin) ? *(const f32x4*)(d) : zero;
            vcc = (cxin && cyin) ? *(const f32x4*)(d + 256) : zero;
            vfc = (fxin && cyin) ? *(const f32x4*)(d + 512) : zero;
            vcf = (cxin && fyin) ? *(const f32x4*)(d + 768) : zero;
        } else {
            const float* d = a.desc_map + (lv.in_off + (long long)frame * dh * dw) * 256 + lane * 4;
            vff = (fxin && fyin) ? *(const f32x4*)(d + (long long)(fy * dw + fx) * 256) : zero;
            vcc = (cxin && cyin) ? *(const f32x4*)(d + (long long)(cy * dw + cx) * 256) : zero;
            vfc = (fxin && cyin) ? *(const f32x4*)(d + (long long)(cy * dw + fx) * 256) : zero;
            vcf = (cxin && fyin) ? *(const f32x4*)(d + (long long)(fy * dw + cx) * 256) : zero;
        }
        const float wff = dx * dy, wcc = (1.0f - dx) * (1.0f - dy), wfc = dx * (1.0f - dy), wcf = (1.0f - dx) * dy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t0 = wff * vff[j], t1 = wcc * vcc[j], t2 = wfc * vfc[j], t3 = wcf * vcf[j];
            o[j] = t0 + t1 + t2 + t3;
        }
    }
    double p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = (double)o[j] * (double)o[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = p[j] + __shfl_xor(p[j], off, 64);
    }
    const double ssum = (p[0] + p[2]) + (p[1] + p[3]);
    const double nrm = sqrt(ssum);
    const float sc = (float)(nrm > DBL_EPSILON ? 1.0 / nrm : 0.0);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = o[j] * sc;
    const long long row = (long long)frame * a.out_frame_stride + base + i;
    *(f32x4*)(a.desc_out + row * 256 + lane * 4) = r;
    if (lane == 0) {
        hfnet_keypoint ko = kp;
        if (a.set_octave) { ko.octave = level; ko.x = kp.x * a.scale_factor[level]; ko.y = kp.y * a.scale_factor[level]; }
        a.kps_out[row] = ko;
    }
}

// free-standing Resampler, same expression order as the reference (and as k_sample)
__global__ __launch_bounds__(256) void k_resampler(const float* __restrict__ data, const float* __restrict__ warp, float* __restrict__ out,
                                                   int dh, int dw, int channels, int npoints) {
    const int b = blockIdx.y;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)npoints * channels) return;
    const int p = (int)(idx / channels), c = (int)(idx - (long long)p * channels);
    const float x = warp[((long long)b * npoints + p) * 2], y = warp[((long long)b * npoints + p) * 2 + 1];
    float o = 0.0f;
    if (x > -1.0f && y > -1.0f && x < (float)dw && y < (float)dh) {
        const int fx = (int)floorf(x), fy = (int)floorf(y), cx = fx + 1, cy = fy + 1;
        const float dx = (float)cx - x, dy = (float)cy - y;
        const float* d = data + (long long)b * dh * dw * channels + c;
        auto pt = [&](int xx, int yy) { return (xx >= 0 && yy >= 0 && xx <= dw - 1 && yy <= dh - 1) ? d[(long long)channels * ((long long)yy * dw + xx)] : 0.0f; };
        const float t0 = dx * dy * pt(fx, fy), t1 = (1.0f - dx) * (1.0f - dy) * pt(cx, cy);
        const float t2 = dx * (1.0f - dy) * pt(fx, cy), t3 = (1.0f - dx) * dy * pt(cx, fy);
        o = t0 + t1 + t2 + t3;
    }
    out[((long long)b * npoints + p) * channels + c] = o;
}

hipError_t launch_resampler(const float* data, const float* warp, float* out, int batch, int dh, int dw, int channels, int npoints, hipStream_t s) {
    if (batch <= 0 || npoints <= 0 || channels <= 0) return hipSuccess;
    dim3 grid((unsigned)(((long long)npoints * channels + 255) / 256), batch);
    hipLaunchKernelGGL(k_resampler, grid, dim3(256), 0, s, data, warp, out, dh, dw, channels, npoints);
    return hipGetLastError();
}

hipError_t launch_sample(const SampleArgs& a, const Geom& g, hipStream_t s) {
    // grid.x covers the largest per-level budget; the caller stores it in kps_stride
    dim3 grid((unsigned)((a.kps_stride + 3) / 4), g.n_levels * g.batch);
    hipLaunchKernelGGL(k_sample, grid, dim3(256), 0, s, a, g);
    return hipGetLastError();
}

}  // namespace hfnet
