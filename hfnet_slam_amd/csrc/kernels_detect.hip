// kernels_detect.hip -- detector tail and keypoint / descriptor post-processing on the GPU.
//
// The reference does all of this on the CPU after copying the dense maps back
// (src/Extractors/HFNetRTModel.cc:134,139-196 == HFNetTFModelV2.cc:111-168); here only the selected
// keypoints and their descriptors ever leave HBM.
#include "kernels.hpp"

#include <cstdlib>

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
    // a cell's 65 logits are only 4-byte aligned: 16 unaligned 16-byte loads + 1 instead of 65 scalar ones (every load
    // instruction of a wave touches 64 different cache lines, the address coalescer is what bounds this kernel)
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        const f32x4u v = *(const f32x4u*)(r + 4 * k4);
#pragma unroll
        for (int c = 0; c < 4; ++c) e[4 * k4 + c] = v[c];
    }
    e[64] = r[64];
    float mx = e[0];
#pragma unroll
    for (int k = 1; k < 65; ++k) mx = fmaxf(mx, e[k]);
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
// layers.py:10-32 with radius 4, iterations 2 (export_model.py:35,37): three dependent 9x9 max-pools,
//   max_mask = scores == pool(scores); supp = pool(max_mask) > 0; ss = supp ? 0 : scores;
//   out = (max_mask | (ss == pool(ss) & !supp)) ? scores : 0            (out-of-image cells never win a max: -inf)
// Three streaming passes without LDS tiles or barriers.  A wave owns a 64-column x 40-row window of the map:
// every lane loads its column (40 coalesced row reads in flight), takes the vertical 9-max of 32 rows in
// registers (22 max per 8 rows), and the horizontal 9-max comes from whole-wave DPP shifts.  The two masks are
// bit columns -- one 32-bit word per (32-row block, column) -- so pass 1 stores one word per lane, pass 2 (the
// dilation of max_mask) is a handful of 64-bit shifts and ORs per column, and pass 3 gets the flags of its 40
// rows from four loads.  Pass 3 rebuilds ss on the fly, pools it and emits the map (optional: only taps read it)
// plus the candidate keys  (~score_bits << 32) | (col * H + row)  -- ascending key order == (response
// descending, column-major index ascending).  One global atomic per wave tile.
#define NMS_RB 32          // output rows of a wave tile (+ 8 halo rows) == rows per mask word
#define NMS_CW 56          // output columns of a wave tile (64 lanes - 2 * 4 halo)

// o[i] = max(v[OFF + i .. OFF + i + 8]) for i < 8: suffix maxima of the first 8, prefix maxima of the next 8
template <int OFF, int NIN>
__device__ __forceinline__ void max9_strip_at(const float (&v)[NIN], float* o) {
    float sfx[8], pfx[8];
    sfx[7] = v[OFF + 7];
#pragma unroll
    for (int i = 6; i >= 0; --i) sfx[i] = fmaxf(v[OFF + i], sfx[i + 1]);
    pfx[0] = v[OFF + 8];
#pragma unroll
    for (int i = 1; i < 8; ++i) pfx[i] = fmaxf(pfx[i - 1], v[OFF + 8 + i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaxf(sfx[i], pfx[i]);
}
// whole-wave shifts by one lane as DPP moves (wave_shl:1 / wave_shr:1): VALU only -- ds_bpermute shuffles go
// through the LDS crossbar, which bounds these kernels otherwise.  A lane without a source lane reads 0
// (bound_ctrl); those lanes are halo lanes whose results are never used.
__device__ __forceinline__ int lane_next_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ int lane_prev_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ float lane_next(float v) { return __int_as_float(lane_next_i(__float_as_int(v))); }
__device__ __forceinline__ float lane_prev(float v) { return __int_as_float(lane_prev_i(__float_as_int(v))); }
// max / OR over lanes [l - 4, l + 4] (valid for lanes 4..59)
__device__ __forceinline__ float hmax9(float c) {
    float r = fmaxf(c, lane_next(c));          // [l, l+1]
    r = fmaxf(c, lane_next(r));                // [l, l+2]
    r = fmaxf(c, lane_next(r));
    r = fmaxf(c, lane_next(r));                // [l, l+4]
    float q = fmaxf(c, lane_prev(c));
    q = fmaxf(c, lane_prev(q));
    q = fmaxf(c, lane_prev(q));
    q = fmaxf(c, lane_prev(q));                // [l-4, l]
    return fmaxf(r, q);
}
// The horizontal 9-max of all NMS_RB rows of a wave tile at once, through a wave-private LDS slice (NMS_RB x NMS_LP floats):
// hmax9 costs nine vector instructions per value, eight of them DPP moves, and the three NMS passes are bound by exactly
// those (0.85-0.95 of the vector issue port busy).  Transposed -- lane (r, h) takes row r, columns 24 h .. 24 h + 39 -- the
// same maxima come out of the register strips the vertical pass uses: 88 max per 32 outputs and lane instead of 288, and the
// transposition itself runs on the LDS port.  max is exact: same values.  o[j] in / out: lane = column, valid for lanes 4..59.
#define NMS_LP 68          // floats per row of the slice: 16-byte reads of 16 consecutive rows cover all banks once
__device__ __forceinline__ void hmax9_rows_lds(float (&o)[NMS_RB], float* __restrict__ slice, int lane) {
    static_assert(NMS_RB == 32, "lane (r, h): 32 rows x two column halves");
#pragma unroll
    for (int j = 0; j < NMS_RB; ++j) slice[j * NMS_LP + lane] = o[j];
    asm volatile("" ::: "memory");                              // (LDS operations of one wave execute in order)
    const int r = lane & 31, c0 = (lane >> 5) * 24;
    float in[40], out[32];
    {
        const float* rp = slice + r * NMS_LP + c0;
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const f32x4 v = *(const f32x4*)(rp + 4 * q);
            in[4 * q] = v[0]; in[4 * q + 1] = v[1]; in[4 * q + 2] = v[2]; in[4 * q + 3] = v[3];
        }
    }
    max9_strip_at<0>(in, out); max9_strip_at<8>(in, out + 8); max9_strip_at<16>(in, out + 16); max9_strip_at<24>(in, out + 24);
    asm volatile("" ::: "memory");                              // every read of the slice is issued before it is overwritten
    {
        float* wp = slice + r * NMS_LP + c0 + 4;                // out[i] is the maximum centred on column c0 + 4 + i
#pragma unroll
        for (int q = 0; q < 8; ++q) *(f32x4*)(wp + 4 * q) = f32x4{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};   // (columns 28..35 twice, same values)
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < NMS_RB; ++j) o[j] = slice[j * NMS_LP + lane];
}
__device__ __forceinline__ unsigned hor9(unsigned c) {
    unsigned r = c | (unsigned)lane_next_i((int)c);
    r = c | (unsigned)lane_next_i((int)r);
    r = c | (unsigned)lane_next_i((int)r);
    r = c | (unsigned)lane_next_i((int)r);
    unsigned q = c | (unsigned)lane_prev_i((int)c);
    q = c | (unsigned)lane_prev_i((int)q);
    q = c | (unsigned)lane_prev_i((int)q);
    q = c | (unsigned)lane_prev_i((int)q);
    return r | q;
}

// A wave tile: the first row / column are wave-uniform, so row addresses are scalar and every load is
//   uniform base + scalar row offset + constant lane offset;
// out-of-image rows / columns are clamped to a valid address and replaced after the load.
struct NmsTile {
    int H, W, gx, cx, gy0, ty, nby, lane;
    bool ok, xin, lane_out;
    long long base;      // first pixel of the image in the score map
    long long wbase;     // first word of the image in the bit-column masks ([row block][column])
};
__device__ __forceinline__ NmsTile nms_tile(const Geom& g) {
    NmsTile t;
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    t.H = lv.H; t.W = lv.W; t.lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int tiles_x = (t.W + NMS_CW - 1) / NMS_CW;
    t.nby = (t.H + NMS_RB - 1) / NMS_RB;
    t.ok = wid < tiles_x * t.nby;
    t.ty = wid / tiles_x;
    const int tx = wid - t.ty * tiles_x;
    t.gx = tx * NMS_CW - 4 + t.lane;
    t.xin = t.gx >= 0 && t.gx < t.W;
    t.lane_out = t.lane >= 4 && t.lane < 4 + NMS_CW && t.xin;
    t.cx = min(max(t.gx, 0), t.W - 1);
    t.gy0 = t.ty * NMS_RB - 4;
    t.base = lv.in_off + (long long)frame * t.H * t.W;
    long long wb = 0;
    for (int l = 0; l < level; ++l) wb += (long long)g.batch * ((g.lv[l].H + NMS_RB - 1) / NMS_RB) * g.lv[l].W;
    t.wbase = wb + (long long)frame * t.nby * t.W;
    return t;
}
// this lane's column of the score map: rows gy0 .. gy0 + 39, -inf outside the image
__device__ __forceinline__ void nms_load_column(const NmsTile& t, const float* __restrict__ src, float (&v)[NMS_RB + 8]) {
    const float NEG = -INFINITY;
    if (t.gy0 >= 0 && t.gy0 + NMS_RB + 8 <= t.H) {                 // interior rows (most tiles): no per-row conditions
        const float* p = src + t.gy0 * t.W + t.cx;
#pragma unroll
        for (int i = 0; i < NMS_RB + 8; ++i) { const float x = p[i * t.W]; v[i] = t.xin ? x : NEG; }
    } else {
#pragma unroll
        for (int i = 0; i < NMS_RB + 8; ++i) {
            const int gy = t.gy0 + i;
            const float x = src[min(max(gy, 0), t.H - 1) * t.W + t.cx];
            v[i] = (gy >= 0 && gy < t.H && t.xin) ? x : NEG;
        }
    }
}
// bits of rows gy0 .. gy0 + 39 of this lane's column (bit i = window row i), 0 outside the image
__device__ __forceinline__ unsigned long long nms_load_bits(const NmsTile& t, const unsigned* __restrict__ words) {
    const unsigned* p = words + t.wbase + t.cx;
    const unsigned cur = p[t.ty * t.W];
    const unsigned prev = t.ty > 0 ? p[(t.ty - 1) * t.W] : 0u;
    const unsigned next = t.ty + 1 < t.nby ? p[(t.ty + 1) * t.W] : 0u;
    const unsigned long long w = ((unsigned long long)prev >> 28) | ((unsigned long long)cur << 4) | ((unsigned long long)next << 36);
    return t.xin ? w : 0ull;
}
__device__ __forceinline__ unsigned nms_row_mask(const NmsTile& t) {     // rows of the tile that are inside the image
    const int left = t.H - t.ty * NMS_RB;
    return left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
}

// pass 1: max_mask = (scores == pool9x9(scores)) as bit columns
__global__ __launch_bounds__(256) void k_nms_mask(const float* __restrict__ dense, unsigned* __restrict__ m0, Geom g) {
    __shared__ __attribute__((aligned(16))) float nms_slice[4][NMS_RB * NMS_LP];      // wave-private: no workgroup barrier anywhere
    const NmsTile t = nms_tile(g);
    if (!t.ok) return;
    float v[NMS_RB + 8];
    nms_load_column(t, dense + t.base, v);
    float o[NMS_RB];
    max9_strip_at<0>(v, o); max9_strip_at<8>(v, o + 8); max9_strip_at<16>(v, o + 16); max9_strip_at<24>(v, o + 24);
    hmax9_rows_lds(o, nms_slice[threadIdx.x >> 6], t.lane);
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < NMS_RB; ++j) bits |= (v[j + 4] == o[j] ? 1u : 0u) << j;
    if (t.lane_out) m0[t.wbase + t.ty * t.W + t.gx] = bits & nms_row_mask(t);
}

// pass 2: supp = dilate9x9(max_mask): vertical on the bits of a column, horizontal across lanes
__global__ __launch_bounds__(256) void k_nms_dilate(const unsigned* __restrict__ m0, unsigned* __restrict__ supp, Geom g) {
    const NmsTile t = nms_tile(g);
    if (!t.ok) return;
    const unsigned long long w = nms_load_bits(t, m0);
    unsigned long long d = w | (w >> 1);
    d |= d >> 2;
    d |= d >> 4;
    d |= w >> 8;                                                    // bit j: window rows j .. j + 8 == tile rows j - 4 .. j + 4
    const unsigned r = hor9((unsigned)d);
    if (t.lane_out) supp[t.wbase + t.ty * t.W + t.gx] = r & nms_row_mask(t);
}

// pass 3: ss = supp ? 0 : scores, pool, select, emit
__global__ __launch_bounds__(256) void k_nms_select(const float* __restrict__ dense, const unsigned* __restrict__ m0, const unsigned* __restrict__ supp,
                                                    float* __restrict__ nms, unsigned long long* __restrict__ cand,
                                                    unsigned int* __restrict__ counters, long long cand_stride, float threshold, Geom g) {
    __shared__ __attribute__((aligned(16))) float nms_slice[4][NMS_RB * NMS_LP];      // wave-private: no workgroup barrier anywhere
    const NmsTile t = nms_tile(g);
    if (!t.ok) return;
    float v[NMS_RB + 8], sc[NMS_RB];
    nms_load_column(t, dense + t.base, v);
    const unsigned m0bits = t.xin ? m0[t.wbase + t.ty * t.W + t.cx] : 0u;
    const unsigned long long sw = nms_load_bits(t, supp);
    const unsigned suppbits = (unsigned)(sw >> 4);
#pragma unroll
    for (int j = 0; j < NMS_RB; ++j) sc[j] = v[j + 4];
#pragma unroll
    for (int i = 0; i < NMS_RB + 8; ++i) v[i] = ((sw >> i) & 1ull) ? 0.0f : v[i];              // (-inf rows have no supp bit)
    float o[NMS_RB];
    max9_strip_at<0>(v, o); max9_strip_at<8>(v, o + 8); max9_strip_at<16>(v, o + 16); max9_strip_at<24>(v, o + 24);
    hmax9_rows_lds(o, nms_slice[threadIdx.x >> 6], t.lane);
    // per-lane bit fields from here on (bit j = tile row j): nothing wave-wide stays alive
    unsigned newmax = 0;
#pragma unroll
    for (int j = 0; j < NMS_RB; ++j) newmax |= (v[j + 4] == o[j] ? 1u : 0u) << j;
    const unsigned live = t.lane_out ? nms_row_mask(t) : 0u;
    const unsigned sel = live & (m0bits | (newmax & ~suppbits));
    unsigned cb = 0;
#pragma unroll
    for (int j = 0; j < NMS_RB; ++j) {
        sc[j] = ((sel >> j) & 1u) ? sc[j] : 0.0f;
        cb |= (sc[j] >= threshold ? 1u : 0u) << j;
    }
    cb &= live;
    if (nms) {                                                      // suppressed map (taps only)
        float* dst = nms + t.base + (t.gy0 + 4) * t.W + t.gx;
#pragma unroll
        for (int j = 0; j < NMS_RB; ++j)
            if ((live >> j) & 1u) dst[j * t.W] = sc[j];
    }
    if (!cand) return;
    // candidates: the order inside the list is free (top-K sorts by key), so a lane's candidates take consecutive
    // slots after those of the lower lanes: one wave scan of the per-lane counts, one global atomic per tile
    const unsigned n = (unsigned)__popc(cb);
    unsigned incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, d, 64);
        if (t.lane >= d) incl += up;
    }
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0) return;                                         // wave-uniform
    unsigned base = 0;
    if (t.lane == 0) base = atomicAdd(&counters[blockIdx.y * HFNET_COUNTER_STRIDE], total);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    unsigned long long* cl = cand + (long long)blockIdx.y * cand_stride + base + (incl - n);
    if (cb) {
        const unsigned idx0 = (unsigned)(t.gx * t.H + t.gy0 + 4);
#pragma unroll
        for (int j = 0; j < NMS_RB; ++j)
            if ((cb >> j) & 1u)
                cl[__popc(cb & ((1u << j) - 1u))] = ((unsigned long long)(~__float_as_uint(sc[j])) << 32) | (idx0 + j);
    }
}

hipError_t launch_nms(const float* dense, float* nms, unsigned* mask0, unsigned* supp, unsigned long long* cand, unsigned int* counters,
                      long long cand_stride, float threshold, const Geom& g, hipStream_t s) {
    int tf = 0;
    for (int l = 0; l < g.n_levels; ++l) tf = max(tf, ((g.lv[l].W + NMS_CW - 1) / NMS_CW) * ((g.lv[l].H + NMS_RB - 1) / NMS_RB));
    const dim3 grid((tf + 3) / 4, g.n_levels * g.batch);
    hipLaunchKernelGGL(k_nms_mask, grid, dim3(256), 0, s, dense, mask0, g);
    hipLaunchKernelGGL(k_nms_dilate, grid, dim3(256), 0, s, mask0, supp, g);
    hipLaunchKernelGGL(k_nms_select, grid, dim3(256), 0, s, dense, mask0, supp, nms, cand, counters, cand_stride, threshold, g);
    return hipGetLastError();
}

// =========================================================================== top-K
// HFNetTFModelV2.cc:144-151.  One 1024-thread workgroup per image.  n <= K: keep everything in scan
// (column-major) order.  n > K: exact radix select of the K smallest 64-bit keys (8 passes of 8 bits),
// then a bitonic sort in LDS -> (response desc, column-major index asc).
#define TOPK_CAP 8192
// the body of k_topk / k_topk_taps: one 1024-thread workgroup per image; returns the number of keypoints written (workgroup-uniform)
__device__ __forceinline__ unsigned int topk_image(const unsigned long long* __restrict__ cand, const unsigned int* __restrict__ counters,
                                                   long long cand_stride, const TopkBudget& kmax_per_level,
                                                   hfnet_keypoint* kps, long long kps_stride, int* __restrict__ n_out, const Geom& g) {
    __shared__ unsigned long long buf[TOPK_CAP];
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sh_prefix;
    __shared__ unsigned int sh_remaining, sh_count;
    const int image = blockIdx.x, level = image / g.batch;
    const int H = g.lv[level].H;
    const unsigned long long* cl = cand + (long long)image * cand_stride;
    const unsigned int n = counters[image * HFNET_COUNTER_STRIDE];
    int K = kmax_per_level.k[level];
    if (K > TOPK_CAP) K = TOPK_CAP;
    if (K > (int)kps_stride) K = (int)kps_stride;              // (never past the image's keypoint slot, whatever the caller's budget)
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
    return m;
}

__global__ __launch_bounds__(1024) void k_topk(const unsigned long long* __restrict__ cand, const unsigned int* __restrict__ counters,
                                               long long cand_stride, TopkBudget kmax_per_level,
                                               hfnet_keypoint* __restrict__ kps, long long kps_stride, int* __restrict__ n_out, Geom g) {
    (void)topk_image(cand, counters, cand_stride, kmax_per_level, kps, kps_stride, n_out, g);
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

// Four tree256_wave sums at once (the four taps of a keypoint), same pairings and therefore the same bits: instead of every
// lane keeping all 16 partial sums through all six butterfly steps (96 cross-lane moves), a lane hands the half of its values
// its partner keeps to the partner and goes on with the other half (a + b == b + a bitwise, so it does not matter which of
// the two lanes of a pair does the addition): 8 + 4 + 2 + 1 moves for the steps 32 / 16 / 8 / 4, then one value per lane for
// the steps 2 / 1 and for (p0 + p2) + (p1 + p3), whose operands sit 8 and 4 lanes apart.  out[t] is wave-uniform.
__device__ __forceinline__ void tree256_wave_x4(const f32x4 (&p)[4], int lane, float (&out)[4]) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    float v8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                     // value index i = 4 tap + j; step 32 splits on the tap's upper bit
        const float lo = p[k >> 2][k & 3], hi = p[2 + (k >> 2)][k & 3];
        v8[k] = (b5 ? hi : lo) + __shfl_xor(b5 ? lo : hi, 32, 64);
    }
    float v4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v4[k] = (b4 ? v8[4 + k] : v8[k]) + __shfl_xor(b4 ? v8[k] : v8[4 + k], 16, 64);     // the tap's lower bit
    float v2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) v2[k] = (b3 ? v4[2 + k] : v4[k]) + __shfl_xor(b3 ? v4[k] : v4[2 + k], 8, 64);      // j's upper bit
    float v = (b2 ? v2[1] : v2[0]) + __shfl_xor(b2 ? v2[0] : v2[1], 4, 64);                                       // j's lower bit
    v = v + __shfl_xor(v, 2, 64);
    v = v + __shfl_xor(v, 1, 64);                     // p[j] of tap (b5, b4), j = (b3, b2), summed over the wave
    v = v + __shfl_xor(v, 8, 64);                     // p0 + p2 | p1 + p3
    v = v + __shfl_xor(v, 4, 64);                     // (p0 + p2) + (p1 + p3)
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * t));
}
// the same for the double-precision sum of cv::normalize: (p0 + p2) + (p1 + p3) of four butterfly sums, in every lane
__device__ __forceinline__ double tree_wave_f64x4(const double (&p)[4], int lane) {
    const bool b5 = lane & 32, b4 = lane & 16;
    double v2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) v2[k] = (b5 ? p[2 + k] : p[k]) + __shfl_xor(b5 ? p[k] : p[2 + k], 32, 64);
    double v = (b4 ? v2[1] : v2[0]) + __shfl_xor(b4 ? v2[0] : v2[1], 16, 64);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    v = v + __shfl_xor(v, 32, 64);                    // p0 + p2 | p1 + p3
    return v + __shfl_xor(v, 16, 64);
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
        const int nl = min(a.n_in[l * g.batch + frame], (int)a.kps_stride);
        if (l < level) base += nl;
        total += nl;
    }
    const int n = min(a.n_in[image], (int)a.kps_stride);        // (a count comes from device memory: never past the image's slot)
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
        if (a.sparse && a.cell_row) {   // de-duplicated taps: the row of every cell is looked up
            const float* d = a.desc_map + ((long long)image * a.kps_stride * 4) * 256 + lane * 4;
            const int* cr = a.cell_row + (long long)image * a.cell_stride;
            // (a row number comes from device memory: kept inside the image's slot whatever it is -- a wrong descriptor is a test failure,
            //  a wild address takes the process down; NOTEBOOK.md R4.8)
            const int rmax = (int)a.kps_stride * 4 - 1;
            // ... and reported: a tap cell without a row means the row list and this keypoint disagree (HFNET_FAULT_SAMPLE_ROW)
            auto row_of = [&](int cell) {
                const int rw = cr[cell];
                if ((rw < 0 || rw > rmax) && a.fault && lane == 0) atomicOr(a.fault, HFNET_FAULT_SAMPLE_ROW);
                return (long long)min(max(rw, 0), rmax) * 256;
            };
            vff = (fxin && fyin) ? *(const f32x4*)(d + row_of(fy * dw + fx)) : zero;
            vcc = (cxin && cyin) ? *(const f32x4*)(d + row_of(cy * dw + cx)) : zero;
            vfc = (fxin && cyin) ? *(const f32x4*)(d + row_of(cy * dw + fx)) : zero;
            vcf = (cxin && fyin) ? *(const f32x4*)(d + row_of(fy * dw + cx)) : zero;
        } else if (a.sparse) {   // rows 4i..4i+3 of the image slot hold the taps (fx,fy) (cx,cy) (fx,cy) (cx,fy)
            const float* d = a.desc_map + (((long long)image * a.kps_stride + i) * 4) * 256 + lane * 4;
            vff = (fxin && fyin) ? *(const f32x4*)(d) : zero;
            vcc = (cxin && cyin) ? *(const f32x4*)(d + 256) : zero;
            vfc = (fxin && cyin) ? *(const f32x4*)(d + 512) : zero;
            vcf = (cxin && fyin) ? *(const f32x4*)(d + 768) : zero;
        }
        if (a.sparse) {
            // the rows come straight from the 1x1 conv: tf.nn.l2_normalize of each (hf_net.py:80) here instead of in a
            // separate pass over them -- same expressions as k_l2norm256 (a skipped tap stays zero)
            f32x4 sq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { sq[0][j] = vff[j] * vff[j]; sq[1][j] = vcc[j] * vcc[j]; sq[2][j] = vfc[j] * vfc[j]; sq[3][j] = vcf[j] * vcf[j]; }
            float ss[4];
            tree256_wave_x4(sq, lane, ss);
            const float i0 = 1.0f / sqrtf(fmaxf(ss[0], 1e-12f)), i1 = 1.0f / sqrtf(fmaxf(ss[1], 1e-12f));
            const float i2 = 1.0f / sqrtf(fmaxf(ss[2], 1e-12f)), i3 = 1.0f / sqrtf(fmaxf(ss[3], 1e-12f));
#pragma unroll
            for (int j = 0; j < 4; ++j) { vff[j] = vff[j] * i0; vcc[j] = vcc[j] * i1; vfc[j] = vfc[j] * i2; vcf[j] = vcf[j] * i3; }
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
    const double ssum = tree_wave_f64x4(p, lane);
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

// ---- distinct tap cells of the selected keypoints (sparse descriptor head).  Keypoints are >= 5 pixels apart (NMS radius
// 4) on an 8-pixel cell grid, so neighbouring keypoints share bilinear taps; real scenes cluster them far more than noise.
// Pass 1 marks the cells any keypoint samples (the same float expressions as k_sample / the gathered conv), pass 2 numbers
// the marked cells of an image in ascending cell order (a spatially sorted row list: good for the conv's L2 reuse too).
// tap `row & 3` of keypoint `row >> 2` of `image`: marks its cell
__device__ __forceinline__ void tap_mark_row(const hfnet_keypoint* kps, int n, long long kps_stride, unsigned char* flags, long long cell_stride,
                                             const Geom& g, int image, int row) {
    const int level = image / g.batch;
    const LevelGeom lv = g.lv[level];                            // H, W: score map; Ho, Wo: cell grid
    const int i = row >> 2, t = row & 3;
    if (i >= min(n, (int)kps_stride)) return;
    const hfnet_keypoint kp = kps[(long long)image * kps_stride + i];
    const int Wc = lv.Wo, Hc = lv.Ho;
    const float sw = ((float)Wc - 1.f) / (float)((float)lv.W - 1.f);
    const float sh = ((float)Hc - 1.f) / (float)((float)lv.H - 1.f);
    const float xf = sw * kp.x, yf = sh * kp.y;
    const int fx = (int)floorf(xf), fy = (int)floorf(yf);
    const int x = fx + ((t == 1 || t == 3) ? 1 : 0), y = fy + ((t == 1 || t == 2) ? 1 : 0);
    if (x >= 0 && x < Wc && y >= 0 && y < Hc) flags[(long long)image * cell_stride + y * Wc + x] = 1;
}
__global__ __launch_bounds__(256) void k_tap_mark(const hfnet_keypoint* __restrict__ kps, const int* __restrict__ n_in, long long kps_stride,
                                                  unsigned char* __restrict__ flags, long long cell_stride, Geom g) {
    tap_mark_row(kps, n_in[blockIdx.y], kps_stride, flags, cell_stride, g, blockIdx.y, blockIdx.x * 256 + threadIdx.x);
}

// The row list of an image holds 4 * kps_stride entries and k_tap_mark sets at most that many flags; the kernel nevertheless bounds
// every row number it writes (a flag array that is not clean -- GPUTEST_r04: a creation-time clear that had not landed yet -- would
// otherwise run the list, and the descriptor head's row buffers after it, past their ends) and reports the overflow.
__device__ __forceinline__ void tap_compact_image(unsigned char* flags, int* __restrict__ cell_row, int* __restrict__ cells,
                                                  int* __restrict__ n_rows, long long cell_stride, long long kps_stride, const Geom& g,
                                                  unsigned int* __restrict__ fault, int image) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int level = image / g.batch;
    const int ncell = g.lv[level].Ho * g.lv[level].Wo;
    unsigned char* f = flags + (long long)image * cell_stride;
    int* cr = cell_row + (long long)image * cell_stride;
    int* cl = cells + (long long)image * kps_stride * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cap = (int)kps_stride * 4;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < ncell; c0 += 1024) {
        const int c = c0 + tid;
        const bool on = c < ncell && f[c] != 0;
        if (c < ncell) f[c] = 0;                                  // (left clean for the next call)
        const unsigned long long mask = __ballot(on);
        const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(mask);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int row = base + woff + prefix;
        const bool fits = row < cap;
        if (c < ncell) cr[c] = on && fits ? row : -1;
        if (on && fits) cl[row] = c;
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int w = 0; w < 16; ++w) tot += wsum[w]; base += tot; }
        __syncthreads();
    }
    if (tid == 0) {
        n_rows[image] = min(base, cap);
        if (base > cap && fault) atomicOr(fault, HFNET_FAULT_TAP_ROWS);
    }
}
__global__ __launch_bounds__(1024) void k_tap_compact(unsigned char* __restrict__ flags, int* __restrict__ cell_row, int* __restrict__ cells,
                                                      int* __restrict__ n_rows, long long cell_stride, long long kps_stride, Geom g,
                                                      unsigned int* __restrict__ fault) {
    tap_compact_image(flags, cell_row, cells, n_rows, cell_stride, kps_stride, g, fault, blockIdx.x);
}
// top-K + the distinct tap cells of the selected keypoints in ONE launch (both are one 1024-thread workgroup per image; asked for since round 3:
// two launches and a dependent hop less per call).  g: H, W = score map, Ho, Wo = cell grid (launch_tap_cells' geometry; top-K reads H only).
__global__ __launch_bounds__(1024) void k_topk_taps(const unsigned long long* __restrict__ cand, const unsigned int* __restrict__ counters,
                                                    long long cand_stride, TopkBudget kmax_per_level, hfnet_keypoint* kps, long long kps_stride,
                                                    int* __restrict__ n_out, unsigned char* flags, int* __restrict__ cell_row, int* __restrict__ cells,
                                                    int* __restrict__ n_rows, long long cell_stride, Geom g, unsigned int* __restrict__ fault) {
    const int m = (int)topk_image(cand, counters, cand_stride, kmax_per_level, kps, kps_stride, n_out, g);
    __threadfence_block();
    __syncthreads();                                              // the keypoints written above are read back by other waves below
    for (int row = threadIdx.x; row < 4 * m; row += 1024) tap_mark_row(kps, m, kps_stride, flags, cell_stride, g, blockIdx.x, row);
    __threadfence_block();
    __syncthreads();
    tap_compact_image(flags, cell_row, cells, n_rows, cell_stride, kps_stride, g, fault, blockIdx.x);
}

hipError_t launch_tap_cells(const hfnet_keypoint* kps, const int* n_in, long long kps_stride, unsigned char* flags, int* cell_row, int* cells,
                            int* n_rows, long long cell_stride, const Geom& g, hipStream_t s, unsigned int* fault) {
    const int images = g.n_levels * g.batch;
    hipLaunchKernelGGL(k_tap_mark, dim3((unsigned)((kps_stride * 4 + 255) / 256), images), dim3(256), 0, s, kps, n_in, kps_stride, flags, cell_stride, g);
    hipLaunchKernelGGL(k_tap_compact, dim3(images), dim3(1024), 0, s, flags, cell_row, cells, n_rows, cell_stride, kps_stride, g, fault);
    return hipGetLastError();
}

hipError_t launch_topk_taps(const unsigned long long* cand, const unsigned int* counters, long long cand_stride, const TopkBudget& kmax_per_level,
                            hfnet_keypoint* kps, long long kps_stride, int* n_out, unsigned char* flags, int* cell_row, int* cells, int* n_rows,
                            long long cell_stride, const Geom& g, hipStream_t s, unsigned int* fault) {
    hipLaunchKernelGGL(k_topk_taps, dim3(g.n_levels * g.batch), dim3(1024), 0, s, cand, counters, cand_stride, kmax_per_level, kps, kps_stride, n_out,
                       flags, cell_row, cells, n_rows, cell_stride, g, fault);
    return hipGetLastError();
}

__global__ void k_bump_seq(int* seq) { *seq = *seq + 1; }
hipError_t launch_bump_seq(int* seq, hipStream_t s) {
    hipLaunchKernelGGL(k_bump_seq, dim3(1), dim3(1), 0, s, seq);
    return hipGetLastError();
}

}  // namespace hfnet
