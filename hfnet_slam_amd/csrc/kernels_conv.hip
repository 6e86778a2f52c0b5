// kernels_conv.hip -- backbone kernels for gfx950 (CDNA4): pyramid resize, stem, 1x1 / 3x3
// convolutions on v_mfma_f32_32x32x2_f32, depthwise 3x3.
//
// Numerics: every accumulation is the fused multiply-add chain the oracle defines
// (oracle/hfnet_oracle.h): BatchNorm is folded into the weights on the host (weights.cpp), every
// accumulator STARTS at the folded bias (the MFMA's C operand) and the f32 MFMA is bit-for-bit a
// k-ordered fmaf chain; the vector kernels use explicit fmaf.  ReLU6 is one v_med3_f32.
// Built with -ffp-contract=off.
//
// Data layout: activations are [pixel][channel] fp32 with the channels of each group of 8 in the
// "physical" order of common.hpp, levels and frames concatenated ([level][frame][y][x][c]).
#include "kernels.hpp"

#include <algorithm>
#include <type_traits>

namespace hfnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float relu6f(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }   // no NaNs on this path

// =========================================================================== pyramid resize
// OpenCV 4.2 cv::resize(INTER_LINEAR) on CV_8UC1: 11-bit fixed-point coefficients, horizontal pass
// to int, vertical pass ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2.  Integer-exact.
// One thread = one output column x 8 consecutive output rows.  Successive output rows mostly share a source row
// (scale 1.2: the lower source row of one output row is the upper one of the next), so the horizontal interpolation
// of a source row is computed once and carried to the next output row: ~2.4 byte loads per output pixel instead of
// 4, and the column tables are read once per thread.  Lanes are consecutive columns (byte-adjacent loads / stores).
#define RESIZE_ROWS 8
__global__ __launch_bounds__(256) void k_resize_u8(const uint8_t* __restrict__ src, int sw, int sh, int s_row, long long s_frame,
                                                   uint8_t* __restrict__ dst, int dw, int dh, int d_row, long long d_frame,
                                                   const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                                   const int* __restrict__ yofs, const short* __restrict__ ibeta) {
    // a thread owns 4 consecutive output columns and RESIZE_ROWS rows and stores packed 32-bit words (destination rows
    // are padded to a multiple of 4 bytes; the columns past dw repeat the last one and land in the padding)
    const int dx4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int dy0 = blockIdx.y * RESIZE_ROWS;
    if (dx4 >= dw) return;
    const uint8_t* sp = src + (long long)blockIdx.z * s_frame;
    unsigned* dp = (unsigned*)(dst + (long long)blockIdx.z * d_frame + dx4);
    int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int dx = min(dx4 + c, dw - 1);
        sx[c] = xofs[dx]; sx1[c] = min(sx[c] + 1, sw - 1);
        a0[c] = ialpha[2 * dx]; a1[c] = ialpha[2 * dx + 1];
    }
    struct H4 { int v[4]; };
    auto hrow = [&](int y) {                                   // horizontal pass of source row y (clamped), 11-bit fixed point
        const uint8_t* rp = sp + (long long)y * s_row;
        H4 h;
#pragma unroll
        for (int c = 0; c < 4; ++c) h.v[c] = rp[sx[c]] * a0[c] + rp[sx1[c]] * a1[c];
        return h;
    };
    int cache_y = -1;
    H4 cache_v = {{0, 0, 0, 0}};                               // horizontal result of the last lower row
#pragma unroll
    for (int i = 0; i < RESIZE_ROWS; ++i) {
        const int dy = dy0 + i;
        if (dy >= dh) break;                                   // uniform
        const int sy = yofs[dy];
        const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        const H4 r0 = y0 == cache_y ? cache_v : hrow(y0);      // uniform condition
        const H4 r1 = y1 == y0 ? r0 : hrow(y1);
        cache_y = y1; cache_v = r1;
        unsigned packed = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int v = (((b0 * (r0.v[c] >> 4)) >> 16) + ((b1 * (r1.v[c] >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            packed |= (unsigned)v << (8 * c);
        }
        dp[((long long)dy * d_row) >> 2] = packed;
    }
}

// ---- the same resize with the source band of a workgroup staged through LDS (calls of many frames).  k_resize_u8 is
// latency-bound: a thread's eight output rows are eight dependent rounds of byte gathers from L2 / HBM (48 us per 128
// frames for 57 MB: 0.15 of the HBM rate).  Here a workgroup owns RESIZE_ROWS output rows over the whole width: the source
// rows they descend from (yofs is non-decreasing: rows clamp(yofs[first]) .. clamp(yofs[last] + 1), ~1.2 x 8 + 2 of them at
// scale 1.2) are read ONCE, as aligned 16-byte pieces issued together -- a piece of a row keeps its offset inside its
// 16-byte line, so that an aligned piece of memory is an aligned piece of LDS; only the bytes [0, sw) of a row are ever
// read: the two partial pieces at a row's ends go byte by byte -- and the gathers come out of LDS.  Per pixel the arithmetic
// is k_resize_u8's.  `pitch` (bytes per band row, a multiple of 16, >= sw + 30) and `cap_rows` (the largest band of the
// level) size the LDS block; both come from the host, which has the tables.
__global__ __launch_bounds__(256) void k_resize_u8_band(const uint8_t* __restrict__ src, int sw, int sh, int s_row, long long s_frame,
                                                        uint8_t* __restrict__ dst, int dw, int dh, int d_row, long long d_frame,
                                                        const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                                        const int* __restrict__ yofs, const short* __restrict__ ibeta, int pitch, int cap_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char band[];
    const int tid = threadIdx.x;
    const int dx4 = tid * 4;                                   // (dw <= 1024: launch check)
    const bool active = dx4 < dw;
    const int dy0 = blockIdx.x * RESIZE_ROWS, dyl = min(dy0 + RESIZE_ROWS, dh) - 1;
    const uint8_t* sp = src + (long long)blockIdx.y * s_frame;
    // column tables first: their latency passes behind the band's
    int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int dx = min(dx4 + c, dw - 1);
        sx[c] = xofs[dx]; sx1[c] = min(sx[c] + 1, sw - 1);
        a0[c] = ialpha[2 * dx]; a1[c] = ialpha[2 * dx + 1];
    }
    // ... and the row tables of all RESIZE_ROWS rows (k_resize_u8 fetches them row by row: eight dependent round trips to L2 per
    // workgroup, which -- not the gathers -- is most of its time); ibeta's pair of shorts as one 4-byte load
    int syv[RESIZE_ROWS], bwv[RESIZE_ROWS];
#pragma unroll
    for (int i = 0; i < RESIZE_ROWS; ++i) {
        const int dyc = min(dy0 + i, dh - 1);
        syv[i] = yofs[dyc];
        bwv[i] = *(const int*)(ibeta + 2 * dyc);
    }
    const int ylo = min(max(syv[0], 0), sh - 1), yhi = min(max(yofs[dyl] + 1, 0), sh - 1);
    const int nrows = min(yhi - ylo + 1, cap_rows);            // (the capacity is the level's largest band: this never cuts)
    const int CH = pitch >> 4;                                 // 16-byte pieces per band row
    const int total = nrows * CH;
    for (int it0 = 0; it0 < total; it0 += 4 * 256) {           // four pieces per thread in flight (a band is ~600 pieces: one pass)
        const uint8_t* rp[4];
        int b0[4], lo[4];
        bool whole[4];
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                           // straight-line loads: every thread reads SOME whole piece of its row
            const int it = it0 + u * 256 + tid;
            const int r = it / CH, c = it - r * CH;
            rp[u] = sp + (long long)(ylo + min(r, nrows - 1)) * s_row;
            const int mis = (int)((size_t)rp[u] & 15);
            b0[u] = c * 16 - mis;                               // row byte of the piece's first byte; rp + b0 is 16-byte aligned
            lo[u] = r * pitch + c * 16;
            const int fw = (16 - mis) & 15, lw = fw + ((sw - fw) & ~15) - 16;      // first / last whole piece of the row (sw >= 32: launch check)
            const int bc = min(max(b0[u], fw), lw);
            whole[u] = it < total && bc == b0[u];
            v[u] = *(const uint4*)(rp[u] + bc);
            if (it >= total) b0[u] = sw;                        // (no piece)
        }
        // (the values are "used" here, all at once: left alone the optimiser sinks each load into the branch that stores it,
        //  where it is waited for on the spot -- four dependent round trips instead of one)
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (whole[u]) *(uint4*)(band + lo[u]) = v[u];
            else if (b0[u] < sw && b0[u] > -16) {              // the partial pieces at a row's two ends: 16 clamped byte loads in flight,
                unsigned char t[16];                            // the bytes inside [0, sw) kept
#pragma unroll
                for (int b = 0; b < 16; ++b) t[b] = rp[u][min(max(b0[u] + b, 0), sw - 1)];
#pragma unroll
                for (int b = 0; b < 16; ++b)
                    if (b0[u] + b >= 0 && b0[u] + b < sw) band[lo[u] + b] = t[b];
            }
        }
    }
    __syncthreads();
    if (!active) return;
    unsigned* dp = (unsigned*)(dst + (long long)blockIdx.y * d_frame + dx4);
    struct H4 { int v[4]; };
    auto hrow = [&](int y) {                                   // horizontal pass of source row y, out of the band
        const unsigned char* lp = band + (y - ylo) * pitch + (int)((size_t)(sp + (long long)y * s_row) & 15);
        H4 h;
#pragma unroll
        for (int c = 0; c < 4; ++c) h.v[c] = lp[sx[c]] * a0[c] + lp[sx1[c]] * a1[c];
        return h;
    };
    int cache_y = -1;
    H4 cache_v = {{0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < RESIZE_ROWS; ++i) {
        const int dy = dy0 + i;
        if (dy >= dh) break;                                   // uniform
        const int sy = syv[i];
        const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
        const int b0 = (short)(bwv[i] & 0xffff), b1 = bwv[i] >> 16;
        const H4 r0 = y0 == cache_y ? cache_v : hrow(y0);      // uniform condition
        const H4 r1 = y1 == y0 ? r0 : hrow(y1);
        cache_y = y1; cache_v = r1;
        unsigned packed = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int v = (((b0 * (r0.v[c] >> 4)) >> 16) + ((b1 * (r1.v[c] >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            packed |= (unsigned)v << (8 * c);
        }
        dp[((long long)dy * d_row) >> 2] = packed;
    }
}

// largest number of source rows a band of RESIZE_ROWS output rows descends from (host table of the level)
int resize_band_rows(const int* yofs, int dh, int sh) {
    int cap = 1;
    for (int dy0 = 0; dy0 < dh; dy0 += RESIZE_ROWS) {
        const int dyl = std::min(dy0 + RESIZE_ROWS, dh) - 1;
        const int ylo = std::min(std::max(yofs[dy0], 0), sh - 1), yhi = std::min(std::max(yofs[dyl] + 1, 0), sh - 1);
        cap = std::max(cap, yhi - ylo + 1);
    }
    return cap;
}

hipError_t launch_resize_u8(const uint8_t* src, int sw, int sh, int s_row, long long s_frame, uint8_t* dst, int dw, int dh,
                            int d_row, long long d_frame, const int* xofs, const short* ialpha, const int* yofs,
                            const short* ibeta, int batch, hipStream_t s, int band_rows) {
    if ((d_row & 3) || (d_frame & 3) || ((size_t)dst & 3)) return hipErrorInvalidValue;      // packed 32-bit stores
    if (band_rows > 0 && dw <= 1024 && sw >= 32) {
        const int pitch = ((sw + 30) / 16 + 1) * 16;
        const size_t lds = (size_t)band_rows * pitch;
        if (lds <= 48 * 1024) {
            hipLaunchKernelGGL(k_resize_u8_band, dim3((dh + RESIZE_ROWS - 1) / RESIZE_ROWS, batch), dim3(256), lds, s, src, sw, sh, s_row, s_frame,
                               dst, dw, dh, d_row, d_frame, xofs, ialpha, yofs, ibeta, pitch, band_rows);
            return hipGetLastError();
        }
    }
    dim3 grid(((dw + 3) / 4 + 255) / 256, (dh + RESIZE_ROWS - 1) / RESIZE_ROWS, batch);
    hipLaunchKernelGGL(k_resize_u8, grid, dim3(256), 0, s, src, sw, sh, s_row, s_frame, dst, dw, dh, d_row, d_frame, xofs, ialpha, yofs, ibeta);
    return hipGetLastError();
}

// ---- the whole pyramid chain in ONE launch (calls of a few frames, where three dependent 7 us launches are 3 % of the
// call).  A workgroup owns PYR_BAND rows of the LAST level and everything above them: it works out which rows of each
// intermediate level those rows descend from (the same yofs tables), computes them level by level from the input image --
// intermediate levels stay in LDS as the next level's source -- and writes every row it computed (rows at band borders are
// written by two workgroups with the same bytes).  Bands tile every level without gaps while a row step of the tables
// is at most 2 (scale < 2).  Per pixel the arithmetic is k_resize_u8's.
#define PYR_BAND 4
struct PyrArgs {
    const uint8_t* src; int s_row; long long s_frame;
    int n;                                            // transitions (levels 1 .. n are produced)
    int w[4], h[4];                                   // sizes of levels 0 .. n
    uint8_t* dst[4]; int d_row[4]; long long d_frame[4];          // [l]: level l, l = 1 .. n
    const int* xofs[4]; const short* ialpha[4]; const int* yofs[4]; const short* ibeta[4];
    int lds_off[4], lds_rows[4];                      // [l]: byte offset / row capacity of level l's band in LDS (l = 1 .. n - 1)
};
__global__ __launch_bounds__(1024) void k_pyramid_chain(PyrArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pyr_lds[];
    __shared__ int lo[4], hi[4];
    const int band = blockIdx.x, frame = blockIdx.y, n = a.n;
    if (threadIdx.x == 0) {
        const int last = (int)gridDim.x - 1;
        lo[n] = band * PYR_BAND; hi[n] = min(lo[n] + PYR_BAND, a.h[n]) - 1;
        for (int l = n; l >= 2; --l) {
            int l0 = min(max(a.yofs[l][lo[l]], 0), a.h[l - 1] - 1), h0 = min(max(a.yofs[l][hi[l]] + 1, 0), a.h[l - 1] - 1);
            if (band == 0) l0 = 0;
            if (band == last) h0 = a.h[l - 1] - 1;
            lo[l - 1] = l0; hi[l - 1] = min(h0, l0 + a.lds_rows[l - 1] - 1);   // (the capacity is sized so that this never cuts)
        }
    }
    __syncthreads();
    const uint8_t* sp0 = a.src + (long long)frame * a.s_frame;
    for (int l = 1; l <= n; ++l) {
        const int sw = a.w[l - 1], sh = a.h[l - 1], dw = a.w[l];
        const uint8_t* sp = l == 1 ? sp0 : pyr_lds + a.lds_off[l - 1];
        const int srow = l == 1 ? a.s_row : sw, sbase = l == 1 ? 0 : lo[l - 1];
        uint8_t* keep = l < n ? pyr_lds + a.lds_off[l] : nullptr;
        uint8_t* gp = a.dst[l] + (long long)frame * a.d_frame[l];
        const int r0 = lo[l], r1 = hi[l], cols = a.d_row[l];
#pragma unroll 2
        for (int idx = threadIdx.x; idx < (r1 - r0 + 1) * cols; idx += 1024) {
            const int ry = idx / cols, dxp = idx - ry * cols, dy = r0 + ry, dx = min(dxp, dw - 1);
            const int sy = a.yofs[l][dy];
            const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
            const int b0 = a.ibeta[l][2 * dy], b1 = a.ibeta[l][2 * dy + 1];
            const int sx = a.xofs[l][dx], sx1 = min(sx + 1, sw - 1);
            const int a0 = a.ialpha[l][2 * dx], a1 = a.ialpha[l][2 * dx + 1];
            const uint8_t* p0 = sp + (long long)(y0 - sbase) * srow;
            const uint8_t* p1 = sp + (long long)(y1 - sbase) * srow;
            const int h0 = p0[sx] * a0 + p0[sx1] * a1, h1 = p1[sx] * a0 + p1[sx1] * a1;
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            gp[(long long)dy * cols + dxp] = (uint8_t)v;                   // (columns past dw repeat the last one: the row padding)
            if (keep && dxp < dw) keep[ry * dw + dxp] = (uint8_t)v;
        }
        __syncthreads();
    }
}

// sizes[l] = (w, h) of level l = 0 .. n; false when the chain does not fit (more than three transitions, a row step above 2, LDS)
bool pyramid_chain_supported(int n, const int* w, const int* h) {
    if (n < 1 || n > 3) return false;
    for (int l = 1; l <= n; ++l) if (h[l - 1] >= 2 * h[l] || w[l] < 1) return false;
    return true;
}
hipError_t launch_pyramid_chain(const uint8_t* src, int s_row, long long s_frame, int n, const int* w, const int* h, uint8_t* const* dst,
                                const int* d_row, const long long* d_frame, const int* const* xofs, const short* const* ialpha,
                                const int* const* yofs, const short* const* ibeta, int batch, hipStream_t s) {
    if (!pyramid_chain_supported(n, w, h)) return hipErrorInvalidValue;
    PyrArgs a;
    a.src = src; a.s_row = s_row; a.s_frame = s_frame; a.n = n;
    for (int l = 0; l <= n; ++l) { a.w[l] = w[l]; a.h[l] = h[l]; }
    size_t lds = 0;
    int rows = PYR_BAND;
    for (int l = n; l >= 1; --l) {
        a.dst[l] = dst[l]; a.d_row[l] = d_row[l]; a.d_frame[l] = d_frame[l];
        a.xofs[l] = xofs[l]; a.ialpha[l] = ialpha[l]; a.yofs[l] = yofs[l]; a.ibeta[l] = ibeta[l];
        a.lds_off[l] = 0; a.lds_rows[l] = 0;
        if (l < n) {                                                           // rows of level l that `rows` rows of level l + 1 can need
            rows = (int)((long long)rows * h[l] / h[l + 1]) + 6;
            a.lds_rows[l] = rows; a.lds_off[l] = (int)lds;
            lds += ((size_t)rows * w[l] + 15) / 16 * 16;
        }
    }
    if (lds > 60 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pyramid_chain, dim3((h[n] + PYR_BAND - 1) / PYR_BAND, batch), dim3(1024), lds, s, a);
    return hipGetLastError();
}

// =========================================================================== stem
// u8 -> (x-128)/128 -> crop to multiples of 8 (Geom carries the cropped size) -> conv 3x3 stride 2
// 1 -> cout, BN, ReLU6.  One thread per output pixel, weights in LDS.  HBM-bound on the output write.
__global__ __launch_bounds__(256) void k_stem(ImageSet imgs, const float* __restrict__ w, const float* __restrict__ bias, int cout,
                                              float* __restrict__ out, Geom g) {
    __shared__ float sw[9 * 64];
    __shared__ float ssh[64];
    for (int i = threadIdx.x; i < 9 * cout; i += 256) sw[i] = w[i];
    if ((int)threadIdx.x < cout) ssh[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= lv.Ho * lv.Wo) return;
    const int oy = idx / lv.Wo, ox = idx - oy * lv.Wo;
    const uint8_t* img = imgs.ptr[level] + (long long)frame * imgs.frame_stride[level];
    const int rs = imgs.row_stride[level];
    float px[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 - lv.pt + ky, ix = ox * 2 - lv.pl + kx;
            float v = 0.0f;
            if (iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W) v = ((float)img[(long long)iy * rs + ix] - 128.0f) * 0.0078125f;
            px[ky * 3 + kx] = v;
        }
    float* op = out + (lv.out_off + (long long)frame * lv.Ho * lv.Wo + idx) * cout;
    for (int c = 0; c < cout; c += 4) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = ssh[c + j];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(px[t], sw[t * cout + c + j], acc);
            r[j] = relu6f(acc);
        }
        *(f32x4*)(op + c) = r;
    }
}

// Channel count known at compile time: the weights are wave-uniform loads with constant offsets (scalar cache, SGPR
// fma operands, no LDS reads), and the workgroup's 256 pixels x COUT channels -- one contiguous block of the output
// tensor -- go through LDS so that the global stores are 16 bytes per lane at consecutive addresses (a thread's own
// pixel is COUT*4 bytes from its neighbour's: written directly, every store instruction touches 64 cache lines).
template <int COUT>
__global__ __launch_bounds__(256) void k_stem_c(ImageSet imgs, const float* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ out, Geom g) {
    constexpr int PS = COUT + 4;                 // LDS pixel stride in floats: 16-byte aligned, 28 words -> conflict-free b128
    static_assert(COUT % 4 == 0 && (PS % 8) == 4, "LDS pixel stride");
    __shared__ __attribute__((aligned(16))) float tile[256 * PS];
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int npix = lv.Ho * lv.Wo;
    const int idx0 = blockIdx.x * 256;
    if (idx0 >= npix) return;
    const int idx = min(idx0 + (int)threadIdx.x, npix - 1);
    const int oy = idx / lv.Wo, ox = idx - oy * lv.Wo;
    const uint8_t* img = imgs.ptr[level] + (long long)frame * imgs.frame_stride[level];
    const int rs = imgs.row_stride[level];
    float px[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 - lv.pt + ky, ix = ox * 2 - lv.pl + kx;
            const bool ok = iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W;
            const float raw = (float)img[(long long)(ok ? iy : 0) * rs + (ok ? ix : 0)];
            px[ky * 3 + kx] = ok ? (raw - 128.0f) * 0.0078125f : 0.0f;
        }
#pragma unroll
    for (int c = 0; c < COUT; c += 4) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = bias[c + j];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(px[t], w[t * COUT + c + j], acc);
            r[j] = relu6f(acc);
        }
        *(f32x4*)(tile + threadIdx.x * PS + c) = r;
    }
    __syncthreads();
    const int nvalid = min(256, npix - idx0);
    f32x4* op = (f32x4*)(out + (lv.out_off + (long long)frame * npix + idx0) * COUT);
#pragma unroll
    for (int k = 0; k < COUT / 4; ++k) {
        const int q = threadIdx.x + k * 256;         // 16-byte piece of the block
        const int p = q / (COUT / 4), part = q - p * (COUT / 4);
        if (p < nvalid) op[q] = *(const f32x4*)(tile + p * PS + part * 4);
    }
}

hipError_t launch_stem(const ImageSet& imgs, const float* w, const float* bias, int cout, float* out, const Geom& g, hipStream_t s) {
    int maxpix = 0;
    for (int l = 0; l < g.n_levels; ++l) maxpix = max(maxpix, g.lv[l].Ho * g.lv[l].Wo);
    dim3 grid((maxpix + 255) / 256, g.n_levels * g.batch);
    if (cout == 24) hipLaunchKernelGGL(k_stem_c<24>, grid, dim3(256), 0, s, imgs, w, bias, out, g);
    else hipLaunchKernelGGL(k_stem, grid, dim3(256), 0, s, imgs, w, bias, cout, out, g);
    return hipGetLastError();
}

// =========================================================================== MFMA convolutions
// One wave owns a 32-pixel x (NT*32)-channel output tile; a workgroup is 4 waves = 128 pixels.
// v_mfma_f32_32x32x2_f32: lane l supplies A[row = l & 31][k = l >> 5] and B[k = l >> 5][col = l & 31];
// D[row][col]: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
// Per 8 input channels: one 16-byte A load per lane (lanes 0-31 physical slots 0-3, lanes 32-63 slots
// 4-7 of the pixel), one 16-byte load of pre-packed B per lane per column tile, then 4 MFMAs per column
// tile whose k-pairs are logical channels (0,1) (2,3) (4,5) (6,7) -- the oracle's order.
struct ConvArgs {
    const float* A;
    const f32x4* W;
    const float* bias;    // folded BatchNorm shift / layer bias: the accumulators start here ([nt_total*32], zero padded)
    const float* res;
    float* out;
    long long P;      // rows (pointwise) -- unused by the 3x3 kernel
    int cin;
    int n;            // valid output columns == output row stride
    int nt_total;
    int relu6;
    int level_tiles[HFNET_MAX_LEVELS];   // 3x3 kernel: 128-row tiles launched per image of each level (exact 1-D grid)
    // pointwise on slotted rows (the tap rows of the sparse descriptor head: slot_rows rows per image, of which the first
    // slot_units[image] * rows_per_unit are in use): 32-row tiles that lie in the unused part of a slot are skipped
    const int* slot_units;
    int slot_rows, rows_per_unit;
};

// does the 32-row tile starting at row0 touch a used row? (wave-uniform)
__device__ __forceinline__ bool tile_in_use(const ConvArgs& a, long long row0, int rows = 32) {
    if (!a.slot_units) return true;
    const long long last = min(row0 + rows, a.P) - 1;
    const int i0 = (int)(row0 / a.slot_rows), i1 = (int)(last / a.slot_rows);
    if ((int)(row0 - (long long)i0 * a.slot_rows) < a.slot_units[i0] * a.rows_per_unit) return true;
    for (int i = i0 + 1; i <= i1; ++i)
        if (a.slot_units[i] > 0) return true;
    return false;
}

// (ReLU6) (+ residual) and store of a wave's 32 x (NT*32) accumulator tile (BatchNorm: folded weights, the accumulators
// started at the folded bias -- conv_acc_init).  VALU instructions compete
// with the f32 MFMAs for the same pipe, so the per-element work is kept minimal: the flags are hoisted into four
// specialised loops, addresses are a uniform 64-bit tile base plus 32-bit lane offsets (row stride multiples are
// scalar), and only the last, partial row tile checks rows.
template <int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[NT], int nt0, long long row_base, long long row_limit,
                                              int half, int r) {
    // the tile's first row is the same for all lanes of the wave: say so, the bases then live in SGPRs
    row_base = ((long long)__builtin_amdgcn_readfirstlane((int)(row_base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)row_base);
    const long long left = row_limit - row_base;
    if (left <= 0) return;
    const int rows = left < 32 ? (int)left : 32;                      // uniform
    float* __restrict__ obase = a.out + row_base * a.n;               // uniform
    const float* __restrict__ rbase = a.res ? a.res + row_base * a.n : nullptr;
    const unsigned n = (unsigned)a.n;
    auto body = [&](auto relu_tag, auto res_tag, auto full_tag) {
        constexpr bool RELU = decltype(relu_tag)::value, RES = decltype(res_tag)::value, FULL = decltype(full_tag)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned col = (unsigned)((nt0 + nt) * 32 + r);
            if (col < n) {
                const unsigned o0 = ((unsigned)(4 * half) * n + col) * 4u;   // byte offsets inside the tile (< 2^32)
                float rv[16];
                if (RES) {                                              // all residual loads first: one latency, not sixteen
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rr = (reg & 3) + 8 * (reg >> 2);
                        rv[reg] = (FULL || rr + 4 * half < rows) ? *(const float*)((const char*)rbase + o0 + (unsigned)rr * n * 4u) : 0.0f;
                    }
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rr = (reg & 3) + 8 * (reg >> 2);          // + 4 * half
                    if (FULL || rr + 4 * half < rows) {
                        const unsigned off = o0 + (unsigned)rr * n * 4u;
                        float v = acc[nt][reg];
                        if (RELU) v = relu6f(v);
                        if (RES) v = v + rv[reg];
                        *(float*)((char*)obase + off) = v;
                    }
                }
            }
        }
    };
    using T = std::true_type; using F = std::false_type;
    const bool full = rows == 32;
    if (a.relu6) {
        if (a.res) { if (full) body(T{}, T{}, T{}); else body(T{}, T{}, F{}); }
        else       { if (full) body(T{}, F{}, T{}); else body(T{}, F{}, F{}); }
    } else {
        if (a.res) { if (full) body(F{}, T{}, T{}); else body(F{}, T{}, F{}); }
        else       { if (full) body(F{}, F{}, T{}); else body(F{}, F{}, F{}); }
    }
}

// accumulators start at the folded bias of the lane's output column (D[row][col]: col = lane & 31)
template <int NT>
__device__ __forceinline__ void conv_acc_init(const ConvArgs& a, f32x16 (&acc)[NT], int nt0, int r) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float b = a.bias[(nt0 + nt) * 32 + r];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = b;
    }
}

template <int NT>
__global__ __launch_bounds__(256, 2) void k_pointwise(ConvArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    const long long row0 = (long long)blockIdx.x * 128 + wave * 32;
    if (row0 >= a.P || !tile_in_use(a, row0)) return;
    const int nt0 = blockIdx.y * NT;
    long long row = row0 + r;
    if (row >= a.P) row = a.P - 1;
    const float* ap = a.A + row * a.cin + half * 4;
    const f32x4* wp = a.W + ((size_t)nt0 * 64 + lane);
    const size_t wstep = (size_t)a.nt_total * 64;
    f32x16 acc[NT];
    conv_acc_init<NT>(a, acc, nt0, r);
    const int KQ = a.cin >> 3;
    // software pipeline: the loads of step kq+1 are issued before the MFMAs of step kq.  (A three-buffer, distance-2
    // pipeline as in k_conv3x3 was measured slower here: these kernels have short k loops and live on occupancy.)
    f32x4 av = *(const f32x4*)(ap);
    f32x4 bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = wp[(size_t)nt * 64];
    for (int kq = 0; kq < KQ; ++kq) {
        f32x4 av_n = av;
        f32x4 bv_n[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv_n[nt] = bv[nt];
        if (kq + 1 < KQ) {
            av_n = *(const f32x4*)(ap + (kq + 1) * 8);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv_n[nt] = wp[(size_t)(kq + 1) * wstep + (size_t)nt * 64];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nt][t], acc[nt], 0, 0, 0);
        av = av_n;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = bv_n[nt];
    }
    conv_epilogue<NT>(a, acc, nt0, row0, a.P, half, r);
}

// GEMM-shaped 1x1 convolutions with more weights than L1 holds (256 -> 256 on the descriptor head's tap rows: 256 KB):
// every wave of k_pointwise fetches its own copy of the weight fragments from L2, 4 KB per k-step and wave, and the
// ~10 bytes per clock a CU gets from L2 bound the kernel at half the MFMA rate.  Here the workgroup stages the weight
// slab of four k-steps (NT x 4 KB) through LDS once -- one coalesced 16-byte load per thread and k-step, double buffered,
// one barrier per slab -- and its four waves read their fragments from there; the activation rows (a wave's own) still
// come straight from memory one k-step ahead.  Same chains, same bits.
template <int NT>
__global__ __launch_bounds__(256, 2) void k_pointwise_wlds(ConvArgs a) {
    constexpr int KS = 4;                                     // k-steps (of 8 channels) per slab
    __shared__ __attribute__((aligned(16))) f32x4 wl[2][KS][NT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    const long long row0 = (long long)blockIdx.x * 128 + wave * 32;
    const int nt0 = blockIdx.y * NT;
    if (!tile_in_use(a, (long long)blockIdx.x * 128, 128)) return;   // (workgroup-uniform: before any barrier)
    const bool active = row0 < a.P && tile_in_use(a, row0);   // (waves without rows still help staging)
    const float* ap = a.A + min(row0 + r, a.P - 1) * a.cin + half * 4;
    const int KQ = a.cin >> 3, n_slabs = (KQ + KS - 1) / KS;
    const size_t wstep = (size_t)a.nt_total * 64;
    // staging role: thread -> (column tile, lane) of one k-step's NT KB (NT * 64 pieces of 16 bytes; 256 threads)
    constexpr int PIECES = NT * 64, PER_T = (PIECES + 255) / 256;
    auto fetch = [&](int slab, f32x4 (&st)[KS][PER_T]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kq = min(slab * KS + ks, KQ - 1);
#pragma unroll
            for (int j = 0; j < PER_T; ++j) {
                const int piece = min(tid + j * 256, PIECES - 1);
                st[ks][j] = a.W[(size_t)kq * wstep + (size_t)nt0 * 64 + piece];
            }
        }
    };
    auto stash = [&](int buf, const f32x4 (&st)[KS][PER_T]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < PER_T; ++j)
                if (tid + j * 256 < PIECES) (&wl[buf][ks][0][0])[tid + j * 256] = st[ks][j];
    };
    f32x16 acc[NT];
    conv_acc_init<NT>(a, acc, nt0, r);
    f32x4 st[KS][PER_T];
    fetch(0, st);
    stash(0, st);
    __syncthreads();
    f32x4 av = *(const f32x4*)(ap);
    for (int slab = 0; slab < n_slabs; ++slab) {
        const int buf = slab & 1;
        if (slab + 1 < n_slabs) fetch(slab + 1, st);           // next slab: in flight while this one is multiplied
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kq = slab * KS + ks;
            if (kq < KQ) {                                    // uniform
                const f32x4 an = *(const f32x4*)(ap + min(kq + 1, KQ - 1) * 8);
                f32x4 bv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = wl[buf][ks][nt][lane];
                if (active) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nt][t], acc[nt], 0, 0, 0);
                }
                av = an;
            }
        }
        if (slab + 1 < n_slabs) stash(buf ^ 1, st);
        __syncthreads();
    }
    if (active) conv_epilogue<NT>(a, acc, nt0, row0, a.P, half, r);
}

// ---- detector tail in one launch: 1x1 conv 128 -> 65 (+bias) -> softmax over the 65 logits -> drop the dustbin -> depth_to_space
// (hf_net.py:84-93).  The GEMM is k_pointwise_wlds<3>'s loop (same chains, same bits); the wave's 32 x 65 logits then go through
// LDS instead of HBM (234 MB written and read back per 64 frames), two lanes per cell take its softmax -- exponentials of the
// lower / upper 32 channels, the left-to-right sum handed from the lower lane to the upper one (k_softmax_d2s' expressions and
// order) -- and store rows 0-3 / 4-7 of the cell's 8 x 8 pixel block.
__device__ __forceinline__ float hf_expf_c(float x) {         // == oracle hfo_expf (as in kernels_detect.hip)
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

__global__ __launch_bounds__(256, 2) void k_det_tail(ConvArgs a, float* __restrict__ dense, Geom g) {
    // 65 columns = two 32-column MFMA tiles + the dustbin.  A third tile would spend 64 MFMAs per 32 cells (a third of the
    // kernel's matrix-core time) on one useful column: the dustbin's chain -- bias, then fma in ascending channel order,
    // exactly what the matrix core does for a column -- runs on the vector ALU instead, 8 fma + 4 half-wave swaps per
    // k-step (a lane's A fragment holds the even channels of its cell, its partner's the odd ones), in both lanes of a pair.
    constexpr int NT = 2, NTW = 3, KS = 4, LP = 65;
    __shared__ __attribute__((aligned(16))) f32x4 wl[2][KS][NTW][64];
    __shared__ float lg[4][32 * LP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];                           // H, W: cell grid; Ho, Wo: dense map (8H, 8W)
    const int ncells = lv.H * lv.W;
    if ((int)blockIdx.x * 128 >= ncells) return;                // workgroup-uniform
    const int c0 = blockIdx.x * 128 + wave * 32;
    const bool active = c0 < ncells;                            // (waves without cells still help staging)
    const long long row_base = lv.in_off + (long long)frame * ncells;
    const float* ap = a.A + (row_base + min(c0 + r, ncells - 1)) * a.cin + half * 4;
    const int KQ = a.cin >> 3, n_slabs = (KQ + KS - 1) / KS;
    const size_t wstep = (size_t)a.nt_total * 64;
    constexpr int PIECES = NTW * 64, PER_T = (PIECES + 255) / 256;
    auto fetch = [&](int slab, f32x4 (&st)[KS][PER_T]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kq = min(slab * KS + ks, KQ - 1);
#pragma unroll
            for (int j = 0; j < PER_T; ++j) st[ks][j] = a.W[(size_t)kq * wstep + min(tid + j * 256, PIECES - 1)];
        }
    };
    auto stash = [&](int buf, const f32x4 (&st)[KS][PER_T]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < PER_T; ++j)
                if (tid + j * 256 < PIECES) (&wl[buf][ks][0][0])[tid + j * 256] = st[ks][j];
    };
    f32x16 acc[NT];
    conv_acc_init<NT>(a, acc, 0, r);
    float dust = a.bias[64];
    f32x4 st[KS][PER_T];
    fetch(0, st);
    stash(0, st);
    __syncthreads();
    f32x4 av = *(const f32x4*)(ap);
    for (int slab = 0; slab < n_slabs; ++slab) {
        const int buf = slab & 1;
        if (slab + 1 < n_slabs) fetch(slab + 1, st);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kq = slab * KS + ks;
            if (kq < KQ) {                                    // uniform
                const f32x4 an = *(const f32x4*)(ap + min(kq + 1, KQ - 1) * 8);
                f32x4 bv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = wl[buf][ks][nt][lane];
                // column 64's weights: the B fragment of the third tile's lanes 0 (even channels) and 32 (odd channels)
                const f32x4 w_even = wl[buf][ks][2][0], w_odd = wl[buf][ks][2][32];
                if (active) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nt][t], acc[nt], 0, 0, 0);
                        const int x = __float_as_int(av[t]);
                        const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // [0]: the lower half's value, [1]: the upper half's, in both
                        dust = fmaf(__int_as_float(sw[0]), w_even[t], dust);
                        dust = fmaf(__int_as_float(sw[1]), w_odd[t], dust);
                    }
                }
                av = an;
            }
        }
        if (slab + 1 < n_slabs) stash(buf ^ 1, st);
        __syncthreads();
    }
    if (!active) return;                                        // (no barrier below: the logits of a wave stay in its own slice)
    // D fragment of the 32x32 tile: column nt * 32 + r, rows (reg & 3) + 8 (reg >> 2) + 4 half
    float* L = lg[wave];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = nt * 32 + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) L[((reg & 3) + 8 * (reg >> 2) + 4 * half) * LP + col] = acc[nt][reg];
    }
    if (!half) L[r * LP + 64] = dust;
    // lane (r, half): cell c0 + r, channels 32 half .. 32 half + 31 (+ the dustbin, channel 64, with the upper half)
    const float* row = L + r * LP + 32 * half;
    float e[33];
#pragma unroll
    for (int k = 0; k < 32; ++k) e[k] = row[k];
    e[32] = half ? row[32] : e[31];                             // (lower half: a copy that changes no maximum and is never summed)
    float mx = e[0];
#pragma unroll
    for (int k = 1; k < 33; ++k) mx = fmaxf(mx, e[k]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
    for (int k = 0; k < 33; ++k) e[k] = hf_expf_c(e[k] - mx);
    // sum = (((0 + e0) + e1) + ... + e64): the lower lane's 32 terms first, then the upper lane continues
    float sum = 0.0f;
    if (!half) {
#pragma unroll
        for (int k = 0; k < 32; ++k) sum = sum + e[k];
    }
    sum = __shfl(sum, r, 64);
    if (half) {
#pragma unroll
        for (int k = 0; k < 33; ++k) sum = sum + e[k];
    }
    sum = __shfl(sum, r + 32, 64);
    const int cell = c0 + r;
    if (cell < ncells) {
        const int cy = cell / lv.W, cx = cell - cy * lv.W;
        float* d = dense + lv.out_off + (long long)frame * lv.Ho * lv.Wo + (long long)(cy * 8 + 4 * half) * lv.Wo + cx * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 u, v;
#pragma unroll
            for (int j = 0; j < 4; ++j) { u[j] = e[i * 8 + j] / sum; v[j] = e[i * 8 + 4 + j] / sum; }
            *(f32x4*)(d + (long long)i * lv.Wo) = u;
            *(f32x4*)(d + (long long)i * lv.Wo + 4) = v;
        }
    }
}

// Low-latency 1x1 convolution for launches that cannot fill the GPU (single frames: the projections of the 30x47 / 15x24
// layers of the global branch are a few dozen workgroups with 288-720 input channels).  k_pointwise prefetches one
// k-step ahead, which is right when other waves fill the gaps; alone on its SIMD a wave then pays one memory latency
// (~0.35 us) per k-step.  This variant keeps NBUF - 1 k-steps of loads in flight in rotating register buffers.
// Same MFMA order, same bits.  (A 16 x 16 tile per wave on v_mfma_f32_16x16x4_f32 -- bit-identical as well, see
// tools/micro/mfma_order.hip -- quarters the MFMA chain but triples the address-coalescer work and measured slower; for the
// big GEMM-shaped launches, e.g. 256 -> 256 on 128k descriptor rows, three steps in flight are no faster than one.)
// Tensors must stay below 2 GB (32-bit lane offsets on scalar bases).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int NT, int NBUF>
__global__ __launch_bounds__(256) void k_pointwise_deep(ConvArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    const long long row0 = (long long)blockIdx.x * 128 + wave * 32;
    if (row0 >= a.P) return;
    const int nt0 = blockIdx.y * NT;
    const long long row = min(row0 + r, a.P - 1);
    // uniform tile base + 32-bit lane offset (the tile's rows span < 4 GB)
    const long long tile_row0 = ((long long)__builtin_amdgcn_readfirstlane((int)(row0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)row0);
    const unsigned aoff = ((unsigned)(row - tile_row0) * (unsigned)a.cin + (unsigned)(half * 4)) * 4u;   // bytes
    const unsigned woff = (unsigned)lane * 16u;
    const char* __restrict__ abase = (const char*)(a.A + tile_row0 * a.cin);                    // uniform
    const char* __restrict__ wbase = (const char*)(a.W + (size_t)nt0 * 64);                     // uniform
    const unsigned wstep = (unsigned)a.nt_total * 64u * 16u;                                    // bytes per k-step
    const int KQ = a.cin >> 3;
    f32x16 acc[NT];
    conv_acc_init<NT>(a, acc, nt0, r);
    f32x4 av[NBUF], bv[NBUF][NT];
    auto load = [&](int kq, auto buf_tag) {
        constexpr int buf = decltype(buf_tag)::value;
        // unconditional (past the end: the last step again, an L1 hit nobody uses): a branch around the loads would make
        // the compiler's wait counts assume the shorter path and drain the queue before every step
        kq = min(kq, KQ - 1);
        av[buf] = *(const f32x4*)(abase + (size_t)kq * 32 + aoff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[buf][nt] = *(const f32x4*)(wbase + (size_t)kq * wstep + nt * 1024 + woff);
    };
    auto compute = [&](int kq, auto buf_tag) {
        constexpr int buf = decltype(buf_tag)::value;
        if (kq < KQ) {                                           // uniform
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][t], bv[buf][nt][t], acc[nt], 0, 0, 0);
        }
    };
    static_for<NBUF - 1>([&](auto i) { load(decltype(i)::value, i); });
    for (int kq = 0; kq < KQ; kq += NBUF) {
        static_for<NBUF>([&](auto i) {
            constexpr int I = decltype(i)::value;
            load(kq + I + NBUF - 1, std::integral_constant<int, (I + NBUF - 1) % NBUF>{});
            __builtin_amdgcn_sched_barrier(0);
            compute(kq + I, i);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    conv_epilogue<NT>(a, acc, nt0, row0, a.P, half, r);
}

// (An LDS-staged variant for wide inputs -- coalesced 128-byte row segments instead of per-lane 16-byte slices -- was
// measured on MI355X and gave no gain; it is not kept.)

// dense 3x3, stride 1, 'SAME' (pad 1): tiles of 32 consecutive pixels of ONE image; out-of-image taps
// contribute fma(0, w, acc) == acc, i.e. they are skipped exactly as the oracle skips them.
// GATHER: the 32 rows of a tile are the bilinear taps of 8 selected keypoints instead of consecutive
// pixels (sparse descriptor head); everything else is identical, so results are bit-identical per cell.
// cells != null: the rows of an image are its de-duplicated tap cells (launch_tap_cells: cells[image * kps_stride * 4 + row] =
// y * Wc + x, n_rows[image] of them) instead of four taps per keypoint
struct TapArgs { const hfnet_keypoint* kps; const int* n_in; long long kps_stride; const int* cells; const int* n_rows; };

template <int NT, bool GATHER>
__global__ __launch_bounds__(256, 2) void k_conv3x3(ConvArgs a, Geom g, TapArgs ta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    // exact 1-D grid, order [level][frame][column-tile group][row tile]: no workgroups are launched for tiles a
    // level does not have (the smaller pyramid levels have a third of level 0's tiles)
    const int G = a.nt_total / NT;
    int level = 0, rest = blockIdx.x;
    for (; level < g.n_levels - 1; ++level) {
        const int per = g.batch * G * a.level_tiles[level];
        if (rest < per) break;
        rest -= per;
    }
    const int tl = a.level_tiles[level];
    const int frame = rest / (G * tl);
    rest -= frame * G * tl;
    const int grp = rest / tl, tile = rest - grp * tl;
    const int image = level * g.batch + frame;
    const LevelGeom lv = g.lv[level];
    const int Hc = GATHER ? lv.Ho : lv.H, Wc = GATHER ? lv.Wo : lv.W;
    // (gathered rows: the count comes from device memory -- bounded by the image's slot of 4 * kps_stride rows)
    const int nrows = GATHER ? min(ta.cells ? ta.n_rows[image] : ta.n_in[image] * 4, (int)ta.kps_stride * 4) : Hc * Wc;
    // (An XCD-aware order -- contiguous runs of row tiles per XCD, both column groups adjacent -- cut this kernel's HBM
    // fetches by 43 % but ran 5-50 % slower: the kernel is issue-bound, not HBM-bound.)
    const int T = (nrows + 127) >> 7;
    if (tile >= T) return;                                      // (gather: an image with fewer keypoints than its level's budget)
    const int nt0 = grp * NT;
    const int p0 = tile * 128 + wave * 32;
    int y, x;
    bool pvalid;
    long long in_base, out_base;
    if (p0 >= nrows) return;
    if (GATHER && ta.cells) {
        pvalid = p0 + r < nrows;
        const int cell = ta.cells[(long long)image * ta.kps_stride * 4 + (pvalid ? p0 + r : 0)];
        y = cell / Wc; x = cell - y * Wc;
        in_base = lv.in_off + (long long)frame * Hc * Wc;
        out_base = (long long)image * ta.kps_stride * 4;
    } else if (GATHER) {
        const int n = nrows >> 2;
        const int row = p0 + r, i = row >> 2, t = row & 3;
        pvalid = i < n;
        const hfnet_keypoint kp = ta.kps[(long long)image * ta.kps_stride + (pvalid ? i : 0)];
        // identical float expressions to k_sample (HFNetTFModelV2.cc:119-120, BaseModel.cc:534-539)
        const float sw = ((float)Wc - 1.f) / (float)((float)lv.W - 1.f);
        const float sh = ((float)Hc - 1.f) / (float)((float)lv.H - 1.f);
        const float xf = sw * kp.x, yf = sh * kp.y;
        const int fx = (int)floorf(xf), fy = (int)floorf(yf);
        x = fx + ((t == 1 || t == 3) ? 1 : 0);
        y = fy + ((t == 1 || t == 2) ? 1 : 0);
        pvalid = pvalid && x >= 0 && x < Wc && y >= 0 && y < Hc;
        if (!pvalid) { x = 0; y = 0; }
        in_base = lv.in_off + (long long)frame * Hc * Wc;
        out_base = (long long)image * ta.kps_stride * 4;
    } else {
        pvalid = (p0 + r) < nrows;
        const int p = pvalid ? p0 + r : nrows - 1;
        y = p / Wc; x = p - y * Wc;
        in_base = lv.in_off + (long long)frame * nrows;
        out_base = lv.out_off + (long long)frame * nrows;
    }
    const f32x4* wp = a.W + ((size_t)nt0 * 64 + lane);
    const size_t wstep = (size_t)a.nt_total * 64;
    f32x16 acc[NT];
    conv_acc_init<NT>(a, acc, nt0, r);
    const int KQ = a.cin >> 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // per-tap source pointer / validity of this lane's pixel
    const float* tap_ptr[9];
    bool tap_ok[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = y + ky - 1, ix = x + kx - 1;
        tap_ok[tap] = pvalid && iy >= 0 && iy < Hc && ix >= 0 && ix < Wc;
        tap_ptr[tap] = a.A + (in_base + (long long)(tap_ok[tap] ? iy * Wc + ix : 0)) * a.cin + half * 4;
    }
    // Software pipeline over the flattened (tap, kq) steps, prefetch distance 2: three operand buffers rotate in a
    // kq loop unrolled by three (no register copies -- a copy would make the compiler wait for the load it was
    // just issued), so a step's operands were requested two steps (32 MFMAs, ~2000 cycles) earlier: an L2 hit
    // takes 0.7-1 us on this part.
    // Loads are unconditional (tap_ptr of an out-of-image tap points at a valid pixel) and the zero of a skipped tap is
    // selected when the operand is used: a load under a branch ends up in a temporary + copies + an early wait.
    auto issue = [&](const float* tp, int kq, size_t wrow, f32x4& av_d, f32x4 (&bv_d)[NT]) {
        av_d = *(const f32x4*)(tp + kq * 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv_d[nt] = wp[wrow + (size_t)nt * 64];
    };
    auto mfma16 = [&](const f32x4& av_u, bool ok, const f32x4 (&bv_u)[NT]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float x = ok ? av_u[t] : 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bv_u[nt][t], acc[nt], 0, 0, 0);
        }
    };
    if (KQ % 3 == 0) {
        f32x4 a0, a1, a2, b0[NT], b1[NT], b2[NT];
        bool o0 = tap_ok[0], o1 = tap_ok[0], o2 = false;
        issue(tap_ptr[0], 0, 0, a0, b0);
        issue(tap_ptr[0], 1, wstep, a1, b1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int tn = tap < 8 ? tap + 1 : 8;
            for (int kq = 0; kq < KQ; kq += 3) {
                const bool wrap = kq + 3 == KQ;                         // the steps after kq + 2 belong to the next tap
                const bool more = !(wrap && tap == 8);                  // (the last two prefetches re-read step 0: harmless)
                const float* tpn = wrap ? tap_ptr[tn] : tap_ptr[tap];
                const bool okn = wrap ? tap_ok[tn] : tap_ok[tap];
                const int kn = (wrap || !more) ? 0 : kq + 3;
                const size_t wn = more ? (size_t)((wrap ? tn : tap) * KQ + kn) * wstep : 0;
                // sched_barrier: the machine scheduler otherwise sinks the loads down to their first use
                issue(tap_ptr[tap], kq + 2, (size_t)(tap * KQ + kq + 2) * wstep, a2, b2); o2 = tap_ok[tap];
                __builtin_amdgcn_sched_barrier(0);
                mfma16(a0, o0, b0);
                __builtin_amdgcn_sched_barrier(0);
                issue(tpn, kn, wn, a0, b0); o0 = okn;
                __builtin_amdgcn_sched_barrier(0);
                mfma16(a1, o1, b1);
                __builtin_amdgcn_sched_barrier(0);
                issue(tpn, kn + 1, wn + wstep, a1, b1); o1 = okn;
                __builtin_amdgcn_sched_barrier(0);
                mfma16(a2, o2, b2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        f32x4 av = tap_ok[0] ? *(const f32x4*)(tap_ptr[0]) : zero;
        f32x4 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = wp[(size_t)nt * 64];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            for (int kq = 0; kq < KQ; ++kq) {
                f32x4 av_n = zero;
                f32x4 bv_n[NT];
                const bool last = (tap == 8) && (kq + 1 == KQ);
                const bool wrap = (kq + 1 == KQ);
                const int tap_n = wrap ? (tap < 8 ? tap + 1 : 8) : tap;
                const int kq_n = wrap ? 0 : kq + 1;
                if (!last) {
                    if (tap_ok[tap_n]) av_n = *(const f32x4*)(tap_ptr[tap_n] + kq_n * 8);
                    const size_t wrow = (size_t)(tap_n * KQ + kq_n) * wstep;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv_n[nt] = wp[wrow + (size_t)nt * 64];
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv_n[nt] = bv[nt];
                }
                mfma16(av, true, bv);
                av = av_n;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = bv_n[nt];
            }
        }
    }
    conv_epilogue<NT>(a, acc, nt0, out_base + p0, out_base + nrows, half, r);
}

// The same convolution with the weights staged through LDS (k_pointwise_wlds' scheme): the workgroup's four waves share
// every weight fragment, so a slab of four k-steps (NT x 4 KB, one coalesced 16-byte load per thread and k-step) crosses
// L2 -> L1 once instead of four times, and the only per-lane global loads left are the activation rows, requested a whole
// slab (64 MFMAs, ~4000 cycles) ahead.  cin % 32 == 0 (a slab never straddles two taps).  Same chains, same bits.
// SPT_T = slabs per tap when known at compile time (0: run-time loop): the slab loop is then fully unrolled and the "next"
// activation registers simply become the current ones (16 v_mov per slab otherwise): -2.5 % (dense) / -3.5 % (gathered).
// The per-value select of out-of-image taps (16 v_cndmask per 64 MFMAs) stays: both ways of getting the zeros from the
// load itself were measured SLOWER -- out-of-range buffer loads +3 % (run-time loop) / +12 % (unrolled), a zero page in
// memory +7-10 %: with the MFMA operands coming straight from load registers the compiler waits on every LDS weight
// fragment right before its first MFMA (21 instead of 7 s_waitcnt per slab), and the MFMA pipe is what this kernel lives on.
template <int NT, bool GATHER, int SPT_T>
__global__ __launch_bounds__(256, 2) void k_conv3x3_wlds(ConvArgs a, Geom g, TapArgs ta) {
    constexpr int KS = 4, UNR = SPT_T > 0 ? SPT_T : 1;
    __shared__ __attribute__((aligned(16))) f32x4 wl[2][KS][NT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    const int G = a.nt_total / NT;
    // workgroup b runs on XCD b % 8 (observed; speed only): every XCD takes one contiguous eighth of the tile list, so that the
    // halo rows a tile shares with its vertical neighbours are fetched into ONE L2 instead of three
    // (time-neutral, but the gathered descriptor taps fetch 69 % less and the dense detector conv 9 % less from beyond L2)
    const int q8 = (int)gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int level = 0, rest = slot < q8 ? xcd * q8 + slot : 8 * q8 + xcd;      // (the last gridDim.x % 8 workgroups keep their place)
    for (; level < g.n_levels - 1; ++level) {
        const int per = g.batch * G * a.level_tiles[level];
        if (rest < per) break;
        rest -= per;
    }
    const int tl = a.level_tiles[level];
    const int frame = rest / (G * tl);
    rest -= frame * G * tl;
    const int grp = rest / tl, tile = rest - grp * tl;
    const int image = level * g.batch + frame;
    const LevelGeom lv = g.lv[level];
    const int Hc = GATHER ? lv.Ho : lv.H, Wc = GATHER ? lv.Wo : lv.W;
    // (gathered rows: the count comes from device memory -- bounded by the image's slot of 4 * kps_stride rows)
    const int nrows = GATHER ? min(ta.cells ? ta.n_rows[image] : ta.n_in[image] * 4, (int)ta.kps_stride * 4) : Hc * Wc;
    const int T = (nrows + 127) >> 7;
    if (tile >= T) return;                                      // workgroup-uniform
    const int nt0 = grp * NT;
    const int p0 = tile * 128 + wave * 32;
    const bool active = p0 < nrows;                             // (a wave without rows still helps staging)
    int y, x;
    bool pvalid;
    long long in_base, out_base;
    if (GATHER && ta.cells) {
        pvalid = active && p0 + r < nrows;
        const int cell = ta.cells[(long long)image * ta.kps_stride * 4 + (pvalid ? p0 + r : 0)];
        y = cell / Wc; x = cell - y * Wc;
        in_base = lv.in_off + (long long)frame * Hc * Wc;
        out_base = (long long)image * ta.kps_stride * 4;
    } else if (GATHER) {
        const int n = nrows >> 2;
        const int row = p0 + r, i = row >> 2, t = row & 3;
        pvalid = active && i < n;
        const hfnet_keypoint kp = ta.kps[(long long)image * ta.kps_stride + (pvalid ? i : 0)];
        // identical float expressions to k_sample (HFNetTFModelV2.cc:119-120, BaseModel.cc:534-539)
        const float sw = ((float)Wc - 1.f) / (float)((float)lv.W - 1.f);
        const float sh = ((float)Hc - 1.f) / (float)((float)lv.H - 1.f);
        const float xf = sw * kp.x, yf = sh * kp.y;
        const int fx = (int)floorf(xf), fy = (int)floorf(yf);
        x = fx + ((t == 1 || t == 3) ? 1 : 0);
        y = fy + ((t == 1 || t == 2) ? 1 : 0);
        pvalid = pvalid && x >= 0 && x < Wc && y >= 0 && y < Hc;
        if (!pvalid) { x = 0; y = 0; }
        in_base = lv.in_off + (long long)frame * Hc * Wc;
        out_base = (long long)image * ta.kps_stride * 4;
    } else {
        pvalid = (p0 + r) < nrows;
        const int p = pvalid ? p0 + r : nrows - 1;
        y = p / Wc; x = p - y * Wc;
        in_base = lv.in_off + (long long)frame * nrows;
        out_base = lv.out_off + (long long)frame * nrows;
    }
    const int KQ = a.cin >> 3, SPT = SPT_T > 0 ? SPT_T : KQ / KS, n_slabs = 9 * SPT;   // slabs per tap
    const size_t wstep = (size_t)a.nt_total * 64;
    constexpr int PIECES = NT * 64, PER_T = (PIECES + 255) / 256;
    auto fetch = [&](int slab, f32x4 (&st)[KS][PER_T]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < PER_T; ++j)
                st[ks][j] = a.W[(size_t)(slab * KS + ks) * wstep + (size_t)nt0 * 64 + min(tid + j * 256, PIECES - 1)];
    };
    auto stash = [&](int buf, const f32x4 (&st)[KS][PER_T]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < PER_T; ++j)
                if (tid + j * 256 < PIECES) (&wl[buf][ks][0][0])[tid + j * 256] = st[ks][j];
    };
    f32x16 acc[NT];
    conv_acc_init<NT>(a, acc, nt0, r);
    // per-tap source pointer of this lane's pixel (an out-of-image tap points at the zero page)
    const float* tap_ptr[9];
    bool tap_ok[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = y + ky - 1, ix = x + kx - 1;
        const bool ok = pvalid && iy >= 0 && iy < Hc && ix >= 0 && ix < Wc;
        tap_ok[tap] = ok;
        tap_ptr[tap] = a.A + (in_base + (long long)(ok ? iy * Wc + ix : 0)) * a.cin + half * 4;
    }
    f32x4 st[KS][PER_T];
    fetch(0, st);
    stash(0, st);
    f32x4 av[KS], an[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) av[ks] = *(const f32x4*)(tap_ptr[0] + ks * 8);
    __syncthreads();
    int slab = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int tn = tap < 8 ? tap + 1 : 8;
#pragma unroll UNR
        for (int sl = 0; sl < SPT; ++sl, ++slab) {
            const int buf = slab & 1;
            const bool wrap = sl + 1 == SPT;
            if (slab + 1 < n_slabs) fetch(slab + 1, st);
            {
                const float* tpn = wrap ? tap_ptr[tn] : tap_ptr[tap];
                const int kn = wrap ? 0 : (sl + 1) * KS;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) an[ks] = *(const f32x4*)(tpn + (kn + ks) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool ok = tap_ok[tap];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f32x4 bv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = wl[buf][ks][nt][lane];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float xv = ok ? av[ks][t] : 0.0f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, bv[nt][t], acc[nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) av[ks] = an[ks];
            if (slab + 1 < n_slabs) stash(buf ^ 1, st);
            __syncthreads();
        }
    }
    if (active) conv_epilogue<NT>(a, acc, nt0, out_base + p0, out_base + nrows, half, r);
}

template <int NT>
static void launch_pw_nt(const ConvArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(k_pointwise<NT>, grid, dim3(256), 0, s, a);
}
template <int NT>
static void launch_c3_nt(const ConvArgs& a, const Geom& g, const TapArgs* ta, dim3 grid, hipStream_t s) {
    if (ta) hipLaunchKernelGGL((k_conv3x3<NT, true>), grid, dim3(256), 0, s, a, g, *ta);
    else { TapArgs none = {nullptr, nullptr, 0, nullptr, nullptr}; hipLaunchKernelGGL((k_conv3x3<NT, false>), grid, dim3(256), 0, s, a, g, none); }
}
static void launch_c3_wlds4(const ConvArgs& a, const Geom& g, const TapArgs* ta, dim3 grid, hipStream_t s) {
    TapArgs none = {nullptr, nullptr, 0, nullptr, nullptr};
    if (a.cin == 96) {                                           // both heads of the network: three slabs per tap, unrolled
        if (ta) hipLaunchKernelGGL((k_conv3x3_wlds<4, true, 3>), grid, dim3(256), 0, s, a, g, *ta);
        else hipLaunchKernelGGL((k_conv3x3_wlds<4, false, 3>), grid, dim3(256), 0, s, a, g, none);
        return;
    }
    if (ta) hipLaunchKernelGGL((k_conv3x3_wlds<4, true, 0>), grid, dim3(256), 0, s, a, g, *ta);
    else hipLaunchKernelGGL((k_conv3x3_wlds<4, false, 0>), grid, dim3(256), 0, s, a, g, none);
}

static ConvArgs make_args(const float* A, const ConvPack& cp, const float* res, float* out, long long P, int relu6) {
    ConvArgs a;
    a.A = A; a.W = (const f32x4*)cp.w; a.bias = cp.bias; a.res = res; a.out = out;
    a.P = P; a.cin = cp.cin; a.n = cp.n; a.nt_total = cp.nt_total; a.relu6 = relu6;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) a.level_tiles[l] = 1;
    a.slot_units = nullptr; a.slot_rows = 0; a.rows_per_unit = 0;
    return a;
}

// column tiles per wave: the packed layout allows any divisor of nt_total; small-M layers (the
// 30x47 / 15x24 global branch) take fewer tiles per wave so that the launch still fills 256 CUs
static int pick_nt(int nt_total, int nt_pref, long long m_tiles) {
    constexpr int max_nt = 4;   // measured on MI355X: <= 4 column tiles per wave (higher occupancy) beats 8
    int best = 1;
    for (int nt = 1; nt <= nt_pref && nt <= max_nt; ++nt) {
        if (nt_total % nt) continue;
        if (m_tiles * (nt_total / nt) >= 2048 || nt == 1) best = nt;
    }
    return best;
}

hipError_t launch_pointwise(const float* A, const ConvPack& cp, const float* residual, float* out, long long P, int relu6,
                            hipStream_t s, const int* slot_units, int slot_rows, int rows_per_unit) {
    if (P <= 0) return hipSuccess;
    ConvArgs a = make_args(A, cp, residual, out, P, relu6);
    if (slot_units && slot_rows > 0) { a.slot_units = slot_units; a.slot_rows = slot_rows; a.rows_per_unit = rows_per_unit; }
    const int nt = pick_nt(cp.nt_total, cp.nt_per_block, (P + 31) / 32);
    // long k chains on few tiles: latency-bound, see k_pointwise_deep
    constexpr long long lowlat_waves = 1024;
    constexpr long long wlds_min_weight_bytes = 32 * 1024;   // (measured: the detector's 128 -> 65 conv, 48 KB, 270 -> 232 us; it was 128 KB)
    if (nt <= 2 && cp.cin >= 192 && (P + 31) / 32 * cp.nt_total < lowlat_waves) {
        dim3 grid((unsigned)((P + 127) / 128), cp.nt_total / nt);
        if (nt == 1) hipLaunchKernelGGL((k_pointwise_deep<1, 8>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_pointwise_deep<2, 8>), grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    // a GEMM-shaped launch whose weights would crowd the activations out of L1 (32 KB): weight slabs through LDS (see k_pointwise_wlds)
    if ((size_t)cp.cin * cp.nt_total * 32 * sizeof(float) >= (size_t)wlds_min_weight_bytes && (P + 127) / 128 * (cp.nt_total / nt) >= 512 &&
        nt >= 2 && nt <= 4) {
        dim3 gw((unsigned)((P + 127) / 128), cp.nt_total / nt);
        if (nt == 2) hipLaunchKernelGGL(k_pointwise_wlds<2>, gw, dim3(256), 0, s, a);
        else if (nt == 3) hipLaunchKernelGGL(k_pointwise_wlds<3>, gw, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_pointwise_wlds<4>, gw, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((P + 127) / 128), cp.nt_total / nt);
    switch (nt) {
        case 1: launch_pw_nt<1>(a, grid, s); break;
        case 2: launch_pw_nt<2>(a, grid, s); break;
        case 3: launch_pw_nt<3>(a, grid, s); break;
        case 4: launch_pw_nt<4>(a, grid, s); break;
        case 5: launch_pw_nt<5>(a, grid, s); break;
        case 6: launch_pw_nt<6>(a, grid, s); break;
        case 7: launch_pw_nt<7>(a, grid, s); break;
        case 8: launch_pw_nt<8>(a, grid, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool det_tail_supported(const ConvPack& cp) { return cp.n == 65 && cp.nt_total == 3 && cp.cin % 8 == 0 && cp.cin >= 8; }

// g: H, W = cell grid, Ho, Wo = dense map (8 H, 8 W), in_off = first cell of the level, out_off = first pixel (launch_softmax_d2s' geometry)
hipError_t launch_det_tail(const float* hidden, const ConvPack& cp, float* dense, const Geom& g, hipStream_t s) {
    if (!det_tail_supported(cp)) return hipErrorInvalidValue;
    int maxcells = 0;
    for (int l = 0; l < g.n_levels; ++l) maxcells = max(maxcells, g.lv[l].H * g.lv[l].W);
    if (maxcells <= 0) return hipSuccess;
    ConvArgs a = make_args(hidden, cp, nullptr, nullptr, 0, 0);
    hipLaunchKernelGGL(k_det_tail, dim3((unsigned)((maxcells + 127) / 128), (unsigned)(g.n_levels * g.batch)), dim3(256), 0, s, a, dense, g);
    return hipGetLastError();
}

static hipError_t launch_conv3x3_any(const float* A, const ConvPack& cp, float* out, int relu6, const Geom& g, const TapArgs* ta,
                                     const int* level_rows, int wlds, hipStream_t s) {
    ConvArgs a = make_args(A, cp, nullptr, out, 0, relu6);
    int ntb = cp.nt_per_block;
    if (ta && cp.nt_total % 4 == 0) ntb = 4;           // gathered rows: four column tiles per wave
    long long tiles = 0;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) {
        a.level_tiles[l] = l < g.n_levels ? std::max((level_rows[l] + 127) / 128, 1) : 1;
        if (l < g.n_levels) tiles += (long long)g.batch * a.level_tiles[l];
    }
    // small batches: a wave's 3x3 chain over 9 * cin is ~50 us long with four column tiles, and a single frame has only
    // ~110 row tiles -- take fewer column tiles per wave until the launch has two workgroups per CU (latency, not throughput)
    constexpr int min_wgs = 512;
    while (ntb > 1 && tiles * (cp.nt_total / ntb) < min_wgs) {
        int next = ntb - 1;
        while (next > 1 && cp.nt_total % next) --next;
        ntb = next;
    }
    const long long wgs = tiles * (cp.nt_total / ntb);
    if (wgs <= 0 || wgs > 0x7fffffffll) return hipErrorInvalidValue;
    dim3 grid((unsigned)wgs, 1, 1);
    if (wlds && ntb == 4 && cp.cin % 32 == 0) { launch_c3_wlds4(a, g, ta, grid, s); return hipGetLastError(); }
    switch (ntb) {
        case 1: launch_c3_nt<1>(a, g, ta, grid, s); break;
        case 2: launch_c3_nt<2>(a, g, ta, grid, s); break;
        case 3: launch_c3_nt<3>(a, g, ta, grid, s); break;
        case 4: launch_c3_nt<4>(a, g, ta, grid, s); break;
        case 5: launch_c3_nt<5>(a, g, ta, grid, s); break;
        case 6: launch_c3_nt<6>(a, g, ta, grid, s); break;
        case 7: launch_c3_nt<7>(a, g, ta, grid, s); break;
        case 8: launch_c3_nt<8>(a, g, ta, grid, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv3x3(const float* A, const ConvPack& cp, float* out, int relu6, const Geom& g, int wlds, hipStream_t s) {
    int rows[HFNET_MAX_LEVELS] = {0};
    for (int l = 0; l < g.n_levels; ++l) rows[l] = g.lv[l].H * g.lv[l].W;
    return launch_conv3x3_any(A, cp, out, relu6, g, nullptr, rows, wlds, s);
}

hipError_t launch_conv3x3_taps(const float* A, const ConvPack& cp, float* out, int relu6, const hfnet_keypoint* kps, const int* n_in,
                               long long kps_stride, const int* level_keypoints, const Geom& g, int wlds, hipStream_t s, const int* cells,
                               const int* n_rows) {
    const TapArgs ta = {kps, n_in, kps_stride, cells, cells ? n_rows : nullptr};
    int rows[HFNET_MAX_LEVELS] = {0};
    for (int l = 0; l < g.n_levels; ++l) rows[l] = 4 * (int)std::min<long long>(level_keypoints[l], kps_stride);
    return launch_conv3x3_any(A, cp, out, relu6, g, &ta, rows, wlds, s);
}

// =========================================================================== split-bf16 GEMMs (engine options desc_bf16x3 / global_bf16x3)
// north_star: "keypoint indices bit-exact ..., descriptor/score tensors within a stated fp32 tolerance".  Everything that DECIDES
// an index (backbone layers 1-7, detector head, NMS, top-K) stays on the exact f32 chains above.  The stages that only produce
// float tensors may -- as an engine OPTION, default off -- run on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16 x the f32
// MFMA rate): every f32 operand is split into two bf16 pieces x = hi + lo + e (round to nearest even twice, |e| <= 2^-16 |x|) and
// a.w ~ ah.wh + ah.wl + al.wh -- three instructions of 32 cycles per 16 k where the f32 form needs eight of 64.  The bf16 x bf16
// products are exact in fp32; dropped are al.wl and the e terms (<= 3 * 2^-16 |a w| per term, worst case) and the sums follow the
// unit's own order: the results are NOT the oracle's bits but within the tolerance stated in include/hfnet_hip.h (tests/).
// Weights are split once per engine (k_repack_bf16x3 from the f32 ConvPack); activations are split in registers by the consuming
// kernel.  K order of a 16-k step s: k = 8 half + p is PHYSICAL slot p of channel group 2 s + half (a lane's 32 consecutive
// bytes of an activation row), so no permutation is needed on either side.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// f32 pack [kq][nt][lane][4 t] (logical channel 2 t + half of group kq) -> [s][nt][hi | lo][lane][8]: element e of lane (half, j)
// is physical slot e of group 2 s + half = logical channel lop(e): e < 4 -> (t = e, half' = 0), else (t = e - 4, half' = 1)
// bias (optional; an odd number of channel groups only): the spare k slot cin of the last step -- element 0 of its upper half -- takes the layer's
// folded bias, so that a kernel that feeds a constant 1 there gets bias + sum from accumulators that start at ZERO (an inline constant: no
// registers, no moves; k_block_fused8's split-bf16 form of the 24-channel layers)
__global__ __launch_bounds__(256) void k_repack_bf16x3(const f32x4* __restrict__ W, int kq_total, int nt_total, bf16x8* __restrict__ out, const float* __restrict__ bias) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int steps = (kq_total + 1) / 2;                          // (an odd number of channel groups: the last step's upper half is zeros)
    if (idx >= (long long)steps * nt_total * 64) return;
    const int lane = (int)(idx & 63), nt = (int)((idx >> 6) % nt_total), s = (int)((idx >> 6) / nt_total);
    const int half = lane >> 5, j = lane & 31, kq = 2 * s + half;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 w0 = kq < kq_total ? W[((size_t)kq * nt_total + nt) * 64 + j] : z4;
    const f32x4 w1 = kq < kq_total ? W[((size_t)kq * nt_total + nt) * 64 + 32 + j] : z4;
    if (bias && kq == kq_total) w0[0] = bias[nt * 32 + j];
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? w0[e] : w1[e - 4];
        hi[e] = (__bf16)v;
        lo[e] = (__bf16)(v - (float)hi[e]);
    }
    out[(((size_t)s * nt_total + nt) * 2 + 0) * 64 + lane] = hi;
    out[(((size_t)s * nt_total + nt) * 2 + 1) * 64 + lane] = lo;
}

hipError_t launch_repack_bf16x3(const ConvPack& cp, void* out, hipStream_t s, int with_bias) {
    const int kq_total = cp.taps * cp.cin / 8;
    if (cp.cin % 8 || (cp.taps != 1 && cp.cin % 16)) return hipErrorInvalidValue;
    if (with_bias && (kq_total % 2 == 0 || cp.taps != 1)) return hipErrorInvalidValue;      // (no spare k slot)
    const long long n = (long long)((kq_total + 1) / 2) * cp.nt_total * 64;
    hipLaunchKernelGGL(k_repack_bf16x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const f32x4*)cp.w, kq_total, cp.nt_total, (bf16x8*)out,
                       with_bias ? cp.bias : nullptr);
    return hipGetLastError();
}
size_t bf16x3_pack_bytes(const ConvPack& cp) { return (size_t)((cp.taps * cp.cin / 8 + 1) / 2) * cp.nt_total * 2 * 64 * 16; }

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two values: hi = round to nearest even (one v_cvt_pk_bf16_f32 for the pair), lo = bf16(v - hi) (the difference is exact in fp32; the f32 of the hi
// pieces is a shift / a mask of the packed word): 3 vector instructions per value
__device__ __forceinline__ void split2(float x, float y, bf16x2& p, bf16x2& q) {
    p[0] = (__bf16)x; p[1] = (__bf16)y;
    const unsigned pu = __builtin_bit_cast(unsigned, p);
    const float hx = __builtin_bit_cast(float, pu << 16), hy = __builtin_bit_cast(float, pu & 0xffff0000u);
    q[0] = (__bf16)(x - hx); q[1] = (__bf16)(y - hy);
}
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bf16x2 p, q;
        split2(j < 2 ? v0[2 * j] : v1[2 * j - 4], j < 2 ? v0[2 * j + 1] : v1[2 * j - 3], p, q);
        hi[2 * j] = p[0]; hi[2 * j + 1] = p[1];
        lo[2 * j] = q[0]; lo[2 * j + 1] = q[1];
    }
}
__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bf16x2 p, q;
        split2(v[2 * j], v[2 * j + 1], p, q);
        hi[2 * j] = p[0]; hi[2 * j + 1] = p[1];
        lo[2 * j] = q[0]; lo[2 * j + 1] = q[1];
    }
}

// A workgroup = 256 rows x 128 columns (wave: 64 rows x 128 columns, eight 32 x 32 accumulator tiles): the weight fragments of a
// slab of two 16-k steps (16 KB: 4 column tiles x {hi, lo} x 2 steps) go through LDS once per workgroup, double buffered, one
// barrier per slab; the activation rows are requested a slab ahead.  GATHER: the rows of an image are its distinct tap cells and
// K walks the nine taps of the 3 x 3 window (out-of-image taps: zeros), as k_conv3x3; otherwise rows are plain [P][cin] rows
// (slotted rows: tiles in the unused part of a slot are skipped).  Bias as the accumulators' start, ReLU6 and stores: conv_epilogue.
template <bool GATHER>
__global__ __launch_bounds__(256, 2) void k_conv_bf16x3(ConvArgs a, const bf16x8* __restrict__ Wb, Geom g, TapArgs ta) {
    constexpr int NT = 4, MT = 2, SS = 2;                       // column tiles / row tiles per wave, steps per slab
    __shared__ __attribute__((aligned(16))) bf16x8 wl[2][SS * NT * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    int grp, tile, Wc = 0, Hc = 0;
    long long nrows, in_base = 0, out_base = 0;
    int image = 0;
    if (GATHER) {
        const int G = (a.nt_total + NT - 1) / NT;
        int level = 0, rest = blockIdx.x;
        for (; level < g.n_levels - 1; ++level) {
            const int per = g.batch * G * a.level_tiles[level];
            if (rest < per) break;
            rest -= per;
        }
        const int tl = a.level_tiles[level];
        const int frame = rest / (G * tl);
        rest -= frame * G * tl;
        grp = rest / tl; tile = rest - grp * tl;
        image = level * g.batch + frame;
        const LevelGeom lv = g.lv[level];
        Hc = lv.Ho; Wc = lv.Wo;
        nrows = min(ta.n_rows[image], (int)ta.kps_stride * 4);
        in_base = lv.in_off + (long long)frame * Hc * Wc;
        out_base = (long long)image * ta.kps_stride * 4;
    } else {
        tile = blockIdx.x; grp = blockIdx.y; nrows = a.P;
    }
    if ((long long)tile * 256 >= nrows) return;                 // (workgroup-uniform)
    const int nt0 = grp * NT, ntv = min(NT, a.nt_total - nt0);  // column tiles of this group that exist (the last group may be ragged)
    const long long p0 = (long long)tile * 256 + wave * 64;
    // a wave without rows still takes part in the staging and the barriers
    bool live = p0 < nrows;
    if (!GATHER && live && !tile_in_use(a, p0, 64)) live = false;
    const int spt = (a.cin + 15) >> 4;                          // 16-k steps per tap (1x1: the last one may hold 8 channels)
    const int n_steps = (GATHER ? 9 : 1) * spt, n_slabs = (n_steps + SS - 1) / SS;
    // ---- this lane's two rows: centre pointers (+ tap validity bits)
    const char* cptr[MT];
    unsigned okbits[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long long row = p0 + 32 * i + r;
        const bool pvalid = live && row < nrows;
        okbits[i] = 0;
        if (GATHER) {
            const int cell = pvalid ? ta.cells[(long long)image * ta.kps_stride * 4 + row] : 0;
            const int y = cell / Wc, x = cell - y * Wc;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                if (pvalid && iy >= 0 && iy < Hc && ix >= 0 && ix < Wc) okbits[i] |= 1u << tap;
            }
            cptr[i] = (const char*)(a.A + (in_base + (long long)y * Wc + x) * a.cin + half * 8);
        } else {
            okbits[i] = pvalid ? 1u : 0u;
            cptr[i] = (const char*)(a.A + (pvalid ? row : 0) * a.cin + half * 8);
        }
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float b = a.bias[(nt0 + min(nt, ntv - 1)) * 32 + r];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][nt][e] = b;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // activation pieces of one slab: [step][row tile][two 16-byte pieces].  A slab's pieces are split into their bf16 fragments
    // first; the registers they came in then take the NEXT slab's loads, which have the slab's 48 MFMAs to arrive.
    f32x4 av[SS][MT][2];
    auto load_a = [&](int slab) {
#pragma unroll
        for (int ss = 0; ss < SS; ++ss) {
            const int s = min(slab * SS + ss, n_steps - 1);
            const int tap = GATHER ? s / spt : 0, cs = s - tap * spt;
            const int toff = GATHER ? ((tap / 3 - 1) * Wc + (tap % 3 - 1)) * a.cin * 4 : 0;     // uniform
            const bool kok = cs * 16 + half * 8 < a.cin;         // (the upper half of a last step of 8 channels: zeros)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bool ok = ((okbits[i] >> tap) & 1u) && kok;
                const char* p = cptr[i] + (ok ? toff : 0) + (kok ? cs * 64 : 0);
                const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + 16);
                av[ss][i][0] = ok ? v0 : zero4; av[ss][i][1] = ok ? v1 : zero4;
            }
        }
    };
    // weight pieces of one slab: 16 pieces of 1 KB ([step][nt][hi | lo]), thread tid takes pieces j * 4 + (tid >> 6), 16 bytes each
    bf16x8 ws[4];
    auto load_w = [&](int slab) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = j * 4 + wave, ss = piece >> 3, nt = (piece >> 1) & 3, hl = piece & 1;
            const int s = min(slab * SS + ss, n_steps - 1);
            ws[j] = Wb[(((size_t)s * a.nt_total + nt0 + min(nt, ntv - 1)) * 2 + hl) * 64 + lane];
        }
    };
    auto stash_w = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wl[buf][(j * 4 + wave) * 64 + lane] = ws[j];
    };
    load_w(0); load_a(0);
    stash_w(0);
    __syncthreads();
    for (int slab = 0; slab < n_slabs; ++slab) {
        const int buf = slab & 1;
        bf16x8 ah[SS][MT], al[SS][MT];
#pragma unroll
        for (int ss = 0; ss < SS; ++ss)
#pragma unroll
            for (int i = 0; i < MT; ++i) split8(av[ss][i][0], av[ss][i][1], ah[ss][i], al[ss][i]);
        __builtin_amdgcn_sched_barrier(0);
        load_w(min(slab + 1, n_slabs - 1)); load_a(min(slab + 1, n_slabs - 1));      // (unconditional: the last pass re-reads its own slab)
        __builtin_amdgcn_sched_barrier(0);
        if (live) {
            // the (step, column tile) pairs of the slab as one sequence: the fragments of pair q + 1 are read from LDS before the MFMAs of
            // pair q are issued (left alone the compiler puts every ds_read right in front of its first use: ~100 exposed cycles per pair)
            const int n_q = min(SS, n_steps - slab * SS) * NT;  // (uniform; an odd step count leaves the last slab half empty)
            bf16x8 bh = wl[buf][0 * 64 + lane], bl = wl[buf][1 * 64 + lane];
#pragma unroll
            for (int q = 0; q < SS * NT; ++q) {
                if (q < n_q) {
                    const int ss = q / NT, nt = q % NT, qn = min(q + 1, SS * NT - 1);
                    const bf16x8 bhn = wl[buf][(qn * 2 + 0) * 64 + lane], bln = wl[buf][(qn * 2 + 1) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                    if (nt < ntv) {                             // (uniform: a ragged last column group)
                        // (the three products of one accumulator are never back to back: a dependent MFMA would wait for its predecessor)
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ss][i], bh, acc[i][nt], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ss][i], bl, acc[i][nt], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ss][i], bh, acc[i][nt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    bh = bhn; bl = bln;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stash_w(buf ^ 1);
        __syncthreads();
    }
    if (live) {
        // (ReLU6 as a median with +-infinity bounds when the layer has none: one instruction either way)
        const float lo6 = a.relu6 ? 0.0f : -INFINITY, hi6 = a.relu6 ? 6.0f : INFINITY;
        const long long n = a.n;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const long long row0 = p0 + 32 * i + 4 * half;
            const int left = (int)min((long long)32, nrows - row0);          // rows of this lane's column that exist (may be <= 0)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = (nt0 + nt) * 32 + r;
                if (nt < ntv && col < a.n) {
                    float* __restrict__ op = a.out + (out_base + row0) * n + col;
                    const float* __restrict__ rp = a.res ? a.res + (out_base + row0) * n + col : nullptr;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rr = (reg & 3) + 8 * (reg >> 2);
                        if (rr < left) {
                            float v = __builtin_amdgcn_fmed3f(acc[i][nt][reg], lo6, hi6);
                            if (rp) v = v + rp[rr * n];
                            op[rr * n] = v;
                        }
                    }
                }
            }
        }
    }
}

// ---- k_conv_bf16x3 for outputs of 256 columns (the descriptor head: 3 x 3 96 -> 256 at the tap cells, 1 x 1 256 -> 256 on the tap rows).
// There every activation row is gathered (and split) once per 128-column group, 32 bytes per lane and k-step straight into registers: the
// kernel is bound by that per-lane gather (~16 B/clk per CU), not by the matrix pipe (0.35 of the bf16 roof).  Here a workgroup owns 128 rows x
// 256 columns: its four waves are two row halves x two column groups, the rows' 16 channels of a k-step are gathered ONCE per workgroup (two
// 16-byte pieces per thread), split ONCE and shared through LDS ([row][hi 16 | lo 16 | pad]: the dense kernel's record); the weight pieces of a
// k-step (8 column tiles x {hi, lo} = 16 KB) arrive by LDS-DMA.  Both double buffered, one barrier per k-step (24 MFMAs per wave); per MFMA the
// gather traffic and the split arithmetic are half of k_conv_bf16x3's.  cin % 16 == 0, nt_total % 8 == 0.
template <bool GATHER>
__global__ __launch_bounds__(256, 2) void k_conv_rows_bf16x3(ConvArgs a, const bf16x8* __restrict__ Wb, Geom g, TapArgs ta) {
    constexpr int NT = 4, MT = 2, REC = 80, ROWS = 128, NTW = 8;   // NTW: column tiles per workgroup
    __shared__ __attribute__((aligned(16))) unsigned char As[2][ROWS * REC];
    __shared__ __attribute__((aligned(16))) bf16x8 wl[2][NTW * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int rh = wave_u & 1, cg = wave_u >> 1;
    int grp, tile, Wc = 0, Hc = 0, image = 0;
    int nrows;                                                   // (launcher: fewer than 2^31 rows, tensors below 4 GB)
    long long in_base = 0, out_base = 0;
    if (GATHER) {
        const int G = a.nt_total / NTW;
        int level = 0, rest = blockIdx.x;
        for (; level < g.n_levels - 1; ++level) {
            const int per = g.batch * G * a.level_tiles[level];
            if (rest < per) break;
            rest -= per;
        }
        const int tl = a.level_tiles[level];
        const int frame = rest / (G * tl);
        rest -= frame * G * tl;
        grp = rest / tl; tile = rest - grp * tl;
        image = level * g.batch + frame;
        const LevelGeom lv = g.lv[level];
        Hc = lv.Ho; Wc = lv.Wo;
        nrows = min(ta.n_rows[image], (int)ta.kps_stride * 4);
        in_base = lv.in_off + (long long)frame * Hc * Wc;
        out_base = (long long)image * ta.kps_stride * 4;
    } else {
        tile = blockIdx.x; grp = blockIdx.y; nrows = (int)a.P;
    }
    const int p0 = tile * ROWS;
    if (p0 >= nrows) return;                                     // (workgroup-uniform)
    if (!GATHER && !tile_in_use(a, p0, ROWS)) return;            // (slotted rows: the whole tile lies in the unused part of a slot)
    const int nt0 = grp * NTW;
    int spt = a.cin >> 4;
    // plain rows, gridDim.z > 1 (the dimensionality-reduction FC: few rows, a long K): part z takes the 16-k steps [z, z + 1) spt / parts and writes
    // its sums (no bias: a.bias is null) to slab z of the output, [parts][P][n]; the caller adds the slabs
    long long part_out = 0;
    unsigned part_shift = 0;
    if (!GATHER && gridDim.z > 1) {
        const int spp = spt / (int)gridDim.z;                    // (launcher: divisible, even)
        part_shift = (unsigned)(blockIdx.z * spp) * 64u;          // bytes into a row
        Wb += (size_t)blockIdx.z * spp * a.nt_total * 2 * 64;
        part_out = (long long)blockIdx.z * a.P * a.n;
        spt = spp;
    }
    const int n_steps = (GATHER ? 9 : 1) * spt;
    unsigned store_tiles = 3u;                                   // which of this wave's two 32-row tiles are stored (slotted rows: decided here, not in the epilogue)
    if (!GATHER) store_tiles = (tile_in_use(a, p0 + rh * 64, 32) ? 1u : 0u) | (tile_in_use(a, p0 + rh * 64 + 32, 32) ? 2u : 0u);
    const char* __restrict__ xb = (const char*)(a.A + in_base * a.cin) + part_shift;      // uniform (plain rows: in_base = 0); lane offsets are 32-bit
    // ---- staging role: this thread's two rows (tid >> 2 and + 64), piece tid & 3 (16 bytes = 4 channels of the k-step's 16).
    // Buffer loads: the per-lane part of the address -- the row's centre cell + the piece, or an offset past the resource's range where the tap lies
    // outside the image / the row does not exist (the load then returns zeros without touching memory) -- is computed ONCE per tap here and stays in
    // registers for the whole K loop; the tap / k-step part moves the scalar base of the resource.  No vector address arithmetic inside the loop: the
    // compiler guards a recycled load-destination register with a vmcnt wait, and such a wait would also wait for the weight requests it cannot see.
    constexpr int NTAP = GATHER ? 9 : 1;
    constexpr unsigned kOutOfRange = 0xffffff00u;
    const unsigned img_bytes = (GATHER ? (unsigned)Hc * (unsigned)Wc * (unsigned)a.cin * 4u : (unsigned)nrows * (unsigned)a.cin * 4u) - part_shift;
    unsigned coff[2], okbits[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = (tid >> 2) + 64 * k;
        const bool pvalid = p0 + row < nrows;
        okbits[k] = 0;
        if (GATHER) {
            const int cell = pvalid ? ta.cells[(long long)image * ta.kps_stride * 4 + p0 + row] : 0;
            const int y = cell / Wc, x = cell - y * Wc;
            coff[k] = ((unsigned)(y * Wc + x) * (unsigned)a.cin + (unsigned)((tid & 3) * 4)) * 4u;
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap) {
                const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                if (pvalid && iy >= 0 && iy < Hc && ix >= 0 && ix < Wc) okbits[k] |= 1u << tap;
            }
        } else {
            okbits[k] = pvalid ? 1u : 0u;
            coff[k] = ((unsigned)(pvalid ? p0 + row : p0) * (unsigned)a.cin + (unsigned)((tid & 3) * 4)) * 4u;     // (launcher: the tensor is < 4 GB)
        }
    }
    // the offsets of one tap: computed once per tap, at the top of the tap BEFORE (in the K loop below), in front of that step's weight requests
    auto tap_offsets = [&](int tap, unsigned (&vo)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) vo[k] = ((okbits[k] >> tap) & 1u) ? coff[k] : kOutOfRange;
    };
    // activation pieces are requested TWO k-steps ahead (two register sets that alternate; the K loop is unrolled so that they keep their names)
    f32x4 aregX[2], aregY[2];
    auto load_a = [&](int tap, int cs, const unsigned (&vo)[2], f32x4 (&areg)[2]) {
        const int toff = GATHER ? ((tap / 3 - 1) * Wc + (tap % 3 - 1)) * a.cin * 4 : 0;      // uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + toff + cs * 64), 0, img_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 2; ++k) areg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo[k], 0, 0));
    };
    auto write_a = [&](int s, int buf, const f32x4 (&areg)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            bf16x4 hi, lo;
            split4(areg[k], hi, lo);                             // (zeros where the tap lies outside the image / the row does not exist)
            unsigned char* dst = As[buf] + ((tid >> 2) + 64 * k) * REC + (tid & 3) * 8;
            *(bf16x4*)dst = hi;
            *(bf16x4*)(dst + 32) = lo;
        }
    };
    // weight pieces of a k-step: 8 column tiles x {hi, lo} = 16 pieces of 1 KB; wave w moves pieces 4 j + w by LDS-DMA (k_conv3x3_dense_bf16x3)
    const unsigned wl_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&wl[0][0];
    const unsigned lane16 = (unsigned)lane * 16u;
    auto load_w = [&](int s, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = j * 4 + wave_u, nt = piece >> 1, hl = piece & 1;
            const bf16x8* src = Wb + (((size_t)s * a.nt_total + nt0 + nt) * 2 + hl) * 64;      // uniform
            const unsigned dst = wl_lds + (unsigned)((buf * NTW * 2 + piece) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(src), "v"(lane16), "s"(dst) : "memory", "m0");
        }
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float b = a.bias ? a.bias[(nt0 + cg * NT + nt) * 32 + r] : 0.0f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][nt][e] = b;
    }
    const int abase = (rh * 64 + r) * REC + half * 16;
    // Request order matters: the weight pieces are inline assembly the compiler does not count, and its own vmcnt waits for the activation registers
    // assume only the loads it knows.  With the weight requests of a step issued BEFORE that step's activation requests, "at most the two newest
    // requests outstanding" -- what the compiler emits in front of write_a, and what the explicit wait in front of the barrier says -- means exactly
    // "everything but the activation pieces two steps ahead has landed".
    unsigned vo_cur[2], vo_nxt[2];
    tap_offsets(0, vo_cur);
    load_w(0, 0);
    load_a(0, 0, vo_cur, aregX);
    load_a(0, 1, vo_cur, aregY);                                  // (spt is even: launcher)
    write_a(0, 0, aregX);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __syncthreads();
    // one k-step: cur is free (its pieces are in LDS) and takes the pieces of step s + 2 = (ltap, lcs) if there is one; nxt holds step s + 1
    auto step = [&](int s, f32x4 (&cur)[2], f32x4 (&nxt)[2], bool load, int ltap, int lcs, const unsigned (&lvo)[2]) {
        const int buf = s & 1;
        if (s + 1 < n_steps) load_w(s + 1, buf ^ 1);
        if (load) load_a(ltap, lcs, lvo, cur);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 ah[MT], al[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            ah[i] = *(const bf16x8*)(As[buf] + abase + i * 32 * REC);
            al[i] = *(const bf16x8*)(As[buf] + abase + i * 32 * REC + 32);
        }
#pragma unroll
        for (int np = 0; np < NT; np += 2) {
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                bh[n2] = wl[buf][((cg * NT + np + n2) * 2 + 0) * 64 + lane];
                bl[n2] = wl[buf][((cg * NT + np + n2) * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][np + n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[n2], acc[i][np + n2], 0, 0, 0);
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][np + n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[n2], acc[i][np + n2], 0, 0, 0);
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][np + n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[n2], acc[i][np + n2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < n_steps) write_a(s + 1, buf ^ 1, nxt);
        // (the two newest requests are the activation pieces of step s + 2 -- when there is no such step, the newest are weight pieces: wait for all)
        if (load) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // (the tap loop stays a runtime loop: unrolled nine times the scalar registers spill into vector ones; the last tap is written out so that
    //  "is there a step s + 2" is a compile-time fact in every copy of the step -- a run-time flag makes the compiler's vmcnt waits conservative)
    auto tap_steps = [&](int tap, auto has_next) {
        constexpr bool NEXT = decltype(has_next)::value;
        const int s0 = tap * spt;
        if (NEXT) tap_offsets(tap + 1, vo_nxt);
#pragma unroll 1
        for (int cs = 0; cs + 2 < spt; cs += 2) {                  // the steps whose look-ahead stays inside this tap
            step(s0 + cs, aregX, aregY, true, tap, cs + 2, vo_cur);
            step(s0 + cs + 1, aregY, aregX, true, tap, cs + 3, vo_cur);
        }
        step(s0 + spt - 2, aregX, aregY, NEXT, tap + 1, 0, vo_nxt);   // the tap's last two steps look ahead into the next tap
        step(s0 + spt - 1, aregY, aregX, NEXT, tap + 1, 1, vo_nxt);
        if (NEXT) {
#pragma unroll
            for (int k = 0; k < 2; ++k) vo_cur[k] = vo_nxt[k];
        }
    };
    int ntap = NTAP;
    asm volatile("" : "+s"(ntap));                                // (opaque: with the loop below folded away for the one-tap form the register allocator spills 37 registers)
#pragma unroll 1
    for (int tap = 0; tap + 1 < ntap; ++tap) tap_steps(tap, std::true_type{});
    tap_steps(ntap - 1, std::false_type{});
    // ---- (ReLU6) and store
    const float lo6 = a.relu6 ? 0.0f : -INFINITY, hi6 = a.relu6 ? 6.0f : INFINITY;
    const long long n = a.n;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row0 = p0 + rh * 64 + 32 * i + 4 * half;
        const int left = min(32, nrows - row0);
        if (!((store_tiles >> i) & 1u)) continue;                  // (uniform: rows in the unused part of a slot are left untouched)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = (nt0 + cg * NT + nt) * 32 + r;
            if (col < a.n) {
                float* __restrict__ op = a.out + part_out + (out_base + row0) * n + col;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rr = (reg & 3) + 8 * (reg >> 2);
                    if (rr < left) op[rr * n] = __builtin_amdgcn_fmed3f(acc[i][nt][reg], lo6, hi6);      // (no residual form: launcher)
                }
            }
        }
    }
}
static bool conv_rows_bf16x3_supported(const ConvPack& cp) { return cp.cin % 32 == 0 && cp.nt_total % 8 == 0 && cp.nt_total * 32 == ((cp.n + 255) / 256) * 256; }

// ---- the dimensionality-reduction FC (layers.py:98-107; engine option global_bf16x3) as k_conv_rows_bf16x3<false> with K split over blockIdx.z.
// Its activations are stored in the f32 FC kernel's slot order (fc_slot_of_logical: slot 4 g + t of a group of 16 holds logical input 4 t + g) and its
// weights as FcPack [k / 16][n / 16][64 lanes][4]; this makes the split-bf16 pieces in the ROWS' memory order, so the activations need no
// permutation: piece (s, nt, hi | lo, lane (half, j)) element e = W[logical input of memory position 16 s + 8 half + e][column 32 nt + j].
__global__ __launch_bounds__(256) void k_repack_fc_bf16x3(const float* __restrict__ w, int n_in, int n_out, bf16x8* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int steps = n_in >> 4, nts = n_out >> 5, ctiles = n_out >> 4;
    if (idx >= (long long)steps * nts * 64) return;
    const int lane = (int)(idx & 63), nt = (int)((idx >> 6) % nts), s = (int)((idx >> 6) / nts);
    const int half = lane >> 5, j = lane & 31, col = nt * 32 + j, ct = col >> 4, c16 = col & 15;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int m = 8 * half + e, g = m >> 2, t = m & 3;      // memory slot m of the group = logical input 4 t + g: FcPack lane 16 g + column, element t
        const float v = w[((((size_t)s * ctiles + ct) * 64) + 16 * g + c16) * 4 + t];
        hi[e] = (__bf16)v;
        lo[e] = (__bf16)(v - (float)hi[e]);
    }
    out[(((size_t)s * nts + nt) * 2 + 0) * 64 + lane] = hi;
    out[(((size_t)s * nts + nt) * 2 + 1) * 64 + lane] = lo;
}
size_t fc_bf16x3_pack_bytes(const FcPack& fc) { return (size_t)(fc.n_in / 16) * (fc.n_out / 32) * 2 * 64 * 16; }
bool fc_bf16x3_supported(const FcPack& fc) { return fc.n_in % (16 * 2 * FC_BF_PARTS) == 0 && fc.n_out % 256 == 0; }
hipError_t launch_repack_fc_bf16x3(const FcPack& fc, void* out, hipStream_t s) {
    if (!fc_bf16x3_supported(fc)) return hipErrorInvalidValue;
    const long long n = (long long)(fc.n_in / 16) * (fc.n_out / 32) * 64;
    hipLaunchKernelGGL(k_repack_fc_bf16x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fc.w, fc.n_in, fc.n_out, (bf16x8*)out);
    return hipGetLastError();
}
// partial[part][frames][n_out] = x[frames][k range of the part] * W (no bias); FC_BF_PARTS parts
hipError_t launch_fc_partials_bf16x3(const float* x, const FcPack& fc, const void* Wb, float* partial, int frames, hipStream_t s) {
    if (!Wb || !fc_bf16x3_supported(fc) || frames <= 0 || (long long)frames * fc.n_in * 4 > 0xffffffffll) return hipErrorInvalidValue;
    ConvArgs a;
    a.A = x; a.W = nullptr; a.bias = nullptr; a.res = nullptr; a.out = partial; a.P = frames; a.cin = fc.n_in; a.n = fc.n_out; a.nt_total = fc.n_out / 32; a.relu6 = 0;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) a.level_tiles[l] = 1;
    a.slot_units = nullptr; a.slot_rows = 0; a.rows_per_unit = 0;
    const Geom g0 = {};
    const TapArgs none = {nullptr, nullptr, 0, nullptr, nullptr};
    hipLaunchKernelGGL((k_conv_rows_bf16x3<false>), dim3((unsigned)((frames + 127) / 128), a.nt_total / 8, FC_BF_PARTS), dim3(256), 0, s, a, (const bf16x8*)Wb, g0, none);
    return hipGetLastError();
}

// 3 x 3 convolution at the distinct tap cells (GATHER) / 1 x 1 convolution on rows, both on split bf16 operands.
// Wb: launch_repack_bf16x3 of `cp`.  Shapes: cin % 16 == 0, nt_total % 4 == 0.
hipError_t launch_conv3x3_cells_bf16x3(const float* A, const ConvPack& cp, const void* Wb, float* out, int relu6, long long kps_stride,
                                       const int* level_keypoints, const Geom& g, const int* cells, const int* n_rows, hipStream_t s) {
    if (cp.taps != 9 || cp.cin % 16 || !cells || !n_rows) return hipErrorInvalidValue;
    ConvArgs a = make_args(A, cp, nullptr, out, 0, relu6);
    const TapArgs ta = {nullptr, nullptr, kps_stride, cells, n_rows};
    if (conv_rows_bf16x3_supported(cp)) {                         // 256-column outputs: rows gathered and split once per workgroup (k_conv_rows_bf16x3)
        long long total = 0;
        for (int l = 0; l < g.n_levels; ++l) {
            if ((long long)g.lv[l].Ho * g.lv[l].Wo * cp.cin * 4 > 0xffffffffll) return hipErrorInvalidValue;
            const int rows = 4 * (int)std::min<long long>(level_keypoints[l], kps_stride);
            a.level_tiles[l] = std::max(1, (rows + 127) / 128);
            total += (long long)g.batch * (cp.nt_total / 8) * a.level_tiles[l];
        }
        if (total <= 0 || total > 0x7fffffffll) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_conv_rows_bf16x3<true>), dim3((unsigned)total), dim3(256), 0, s, a, (const bf16x8*)Wb, g, ta);
        return hipGetLastError();
    }
    const int groups = (cp.nt_total + 3) / 4;
    long long total = 0;
    for (int l = 0; l < g.n_levels; ++l) {
        const int rows = 4 * (int)std::min<long long>(level_keypoints[l], kps_stride);
        a.level_tiles[l] = std::max(1, (rows + 255) / 256);
        total += (long long)g.batch * groups * a.level_tiles[l];
    }
    if (total <= 0 || total > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_conv_bf16x3<true>), dim3((unsigned)total), dim3(256), 0, s, a, (const bf16x8*)Wb, g, ta);
    return hipGetLastError();
}
hipError_t launch_pointwise_bf16x3(const float* A, const ConvPack& cp, const void* Wb, const float* residual, float* out, long long P, int relu6,
                                   hipStream_t s, const int* slot_units, int slot_rows, int rows_per_unit) {
    if (P <= 0) return hipSuccess;
    if (cp.taps != 1 || cp.cin % 8) return hipErrorInvalidValue;
    ConvArgs a = make_args(A, cp, residual, out, P, relu6);
    if (slot_units && slot_rows > 0) { a.slot_units = slot_units; a.slot_rows = slot_rows; a.rows_per_unit = rows_per_unit; }
    const Geom g0 = {};
    const TapArgs none = {nullptr, nullptr, 0, nullptr, nullptr};
    if (conv_rows_bf16x3_supported(cp) && !residual && (long long)P * cp.cin * 4 <= 0xffffffffll && (long long)P * cp.n * 4 <= 0xffffffffll && P >= 4096) {
        hipLaunchKernelGGL((k_conv_rows_bf16x3<false>), dim3((unsigned)((P + 127) / 128), cp.nt_total / 8), dim3(256), 0, s, a, (const bf16x8*)Wb, g0, none);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((k_conv_bf16x3<false>), dim3((unsigned)((P + 255) / 256), (cp.nt_total + 3) / 4), dim3(256), 0, s, a, (const bf16x8*)Wb, g0, none);
    return hipGetLastError();
}

// ---- detector tail on split bf16 operands (engine option scores_bf16x3): k_det_tail with the 1 x 1 conv 128 -> 65 as three products on the
// bf16 matrix pipe.  All of the conv's weight pieces (8 steps x 3 column tiles x {hi, lo} = 48 KB) arrive by LDS-DMA once per workgroup; the
// dustbin column is simply column 0 of the third tile (an MFMA more per step is cheaper here than the vector-ALU chain of the exact form);
// the logits then overlay the weight block in LDS and go through k_det_tail's softmax / depth_to_space, expression for expression.
__global__ __launch_bounds__(256, 2) void k_det_tail_bf16x3(ConvArgs a, const bf16x8* __restrict__ Wb, float* __restrict__ dense, Geom g) {
    constexpr int NTW = 3, STEPS = 8, LP = 65;                 // cin == 128 (launcher)
    __shared__ __attribute__((aligned(16))) unsigned char smem[STEPS * NTW * 2 * 1024];
    const bf16x8* wl = (const bf16x8*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];                           // H, W: cell grid; Ho, Wo: dense map (8H, 8W)
    const int ncells = lv.H * lv.W;
    if ((int)blockIdx.x * 128 >= ncells) return;                // workgroup-uniform
    const int c0 = blockIdx.x * 128 + wave * 32;
    const bool active = c0 < ncells;
    {   // 48 pieces of 1 KB, twelve per wave
        const unsigned wl_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
        const unsigned lane16 = (unsigned)lane * 16u;
#pragma unroll
        for (int j = 0; j < STEPS * NTW * 2 / 4; ++j) {
            const int piece = j * 4 + wave_u;
            const bf16x8* src = Wb + (size_t)piece * 64;
            const unsigned dst = wl_lds + (unsigned)(piece * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(src), "v"(lane16), "s"(dst) : "memory", "m0");
        }
    }
    const long long row_base = lv.in_off + (long long)frame * ncells;
    const float* ap = a.A + (row_base + min(c0 + r, ncells - 1)) * a.cin + half * 8;
    f32x4 av[STEPS][2];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) { av[s][0] = *(const f32x4*)(ap + s * 16); av[s][1] = *(const f32x4*)(ap + s * 16 + 4); }
    f32x16 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const float b = a.bias[nt * 32 + r];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = b;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            bf16x8 ah, al;
            split8(av[s][0], av[s][1], ah, al);
            bf16x8 bh[NTW], bl[NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                bh[nt] = wl[((s * NTW + nt) * 2 + 0) * 64 + lane];
                bl[nt] = wl[((s * NTW + nt) * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[nt], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[nt], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[nt], acc[nt], 0, 0, 0);
        }
    }
    __syncthreads();                                            // every wave is done with the weights: the logits take their place
    if (!active) return;
    float* L = (float*)smem + wave * (32 * LP);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = nt * 32 + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) L[((reg & 3) + 8 * (reg >> 2) + 4 * half) * LP + col] = acc[nt][reg];
    }
    if (r == 0) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) L[((reg & 3) + 8 * (reg >> 2) + 4 * half) * LP + 64] = acc[2][reg];
    }
    asm volatile("" ::: "memory");
    // ---- k_det_tail's softmax / depth_to_space: lane (r, half): cell c0 + r, channels 32 half .. 32 half + 31 (+ the dustbin with the upper half)
    const float* row = L + r * LP + 32 * half;
    float e[33];
#pragma unroll
    for (int k = 0; k < 32; ++k) e[k] = row[k];
    e[32] = half ? row[32] : e[31];
    float mx = e[0];
#pragma unroll
    for (int k = 1; k < 33; ++k) mx = fmaxf(mx, e[k]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
    for (int k = 0; k < 33; ++k) e[k] = hf_expf_c(e[k] - mx);
    float sum = 0.0f;
    if (!half) {
#pragma unroll
        for (int k = 0; k < 32; ++k) sum = sum + e[k];
    }
    sum = __shfl(sum, r, 64);
    if (half) {
#pragma unroll
        for (int k = 0; k < 33; ++k) sum = sum + e[k];
    }
    sum = __shfl(sum, r + 32, 64);
    const int cell = c0 + r;
    if (cell < ncells) {
        const int cy = cell / lv.W, cx = cell - cy * lv.W;
        float* d = dense + lv.out_off + (long long)frame * lv.Ho * lv.Wo + (long long)(cy * 8 + 4 * half) * lv.Wo + cx * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 u, v;
#pragma unroll
            for (int j = 0; j < 4; ++j) { u[j] = e[i * 8 + j] / sum; v[j] = e[i * 8 + 4 + j] / sum; }
            *(f32x4*)(d + (long long)i * lv.Wo) = u;
            *(f32x4*)(d + (long long)i * lv.Wo + 4) = v;
        }
    }
}

bool det_tail_bf16x3_supported(const ConvPack& cp) { return det_tail_supported(cp) && cp.cin == 128; }
hipError_t launch_det_tail_bf16x3(const float* hidden, const ConvPack& cp, const void* Wb, float* dense, const Geom& g, hipStream_t s) {
    if (!Wb || !det_tail_bf16x3_supported(cp)) return hipErrorInvalidValue;
    int maxcells = 0;
    for (int l = 0; l < g.n_levels; ++l) maxcells = max(maxcells, g.lv[l].H * g.lv[l].W);
    if (maxcells <= 0) return hipSuccess;
    ConvArgs a = make_args(hidden, cp, nullptr, nullptr, 0, 0);
    hipLaunchKernelGGL(k_det_tail_bf16x3, dim3((unsigned)((maxcells + 127) / 128), (unsigned)(g.n_levels * g.batch)), dim3(256), 0, s, a, (const bf16x8*)Wb, dense, g);
    return hipGetLastError();
}

// ---- dense 3 x 3 convolution on split bf16 operands (engine option scores_bf16x3: the detector head's 96 -> 128 conv, the launch a
// call spends most of its time in).  k_conv_bf16x3<GATHER> above fetches every activation row from memory once per TAP and splits it
// once per tap (nine times each), 32 bytes per lane and k-step straight from L1 / L2: it is bound by that gather.  A DENSE map has
// what the gather lacks -- neighbours share their windows -- so here a workgroup stages the halo of its 256 consecutive pixels
// through LDS, ALREADY SPLIT: every input value crosses L2 -> CU once per workgroup, is split once, and the nine taps of a k-step
// read their A fragments from LDS.
//   K order: chunk-major (16 input channels at a time: one 16-k step per tap), the tolerance mode's own order -- LDS holds one
//     chunk of the halo ([cell][hi 16 | lo 16 | pad]: 80 bytes per cell, conflict-free ds_read_b128 for 32 consecutive cells).
//   Halo in PADDED linear coordinates q = (y + 1) (W + 2) + (x + 1) of a map with a zero border: a tap is a uniform offset
//     (ky - 1) (W + 2) + (kx - 1) for every pixel, border cells are written as zeros by the staging pass, the main loop has no
//     per-lane mask or select at all; a tile of 256 pixels needs the cells q(first) - (W + 3) .. q(last) + (W + 3).
//   Weights: launch_repack_bf16x3's pieces (step s = tap * steps_per_tap + chunk), two taps (16 KB) per slab through LDS, double
//     buffered, one barrier per slab; the next chunk's activations are requested during the chunk's last slab into registers and
//     split + written between the two barriers of the chunk boundary (the second workgroup of the CU computes meanwhile).
//   Wave tile 64 pixels x 128 columns (eight 32 x 32 accumulators): 24 MFMAs per tap against 4 + 8 ds_read_b128.
// CELLS: LDS capacity in halo cells (the launcher checks the geometry against it).
template <int CELLS>
__global__ __launch_bounds__(256, 2) void k_conv3x3_dense_bf16x3(ConvArgs a, const bf16x8* __restrict__ Wb, Geom g) {
    constexpr int NT = 4, MT = 2, REC = 80, TS = 2, NP = CELLS * 4 / 256;     // NP: 16-byte activation pieces per thread and chunk
    __shared__ __attribute__((aligned(16))) unsigned char As[CELLS * REC];
    __shared__ __attribute__((aligned(16))) bf16x8 wl[2][TS * NT * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    const int G = a.nt_total / NT;
    // XCD mapping as k_conv3x3_wlds: every XCD takes one contiguous eighth of the tile list (vertical neighbours share an L2)
    const int q8 = (int)gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int level = 0, rest = slot < q8 ? xcd * q8 + slot : 8 * q8 + xcd;
    for (; level < g.n_levels - 1; ++level) {
        const int per = g.batch * G * a.level_tiles[level];
        if (rest < per) break;
        rest -= per;
    }
    const int tl = a.level_tiles[level];
    const int frame = rest / (G * tl);
    rest -= frame * G * tl;
    const int grp = rest / tl, tile = rest - grp * tl;
    const LevelGeom lv = g.lv[level];
    const int Hc = lv.H, Wc = lv.W, Wp = Wc + 2, nrows = Hc * Wc;
    const int p0 = tile * 256;
    if (p0 >= nrows) return;                                      // (workgroup-uniform)
    const int nt0 = grp * NT;
    const int y0 = p0 / Wc, x0 = p0 - y0 * Wc;
    const int plast = min(p0 + 255, nrows - 1), yl = plast / Wc, xl = plast - yl * Wc;
    const int qbase = y0 * Wp + x0;                               // padded index of the first staged cell: q(p0) - (Wp + 1)
    const int ncell = (yl + 2) * Wp + xl + 2 - qbase + 1;         // ... up to q(plast) + Wp + 1   (<= CELLS: launcher)
    const char* __restrict__ xb = (const char*)(a.A + (lv.in_off + (long long)frame * nrows) * a.cin);      // uniform; lane offsets are 32-bit
    // ---- staging role: piece e = tid + 256 k is 16 bytes (4 channels) of cell e >> 2
    unsigned aoff[NP];
    unsigned okbits = 0, wrbits = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int e = tid + 256 * k, j = e >> 2, pc = e & 3;
        const int q = qbase + j, yq = q / Wp, xq = q - yq * Wp;
        const int y = yq - 1, x = xq - 1;
        const bool in = j < ncell && y >= 0 && y < Hc && x >= 0 && x < Wc;
        aoff[k] = in ? ((unsigned)(y * Wc + x) * (unsigned)a.cin + (unsigned)(pc * 4)) * 4u : 0u;
        if (in) okbits |= 1u << k;
        if (j < ncell) wrbits |= 1u << k;
    }
    f32x4 areg[NP];
    auto load_a = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < NP; ++k) areg[k] = *(const f32x4*)(xb + aoff[k] + chunk * 64);     // (outside cells re-read the image's first pixel: discarded)
    };
    auto write_a = [&]() {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if ((wrbits >> k) & 1u) {
                const int e = tid + 256 * k;
                const bool ok = (okbits >> k) & 1u;
                bf16x4 hi, lo;
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                split4(ok ? areg[k] : z4, hi, lo);
                unsigned char* dst = As + (e >> 2) * REC + (e & 3) * 8;
                *(bf16x4*)dst = hi;
                *(bf16x4*)(dst + 32) = lo;
            }
        }
    };
    // ---- this lane's two pixels: LDS offset of the window CENTRE's hi fragment
    int abase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int p = min(p0 + wave * 64 + 32 * i + r, nrows - 1);
        const int y = p / Wc, x = p - y * Wc;
        abase[i] = ((y + 1) * Wp + (x + 1) - qbase) * REC + half * 16;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float b = a.bias[(nt0 + nt) * 32 + r];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][nt][e] = b;
    }
    const int spt = a.cin >> 4;                                   // 16-k steps per tap == chunks
    // weight pieces of one slab: TS taps x 4 column tiles x {hi, lo} = 16 pieces of 1 KB; wave w moves pieces j * 4 + w straight from
    // memory into LDS (global_load_lds_dwordx4: destination = a wave-uniform base + 16 lane; no registers, no ds_write).  Ordering: a
    // slab's pieces are requested at the top of the slab BEFORE it, into the buffer whose last readers passed the previous barrier; every
    // wave waits for its own requests (vmcnt(0)) in front of the slab's closing barrier; the reads come after that barrier.
    // (The instruction is written as inline assembly: behind the compiler's own builtin every later ds_read of ANY LDS array waits for
    //  vmcnt(0) -- the request that was just issued for the NEXT slab -- and the latency of the weight stream is exposed once per slab.
    //  The compiler does not count these requests; its own vmcnt waits for the ordinary loads can then only be stricter than needed.)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned wl_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&wl[0][0];
    const unsigned lane16 = (unsigned)lane * 16u;
    auto load_w = [&](int chunk, int tap0, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = j * 4 + wave_u, tt = piece >> 3, nt = (piece >> 1) & 3, hl = piece & 1;
            const int s = min(tap0 + tt, 8) * spt + chunk;
            const bf16x8* src = Wb + (((size_t)s * a.nt_total + nt0 + nt) * 2 + hl) * 64;      // uniform
            const unsigned dst = wl_lds + (unsigned)((buf * TS * NT * 2 + piece) * 1024);       // uniform
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(src), "v"(lane16), "s"(dst) : "memory", "m0");
        }
    };
    load_a(0);
    load_w(0, 0, 0);
    write_a();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int chunk = 0; chunk < spt; ++chunk) {
#pragma unroll
        for (int sl = 0; sl < 5; ++sl) {                          // slabs of taps {0,1} {2,3} {4,5} {6,7} {8}
            const bool last_slab = sl == 4;
            const bool more = !(last_slab && chunk + 1 == spt);
            // (the activation requests first: the compiler guards their registers with vmcnt waits that would otherwise stall on the weight requests)
            if (last_slab && chunk + 1 < spt) load_a(chunk + 1);
            if (more) load_w(last_slab ? chunk + 1 : chunk, last_slab ? 0 : 2 * sl + 2, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < TS; ++tt) {
                const int tap = 2 * sl + tt;
                if (tap < 9) {
                    const int toff = ((tap / 3 - 1) * Wp + (tap % 3 - 1)) * REC;      // uniform
                    bf16x8 ah[MT], al[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        ah[i] = *(const bf16x8*)(As + abase[i] + toff);
                        al[i] = *(const bf16x8*)(As + abase[i] + toff + 32);
                    }
                    // two column tiles at a time (their hi / lo fragments: 16 registers): the three products of one accumulator are four
                    // MFMAs apart
#pragma unroll
                    for (int np = 0; np < NT; np += 2) {
                        bf16x8 bh[2], bl[2];
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2) {
                            bh[n2] = wl[buf][((tt * NT + np + n2) * 2 + 0) * 64 + lane];
                            bl[n2] = wl[buf][((tt * NT + np + n2) * 2 + 1) * 64 + lane];
                        }
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                            for (int i = 0; i < MT; ++i) acc[i][np + n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[n2], acc[i][np + n2], 0, 0, 0);
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                            for (int i = 0; i < MT; ++i) acc[i][np + n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[n2], acc[i][np + n2], 0, 0, 0);
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                            for (int i = 0; i < MT; ++i) acc[i][np + n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[n2], acc[i][np + n2], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's weight pieces (and, at a chunk's end, its activation pieces) have landed
            __syncthreads();
            if (last_slab && chunk + 1 < spt) {                    // every wave is done with this chunk's halo: the next one's goes in
                write_a();
                __syncthreads();
            }
            buf ^= 1;
        }
    }
    // ---- (ReLU6) and store
    const float lo6 = a.relu6 ? 0.0f : -INFINITY, hi6 = a.relu6 ? 6.0f : INFINITY;
    float* __restrict__ ob = a.out + (lv.out_off + (long long)frame * nrows) * a.n;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row0 = p0 + wave * 64 + 32 * i + 4 * half;
        const int left = nrows - row0;                             // rows of this lane's column that exist (may be <= 0)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = (nt0 + nt) * 32 + r;
            if (col < a.n) {
                float* __restrict__ op = ob + (long long)row0 * a.n + col;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rr = (reg & 3) + 8 * (reg >> 2);
                    if (rr < left) op[rr * a.n] = __builtin_amdgcn_fmed3f(acc[i][nt][reg], lo6, hi6);
                }
            }
        }
    }
}

bool conv3x3_dense_bf16x3_supported(const ConvPack& cp, const Geom& g) {
    if (cp.taps != 9 || cp.cin % 16 || cp.nt_total % 4) return false;
    for (int l = 0; l < g.n_levels; ++l) {
        const int Wc = g.lv[l].W, Wp = Wc + 2;
        if (Wc < 1 || g.lv[l].H < 1) return false;
        const int span = (255 + Wc - 1) / Wc;                       // image rows a tile's last pixel can lie below its first
        if (255 + 2 * span + 2 * Wp + 3 > 512) return false;         // halo cells of the worst tile against the kernel's LDS block
        if ((long long)g.lv[l].H * Wc * cp.cin * 4 > 0xffffffffll) return false;     // 32-bit lane offsets inside one image
    }
    return true;
}

// out[pixel][n] = act(3 x 3 conv of A) for every pixel of every level / frame, split-bf16 operands (Wb: launch_repack_bf16x3 of cp)
hipError_t launch_conv3x3_dense_bf16x3(const float* A, const ConvPack& cp, const void* Wb, float* out, int relu6, const Geom& g, hipStream_t s) {
    if (!Wb || !conv3x3_dense_bf16x3_supported(cp, g)) return hipErrorInvalidValue;
    ConvArgs a = make_args(A, cp, nullptr, out, 0, relu6);
    long long total = 0;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) {
        a.level_tiles[l] = l < g.n_levels ? std::max((g.lv[l].H * g.lv[l].W + 255) / 256, 1) : 1;
        if (l < g.n_levels) total += (long long)g.batch * (cp.nt_total / 4) * a.level_tiles[l];
    }
    if (total <= 0 || total > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_conv3x3_dense_bf16x3<512>), dim3((unsigned)total), dim3(256), 0, s, a, (const bf16x8*)Wb, g);
    return hipGetLastError();
}

// =========================================================================== depthwise 3x3
// One thread = 4 channels x a vertical strip of R output pixels of one column: the (R s + 2) x 3 input pieces and the nine
// weight pieces are loaded once and serve R outputs (a thread per output re-read every input piece up to nine times
// through L1, which bounded the kernel).  Consecutive threads take consecutive channel quads of a pixel: 16-byte pieces
// of one contiguous row.  Out-of-image taps contribute fma(0, w, acc) == acc: the oracle skips them.  Per output the
// chain is bias, then the taps in (ky, kx) order.
template <int STRIDE, int R>
__global__ __launch_bounds__(256) void k_depthwise(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ out, int C, Geom g) {
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int c4 = C >> 2, strips = (lv.Ho + R - 1) / R;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)strips * lv.Wo * c4) return;
    const int sp = (int)(idx / c4), cq = (int)(idx - (long long)sp * c4);
    const int st = sp / lv.Wo, ox = sp - st * lv.Wo, oy0 = st * R;
    const float* ip = in + (lv.in_off + (long long)frame * lv.H * lv.W) * C + cq * 4;
    constexpr int NR = (R - 1) * STRIDE + 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 x[NR][3];
#pragma unroll
    for (int ry = 0; ry < NR; ++ry) {
        const int iy = oy0 * STRIDE - lv.pt + ry;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * STRIDE - lv.pl + kx;
            const bool ok = iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W;
            x[ry][kx] = ok ? *(const f32x4*)(ip + (long long)(iy * lv.W + ix) * C) : zero;
        }
    }
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *(const f32x4*)(w + t * C + cq * 4);
    const f32x4 bv = *(const f32x4*)(bias + cq * 4);
    float* op = out + (lv.out_off + (long long)frame * lv.Ho * lv.Wo) * C + cq * 4;
#pragma unroll
    for (int o = 0; o < R; ++o) {
        if (oy0 + o >= lv.Ho) break;
        f32x4 acc = bv;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(x[o * STRIDE + ky][kx][j], wv[ky * 3 + kx][j], acc[j]);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = relu6f(acc[j]);
        *(f32x4*)(op + (long long)((oy0 + o) * lv.Wo + ox) * C) = r;
    }
}

hipError_t launch_depthwise(const float* in, const DwPack& dp, int stride, float* out, const Geom& g, hipStream_t s) {
    constexpr int R = 5;                                      // (15-row maps of the global branch: three strips)
    long long maxwork = 0;
    for (int l = 0; l < g.n_levels; ++l) maxwork = max(maxwork, (long long)((g.lv[l].Ho + R - 1) / R) * g.lv[l].Wo * (dp.c / 4));
    dim3 grid((unsigned)((maxwork + 255) / 256), g.n_levels * g.batch);
    if (stride == 1) hipLaunchKernelGGL((k_depthwise<1, R>), grid, dim3(256), 0, s, in, dp.w, dp.bias, out, dp.c, g);
    else if (stride == 2) hipLaunchKernelGGL((k_depthwise<2, R>), grid, dim3(256), 0, s, in, dp.w, dp.bias, out, dp.c, g);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// =========================================================================== channel permutation
__global__ __launch_bounds__(256) void k_permute_channels(const float* __restrict__ in, float* __restrict__ out, long long total, int C,
                                                          int to_logical) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long p = i / C;
    const int c = (int)(i - p * C);
    const int rr = c & 7;
    // physical slot of logical channel c, and logical channel held by physical slot c
    const int phys = (c & ~7) | ((c & 1) << 2) | (rr >> 1);
    const int logi = (c & ~7) | (rr < 4 ? 2 * rr : 2 * (rr - 4) + 1);
    out[i] = in[p * C + (to_logical ? phys : logi)];
}

hipError_t launch_permute_channels(const float* in, float* out, long long P, int C, int to_logical, hipStream_t s) {
    const long long total = P * C;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_permute_channels, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, C, to_logical);
    return hipGetLastError();
}

}  // namespace hfnet
