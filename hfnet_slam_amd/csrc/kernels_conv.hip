// kernels_conv.hip -- backbone kernels for gfx950 (CDNA4): pyramid resize, stem, 1x1 / 3x3
// convolutions on v_mfma_f32_32x32x2_f32, depthwise 3x3.
//
// Numerics: every accumulation is the fused multiply-add chain the oracle defines
// (oracle/hfnet_oracle.h): the f32 MFMA is bit-for-bit a k-ordered fmaf chain, the vector
// kernels use explicit fmaf.  Built with -ffp-contract=off.
//
// Data layout: activations are [pixel][channel] fp32 with the channels of each group of 8 in the
// "physical" order of common.hpp, levels and frames concatenated ([level][frame][y][x][c]).
#include "kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace hfnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

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

hipError_t launch_resize_u8(const uint8_t* src, int sw, int sh, int s_row, long long s_frame, uint8_t* dst, int dw, int dh,
                            int d_row, long long d_frame, const int* xofs, const short* ialpha, const int* yofs,
                            const short* ibeta, int batch, hipStream_t s) {
    if ((d_row & 3) || (d_frame & 3) || ((size_t)dst & 3)) return hipErrorInvalidValue;      // packed 32-bit stores
    dim3 grid(((dw + 3) / 4 + 255) / 256, (dh + RESIZE_ROWS - 1) / RESIZE_ROWS, batch);
    hipLaunchKernelGGL(k_resize_u8, grid, dim3(256), 0, s, src, sw, sh, s_row, s_frame, dst, dw, dh, d_row, d_frame, xofs, ialpha, yofs, ibeta);
    return hipGetLastError();
}

// =========================================================================== stem
// u8 -> (x-128)/128 -> crop to multiples of 8 (Geom carries the cropped size) -> conv 3x3 stride 2
// 1 -> cout, BN, ReLU6.  One thread per output pixel, weights in LDS.  HBM-bound on the output write.
__global__ __launch_bounds__(256) void k_stem(ImageSet imgs, const float* __restrict__ w, const float* __restrict__ scale,
                                              const float* __restrict__ shift, int cout, float* __restrict__ out, Geom g) {
    __shared__ float sw[9 * 64];
    __shared__ float ssc[64], ssh[64];
    for (int i = threadIdx.x; i < 9 * cout; i += 256) sw[i] = w[i];
    if (threadIdx.x < cout) { ssc[threadIdx.x] = scale[threadIdx.x]; ssh[threadIdx.x] = shift[threadIdx.x]; }
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
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(px[t], sw[t * cout + c + j], acc);
            r[j] = relu6f(fmaf(acc, ssc[c + j], ssh[c + j]));
        }
        *(f32x4*)(op + c) = r;
    }
}

// Channel count known at compile time: the weights are wave-uniform loads with constant offsets (scalar cache, SGPR
// fma operands, no LDS reads), and the workgroup's 256 pixels x COUT channels -- one contiguous block of the output
// tensor -- go through LDS so that the global stores are 16 bytes per lane at consecutive addresses (a thread's own
// pixel is COUT*4 bytes from its neighbour's: written directly, every store instruction touches 64 cache lines).
template <int COUT>
__global__ __launch_bounds__(256) void k_stem_c(ImageSet imgs, const float* __restrict__ w, const float* __restrict__ scale,
                                                const float* __restrict__ shift, float* __restrict__ out, Geom g) {
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
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(px[t], w[t * COUT + c + j], acc);
            r[j] = relu6f(fmaf(acc, scale[c + j], shift[c + j]));
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

hipError_t launch_stem(const ImageSet& imgs, const float* w, const float* scale, const float* shift, int cout, float* out,
                       const Geom& g, hipStream_t s) {
    int maxpix = 0;
    for (int l = 0; l < g.n_levels; ++l) maxpix = max(maxpix, g.lv[l].Ho * g.lv[l].Wo);
    dim3 grid((maxpix + 255) / 256, g.n_levels * g.batch);
    if (cout == 24) hipLaunchKernelGGL(k_stem_c<24>, grid, dim3(256), 0, s, imgs, w, scale, shift, out, g);
    else hipLaunchKernelGGL(k_stem, grid, dim3(256), 0, s, imgs, w, scale, shift, cout, out, g);
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
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    long long P;      // rows (pointwise) -- unused by the 3x3 kernel
    int cin;
    int n;            // valid output columns == output row stride
    int nt_total;
    int relu6;
    int level_tiles[HFNET_MAX_LEVELS];   // 3x3 kernel: 128-row tiles launched per image of each level (exact 1-D grid)
};

// BN (+ ReLU6) (+ residual) and store of a wave's 32 x (NT*32) accumulator tile.  VALU instructions compete
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
                const float sc = a.scale[col], sh = a.shift[col];
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
                        float v = fmaf(acc[nt][reg], sc, sh);
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

template <int NT>
__global__ __launch_bounds__(256, 2) void k_pointwise(ConvArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    const long long row0 = (long long)blockIdx.x * 128 + wave * 32;
    if (row0 >= a.P) return;
    const int nt0 = blockIdx.y * NT;
    long long row = row0 + r;
    if (row >= a.P) row = a.P - 1;
    const float* ap = a.A + row * a.cin + half * 4;
    const f32x4* wp = a.W + ((size_t)nt0 * 64 + lane);
    const size_t wstep = (size_t)a.nt_total * 64;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
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
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
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
struct TapArgs { const hfnet_keypoint* kps; const int* n_in; long long kps_stride; };

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
    const int nrows = GATHER ? ta.n_in[image] * 4 : Hc * Wc;
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
    if (GATHER) {
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
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
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

template <int NT>
static void launch_pw_nt(const ConvArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(k_pointwise<NT>, grid, dim3(256), 0, s, a);
}
template <int NT>
static void launch_c3_nt(const ConvArgs& a, const Geom& g, const TapArgs* ta, dim3 grid, hipStream_t s) {
    if (ta) hipLaunchKernelGGL((k_conv3x3<NT, true>), grid, dim3(256), 0, s, a, g, *ta);
    else { TapArgs none = {nullptr, nullptr, 0}; hipLaunchKernelGGL((k_conv3x3<NT, false>), grid, dim3(256), 0, s, a, g, none); }
}

static ConvArgs make_args(const float* A, const ConvPack& cp, const float* res, float* out, long long P, int relu6) {
    ConvArgs a;
    a.A = A; a.W = (const f32x4*)cp.w; a.scale = cp.scale; a.shift = cp.shift; a.res = res; a.out = out;
    a.P = P; a.cin = cp.cin; a.n = cp.n; a.nt_total = cp.nt_total; a.relu6 = relu6;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) a.level_tiles[l] = 1;
    return a;
}

// column tiles per wave: the packed layout allows any divisor of nt_total; small-M layers (the
// 30x47 / 15x24 global branch) take fewer tiles per wave so that the launch still fills 256 CUs
static int pick_nt(int nt_total, int nt_pref, long long m_tiles) {
    static const int max_nt = []() { const char* v = getenv("HFNET_MAX_NT"); return v ? atoi(v) : 4; }();   // measured on MI355X: <= 4 column tiles per wave (higher occupancy) beats 8
    int best = 1;
    for (int nt = 1; nt <= nt_pref && nt <= max_nt; ++nt) {
        if (nt_total % nt) continue;
        if (m_tiles * (nt_total / nt) >= 2048 || nt == 1) best = nt;
    }
    return best;
}

hipError_t launch_pointwise(const float* A, const ConvPack& cp, const float* residual, float* out, long long P, int relu6,
                            hipStream_t s) {
    if (P <= 0) return hipSuccess;
    const ConvArgs a = make_args(A, cp, residual, out, P, relu6);
    const int nt = pick_nt(cp.nt_total, cp.nt_per_block, (P + 31) / 32);
    // long k chains on few tiles: latency-bound, see k_pointwise_deep
    static const long long lowlat_waves = []() { const char* v = getenv("HFNET_PWD_WAVES"); return v ? atoll(v) : 1024ll; }();
    if (nt <= 2 && cp.cin >= 192 && (P + 31) / 32 * cp.nt_total < lowlat_waves) {
        dim3 grid((unsigned)((P + 127) / 128), cp.nt_total / nt);
        if (nt == 1) hipLaunchKernelGGL((k_pointwise_deep<1, 8>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_pointwise_deep<2, 8>), grid, dim3(256), 0, s, a);
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

static hipError_t launch_conv3x3_any(const float* A, const ConvPack& cp, float* out, int relu6, const Geom& g, const TapArgs* ta,
                                     const int* level_rows, hipStream_t s) {
    ConvArgs a = make_args(A, cp, nullptr, out, 0, relu6);
    int ntb = cp.nt_per_block;
    if (!ta) { static const int t = []() { const char* v = getenv("HFNET_CONV3_NT"); return v ? atoi(v) : 0; }(); if (t > 0 && cp.nt_total % t == 0) ntb = t; }
    if (ta) { static const int t = []() { const char* v = getenv("HFNET_TAPS_NT"); return v ? atoi(v) : 4; }(); if (t > 0 && cp.nt_total % t == 0) ntb = t; }
    long long tiles = 0;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) {
        a.level_tiles[l] = l < g.n_levels ? std::max((level_rows[l] + 127) / 128, 1) : 1;
        if (l < g.n_levels) tiles += (long long)g.batch * a.level_tiles[l];
    }
    // small batches: a wave's 3x3 chain over 9 * cin is ~50 us long with four column tiles, and a single frame has only
    // ~110 row tiles -- take fewer column tiles per wave until the launch has two workgroups per CU (latency, not throughput)
    static const int min_wgs = []() { const char* v = getenv("HFNET_CONV3_MIN_WGS"); return v ? atoi(v) : 512; }();
    while (ntb > 1 && tiles * (cp.nt_total / ntb) < min_wgs) {
        int next = ntb - 1;
        while (next > 1 && cp.nt_total % next) --next;
        ntb = next;
    }
    const long long wgs = tiles * (cp.nt_total / ntb);
    if (wgs <= 0 || wgs > 0x7fffffffll) return hipErrorInvalidValue;
    dim3 grid((unsigned)wgs, 1, 1);
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

hipError_t launch_conv3x3(const float* A, const ConvPack& cp, float* out, int relu6, const Geom& g, hipStream_t s) {
    int rows[HFNET_MAX_LEVELS] = {0};
    for (int l = 0; l < g.n_levels; ++l) rows[l] = g.lv[l].H * g.lv[l].W;
    return launch_conv3x3_any(A, cp, out, relu6, g, nullptr, rows, s);
}

hipError_t launch_conv3x3_taps(const float* A, const ConvPack& cp, float* out, int relu6, const hfnet_keypoint* kps, const int* n_in,
                               long long kps_stride, const int* level_keypoints, const Geom& g, hipStream_t s) {
    const TapArgs ta = {kps, n_in, kps_stride};
    int rows[HFNET_MAX_LEVELS] = {0};
    for (int l = 0; l < g.n_levels; ++l) rows[l] = 4 * (int)std::min<long long>(level_keypoints[l], kps_stride);
    return launch_conv3x3_any(A, cp, out, relu6, g, &ta, rows, s);
}

// =========================================================================== fused inverted-residual block
// conv_blocks.py:163-312 in ONE launch: [1x1 expand + BN + ReLU6] -> depthwise 3x3 + BN + ReLU6 ->
// 1x1 project + BN [+ input].  The expanded tensor (6x the block input, written and read twice by the
// unfused chain) never leaves the CU: per 32-channel chunk of the expansion
//   stage 1  MFMA: expand the (TH*s+2) x (TW*s+2) halo tile of the input into LDS (out-of-image halo = 0,
//            which is what 'SAME' padding of the depthwise conv sees)
//   stage 2  VALU: depthwise 3x3 from LDS to LDS
//   stage 3  MFMA: accumulate the chunk into the projection (k-order = expansion channel order, chunks in
//            order, so the chain is the oracle's)
// Algorithmic HBM traffic per block drops from in + 4*expanded + out to in*(halo) + out.
struct FusedArgs {
    const float* X;
    const f32x4* Wex; const float* ex_scale; const float* ex_shift; int ex_nt_total;
    const float* Wdw; const float* dw_scale; const float* dw_shift;
    const f32x4* Wpr; const float* pr_scale; const float* pr_shift; int pr_nt_total;
    float* out;
    int cin, cexp, cout, residual, has_expand;
    int ablate;   // diagnostics: bit0 skip stage 1, bit1 skip stage 2, bit2 skip stage 3 (results are then wrong)
};

template <int STRIDE, int TH, int TW, int NTO>
__global__ __launch_bounds__(256, 2) void k_block_fused(FusedArgs a, Geom g) {
    constexpr int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3, IPIX = IH * IW, OPIX = TH * TW, CEP = 36;
    constexpr int MT_IN = (IPIX + 31) / 32, MT_OUT = OPIX / 32;
    static_assert(OPIX % 32 == 0 && MT_OUT <= 4, "output tile must be 32..128 pixels");
    __shared__ __attribute__((aligned(16))) float E[IPIX * CEP];
    __shared__ __attribute__((aligned(16))) float D[OPIX * CEP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int tiles_x = (lv.Wo + TW - 1) / TW, tiles_y = (lv.Ho + TH - 1) / TH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * STRIDE - lv.pt, ix0 = ox0 * STRIDE - lv.pl;
    const long long in_base = lv.in_off + (long long)frame * lv.H * lv.W;
    const long long out_base = lv.out_off + (long long)frame * lv.Ho * lv.Wo;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 pacc[NTO];
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[nt][i] = 0.0f;
    const int n_chunks = a.has_expand ? a.ex_nt_total : 1;
    const int KQ = a.cin >> 3;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int ch0 = chunk * 32;
        if (ch0 >= a.cexp) break;                       // all-padding column tile
        // ---- stage 1: expansion of the halo tile -> E
        if (a.ablate & 1) {
        } else if (a.has_expand) {
            const float sc = a.ex_scale[ch0 + r], sh = a.ex_shift[ch0 + r];
            for (int mt = wave; mt < MT_IN; mt += 4) {
                const int hp = mt * 32 + r;
                const int hy = hp / IW, hx = hp - hy * IW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                const bool ok = hp < IPIX && iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W;
                const float* ap = a.X + (in_base + (long long)(ok ? iy * lv.W + ix : 0)) * a.cin + half * 4;
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
                for (int kq = 0; kq < KQ; ++kq) {
                    f32x4 av = zero4;
                    if (ok) av = *(const f32x4*)(ap + kq * 8);
                    const f32x4 bv = a.Wex[((size_t)kq * a.ex_nt_total + chunk) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int hp2 = mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    if (hp2 < IPIX) {
                        const int hy2 = hp2 / IW, hx2 = hp2 - hy2 * IW;
                        const int iy2 = iy0 + hy2, ix2 = ix0 + hx2;
                        const bool in2 = iy2 >= 0 && iy2 < lv.H && ix2 >= 0 && ix2 < lv.W;
                        E[hp2 * CEP + r] = in2 ? relu6f(fmaf(acc[reg], sc, sh)) : 0.0f;
                    }
                }
            }
        } else {
            for (int idx = threadIdx.x; idx < IPIX * 8; idx += 256) {
                const int hp = idx >> 3, c4 = idx & 7;
                const int hy = hp / IW, hx = hp - hy * IW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                f32x4 v = zero4;
                if (c4 * 4 < a.cexp && iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W)
                    v = *(const f32x4*)(a.X + (in_base + (long long)iy * lv.W + ix) * a.cin + c4 * 4);
                *(f32x4*)(E + hp * CEP + c4 * 4) = v;
            }
        }
        __syncthreads();
        // ---- stage 2: depthwise 3x3 + BN + ReLU6, E -> D
        if (!(a.ablate & 2))
        for (int idx = threadIdx.x; idx < OPIX * 8; idx += 256) {
            const int op = idx >> 3, c4 = idx & 7;
            const int c = ch0 + c4 * 4;
            f32x4 o = zero4;
            if (c < a.cexp) {
                const int oy = op / TW, ox = op - oy * TW;
                f32x4 acc = zero4;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 ev = *(const f32x4*)(E + ((oy * STRIDE + ky) * IW + ox * STRIDE + kx) * CEP + c4 * 4);
                        const f32x4 wv = *(const f32x4*)(a.Wdw + (ky * 3 + kx) * a.cexp + c);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = fmaf(ev[j], wv[j], acc[j]);
                    }
                const f32x4 dsc = *(const f32x4*)(a.dw_scale + c), dsh = *(const f32x4*)(a.dw_shift + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = relu6f(fmaf(acc[j], dsc[j], dsh[j]));
            }
            *(f32x4*)(D + op * CEP + c4 * 4) = o;
        }
        __syncthreads();
        // ---- stage 3: projection, accumulate this chunk's channels
        if (wave < MT_OUT && !(a.ablate & 4)) {
            const int kqc = min(4, (a.cexp - ch0) >> 3);
            for (int kq = 0; kq < kqc; ++kq) {
                const f32x4 av = *(const f32x4*)(D + (wave * 32 + r) * CEP + kq * 8 + half * 4);
                f32x4 bv[NTO];
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) bv[nt] = a.Wpr[((size_t)(chunk * 4 + kq) * a.pr_nt_total + nt) * 64 + lane];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nt][t], pacc[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- epilogue: BN (+ residual), store
    if (wave < MT_OUT) {
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            const int col = nt * 32 + r;
            if (col >= a.cout) continue;
            const float sc = a.pr_scale[col], sh = a.pr_shift[col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int op = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                const int oy = oy0 + op / TW, ox = ox0 + op % TW;
                if (oy >= lv.Ho || ox >= lv.Wo) continue;
                float v = fmaf(pacc[nt][reg], sc, sh);
                if (a.residual) v = v + a.X[(in_base + (long long)oy * lv.W + ox) * a.cin + col];
                a.out[(out_base + (long long)oy * lv.Wo + ox) * a.cout + col] = v;
            }
        }
    }
}

// ---- v2 of the fused block for the shapes of the high-resolution layers (cin = 8*KQT known at
// compile time).  Differences to the generic kernel above:
//  * the expanded halo tile is kept channel-major in LDS (ET[channel][padded position]); the MFMA
//    D fragment (lane = channel, 4 consecutive rows per register quad) goes out as ds_write_b128;
//  * the depthwise stage runs one thread per (channel, output row): three input rows are read once as
//    16-byte LDS loads and slide along x in registers; the 9 taps + BN live in registers per chunk;
//  * the block input (A fragments of every halo M-tile of the wave) is loaded once and stays in
//    registers across chunks; the chunk's expand weights are loaded once per chunk, not per M-tile.
struct TileSplit { int first[4], count[4], maxc; };
// halo M-tiles per MFMA wave: the split that minimises the largest per-wave MFMA count (stage-1 MFMAs + the
// stage-3 MFMAs of the waves that own an output tile); ties go to the smaller register footprint (max tiles per wave)
constexpr TileSplit split_tiles(int mt_in, int mt_out, int c1, int c3) {
    TileSplit sp{};
    long best = -1;
    for (int n0 = 0; n0 <= mt_in; ++n0)
        for (int n1 = 0; n0 + n1 <= mt_in; ++n1)
            for (int n2 = 0; n0 + n1 + n2 <= mt_in; ++n2) {
                const int cnt[4] = {n0, n1, n2, mt_in - n0 - n1 - n2};
                int maxload = 0, maxc = 0;
                long sq = 0;
                for (int w = 0; w < 4; ++w) {
                    const int load = cnt[w] * c1 + (w < mt_out ? c3 : 0);
                    if (load > maxload) maxload = load;
                    if (cnt[w] > maxc) maxc = cnt[w];
                    sq += (long)load * load;
                }
                const long key = ((long)maxload * 64 + maxc) * 1000000 + sq;
                if (best < 0 || key < best) {
                    best = key;
                    for (int w = 0; w < 4; ++w) sp.count[w] = cnt[w];
                    sp.maxc = maxc;
                }
            }
    int f = 0;
    for (int w = 0; w < 4; ++w) { sp.first[w] = f; f += sp.count[w]; }
    return sp;
}

// workgroups per CU the register budget is tuned for: three where the LDS tile allows it (f32 MFMA and VALU work
// share one issue pipe, so more resident waves is what hides the LDS / barrier latencies).  "Diet": the widest
// high-resolution block does not keep its input fragments and projection weights in registers across the chunk
// loop (236 VGPRs, 2 workgroups per CU) but re-reads them from L1 / L2 when they are used (<= 168, 3 workgroups).
template <int STRIDE, int NTO, int KQT, int TW>
constexpr bool fused2_diet() { return (STRIDE == 1 && (KQT >= 6 || NTO >= 2)) || KQT >= 12; }    // layers 6, 7, 8, 9-14
template <int STRIDE, int NTO, int KQT, int TW>
constexpr int fused2_min_blocks() {
    return (STRIDE == 2 && TW == 8) || (STRIDE == 1 && KQT <= 3 && NTO == 1) || fused2_diet<STRIDE, NTO, KQT, TW>() ? 3 : 2;
}

template <int STRIDE, int NTO, int KQT, bool HAS_EXPAND, int TW>
__global__ __launch_bounds__(256, (fused2_min_blocks<STRIDE, NTO, KQT, TW>())) void k_block_fused2(FusedArgs a, Geom g) {
    constexpr int TH = 8;
    // halo rows are stored back to back; an even row length keeps the depthwise stage's row reads 8-byte aligned
    // (odd widths get one padding column: fewer halo M-tiles than padding to a multiple of 4)
    constexpr int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3, IWP = (IW + 1) / 2 * 2, NPOS = IH * IWP;
    constexpr int MT_IN = (NPOS + 31) / 32, OPIX = TH * TW, MT_OUT = OPIX / 32, CEP = 36;
    constexpr int EP = ((MT_IN * 32 / 4) % 2 == 1) ? MT_IN * 32 : MT_IN * 32 + 4;   // per-channel stride, EP/4 odd
    // halo M-tiles per wave.  Stage 3 of chunk c runs in the same barrier phase as stage 1 of chunk c+1
    // (see the loop), and only waves < MT_OUT have stage-3 work, so those waves own fewer halo tiles.
    constexpr TileSplit SP = split_tiles(MT_IN, MT_OUT, KQT * 4, NTO * 16);
    constexpr int MTC0 = SP.count[0], MTC1 = SP.count[1], MTC2 = SP.count[2], MTC3 = SP.count[3], MTW = SP.maxc;
    static_assert(MTC0 + MTC1 + MTC2 + MTC3 == MT_IN && MTW <= 5, "halo tile distribution");
    __shared__ __attribute__((aligned(16))) float ET[32 * EP];
    __shared__ __attribute__((aligned(16))) float D[OPIX * CEP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    const int mt_first = wave == 0 ? 0 : wave == 1 ? MTC0 : wave == 2 ? MTC0 + MTC1 : MTC0 + MTC1 + MTC2;
    int mt_count = wave == 0 ? MTC0 : wave == 1 ? MTC1 : wave == 2 ? MTC2 : MTC3;
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int tiles_x = (lv.Wo + TW - 1) / TW, tiles_y = (lv.Ho + TH - 1) / TH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * STRIDE - lv.pt, ix0 = ox0 * STRIDE - lv.pl;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= lv.H && ix0 + IW <= lv.W;
    // Tiles hanging over the bottom edge (the pyramid levels are not multiples of the tile height: up to 30 % of a level's
    // tile area at 1/8 resolution): halo M-tiles below the last needed input row, depthwise rows and projection M-tiles
    // below the last output row are skipped.  Skipped regions of ET / D keep stale values that only ever feed rows of
    // MFMA tiles which are never stored (a row of A only affects the same row of D).
    const int rows_valid = min(TH, lv.Ho - oy0);                               // uniform, >= 1
    {
        const int hy_max = (rows_valid - 1) * STRIDE + 2;                      // last halo row any valid output row reads
        const int n_live = ((hy_max + 1) * IWP + 31) >> 5;                     // halo M-tiles covering positions < (hy_max + 1) * IWP
        mt_count = max(0, min(mt_count, n_live - mt_first));
    }
    const bool out_live = wave < MT_OUT && (wave * 32) / TW < rows_valid;      // this wave's projection tile has a valid row
    const long long in_base = lv.in_off + (long long)frame * lv.H * lv.W;
    const long long out_base = lv.out_off + (long long)frame * lv.Ho * lv.Wo;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 pacc[NTO];
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[nt][i] = 0.0f;
    // A fragments of this wave's halo M-tiles: kept for all chunks, or (diet) only their addresses
    constexpr int KQA = HAS_EXPAND ? KQT : 1;
    constexpr bool DIET = HAS_EXPAND && fused2_diet<STRIDE, NTO, KQT, TW>();
    f32x4 afrag[DIET ? 1 : MTW][KQA];
    const float* aptr[MTW];
    bool aok[MTW];
    if (HAS_EXPAND) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int mt = mt_first + m;
            const int pp = mt * 32 + r;
            const int hy = pp / IWP, hx = pp - hy * IWP;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = m < mt_count && hy < IH && hx < IW && iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W;
            const float* ap = a.X + (in_base + (long long)(ok ? iy * lv.W + ix : 0)) * a.cin + half * 4;
            aptr[m] = ap; aok[m] = ok;
            if (!DIET) {
#pragma unroll
                for (int kq = 0; kq < KQA; ++kq) afrag[m][kq] = ok ? *(const f32x4*)(ap + kq * 8) : zero4;
            }
        }
    }
    const int dc = threadIdx.x & 31, doy = threadIdx.x >> 5;      // depthwise role: channel lane, output row
    const int n_chunks_all = HAS_EXPAND ? a.ex_nt_total : 1;
    const int n_chunks = min(n_chunks_all, (a.cexp + 31) >> 5);   // skip all-padding column tiles

    // ---- stage 1 of one chunk: expansion of the halo tile -> ET (channel-major)
    // expansion weights / BN of a chunk.  With few input channels (KQT <= 3) the next chunk's set is prefetched a whole
    // phase ahead (an L2 hit takes 0.7-1 us here, a phase is 1-2 us); the wide layers have no registers to spare.
    constexpr bool PREFETCH_B = false;    // measured on L03-L06: no gain (the other resident workgroups already cover the wait)
    f32x4 bpre[KQA];
    float scpre = 0.f, shpre = 0.f;
    auto fetch_b = [&](int chunk) {
#pragma unroll
        for (int kq = 0; kq < KQA; ++kq) bpre[kq] = a.Wex[((size_t)kq * a.ex_nt_total + chunk) * 64 + lane];
        scpre = a.ex_scale[chunk * 32 + r]; shpre = a.ex_shift[chunk * 32 + r];
    };
    auto stage1 = [&](int chunk) {
        if (HAS_EXPAND) {
            if (!PREFETCH_B) fetch_b(chunk);
            f32x4 bfrag[KQA];
#pragma unroll
            for (int kq = 0; kq < KQA; ++kq) bfrag[kq] = bpre[kq];
            const float sc = scpre, sh = shpre;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                if (m < mt_count) {
                    const int mt = mt_first + m;
                    f32x4 af[KQA];
#pragma unroll
                    for (int kq = 0; kq < KQA; ++kq) {
                        if (DIET) { const f32x4 v = *(const f32x4*)(aptr[m] + kq * 8); af[kq] = aok[m] ? v : zero4; }
                        else af[kq] = afrag[m][kq];
                    }
                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][0], bfrag[0][0], zero16, 0, 0, 0);
#pragma unroll
                    for (int kq = 0; kq < KQA; ++kq)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (kq + t > 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kq][t], bfrag[kq][t], acc, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pp = mt * 32 + 8 * q + 4 * half;
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = relu6f(fmaf(acc[4 * q + i], sc, sh));
                        *(f32x4*)(ET + r * EP + pp) = v;        // border tiles: out-of-image positions are zeroed by zero_border()
                    }
                }
            }
        } else {
            // no expansion conv: the block input itself is the depthwise input
            for (int idx = threadIdx.x; idx < NPOS * 8; idx += 256) {
                const int pp = idx >> 3, c4 = idx & 7;
                const int hy = pp / IWP, hx = pp - hy * IWP;
                const int iy = iy0 + hy, ix = ix0 + hx;
                f32x4 v = zero4;
                if (c4 * 4 < a.cexp && hx < IW && iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W)
                    v = *(const f32x4*)(a.X + (in_base + (long long)iy * lv.W + ix) * a.cin + c4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) ET[(c4 * 4 + j) * EP + pp] = v[j];
            }
        }
    };

    // tiles that touch the image border: the expansion of an out-of-image halo position must be 0 (the
    // depthwise conv's 'SAME' padding), not relu6(shift).  Kept out of the MFMA epilogue: one extra pass + barrier,
    // executed only by border workgroups.
    auto zero_border = [&]() {
        __syncthreads();
        for (int pp = threadIdx.x; pp < NPOS; pp += 256) {
            const int hy = pp / IWP, hx = pp - hy * IWP;
            const int iy = iy0 + hy, ix = ix0 + hx;
            if (iy < 0 || iy >= lv.H || ix < 0 || ix >= lv.W)
                for (int c = 0; c < 32; ++c) ET[c * EP + pp] = 0.0f;
        }
    };

    if (PREFETCH_B) fetch_b(0);
    if (!(a.ablate & 1)) stage1(0);
    if (HAS_EXPAND && !interior) zero_border();
    __syncthreads();
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int ch0 = chunk * 32;
        if (PREFETCH_B && chunk + 1 < n_chunks) fetch_b(chunk + 1);
        // depthwise taps / BN of this thread's channel and the projection weights of this chunk
        const int dch = ch0 + dc;
        const bool dact = dch < a.cexp;
        float dwt[9], dsc = 0.f, dsh = 0.f;
        if (dact) {
            const float* wdp = a.Wdw + dch;
#pragma unroll
            for (int t = 0; t < 9; ++t) dwt[t] = wdp[t * a.cexp];
            dsc = a.dw_scale[dch]; dsh = a.dw_shift[dch];
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) dwt[t] = 0.f;
        }
        const int kqc = min(4, (a.cexp - ch0) >> 3);
        f32x4 pfrag[4][NTO];
        auto fetch_p = [&]() {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
                    pfrag[kq][nt] = kq < kqc ? a.Wpr[((size_t)(chunk * 4 + kq) * a.pr_nt_total + nt) * 64 + lane] : zero4;
        };
        if (out_live && !DIET) fetch_p();
        // ---- stage 2: thread = (channel dc, output row doy), ET -> D
        if (!(a.ablate & 2) && doy < rows_valid) {
            float row[3][IWP];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* rp = ET + dc * EP + (doy * STRIDE + ky) * IWP;
                if constexpr (IWP % 4 == 0) {
#pragma unroll
                    for (int qx = 0; qx < IWP / 4; ++qx) {
                        const f32x4 v = *(const f32x4*)(rp + qx * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) row[ky][qx * 4 + j] = v[j];
                    }
                } else {
#pragma unroll
                    for (int qx = 0; qx < IWP / 2; ++qx) {
                        const float2 v = *(const float2*)(rp + qx * 2);
                        row[ky][qx * 2] = v.x; row[ky][qx * 2 + 1] = v.y;
                    }
                }
            }
#pragma unroll
            for (int ox = 0; ox < TW; ++ox) {
                float acc = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc = fmaf(row[ky][ox * STRIDE + kx], dwt[ky * 3 + kx], acc);
                D[(doy * TW + ox) * CEP + dc] = relu6f(fmaf(acc, dsc, dsh));    // inactive channel: taps, scale, shift are all 0 -> 0
            }
        }
        __syncthreads();          // D complete, ET free
        // ---- stage 3 of this chunk (reads D) and stage 1 of the next one (writes ET) share this phase
        if (out_live && !(a.ablate & 4)) {
            if (DIET) fetch_p();
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                if (kq < kqc) {
                    const f32x4 av = *(const f32x4*)(D + (wave * 32 + r) * CEP + kq * 8 + half * 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], pfrag[kq][nt][t], pacc[nt], 0, 0, 0);
                }
            }
        }
        if (chunk + 1 < n_chunks && !(a.ablate & 1)) {
            stage1(chunk + 1);
            if (HAS_EXPAND && !interior) zero_border();
        }
        __syncthreads();          // ET complete, D free
    }
    if (out_live && !(a.ablate & 8)) {
        float* obase = a.out + out_base * a.cout;                      // uniform
        const float* rbase = a.X + in_base * a.cin;                    // uniform (residual: same spatial size, cin == cout)
        const bool full = oy0 + TH <= lv.Ho && ox0 + TW <= lv.Wo;
        if constexpr ((TW & (TW - 1)) == 0) {
            // TW a power of two: register reg of the D fragment is tile row (wave*32 + c) / TW, column 4*half + c % TW with
            // c = (reg & 3) + 8 * (reg >> 2) -- the row does not depend on the lane.  One per-lane byte offset, the per-register
            // part is scalar; partial tiles check the row with a scalar compare and the column per lane.
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            const int oyb = oy0 + (wv * 32) / TW, oxl = ox0 + 4 * half;
            const unsigned cout4 = (unsigned)a.cout * 4u;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const int col = nt * 32 + r;
                if (col < a.cout) {
                    const float sc = a.pr_scale[col], sh = a.pr_shift[col];
                    const unsigned lane_off = (unsigned)(oyb * lv.Wo + oxl) * cout4 + (unsigned)col * 4u;
                    float rv[16];
                    if (a.residual) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            constexpr int dummy = 0; (void)dummy;
                            const int c = (reg & 3) + 8 * (reg >> 2), ry = c / TW, rx = c % TW;
                            const bool ok = full || (oyb + ry < lv.Ho && oxl + rx < lv.Wo);
                            rv[reg] = ok ? *(const float*)((const char*)rbase + lane_off + (unsigned)(ry * lv.Wo + rx) * cout4) : 0.0f;
                        }
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int c = (reg & 3) + 8 * (reg >> 2), ry = c / TW, rx = c % TW;
                        if (full || (oyb + ry < lv.Ho && oxl + rx < lv.Wo)) {
                            float v = fmaf(pacc[nt][reg], sc, sh);
                            if (a.residual) v = v + rv[reg];
                            *(float*)((char*)obase + lane_off + (unsigned)(ry * lv.Wo + rx) * cout4) = v;
                        }
                    }
                }
            }
        } else {
            // op = wave*32 + (reg&3) + 8*(reg>>2) + 4*half  ->  (oy, ox)
            const int opl = wave * 32 + 4 * half;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const int col = nt * 32 + r;
                if (col < a.cout) {
                    const float sc = a.pr_scale[col], sh = a.pr_shift[col];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int op = opl + (reg & 3) + 8 * (reg >> 2);
                        const int oy = oy0 + op / TW, ox = ox0 + op % TW;
                        if (full || (oy < lv.Ho && ox < lv.Wo)) {
                            const int off = (oy * lv.Wo + ox) * a.cout + col;
                            float v = fmaf(pacc[nt][reg], sc, sh);
                            if (a.residual) v = v + rbase[off];
                            obase[off] = v;
                        }
                    }
                }
            }
        }
    }
}

template <int STRIDE, int NTO, int KQT, bool HAS_EXPAND, int TW = (STRIDE == 1 ? 16 : 8)>
static hipError_t launch_block_fused2_t(const FusedArgs& a, const Geom& g, hipStream_t s) {
    constexpr int TH = 8;
    int maxtiles = 0;
    for (int l = 0; l < g.n_levels; ++l) maxtiles = max(maxtiles, ((g.lv[l].Wo + TW - 1) / TW) * ((g.lv[l].Ho + TH - 1) / TH));
    dim3 grid(maxtiles, g.n_levels * g.batch);
    hipLaunchKernelGGL((k_block_fused2<STRIDE, NTO, KQT, HAS_EXPAND, TW>), grid, dim3(256), 0, s, a, g);
    return hipGetLastError();
}

// ---- v3 of the fused block: wave-specialised.  A workgroup is 8 waves, two per SIMD: waves 0-3 only issue the
// two MFMA stages (expansion of chunk p, projection of chunk p-2), waves 4-7 only run the depthwise stage (chunk
// p-1) on the vector ALUs and stage the next projection weights into LDS.  ET / D / projection weights are
// double-buffered, so the three stages of three consecutive chunks run in the same barrier phase: the matrix
// pipe of every SIMD always has a wave with MFMA work while the VALU work of the other wave co-issues, and
// there is one barrier per chunk instead of two.  (v2 alternates the stages inside every wave; with 2
// workgroups per CU the phases overlap only by chance and the matrix pipe idles ~50 % of the time.)
template <int STRIDE, int NTO, int KQT, int TW>
__global__ __launch_bounds__(512) void k_block_fused3(FusedArgs a, Geom g) {
    constexpr int TH = 8;
    constexpr int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3, IWP = (IW % 2 == 0) ? IW : (IW + 3) / 4 * 4, NPOS = IH * IWP;
    constexpr int MT_IN = (NPOS + 31) / 32, OPIX = TH * TW, MT_OUT = OPIX / 32, CEP = 36;
    constexpr int EP = ((MT_IN * 32 / 4) % 2 == 1) ? MT_IN * 32 : MT_IN * 32 + 4;   // per-channel stride, EP/4 odd
    constexpr TileSplit SP = split_tiles(MT_IN, MT_OUT, KQT * 4, NTO * 16);
    constexpr int MTW = SP.maxc;
    static_assert(MT_OUT <= 4 && OPIX % 32 == 0 && MTW <= 5, "output tile / halo tile split");
    __shared__ __attribute__((aligned(16))) float ET[2][32 * EP];
    __shared__ __attribute__((aligned(16))) float D[2][OPIX * CEP];
    __shared__ f32x4 WP[2][4 * NTO * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int tiles_x = (lv.Wo + TW - 1) / TW, tiles_y = (lv.Ho + TH - 1) / TH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * STRIDE - lv.pt, ix0 = ox0 * STRIDE - lv.pl;
    const long long in_base = lv.in_off + (long long)frame * lv.H * lv.W;
    const long long out_base = lv.out_off + (long long)frame * lv.Ho * lv.Wo;
    const int n_chunks = min(a.ex_nt_total, (a.cexp + 31) >> 5);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    if (wave < 4) {
        // ================================================================ MFMA waves
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int mt_first = wave == 0 ? SP.first[0] : wave == 1 ? SP.first[1] : wave == 2 ? SP.first[2] : SP.first[3];
        const int mt_count = wave == 0 ? SP.count[0] : wave == 1 ? SP.count[1] : wave == 2 ? SP.count[2] : SP.count[3];
        f32x16 pacc[NTO];
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) pacc[nt] = zero16;
        // block input: the A fragments of this wave's halo M-tiles stay in registers for all chunks
        f32x4 afrag[MTW][KQT];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int pp = (mt_first + m) * 32 + r;
            const int hy = pp / IWP, hx = pp - hy * IWP;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = m < mt_count && hy < IH && hx < IW && iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W;
            const float* ap = a.X + (in_base + (long long)(ok ? iy * lv.W + ix : 0)) * a.cin + half * 4;
#pragma unroll
            for (int kq = 0; kq < KQT; ++kq) afrag[m][kq] = ok ? *(const f32x4*)(ap + kq * 8) : zero4;
        }
        f32x4 bcur[KQT], bnxt[KQT];
        float sc, sh, scn = 0.f, shn = 0.f;
#pragma unroll
        for (int kq = 0; kq < KQT; ++kq) { bcur[kq] = a.Wex[((size_t)kq * a.ex_nt_total) * 64 + lane]; bnxt[kq] = zero4; }
        sc = a.ex_scale[r]; sh = a.ex_shift[r];
        for (int p = 0; p < n_chunks + 2; ++p) {
            const int buf = p & 1;
            if (p + 1 < n_chunks) {                                        // expansion weights of the next chunk
#pragma unroll
                for (int kq = 0; kq < KQT; ++kq) bnxt[kq] = a.Wex[((size_t)kq * a.ex_nt_total + p + 1) * 64 + lane];
                scn = a.ex_scale[(p + 1) * 32 + r]; shn = a.ex_shift[(p + 1) * 32 + r];
            }
            if (p >= 2 && wave < MT_OUT) {                                 // stage 3: projection of chunk p-2
                const int kqc = min(4, (a.cexp - (p - 2) * 32) >> 3);
                const float* dp = D[buf] + (wave * 32 + r) * CEP + half * 4;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    if (kq < kqc) {
                        const f32x4 av = *(const f32x4*)(dp + kq * 8);
                        f32x4 bv[NTO];
#pragma unroll
                        for (int nt = 0; nt < NTO; ++nt) bv[nt] = WP[buf][(kq * NTO + nt) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nt][t], pacc[nt], 0, 0, 0);
                    }
                }
            }
            if (p < n_chunks) {                                            // stage 1: expansion of chunk p -> ET[buf]
                float* et = ET[buf] + r * EP + 4 * half + mt_first * 32;
                // straight-line code per tile count: the chain of tile m+1 is issued before the epilogue of tile m,
                // so the BN / ReLU6 / LDS writes run in the shadow of the next tile's MFMAs
                auto chain = [&](int m) {
                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][0][0], bcur[0][0], zero16, 0, 0, 0);
#pragma unroll
                    for (int kq = 0; kq < KQT; ++kq)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (kq + t > 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][kq][t], bcur[kq][t], acc, 0, 0, 0);
                    return acc;
                };
                auto epi = [&](int m, const f32x16& acc) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = relu6f(fmaf(acc[4 * q + i], sc, sh));
                        *(f32x4*)(et + m * 32 + 8 * q) = v;               // out-of-image positions are masked by stage 2
                    }
                };
                auto stage1 = [&](auto cnt) {
                    constexpr int CNT = decltype(cnt)::value;
                    if constexpr (CNT > 0 && CNT <= MTW) {
                        f32x16 prev = chain(0);
#pragma unroll
                        for (int m = 1; m < CNT; ++m) {
                            const f32x16 cur = chain(m);
                            epi(m - 1, prev);
                            prev = cur;
                            // issue order inside this pair: one MFMA, then its share of the 32 epilogue VALU ops / 4 LDS writes
                            constexpr int NM = KQT * 4, VPM = (32 + NM - 1) / NM, DSI = NM / 4;
#pragma unroll
                            for (int i = 0; i < NM; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                                if (i % DSI == DSI - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                            }
                        }
                        epi(CNT - 1, prev);
                    }
                };
                switch (mt_count) {
                    case 1: stage1(std::integral_constant<int, 1>{}); break;
                    case 2: stage1(std::integral_constant<int, 2>{}); break;
                    case 3: stage1(std::integral_constant<int, 3>{}); break;
                    case 4: stage1(std::integral_constant<int, 4>{}); break;
                    case 5: stage1(std::integral_constant<int, 5>{}); break;
                    default: break;
                }
            }
            if (p + 1 < n_chunks) {
#pragma unroll
                for (int kq = 0; kq < KQT; ++kq) bcur[kq] = bnxt[kq];
                sc = scn; sh = shn;
            }
            __syncthreads();
        }
        // ---- epilogue: BN (+ residual), store
        if (wave < MT_OUT) {
            float* obase = a.out + out_base * a.cout;
            const float* rbase = a.X + in_base * a.cin;                    // residual: same spatial size, cin == cout
            const bool full = oy0 + TH <= lv.Ho && ox0 + TW <= lv.Wo;
            const int opl = wave * 32 + 4 * half;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const int col = nt * 32 + r;
                if (col < a.cout) {
                    const float psc = a.pr_scale[col], psh = a.pr_shift[col];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int op = opl + (reg & 3) + 8 * (reg >> 2);
                        const int oy = oy0 + op / TW, ox = ox0 + op % TW;
                        if (full || (oy < lv.Ho && ox < lv.Wo)) {
                            const int off = (oy * lv.Wo + ox) * a.cout + col;
                            float v = fmaf(pacc[nt][reg], psc, psh);
                            if (a.residual) v = v + rbase[off];
                            obase[off] = v;
                        }
                    }
                }
            }
        }
    } else {
        // ================================================================ depthwise waves
        const int vt = threadIdx.x - 256, dc = vt & 31, doy = vt >> 5;    // role: (channel lane, output row)
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= lv.H && ix0 + IW <= lv.W;
        float dwt[9], dsc = 0.f, dsh = 0.f, dwn[9], dscn = 0.f, dshn = 0.f;
        bool dact = false, dactn = false;
        f32x4 preg[NTO];
#pragma unroll
        for (int t = 0; t < 9; ++t) { dwt[t] = 0.f; dwn[t] = 0.f; }
#pragma unroll
        for (int j = 0; j < NTO; ++j) preg[j] = zero4;
        for (int p = 0; p < n_chunks + 2; ++p) {
            const bool work = p >= 1 && p <= n_chunks;
            if (work) {                                                    // publish what was fetched one phase ago
#pragma unroll
                for (int j = 0; j < NTO; ++j) WP[(p - 1) & 1][vt + j * 256] = preg[j];
#pragma unroll
                for (int t = 0; t < 9; ++t) dwt[t] = dwn[t];
                dsc = dscn; dsh = dshn; dact = dactn;
            }
            if (p < n_chunks) {                                            // fetch chunk p: projection weights, depthwise taps / BN
                const int kqc = min(4, (a.cexp - p * 32) >> 3);
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    const int idx = vt + j * 256, kq = idx / (NTO * 64);
                    preg[j] = kq < kqc ? a.Wpr[(size_t)p * 4 * NTO * 64 + idx] : zero4;
                }
                const int dch = p * 32 + dc;
                dactn = dch < a.cexp;
                if (dactn) {
                    const float* wdp = a.Wdw + dch;
#pragma unroll
                    for (int t = 0; t < 9; ++t) dwn[t] = wdp[t * a.cexp];
                    dscn = a.dw_scale[dch]; dshn = a.dw_shift[dch];
                }
            }
            if (work) {                                                    // stage 2 of chunk p-1: ET -> D
                const int buf = (p - 1) & 1;
                float row[3][IWP];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* rp = ET[buf] + dc * EP + (doy * STRIDE + ky) * IWP;
                    if constexpr (IWP % 4 == 0) {
#pragma unroll
                        for (int qx = 0; qx < IWP / 4; ++qx) {
                            const f32x4 v = *(const f32x4*)(rp + qx * 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) row[ky][qx * 4 + j] = v[j];
                        }
                    } else {
#pragma unroll
                        for (int qx = 0; qx < IWP / 2; ++qx) {
                            const float2 v = *(const float2*)(rp + qx * 2);
                            row[ky][qx * 2] = v.x; row[ky][qx * 2 + 1] = v.y;
                        }
                    }
                }
                if (!interior) {
                    // the expansion of an out-of-image halo position must count as 0 ('SAME' padding of the depthwise conv)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const bool rok = (unsigned)(iy0 + doy * STRIDE + ky) < (unsigned)lv.H;
#pragma unroll
                        for (int x = 0; x < IW; ++x) row[ky][x] = (rok && (unsigned)(ix0 + x) < (unsigned)lv.W) ? row[ky][x] : 0.0f;
                    }
                }
                float* dp = D[buf] + (doy * TW) * CEP + dc;
#pragma unroll
                for (int ox = 0; ox < TW; ++ox) {
                    float acc = 0.0f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc = fmaf(row[ky][ox * STRIDE + kx], dwt[ky * 3 + kx], acc);
                    dp[ox * CEP] = dact ? relu6f(fmaf(acc, dsc, dsh)) : 0.0f;
                }
            }
            __syncthreads();
        }
    }
}

// ---- v3p: the same wave-specialised pipeline, persistent over tiles.  One workgroup per CU walks tiles
// b, b + gridDim.x, ...; the global phase counter keeps running across tile boundaries, so the expansion of the
// next tile's first chunks runs while the projection of the previous tile drains (a tile costs n_chunks phases
// instead of n_chunks + 2), the next tile's input fragments are prefetched a whole tile ahead and the BN / store
// epilogue overlaps the next tile's MFMAs.
struct TileGeo {
    int oy0, ox0, iy0, ix0, H, W, Ho, Wo;
    long long in_base, out_base;
};
template <int STRIDE, int TH, int TW>
__device__ __forceinline__ TileGeo decode_tile(const Geom& g, int t) {
    int l = 0, rem = t, tx = 1, per = 1;
    for (;; ++l) {
        tx = (g.lv[l].Wo + TW - 1) / TW;
        per = tx * ((g.lv[l].Ho + TH - 1) / TH);
        if (l == g.n_levels - 1 || rem < per * g.batch) break;
        rem -= per * g.batch;
    }
    const LevelGeom lv = g.lv[l];
    const int frame = rem / per, tile = rem - frame * per;
    const int tyi = tile / tx, txi = tile - tyi * tx;
    TileGeo o;
    o.oy0 = tyi * TH; o.ox0 = txi * TW;
    o.iy0 = o.oy0 * STRIDE - lv.pt; o.ix0 = o.ox0 * STRIDE - lv.pl;
    o.H = lv.H; o.W = lv.W; o.Ho = lv.Ho; o.Wo = lv.Wo;
    o.in_base = lv.in_off + (long long)frame * lv.H * lv.W;
    o.out_base = lv.out_off + (long long)frame * lv.Ho * lv.Wo;
    return o;
}

template <int STRIDE, int NTO, int KQT, int TW, bool RESID>
__global__ __launch_bounds__(512) void k_block_fused3p(FusedArgs a, Geom g, int total_tiles) {
    constexpr int TH = 8;
    constexpr int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3, IWP = (IW % 2 == 0) ? IW : (IW + 3) / 4 * 4, NPOS = IH * IWP;
    constexpr int MT_IN = (NPOS + 31) / 32, OPIX = TH * TW, MT_OUT = OPIX / 32, CEP = 36;
    constexpr int EP = ((MT_IN * 32 / 4) % 2 == 1) ? MT_IN * 32 : MT_IN * 32 + 4;
    constexpr TileSplit SP = split_tiles(MT_IN, MT_OUT, KQT * 4, NTO * 16);
    constexpr int MTW = SP.maxc;
    static_assert(MT_OUT <= 4 && OPIX % 32 == 0 && MTW <= 5, "output tile / halo tile split");
    __shared__ __attribute__((aligned(16))) float ET[2][32 * EP];
    __shared__ __attribute__((aligned(16))) float D[2][OPIX * CEP];
    __shared__ f32x4 WP[2][4 * NTO * 64];      // projection weights of a chunk (B fragments, [kq][nt][lane])
    __shared__ f32x4 WB[2][KQT * 64];          // expansion weights of a chunk ([kq][lane])
    __shared__ float WS[2][64];                // expansion BN scale [0..31] / shift [32..63] of a chunk
    const int lane = threadIdx.x & 63, hw_wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    // role mapping experiment (ablate bit 4): MFMA waves = even hardware waves instead of waves 0-3
    const bool alt = (a.ablate & 16) != 0;
    const bool is_mfma = alt ? !(hw_wave & 1) : hw_wave < 4;
    const int wave = alt ? (hw_wave >> 1) : (hw_wave & 3);
    const int n = min(a.ex_nt_total, (a.cexp + 31) >> 5);                         // chunks per tile
    const int m_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (m_tiles <= 0) return;
    const int n_phases = m_tiles * n + 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    if (is_mfma) {
        // ================================================================ MFMA waves
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int mt_first = wave == 0 ? SP.first[0] : wave == 1 ? SP.first[1] : wave == 2 ? SP.first[2] : SP.first[3];
        const int mt_count = wave == 0 ? SP.count[0] : wave == 1 ? SP.count[1] : wave == 2 ? SP.count[2] : SP.count[3];
        f32x16 pacc[NTO];
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) pacc[nt] = zero16;
        f32x4 afrag[MTW][KQT], anext[MTW][KQT];
        auto load_a = [&](int tile_id) {
            const TileGeo tg = decode_tile<STRIDE, TH, TW>(g, tile_id);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int pp = (mt_first + m) * 32 + r;
                const int hy = pp / IWP, hx = pp - hy * IWP;
                const int iy = tg.iy0 + hy, ix = tg.ix0 + hx;
                const bool ok = m < mt_count && hy < IH && hx < IW && iy >= 0 && iy < tg.H && ix >= 0 && ix < tg.W;
                const float* ap = a.X + (tg.in_base + (long long)(ok ? iy * tg.W + ix : 0)) * a.cin + half * 4;
#pragma unroll
                for (int kq = 0; kq < KQT; ++kq) anext[m][kq] = ok ? *(const f32x4*)(ap + kq * 8) : zero4;
            }
        };
        load_a(blockIdx.x);
        __syncthreads();                                                   // WB[0] / WS[0] published by the depthwise waves
        int t1 = 0, c1 = 0;          // stage-1 position (tile, chunk) of this phase
        int t3 = 0, c3 = 0;          // stage-3 position, valid from phase 2 on
        for (int ph = 0; ph < n_phases; ++ph) {
            const int buf = ph & 1;
            const bool s1 = t1 < m_tiles;
            if (s1 && c1 == 0) {                                           // new tile: take the prefetched input, prefetch the next one
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int kq = 0; kq < KQT; ++kq) afrag[m][kq] = anext[m][kq];
                if (t1 + 1 < m_tiles) load_a(blockIdx.x + (t1 + 1) * gridDim.x);
            }
            f32x4 resid[RESID ? NTO : 1][4];                             // residual of the tile that completes in this phase
            const bool last3 = ph >= 2 && c3 == n - 1;
            TileGeo tg3;
            if (last3) tg3 = decode_tile<STRIDE, TH, TW>(g, blockIdx.x + t3 * gridDim.x);
            if (RESID && last3 && wave < MT_OUT) {
                const float* rbase = a.X + tg3.in_base * a.cin;            // same spatial size, cin == cout
                const int opl = wave * 32 + 4 * half;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int op = opl + (reg & 3) + 8 * (reg >> 2);
                        const int oy = min(tg3.oy0 + op / TW, tg3.Ho - 1), ox = min(tg3.ox0 + op % TW, tg3.Wo - 1);
                        const int col = min(nt * 32 + r, a.cout - 1);
                        resid[RESID ? nt : 0][reg >> 2][reg & 3] = rbase[(oy * tg3.Wo + ox) * a.cout + col];
                    }
            }
            if (ph >= 2 && wave < MT_OUT && !(a.ablate & 4)) {            // stage 3: projection of chunk c3 of tile t3
                const int kqc = min(4, (a.cexp - c3 * 32) >> 3);
                const float* dp = D[buf] + (wave * 32 + r) * CEP + half * 4;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    if (kq < kqc) {
                        const f32x4 av = *(const f32x4*)(dp + kq * 8);
                        f32x4 bv[NTO];
#pragma unroll
                        for (int nt = 0; nt < NTO; ++nt) bv[nt] = WP[buf][(kq * NTO + nt) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nt][t], pacc[nt], 0, 0, 0);
                    }
                }
            }
            if (s1 && !(a.ablate & 1)) {                                   // stage 1: expansion of chunk c1 of tile t1 -> ET[buf]
                float* et = ET[buf] + r * EP + 4 * half + mt_first * 32;
                f32x4 bcur[KQT];
#pragma unroll
                for (int kq = 0; kq < KQT; ++kq) bcur[kq] = WB[buf][kq * 64 + lane];
                const float sc = WS[buf][r], sh = WS[buf][32 + r];
                auto chain = [&](int m) {
                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][0][0], bcur[0][0], zero16, 0, 0, 0);
#pragma unroll
                    for (int kq = 0; kq < KQT; ++kq)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (kq + t > 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][kq][t], bcur[kq][t], acc, 0, 0, 0);
                    return acc;
                };
                auto epi = [&](int m, const f32x16& acc) {
                    if (a.ablate & 32) { if (acc[0] == 1234.5f) et[m] = acc[1]; return; }    // diagnostics: MFMA chains only
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = relu6f(fmaf(acc[4 * q + i], sc, sh));
                        *(f32x4*)(et + m * 32 + 8 * q) = v;
                    }
                };
                auto stage1 = [&](auto cnt) {
                    constexpr int CNT = decltype(cnt)::value;
                    if constexpr (CNT > 0 && CNT <= MTW) {
                        f32x16 prev = chain(0);
#pragma unroll
                        for (int m = 1; m < CNT; ++m) {
                            const f32x16 cur = chain(m);
                            epi(m - 1, prev);
                            prev = cur;
                            constexpr int NM = KQT * 4, VPM = (32 + NM - 1) / NM, DSI = NM / 4;
#pragma unroll
                            for (int i = 0; i < NM; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                                if (i % DSI == DSI - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                            }
                        }
                        epi(CNT - 1, prev);
                    }
                };
                switch (mt_count) {
                    case 1: stage1(std::integral_constant<int, 1>{}); break;
                    case 2: stage1(std::integral_constant<int, 2>{}); break;
                    case 3: stage1(std::integral_constant<int, 3>{}); break;
                    case 4: stage1(std::integral_constant<int, 4>{}); break;
                    case 5: stage1(std::integral_constant<int, 5>{}); break;
                    default: break;
                }
            }
            if (last3 && wave < MT_OUT) {                                  // tile t3 complete: BN (+ residual), store, reset
                float* obase = a.out + tg3.out_base * a.cout;
                const bool full = tg3.oy0 + TH <= tg3.Ho && tg3.ox0 + TW <= tg3.Wo;
                const int opl = wave * 32 + 4 * half;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    const int col = nt * 32 + r;
                    if (col < a.cout && !(a.ablate & 8)) {
                        const float psc = a.pr_scale[col], psh = a.pr_shift[col];
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int op = opl + (reg & 3) + 8 * (reg >> 2);
                            const int oy = tg3.oy0 + op / TW, ox = tg3.ox0 + op % TW;
                            if (full || (oy < tg3.Ho && ox < tg3.Wo)) {
                                float v = fmaf(pacc[nt][reg], psc, psh);
                                if (RESID) v = v + resid[RESID ? nt : 0][reg >> 2][reg & 3];
                                obase[(oy * tg3.Wo + ox) * a.cout + col] = v;
                            }
                        }
                    }
                    pacc[nt] = zero16;
                }
            }
            if (s1) { if (++c1 == n) { c1 = 0; ++t1; } }
            if (ph >= 2) { if (++c3 == n) { c3 = 0; ++t3; } }
            __syncthreads();
        }
    } else {
        // ================================================================ depthwise waves
        const int vt = wave * 64 + lane, dc = vt & 31, doy = vt >> 5;    // role: (channel lane, output row)
        float dwt[9], dsc = 0.f, dsh = 0.f, dwn[9], dscn = 0.f, dshn = 0.f;
        bool dact = false, dactn = false;
        f32x4 preg[NTO];
#pragma unroll
        for (int t = 0; t < 9; ++t) { dwt[t] = 0.f; dwn[t] = 0.f; }
#pragma unroll
        for (int j = 0; j < NTO; ++j) preg[j] = zero4;
        int t2 = 0, c2 = 0;          // stage-2 position, valid from phase 1 on
        int cf = 0;                  // chunk whose projection / depthwise weights are fetched in this phase (the stage-1 chunk)
        // expansion weights run two phases ahead of their stage 1: fetched at ph-2, published at ph-1 into WB[ph & 1]
        constexpr int BJ = (KQT * 64 + 255) / 256;
        f32x4 breg[BJ];
        float sreg = 0.f;
        auto fetch_b = [&](int chunk) {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int idx = vt + j * 256, kq = idx >> 6, ln = idx & 63;
                breg[j] = idx < KQT * 64 ? a.Wex[((size_t)kq * a.ex_nt_total + chunk) * 64 + ln] : zero4;
            }
            if (vt < 64) sreg = vt < 32 ? a.ex_scale[chunk * 32 + vt] : a.ex_shift[chunk * 32 + vt - 32];
        };
        auto publish_b = [&](int b) {
#pragma unroll
            for (int j = 0; j < BJ; ++j) { const int idx = vt + j * 256; if (idx < KQT * 64) WB[b][idx] = breg[j]; }
            if (vt < 64) WS[b][vt] = sreg;
        };
        fetch_b(0); publish_b(0);
        int cb = n > 1 ? 1 : 0;      // chunk fetched next
        fetch_b(cb); if (++cb == n) cb = 0;
        __syncthreads();
        int iy0 = 0, ix0 = 0, H = 0, W = 0;
        bool interior = true;
        for (int ph = 0; ph < n_phases; ++ph) {
            const bool work = ph >= 1 && ph <= m_tiles * n;
            publish_b((ph + 1) & 1);                                       // expansion weights of the next phase's chunk
            fetch_b(cb); if (++cb == n) cb = 0;
            if (work) {                                                    // publish what was fetched one phase ago
#pragma unroll
                for (int j = 0; j < NTO; ++j) WP[(ph - 1) & 1][vt + j * 256] = preg[j];
#pragma unroll
                for (int t = 0; t < 9; ++t) dwt[t] = dwn[t];
                dsc = dscn; dsh = dshn; dact = dactn;
                if (c2 == 0) {
                    const TileGeo tg = decode_tile<STRIDE, TH, TW>(g, blockIdx.x + t2 * gridDim.x);
                    iy0 = tg.iy0; ix0 = tg.ix0; H = tg.H; W = tg.W;
                    interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= H && ix0 + IW <= W;
                }
            }
            if (ph < m_tiles * n) {                                        // fetch chunk cf: projection weights, depthwise taps / BN
                const int kqc = min(4, (a.cexp - cf * 32) >> 3);
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    const int idx = vt + j * 256, kq = idx / (NTO * 64);
                    preg[j] = kq < kqc ? a.Wpr[(size_t)cf * 4 * NTO * 64 + idx] : zero4;
                }
                const int dch = cf * 32 + dc;
                dactn = dch < a.cexp;
                if (dactn) {
                    const float* wdp = a.Wdw + dch;
#pragma unroll
                    for (int t = 0; t < 9; ++t) dwn[t] = wdp[t * a.cexp];
                    dscn = a.dw_scale[dch]; dshn = a.dw_shift[dch];
                }
                if (++cf == n) cf = 0;
            }
            if (work && !(a.ablate & 2)) {                                 // stage 2 of chunk c2 of tile t2: ET -> D
                const int buf = (ph - 1) & 1;
                float row[3][IWP];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* rp = ET[buf] + dc * EP + (doy * STRIDE + ky) * IWP;
                    if constexpr (IWP % 4 == 0) {
#pragma unroll
                        for (int qx = 0; qx < IWP / 4; ++qx) {
                            const f32x4 v = *(const f32x4*)(rp + qx * 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) row[ky][qx * 4 + j] = v[j];
                        }
                    } else {
#pragma unroll
                        for (int qx = 0; qx < IWP / 2; ++qx) {
                            const float2 v = *(const float2*)(rp + qx * 2);
                            row[ky][qx * 2] = v.x; row[ky][qx * 2 + 1] = v.y;
                        }
                    }
                }
                if (!interior) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const bool rok = (unsigned)(iy0 + doy * STRIDE + ky) < (unsigned)H;
#pragma unroll
                        for (int x = 0; x < IW; ++x) row[ky][x] = (rok && (unsigned)(ix0 + x) < (unsigned)W) ? row[ky][x] : 0.0f;
                    }
                }
                float* dp = D[buf] + (doy * TW) * CEP + dc;
#pragma unroll
                for (int ox = 0; ox < TW; ++ox) {
                    float acc = 0.0f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc = fmaf(row[ky][ox * STRIDE + kx], dwt[ky * 3 + kx], acc);
                    dp[ox * CEP] = dact ? relu6f(fmaf(acc, dsc, dsh)) : 0.0f;
                }
            }
            if (work) { if (++c2 == n) { c2 = 0; ++t2; } }
            __syncthreads();
        }
    }
}

template <int STRIDE, int NTO, int KQT, int TW>
static hipError_t launch_block_fused3p_t(const FusedArgs& a, const Geom& g, hipStream_t s) {
    if (a.residual && (a.cin != a.cout || STRIDE != 1)) return hipErrorInvalidValue;
    constexpr int TH = 8;
    int total = 0;
    for (int l = 0; l < g.n_levels; ++l) total += ((g.lv[l].Wo + TW - 1) / TW) * ((g.lv[l].Ho + TH - 1) / TH) * g.batch;
    static const int n_cu = []() { const char* v = getenv("HFNET_FUSE3_WGS"); return v ? atoi(v) : 256; }();
    const int grid = total < n_cu ? total : n_cu;
    if (grid <= 0) return hipSuccess;
    if (a.residual) hipLaunchKernelGGL((k_block_fused3p<STRIDE, NTO, KQT, TW, true>), dim3(grid), dim3(512), 0, s, a, g, total);
    else hipLaunchKernelGGL((k_block_fused3p<STRIDE, NTO, KQT, TW, false>), dim3(grid), dim3(512), 0, s, a, g, total);
    return hipGetLastError();
}

template <int STRIDE, int NTO, int KQT, int TW>
static hipError_t launch_block_fused3_t(const FusedArgs& a, const Geom& g, hipStream_t s) {
    constexpr int TH = 8;
    int maxtiles = 0;
    for (int l = 0; l < g.n_levels; ++l) maxtiles = max(maxtiles, ((g.lv[l].Wo + TW - 1) / TW) * ((g.lv[l].Ho + TH - 1) / TH));
    dim3 grid(maxtiles, g.n_levels * g.batch);
    hipLaunchKernelGGL((k_block_fused3<STRIDE, NTO, KQT, TW>), grid, dim3(512), 0, s, a, g);
    return hipGetLastError();
}

// ---- layer_2 (expanded_conv with expansion factor 1: no expand conv, hf_net.py:31-33): depthwise 3x3 +
// BN + ReLU6 on CIN channels, then the 1x1 projection CIN -> COUT + BN, stride 1.  Its tensors are the
// largest of the network (1/2 resolution) and its arithmetic the smallest: one thread per output pixel on
// the vector ALUs, projection weights in LDS (wave-uniform reads), the fma chain over the CIN depthwise
// outputs in logical channel order (the oracle's), 16-byte fully coalesced loads and stores.  HBM-bound.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_block_noexpand(const float* __restrict__ X, float* __restrict__ out,
                                                        const float* __restrict__ wd /*[9][CIN] phys*/, const float* __restrict__ dsc,
                                                        const float* __restrict__ dsh, const float* __restrict__ wp /*[CIN logical][COUT phys]*/,
                                                        const float* __restrict__ psc, const float* __restrict__ psh, Geom g) {
    // 16x16 output tile per workgroup; the 18x18 input halo tile is staged through LDS with coalesced 96-byte
    // pixel rows, so every input byte crosses HBM ~1.27x instead of up to 9x.  Weight indices are compile-time
    // constants: those loads are wave-uniform and go through the scalar cache into SGPR fma operands.
    constexpr int T = 16, SH = T + 2, CP = CIN + 4;
    __shared__ __attribute__((aligned(16))) float tile[SH * SH * CP];
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int tiles_x = (lv.Wo + T - 1) / T;
    if ((int)blockIdx.x >= tiles_x * ((lv.Ho + T - 1) / T)) return;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int oy0 = tyi * T, ox0 = txi * T;
    const float* xin = X + (lv.in_off + (long long)frame * lv.H * lv.W) * CIN;
    // a halo row is SH pixels = SH * CIN / 4 consecutive 16-byte pieces in memory: thread = (row parity, piece), so the
    // piece -> (pixel, channel quad) split is done once per thread and a load costs an add
    {
        constexpr int PPR = SH * (CIN / 4);                    // pieces per halo row (108)
        static_assert(2 * PPR <= 256, "two halo rows per pass");
        const int rsel = threadIdx.x >= PPR ? 1 : 0, piece = threadIdx.x - rsel * PPR;
        const int hx = piece / (CIN / 4), c4 = piece - hx * (CIN / 4);
        const int ix = ox0 - lv.pl + hx;
        const bool xok = threadIdx.x < 2 * PPR && ix >= 0 && ix < lv.W;
        const float* colp = xin + (long long)(xok ? ix : 0) * CIN + c4 * 4;
        float* tp = tile + hx * CP + c4 * 4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (threadIdx.x < 2 * PPR) {
#pragma unroll
            for (int r2 = 0; r2 < SH / 2; ++r2) {
                const int hy = r2 * 2 + rsel, iy = oy0 - lv.pt + hy;
                const bool ok = xok && iy >= 0 && iy < lv.H;
                const f32x4 v = *(const f32x4*)(colp + (long long)(ok ? iy : 0) * lv.W * CIN);
                *(f32x4*)(tp + hy * SH * CP) = ok ? v : zero;
            }
        }
    }
    __syncthreads();
    const int ty = threadIdx.x / T, tx = threadIdx.x - ty * T;
    // The weights are wave-uniform scalar loads (SGPR fma operands).  Left alone, the scheduler hoists all ~650 of them
    // to the top and spills them to VGPR lanes (v_writelane / v_readlane: more VALU work than the convolution itself),
    // so they are fetched one step ahead of their use and scheduling barriers keep each batch where it is.
    float d[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) d[c] = 0.0f;
    float wc[CIN], wn[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) wc[c] = wd[c];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {                    // out-of-image taps are zeros in the tile: fma(0, w, d) == d
        if (tap < 8) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) wn[c] = wd[(tap + 1) * CIN + c];
        } else {
#pragma unroll
            for (int c = 0; c < CIN; ++c) wn[c] = dsc[c];
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        const float* xp = tile + ((ty + tap / 3) * SH + tx + tap % 3) * CP;
#pragma unroll
        for (int c4 = 0; c4 < CIN / 4; ++c4) {
            const f32x4 xv = *(const f32x4*)(xp + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[c4 * 4 + j] = fmaf(xv[j], wc[c4 * 4 + j], d[c4 * 4 + j]);
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CIN; ++c) wc[c] = wn[c];
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c) wn[c] = dsh[c];
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CIN; ++c) d[c] = relu6f(fmaf(d[c], wc[c], wn[c]));
    __builtin_amdgcn_sched_barrier(0);
    float acc[COUT];
#pragma unroll
    for (int n = 0; n < COUT; ++n) acc[n] = 0.0f;
    float pc[COUT], pn[COUT];
#pragma unroll
    for (int n = 0; n < COUT; ++n) pc[n] = wp[n];
#pragma unroll
    for (int k = 0; k < CIN; ++k) {                       // logical channel k sits in physical slot phys(k)
        if (k + 1 < CIN) {
#pragma unroll
            for (int n = 0; n < COUT; ++n) pn[n] = wp[(k + 1) * COUT + n];
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        const int pk = (k & ~7) | ((k & 1) << 2) | ((k & 7) >> 1);
        const float dk = d[pk];
#pragma unroll
        for (int n = 0; n < COUT; ++n) acc[n] = fmaf(dk, pc[n], acc[n]);
        // pin this row's fmas here (pure arithmetic is not ordered by the barriers: the DAG scheduler would sink all of
        // it below the last load and every row would be spilled in between)
        static_assert(COUT == 16, "accumulator pinning is written for 16 outputs");
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                          "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < COUT; ++n) pc[n] = pn[n];
    }
#pragma unroll
    for (int n = 0; n < COUT; ++n) pc[n] = psc[n];
    // pc now holds the projection BN scale
    // output through LDS: a tile row is T pixels x COUT channels = one contiguous 1 KB run of the output tensor;
    // written back as consecutive 16-byte pieces per lane instead of four 64-byte-strided stores per thread
    constexpr int OP = COUT + 4;                           // 20 words: conflict-free b128
    __syncthreads();                                       // every thread is done reading the input tile
    float* ot = tile;
#pragma unroll
    for (int n4 = 0; n4 < COUT / 4; ++n4) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(acc[n4 * 4 + j], pc[n4 * 4 + j], psh[n4 * 4 + j]);
        *(f32x4*)(ot + threadIdx.x * OP + n4 * 4) = v;
    }
    __syncthreads();
    float* obase = out + (lv.out_off + (long long)frame * lv.Ho * lv.Wo) * COUT;
    const int cols = min(T, lv.Wo - ox0);                  // valid pixels per tile row
#pragma unroll
    for (int k = 0; k < COUT / 4; ++k) {
        const int q = threadIdx.x + k * 256;               // piece of the tile: row = q / (T * COUT / 4)
        const int row = q / (T * COUT / 4), rem = q - row * (T * COUT / 4);
        const int px = rem / (COUT / 4), part = rem - px * (COUT / 4);
        if (oy0 + row < lv.Ho && px < cols)
            *(f32x4*)(obase + ((long long)(oy0 + row) * lv.Wo + ox0 + px) * COUT + part * 4) = *(const f32x4*)(ot + (row * T + px) * OP + part * 4);
    }
}

// ---- stem + layer_2 in one launch: u8 image -> [(x-128)/128, conv 3x3/2 1->CS, BN, ReLU6] -> depthwise 3x3 +
// BN + ReLU6 -> 1x1 CS->COUT + BN.  The half-resolution CS-channel stem tensor (the largest activation of the
// network: 8.7 MB per 752x480 frame) is produced into LDS for an 18x18 halo tile and consumed from there; only
// the u8 image is read and the COUT-channel layer_2 output written.  Vector-ALU kernel (K = 9 / 9 / CS), one
// thread per output pixel of a 16x16 tile; every fma chain is the oracle's.
struct StemBlockArgs {
    const float* stem_w; const float* stem_scale; const float* stem_shift;      // [9][CS] physical order
    const float* dw_w; const float* dw_scale; const float* dw_shift;            // [9][CS] physical
    const float* pr_w; const float* pr_scale; const float* pr_shift;            // [CS logical][COUT physical]
    float* out;
};

// Built on k_block_noexpand: its halo-tile load is replaced by the stem
// convolution of the 18 x 18 halo positions straight from the u8 image (wave-uniform scalar weights; the 324 positions
// x 2 channel halves are 12 wave-sized work units, three per wave), everything after it is k_block_noexpand unchanged.
// The 24-channel half-resolution stem tensor -- 690 MB per 32-frame step, written by one kernel and read by the next --
// never exists.
template <int CS, int COUT>
__global__ __launch_bounds__(256) void k_stem_block2(ImageSet imgs, const float* __restrict__ stem_w, const float* __restrict__ stem_scale,
                                                     const float* __restrict__ stem_shift, const float* __restrict__ wd,
                                                     const float* __restrict__ dsc, const float* __restrict__ dsh, const float* __restrict__ wp,
                                                     const float* __restrict__ psc, const float* __restrict__ psh, float* __restrict__ out,
                                                     Geom gs /*image -> stem*/, Geom gb /*stem -> layer_2*/) {
    constexpr int T = 16, SH = T + 2, SP = SH * SH, CP = CS + 4, CH = CS / 2;
    static_assert(CS == 24 && COUT == 16, "written for the 0.75-width network");
    __shared__ __attribute__((aligned(16))) float tile[SP * CP];
    const int image = blockIdx.y, level = image / gs.batch, frame = image - level * gs.batch;
    const LevelGeom ls = gs.lv[level], lb = gb.lv[level];     // ls: H,W image (cropped), Ho,Wo stem; lb: H,W stem, Ho,Wo out (same size)
    const int tiles_x = (lb.Wo + T - 1) / T;
    if ((int)blockIdx.x >= tiles_x * ((lb.Ho + T - 1) / T)) return;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int oy0 = tyi * T, ox0 = txi * T;
    const int sy0 = oy0 - lb.pt, sx0 = ox0 - lb.pl;            // first stem row / col of the halo tile
    const uint8_t* img = imgs.ptr[level] + (long long)frame * imgs.frame_stride[level];
    const int rs = imgs.row_stride[level];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- stem on the halo tile: unit u = wave + 4 * pass covers positions [54 * (u % 6), +54) and channel half u / 6
    constexpr int CHUNK = SP / 6;                              // 54 positions per unit
    static_assert(CHUNK * 6 == SP && CHUNK <= 64, "halo positions split into six wave-sized chunks");
    // tiles whose halo and image patch lie inside the maps (the vast majority) skip every bounds check and select
    const int iy_first = sy0 * 2 - ls.pt, ix_first = sx0 * 2 - ls.pl;
    const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + SH <= ls.Ho && sx0 + SH <= ls.Wo && iy_first >= 0 && ix_first >= 0 &&
                          iy_first + 2 * SH < ls.H && ix_first + 2 * SH < ls.W;     // uniform
    auto stem_passes = [&](auto interior_tag) {
        constexpr bool INT = decltype(interior_tag)::value;
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            const int u = wave + 4 * pass, chunk = u % 6, hsel = u / 6;          // uniform
            const int p = chunk * CHUNK + min(lane, CHUNK - 1);
            const int hy = p / SH, hx = p - hy * SH;
            const int sy = sy0 + hy, sx = sx0 + hx;
            const bool in = INT || (sy >= 0 && sy < ls.Ho && sx >= 0 && sx < ls.Wo);
            float px[9];
            if (INT) {
                const uint8_t* ip = img + (long long)(sy * 2 - ls.pt) * rs + (sx * 2 - ls.pl);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) px[ky * 3 + kx] = ((float)ip[ky * rs + kx] - 128.0f) * 0.0078125f;
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int iy = sy * 2 - ls.pt + ky, ix = sx * 2 - ls.pl + kx;
                        const bool ok = in && iy >= 0 && iy < ls.H && ix >= 0 && ix < ls.W;
                        const float raw = (float)img[(long long)(ok ? iy : 0) * rs + (ok ? ix : 0)];
                        px[ky * 3 + kx] = ok ? (raw - 128.0f) * 0.0078125f : 0.0f;
                    }
            }
            // weights of four channels at a time as wave-uniform 16-byte scalar loads (36 + 8 SGPRs per group: a second set
            // prefetched ahead does not fit next to the geometry without spilling to VGPR lanes)
            const f32x4* __restrict__ w4 = (const f32x4*)(stem_w + hsel * CH);    // uniform; row t is w4[t * CS / 4 + group]
            const f32x4* __restrict__ sc4 = (const f32x4*)(stem_scale + hsel * CH);
            const f32x4* __restrict__ sh4 = (const f32x4*)(stem_shift + hsel * CH);
            float* tp = tile + p * CP + hsel * CH;
#pragma unroll
            for (int grp = 0; grp < CH / 4; ++grp) {
                f32x4 wq[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wq[t] = w4[t * (CS / 4) + grp];
                const f32x4 scq = sc4[grp], shq = sh4[grp];
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = 0.0f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc = fmaf(px[t], wq[t][j], acc);
                    float v = relu6f(fmaf(acc, scq[j], shq[j]));
                    if (!INT) {
                        asm volatile("" : "+v"(v));                                // computed by every lane: a select, not a branch
                        v = in ? v : 0.0f;                                         // outside the stem map: the depthwise conv's zero padding
                    }
                    r[j] = v;
                }
                if (lane < CHUNK) *(f32x4*)(tp + grp * 4) = r;
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (interior) stem_passes(std::true_type{}); else stem_passes(std::false_type{});
    __syncthreads();
    const int ty = threadIdx.x / T, tx = threadIdx.x - ty * T;
    constexpr int CIN = CS;
    float d[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) d[c] = 0.0f;
    float wc[CIN], wn[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) wc[c] = wd[c];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {                    // (see k_block_noexpand for the weight staging)
        if (tap < 8) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) wn[c] = wd[(tap + 1) * CIN + c];
        } else {
#pragma unroll
            for (int c = 0; c < CIN; ++c) wn[c] = dsc[c];
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        const float* xp = tile + ((ty + tap / 3) * SH + tx + tap % 3) * CP;
#pragma unroll
        for (int c4 = 0; c4 < CIN / 4; ++c4) {
            const f32x4 xv = *(const f32x4*)(xp + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[c4 * 4 + j] = fmaf(xv[j], wc[c4 * 4 + j], d[c4 * 4 + j]);
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CIN; ++c) wc[c] = wn[c];
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c) wn[c] = dsh[c];
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CIN; ++c) d[c] = relu6f(fmaf(d[c], wc[c], wn[c]));
    __builtin_amdgcn_sched_barrier(0);
    float acc[COUT];
#pragma unroll
    for (int n = 0; n < COUT; ++n) acc[n] = 0.0f;
    float pc[COUT], pn[COUT];
#pragma unroll
    for (int n = 0; n < COUT; ++n) pc[n] = wp[n];
#pragma unroll
    for (int k = 0; k < CIN; ++k) {                       // logical channel k sits in physical slot phys(k)
        if (k + 1 < CIN) {
#pragma unroll
            for (int n = 0; n < COUT; ++n) pn[n] = wp[(k + 1) * COUT + n];
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        const int pk = (k & ~7) | ((k & 1) << 2) | ((k & 7) >> 1);
        const float dk = d[pk];
#pragma unroll
        for (int n = 0; n < COUT; ++n) acc[n] = fmaf(dk, pc[n], acc[n]);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                          "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < COUT; ++n) pc[n] = pn[n];
    }
#pragma unroll
    for (int n = 0; n < COUT; ++n) pc[n] = psc[n];
    constexpr int OP = COUT + 4;
    __syncthreads();                                       // every thread is done reading the stem tile
    float* ot = tile;
#pragma unroll
    for (int n4 = 0; n4 < COUT / 4; ++n4) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(acc[n4 * 4 + j], pc[n4 * 4 + j], psh[n4 * 4 + j]);
        *(f32x4*)(ot + threadIdx.x * OP + n4 * 4) = v;
    }
    __syncthreads();
    float* obase = out + (lb.out_off + (long long)frame * lb.Ho * lb.Wo) * COUT;
    const int cols = min(T, lb.Wo - ox0);
#pragma unroll
    for (int k = 0; k < COUT / 4; ++k) {
        const int q = threadIdx.x + k * 256;
        const int row = q / (T * COUT / 4), rem = q - row * (T * COUT / 4);
        const int px2 = rem / (COUT / 4), part = rem - px2 * (COUT / 4);
        if (oy0 + row < lb.Ho && px2 < cols)
            *(f32x4*)(obase + ((long long)(oy0 + row) * lb.Wo + ox0 + px2) * COUT + part * 4) = *(const f32x4*)(ot + (row * T + px2) * OP + part * 4);
    }
}

bool stem_block_fusable(int stem_out, const BlockPack& b) {
    return stem_out == 24 && !b.has_expand && b.stride == 1 && !b.residual && b.cin == 24 && b.cout == 16 && b.pr_logical != nullptr;
}

hipError_t launch_stem_block(const ImageSet& imgs, const float* stem_w, const float* stem_scale, const float* stem_shift, const BlockPack& b,
                             float* out, const Geom& g_stem, const Geom& g_block, hipStream_t s) {
    StemBlockArgs a;
    a.stem_w = stem_w; a.stem_scale = stem_scale; a.stem_shift = stem_shift;
    a.dw_w = b.dw.w; a.dw_scale = b.dw.scale; a.dw_shift = b.dw.shift;
    a.pr_w = b.pr_logical; a.pr_scale = b.pr.scale; a.pr_shift = b.pr.shift;
    a.out = out;
    int maxtiles = 0;
    for (int l = 0; l < g_block.n_levels; ++l) maxtiles = max(maxtiles, ((g_block.lv[l].Wo + 15) / 16) * ((g_block.lv[l].Ho + 15) / 16));
    hipLaunchKernelGGL((k_stem_block2<24, 16>), dim3(maxtiles, g_block.n_levels * g_block.batch), dim3(256), 0, s, imgs, a.stem_w, a.stem_scale,
                                    a.stem_shift, a.dw_w, a.dw_scale, a.dw_shift, a.pr_w, a.pr_scale, a.pr_shift, a.out, g_stem, g_block);
    return hipGetLastError();
}

template <int STRIDE, int TH, int TW>
static hipError_t launch_block_fused_t(const FusedArgs& a, const Geom& g, int nto, hipStream_t s) {
    int maxtiles = 0;
    for (int l = 0; l < g.n_levels; ++l) maxtiles = max(maxtiles, ((g.lv[l].Wo + TW - 1) / TW) * ((g.lv[l].Ho + TH - 1) / TH));
    dim3 grid(maxtiles, g.n_levels * g.batch);
    switch (nto) {
        case 1: hipLaunchKernelGGL((k_block_fused<STRIDE, TH, TW, 1>), grid, dim3(256), 0, s, a, g); break;
        case 2: hipLaunchKernelGGL((k_block_fused<STRIDE, TH, TW, 2>), grid, dim3(256), 0, s, a, g); break;
        case 3: hipLaunchKernelGGL((k_block_fused<STRIDE, TH, TW, 3>), grid, dim3(256), 0, s, a, g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool block_fusable(const BlockPack& b) {
    const int nto = (b.cout + 31) / 32;
    if (nto > 3 || (b.stride != 1 && b.stride != 2)) return false;
    if (!b.has_expand && b.expand > 32) return false;
    if (b.stride == 2 && b.cin > 24 && !(b.cin == 96 && nto == 2)) return false;   // wide stride-2 blocks: only layer_8's shape has a (register-diet) variant
    return b.cin % 8 == 0 && b.expand % 8 == 0;
}

hipError_t launch_block_fused(const float* X, const BlockPack& b, float* out, const Geom& g, hipStream_t s) {
    FusedArgs a;
    a.X = X;
    a.Wex = (const f32x4*)b.ex.w; a.ex_scale = b.ex.scale; a.ex_shift = b.ex.shift; a.ex_nt_total = b.ex.nt_total;
    a.Wdw = b.dw.w; a.dw_scale = b.dw.scale; a.dw_shift = b.dw.shift;
    a.Wpr = (const f32x4*)b.pr.w; a.pr_scale = b.pr.scale; a.pr_shift = b.pr.shift; a.pr_nt_total = b.pr.nt_total;
    { const char* v = getenv("HFNET_FUSE_ABLATE"); a.ablate = v ? atoi(v) : 0; }
    a.out = out; a.cin = b.cin; a.cexp = b.expand; a.cout = b.cout; a.residual = b.residual; a.has_expand = b.has_expand;
    const int nto = (b.cout + 31) / 32;
    static const bool use_v2 = []() { const char* v = getenv("HFNET_FUSE_V2"); return v ? atoi(v) != 0 : true; }();
    if (use_v2 && !b.has_expand && b.stride == 1 && !b.residual && b.cin == 24 && b.cout == 16 && b.pr_logical) {
        int maxtiles = 0;
        for (int l = 0; l < g.n_levels; ++l) maxtiles = max(maxtiles, ((g.lv[l].Wo + 15) / 16) * ((g.lv[l].Ho + 15) / 16));
        hipLaunchKernelGGL((k_block_noexpand<24, 16>), dim3(maxtiles, g.n_levels * g.batch), dim3(256), 0, s, a.X, a.out, a.Wdw, a.dw_scale,
                           a.dw_shift, (const float*)b.pr_logical, a.pr_scale, a.pr_shift, g);
        return hipGetLastError();
    }
    const bool use_v3 = []() { const char* v = getenv("HFNET_FUSE_V3"); return v ? atoi(v) != 0 : false; }();   // experiment; read per launch so tests can switch it
    if (use_v3 && b.has_expand && b.pr.nt_total == nto) {
        const int kq = b.cin / 8, st = b.stride;
        const int tw3 = []() { const char* v = getenv("HFNET_FUSE3_S2_TW"); return v ? atoi(v) : 12; }();
        const bool persist = []() { const char* v = getenv("HFNET_FUSE3_PERSIST"); return v ? atoi(v) != 0 : true; }();
        if (persist) {
            if (st == 2 && kq == 2 && nto == 1) return launch_block_fused3p_t<2, 1, 2, 8>(a, g, s);     // (8x12 tiles do not fit the LDS)
            if (st == 2 && kq == 3 && nto == 1) return launch_block_fused3p_t<2, 1, 3, 8>(a, g, s);
            if (st == 1 && kq == 3 && nto == 1) return launch_block_fused3p_t<1, 1, 3, 16>(a, g, s);
            if (st == 1 && kq == 3 && nto == 2) return launch_block_fused3p_t<1, 2, 3, 16>(a, g, s);
            if (st == 1 && kq == 6 && nto == 3) return launch_block_fused3p_t<1, 3, 6, 16>(a, g, s);
            if (st == 1 && kq == 6 && nto == 2) return launch_block_fused3p_t<1, 2, 6, 16>(a, g, s);
        }
        if (st == 2 && kq == 2 && nto == 1) return tw3 == 12 ? launch_block_fused3_t<2, 1, 2, 12>(a, g, s) : launch_block_fused3_t<2, 1, 2, 8>(a, g, s);
        if (st == 2 && kq == 3 && nto == 1) return tw3 == 12 ? launch_block_fused3_t<2, 1, 3, 12>(a, g, s) : launch_block_fused3_t<2, 1, 3, 8>(a, g, s);
        if (st == 1 && kq == 3 && nto == 1) return launch_block_fused3_t<1, 1, 3, 16>(a, g, s);
        if (st == 1 && kq == 3 && nto == 2) return launch_block_fused3_t<1, 2, 3, 16>(a, g, s);
        if (st == 1 && kq == 6 && nto == 3) return launch_block_fused3_t<1, 3, 6, 16>(a, g, s);
        if (st == 1 && kq == 6 && nto == 2) return launch_block_fused3_t<1, 2, 6, 16>(a, g, s);
        if (st == 1 && kq == 9 && nto == 3) return launch_block_fused3_t<1, 3, 9, 16>(a, g, s);
    }
    if (use_v2) {
        const int kq = b.cin / 8, st = b.stride;
        if (!b.has_expand && st == 1 && nto == 1) return launch_block_fused2_t<1, 1, 0, false>(a, g, s);
        static const int tw2 = []() { const char* v = getenv("HFNET_FUSE_S2_TW"); return v ? atoi(v) : 8; }();   // 8x8 tiles: 3 workgroups per CU beat 8x12 at 2
        if (b.has_expand && st == 2 && kq == 2 && nto == 1 && tw2 == 12) return launch_block_fused2_t<2, 1, 2, true, 12>(a, g, s);
        if (b.has_expand && st == 2 && kq == 3 && nto == 1 && tw2 == 12) return launch_block_fused2_t<2, 1, 3, true, 12>(a, g, s);
        if (b.has_expand && st == 2 && kq == 2 && nto == 1) return launch_block_fused2_t<2, 1, 2, true>(a, g, s);
        if (b.has_expand && st == 1 && kq == 3 && nto == 1) return launch_block_fused2_t<1, 1, 3, true>(a, g, s);
        if (b.has_expand && st == 2 && kq == 3 && nto == 1) return launch_block_fused2_t<2, 1, 3, true>(a, g, s);
        if (b.has_expand && st == 1 && kq == 3 && nto == 2) return launch_block_fused2_t<1, 2, 3, true>(a, g, s);
        if (b.has_expand && st == 1 && kq == 6 && nto == 3) return launch_block_fused2_t<1, 3, 6, true>(a, g, s);
        if (b.has_expand && st == 1 && kq == 6 && nto == 2) return launch_block_fused2_t<1, 2, 6, true>(a, g, s);
        if (b.has_expand && st == 1 && kq == 9 && nto == 3) return launch_block_fused2_t<1, 3, 9, true>(a, g, s);
        if (b.has_expand && st == 2 && kq == 12 && nto == 2) return launch_block_fused2_t<2, 2, 12, true>(a, g, s);
    }
    if (b.stride == 1) return launch_block_fused_t<1, 8, 16>(a, g, nto, s);
    return launch_block_fused_t<2, 8, 8>(a, g, nto, s);
}

// =========================================================================== depthwise 3x3
// One thread per (output pixel, 4 channels).  HBM / L2-bound: 9 (stride 1) or 2.25 (stride 2)
// cached reads and one write per output element.
__global__ __launch_bounds__(256) void k_depthwise(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ out, int C, int stride, Geom g) {
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int c4 = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)lv.Ho * lv.Wo * c4) return;
    const int op = (int)(idx / c4), cq = (int)(idx - (long long)op * c4);
    const int oy = op / lv.Wo, ox = op - oy * lv.Wo;
    const float* ip = in + (lv.in_off + (long long)frame * lv.H * lv.W) * C + cq * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - lv.pt + ky;
        if (iy < 0 || iy >= lv.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride - lv.pl + kx;
            if (ix < 0 || ix >= lv.W) continue;
            const f32x4 xv = *(const f32x4*)(ip + (long long)(iy * lv.W + ix) * C);
            const f32x4 wv = *(const f32x4*)(w + (ky * 3 + kx) * C + cq * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv[j], wv[j], acc[j]);
        }
    }
    const f32x4 sc = *(const f32x4*)(scale + cq * 4), sh = *(const f32x4*)(shift + cq * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = relu6f(fmaf(acc[j], sc[j], sh[j]));
    *(f32x4*)(out + (lv.out_off + (long long)frame * lv.Ho * lv.Wo + op) * C + cq * 4) = o;
}

hipError_t launch_depthwise(const float* in, const DwPack& dp, int stride, float* out, const Geom& g, hipStream_t s) {
    long long maxwork = 0;
    for (int l = 0; l < g.n_levels; ++l) maxwork = max(maxwork, (long long)g.lv[l].Ho * g.lv[l].Wo * (dp.c / 4));
    dim3 grid((unsigned)((maxwork + 255) / 256), g.n_levels * g.batch);
    hipLaunchKernelGGL(k_depthwise, grid, dim3(256), 0, s, in, dp.w, dp.scale, dp.shift, out, dp.c, stride, g);
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
