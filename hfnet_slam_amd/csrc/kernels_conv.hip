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

namespace hfnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// =========================================================================== pyramid resize
// OpenCV 4.2 cv::resize(INTER_LINEAR) on CV_8UC1: 11-bit fixed-point coefficients, horizontal pass
// to int, vertical pass ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2.  Integer-exact.
__global__ __launch_bounds__(256) void k_resize_u8(const uint8_t* __restrict__ src, int sw, int sh, int s_row, long long s_frame,
                                                   uint8_t* __restrict__ dst, int dw, int dh, int d_row, long long d_frame,
                                                   const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                                   const int* __restrict__ yofs, const short* __restrict__ ibeta) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const uint8_t* sp = src + (long long)blockIdx.z * s_frame;
    const int sy = yofs[dy];
    const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1);
    const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
    const uint8_t* r0p = sp + (long long)y0 * s_row;
    const uint8_t* r1p = sp + (long long)y1 * s_row;
    const int r0 = r0p[sx] * a0 + r0p[sx1] * a1;
    const int r1 = r1p[sx] * a0 + r1p[sx1] * a1;
    int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    dst[(long long)blockIdx.z * d_frame + (long long)dy * d_row + dx] = (uint8_t)v;
}

hipError_t launch_resize_u8(const uint8_t* src, int sw, int sh, int s_row, long long s_frame, uint8_t* dst, int dw, int dh,
                            int d_row, long long d_frame, const int* xofs, const short* ialpha, const int* yofs,
                            const short* ibeta, int batch, hipStream_t s) {
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, batch);
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

hipError_t launch_stem(const ImageSet& imgs, const float* w, const float* scale, const float* shift, int cout, float* out,
                       const Geom& g, hipStream_t s) {
    int maxpix = 0;
    for (int l = 0; l < g.n_levels; ++l) maxpix = max(maxpix, g.lv[l].Ho * g.lv[l].Wo);
    dim3 grid((maxpix + 255) / 256, g.n_levels * g.batch);
    hipLaunchKernelGGL(k_stem, grid, dim3(256), 0, s, imgs, w, scale, shift, cout, out, g);
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
};

template <int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[NT], int nt0, long long row_base, long long row_limit,
                                              int half, int r) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = (nt0 + nt) * 32 + r;
        if (col >= a.n) continue;
        const float sc = a.scale[col], sh = a.shift[col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const long long row = row_base + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            if (row >= row_limit) continue;
            float v = fmaf(acc[nt][reg], sc, sh);
            if (a.relu6) v = relu6f(v);
            if (a.res) v = v + a.res[row * a.n + col];
            a.out[row * a.n + col] = v;
        }
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
    for (int kq = 0; kq < KQ; ++kq) {
        const f32x4 av = *(const f32x4*)(ap + kq * 8);
        f32x4 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = wp[(size_t)kq * wstep + (size_t)nt * 64];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nt][t], acc[nt], 0, 0, 0);
    }
    conv_epilogue<NT>(a, acc, nt0, row0, a.P, half, r);
}

// dense 3x3, stride 1, 'SAME' (pad 1): tiles of 32 consecutive pixels of ONE image; out-of-image taps
// contribute fma(0, w, acc) == acc, i.e. they are skipped exactly as the oracle skips them.
template <int NT>
__global__ __launch_bounds__(256, 2) void k_conv3x3(ConvArgs a, Geom g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r = lane & 31;
    const int image = blockIdx.z, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int npix = lv.H * lv.W;
    const int p0 = blockIdx.x * 128 + wave * 32;
    if (p0 >= npix) return;
    const int nt0 = blockIdx.y * NT;
    const bool pvalid = (p0 + r) < npix;
    const int p = pvalid ? p0 + r : npix - 1;
    const int y = p / lv.W, x = p - y * lv.W;
    const long long in_base = lv.in_off + (long long)frame * npix;
    const f32x4* wp = a.W + ((size_t)nt0 * 64 + lane);
    const size_t wstep = (size_t)a.nt_total * 64;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
    const int KQ = a.cin >> 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = y + ky - 1, ix = x + kx - 1;
        const bool ok = pvalid && iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W;
        const float* ap = a.A + (in_base + (long long)(ok ? iy * lv.W + ix : 0)) * a.cin + half * 4;
        for (int kq = 0; kq < KQ; ++kq) {
            f32x4 av = zero;
            if (ok) av = *(const f32x4*)(ap + kq * 8);
            f32x4 bv[NT];
            const size_t wrow = (size_t)(tap * KQ + kq) * wstep;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = wp[wrow + (size_t)nt * 64];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nt][t], acc[nt], 0, 0, 0);
        }
    }
    const long long out_base = lv.out_off + (long long)frame * npix;
    conv_epilogue<NT>(a, acc, nt0, out_base + p0, out_base + npix, half, r);
}

template <int NT>
static void launch_pw_nt(const ConvArgs& a, dim3 grid, hipStream_t s) { hipLaunchKernelGGL(k_pointwise<NT>, grid, dim3(256), 0, s, a); }
template <int NT>
static void launch_c3_nt(const ConvArgs& a, const Geom& g, dim3 grid, hipStream_t s) { hipLaunchKernelGGL(k_conv3x3<NT>, grid, dim3(256), 0, s, a, g); }

static ConvArgs make_args(const float* A, const ConvPack& cp, const float* res, float* out, long long P, int relu6) {
    ConvArgs a;
    a.A = A; a.W = (const f32x4*)cp.w; a.scale = cp.scale; a.shift = cp.shift; a.res = res; a.out = out;
    a.P = P; a.cin = cp.cin; a.n = cp.n; a.nt_total = cp.nt_total; a.relu6 = relu6;
    return a;
}

hipError_t launch_pointwise(const float* A, const ConvPack& cp, const float* residual, float* out, long long P, int relu6,
                            hipStream_t s) {
    if (P <= 0) return hipSuccess;
    const ConvArgs a = make_args(A, cp, residual, out, P, relu6);
    dim3 grid((unsigned)((P + 127) / 128), cp.nt_total / cp.nt_per_block);
    switch (cp.nt_per_block) {
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

hipError_t launch_conv3x3(const float* A, const ConvPack& cp, float* out, int relu6, const Geom& g, hipStream_t s) {
    const ConvArgs a = make_args(A, cp, nullptr, out, 0, relu6);
    int maxpix = 0;
    for (int l = 0; l < g.n_levels; ++l) maxpix = max(maxpix, g.lv[l].H * g.lv[l].W);
    dim3 grid((maxpix + 127) / 128, cp.nt_total / cp.nt_per_block, g.n_levels * g.batch);
    switch (cp.nt_per_block) {
        case 1: launch_c3_nt<1>(a, g, grid, s); break;
        case 2: launch_c3_nt<2>(a, g, grid, s); break;
        case 3: launch_c3_nt<3>(a, g, grid, s); break;
        case 4: launch_c3_nt<4>(a, g, grid, s); break;
        case 5: launch_c3_nt<5>(a, g, grid, s); break;
        case 6: launch_c3_nt<6>(a, g, grid, s); break;
        case 7: launch_c3_nt<7>(a, g, grid, s); break;
        case 8: launch_c3_nt<8>(a, g, grid, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
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
