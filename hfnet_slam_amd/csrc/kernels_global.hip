// kernels_global.hip -- NetVLAD global-descriptor head (hfnet/models/utils/layers.py:57-109).
// The memberships 1x1 conv runs on the MFMA pointwise kernel; everything after it is here.
#include "kernels.hpp"

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

// softmax over the n (<= 64) leading entries of every row, left-to-right sum (layers.py:75)
__global__ __launch_bounds__(256) void k_softmax_rows(float* __restrict__ x, long long rows, int n, int ld) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float* r = x + row * ld;
    float mx = r[0];
    for (int k = 1; k < n; ++k) mx = fmaxf(mx, r[k]);
    float sum = 0.0f;
    for (int k = 0; k < n; ++k) { const float e = hf_expf_g(r[k] - mx); r[k] = e; sum = sum + e; }
    for (int k = 0; k < n; ++k) r[k] = r[k] / sum;
}

hipError_t launch_softmax_rows(float* x, long long rows, int n, int ld, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, x, rows, n, ld);
    return hipGetLastError();
}

// desc[k][d] = sum_p (c[k][d] - f[p][d]) * m[p][k], pixels left to right (layers.py:82-87).
// feat is in the device channel layout; thread handles physical slot pd == logical channel d.  The
// chain over pixels is sequential by definition; the loads are unrolled 8 deep to hide their latency.
__global__ __launch_bounds__(256) void k_vlad_aggregate(const float* __restrict__ feat, const float* __restrict__ memb,
                                                        const float* __restrict__ clusters, float* __restrict__ out, int P, int D, int K) {
    const int frame = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= K * D) return;
    const int k = t / D, pd = t - k * D;
    const int rr = pd & 7;
    const int d = (pd & ~7) | (rr < 4 ? 2 * rr : 2 * (rr - 4) + 1);
    const float c = clusters[k * D + d];
    const float* f = feat + (long long)frame * P * D + pd;
    const float* m = memb + (long long)frame * P * K + k;
    float acc = 0.0f;
    int p = 0;
    for (; p + 8 <= P; p += 8) {
        float fv[8], mv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { fv[j] = f[(long long)(p + j) * D]; mv[j] = m[(long long)(p + j) * K]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float r = c - fv[j]; const float tt = r * mv[j]; acc = acc + tt; }
    }
    for (; p < P; ++p) { const float r = c - f[(long long)p * D]; const float tt = r * m[(long long)p * K]; acc = acc + tt; }
    out[(long long)frame * K * D + k * D + d] = acc;
}

// block-wide tree256 sum of squares of v[0..n): partial tid accumulates elements tid + 256 j
__device__ __forceinline__ float block_sumsq_tree256(const float* v, int n, float* red) {
    float p = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) p = fmaf(v[i], v[i], p);
    red[threadIdx.x] = p;
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
__global__ __launch_bounds__(256) void k_vlad_norm(const float* __restrict__ raw, float* __restrict__ vlad_tap, float* __restrict__ out,
                                                   int D, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* v = smem;            // K*D
    float* red = smem + K * D;  // 256
    const int frame = blockIdx.x, N = K * D;
    const float* src = raw + (long long)frame * N;
    for (int i = threadIdx.x; i < N; i += 256) v[i] = src[i];
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        float ss = 0.0f;
        for (int k = 0; k < K; ++k) ss = fmaf(v[k * D + d], v[k * D + d], ss);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        for (int k = 0; k < K; ++k) v[k * D + d] = v[k * D + d] * inv;
    }
    __syncthreads();
    float ss = block_sumsq_tree256(v, N, red);
    float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int i = threadIdx.x; i < N; i += 256) v[i] = v[i] * inv;
    __syncthreads();
    if (vlad_tap) for (int i = threadIdx.x; i < N; i += 256) vlad_tap[(long long)frame * N + i] = v[i];
    ss = block_sumsq_tree256(v, N, red);
    inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int i = threadIdx.x; i < N; i += 256) out[(long long)frame * N + i] = v[i] * inv;
}

hipError_t launch_vlad(const float* feat, const float* memb, const float* clusters, float* vlad_tap, float* out, float* scratch,
                       int frames, int P, int D, int K, hipStream_t s) {
    if (frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_vlad_aggregate, dim3((K * D + 255) / 256, frames), dim3(256), 0, s, feat, memb, clusters, scratch, P, D, K);
    const size_t lds = (size_t)(K * D + 256) * sizeof(float);
    hipLaunchKernelGGL(k_vlad_norm, dim3(frames), dim3(256), lds, s, scratch, vlad_tap, out, D, K);
    return hipGetLastError();
}

// y[f][j] = tree256_dot(x[f], wt[j]) + b[j]  (slim.fully_connected, layers.py:99-107).
// A wave owns JW = 4 outputs and up to FB = 8 frames per pass: lane l holds the tree256 partials
// 4l..4l+3 of every (output, frame) pair; per 256 inputs it issues 4 weight + FB activation 16-byte
// loads for 4*FB*4 fmas, so the activations (L2-resident) are not re-streamed once per output.  The
// 126 MB of transposed weights are read once per pass: HBM-bound.
__global__ __launch_bounds__(256) void k_fc(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                            float* __restrict__ y, int frames, int n_in, int n_out) {
    constexpr int JW = 4, FB = 8;
    const int j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * JW;
    if (j0 >= n_out) return;
    const int lane = threadIdx.x & 63;
    const float* w[JW];
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) w[jj] = wt + (long long)min(j0 + jj, n_out - 1) * n_in + lane * 4;
    for (int f0 = 0; f0 < frames; f0 += FB) {
        const int nf = min(FB, frames - f0);
        f32x4 p[JW][FB];
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
#pragma unroll
            for (int f = 0; f < FB; ++f) p[jj][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < n_in; q += 256) {
            f32x4 wv[JW], xv[FB];
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) wv[jj] = *(const f32x4*)(w[jj] + q);
#pragma unroll
            for (int f = 0; f < FB; ++f) xv[f] = *(const f32x4*)(x + (long long)(f0 + min(f, nf - 1)) * n_in + q + lane * 4);
#pragma unroll
            for (int jj = 0; jj < JW; ++jj)
#pragma unroll
                for (int f = 0; f < FB; ++f)
#pragma unroll
                    for (int c = 0; c < 4; ++c) p[jj][f][c] = fmaf(xv[f][c], wv[jj][c], p[jj][f][c]);
        }
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
#pragma unroll
            for (int f = 0; f < FB; ++f) {
                f32x4 t = p[jj][f];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = t[c] + __shfl_xor(t[c], off, 64);
                }
                const float a = t[0] + t[2], b = t[1] + t[3];
                if (lane == 0 && f < nf && j0 + jj < n_out) y[(long long)(f0 + f) * n_out + j0 + jj] = (a + b) + bias[j0 + jj];
            }
    }
}

__global__ __launch_bounds__(256) void k_l2norm_vec(const float* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ float red[256];
    const float* v = in + (long long)blockIdx.x * n;
    const float ss = block_sumsq_tree256(v, n, red);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int i = threadIdx.x; i < n; i += 256) out[(long long)blockIdx.x * n + i] = v[i] * inv;
}

hipError_t launch_fc_l2(const float* x, const float* wt, const float* bias, float* y_raw, float* out, int frames, int n_in,
                        int n_out, hipStream_t s) {
    if (frames <= 0) return hipSuccess;
    if (n_in % 256 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fc, dim3((n_out + 15) / 16), dim3(256), 0, s, x, wt, bias, y_raw, frames, n_in, n_out);
    hipLaunchKernelGGL(k_l2norm_vec, dim3(frames), dim3(256), 0, s, y_raw, out, n_out);
    return hipGetLastError();
}

}  // namespace hfnet
