// kernels_tail.hip -- single-frame kernels for layers 8-18 (the global branch): what Tracking pays per frame
// (Tracking.cc:850-896, Frame.cc:420-427) is a chain of ~35 small launches on a 30 x 47 / 15 x 24 map, every one of them
// at the dependent-launch floor or as long as its own accumulation chain.  A 1x1 projection over K expanded channels on
// v_mfma_f32_32x32x2_f32 is K / 2 DEPENDENT instructions of 64 cycles each (K = 720: 10 us however few pixels there are);
// v_mfma_f32_16x16x4_f32 consumes four k per 32 cycles -- the same k-ascending fma chain from the bias
// (tools/micro/mfma_order.hip), a quarter of the latency.  Its A operand wants four LOGICAL channels 4 apart in one 16-byte
// piece, which no tensor in HBM provides, so the depthwise convolution that produces the projection's input runs in the
// same workgroup and leaves its result in LDS in exactly that order:
//
//   k_dwproject   depthwise 3x3 + BN + ReLU6  ->  1x1 project + BN [+ block input]  [-> the NEXT block's 1x1 expansion + BN + ReLU6]
//                                                                                     (conv_blocks.py:263-311)
//
// One launch per block instead of three (the next block's expansion only needs the projected pixel itself), the depthwise
// tensor never reaches HBM, and the projection's chain is four times shorter.
// Numerics: every output is the oracle's chain -- depthwise: bias, then the taps in (ky, kx) order; projection: bias, then
// the expanded channels in ascending logical order, one fma each; the residual is added last.
#include "kernels.hpp"

#include <mutex>

namespace hfnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu6t(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }
__device__ __forceinline__ float hf_expf_t(float x) {   // == oracle hfo_expf (as in kernels_global.hip / kernels_detect.hip)
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
__device__ __forceinline__ int d_logical_of_phys(int p) { const int r = p & 7; return (p & ~7) | (r < 4 ? 2 * r : 2 * (r - 4) + 1); }
// slot of logical channel l inside its group of 16: lane group g = l % 4 of the 16x16x4 MFMA finds channels g, 4 + g, 8 + g, 12 + g
// (its operands of four consecutive MFMAs) in slots 4 g .. 4 g + 3
__device__ __forceinline__ int d_slot16_of_logical(int l) { const int r = l & 15; return (l & ~15) | ((r & 3) << 2) | (r >> 2); }

struct DwProjArgs {
    const float* E;          // expanded tensor [pixels of the block input][cexp], device channel order, after ReLU6
    const float* Wdw;        // [9][cexp] device order, BN folded
    const float* dw_bias;    // [cexp]
    const f32x4* Wpr;        // ConvPack16: [cexp / 16][n16][64][4]
    const float* pr_bias;    // [>= n16 * 16] device column order
    const float* R;          // block input for the residual ([pixels][cout]) or null
    float* out;              // [pixels of the block output][cout]
    int cexp, cout, n16, nsplit;
    // NEXT: the 1x1 convolution that consumes this block's output (the next block's expansion, or the NetVLAD memberships
    // conv after layer 18), evaluated on the tile while it is still in LDS
    const f32x4* Wnx;        // ConvPack16: [ceil(cout / 16)][nx_n16][64][4]
    const float* nx_bias;
    float* nx_out;           // [pixels of the block output][nx_ld]
    int nx_n, nx_n16, nx_ld, nx_relu;
    int nx_softmax;          // the rows of nx_out are softmaxed (the NetVLAD memberships, layers.py:75; nx_n <= 64, one workgroup per tile)
};

// TH x 8 output pixels per workgroup (16: one MFMA row tile), 8 waves.
//   phase 1  the (TH - 1) s + 3 rows x 7 s + 3 columns of the expanded tensor go through LDS in chunks of 128 channels
//            (coalesced 512-byte row pieces, all of them requested before the first is used); thread =
//            (pixel, channel quad): nine 16-byte LDS reads, 36 fma, ReLU6; the result lands in D[pixel][slot] with the
//            channels of every group of 16 in MFMA slot order
//   phase 2  a wave owns 16-column output tiles of the projection: K / 16 steps of one 16-byte LDS read (A), one 16-byte
//            weight load (B, eight steps ahead) and four dependent MFMAs; + residual, store
//   phase 3  (NEXT) the projected tile, kept in LDS in slot order as well, times the next 1x1 convolution's weights: the
//            workgroups that share a pixel tile (nsplit of them, each repeats phases 1 and 2) take a range of its column tiles
template <int STRIDE, bool NEXT>
__global__ __launch_bounds__(512) void k_dwproject(DwProjArgs a, Geom g) {
    constexpr int TH = 2, TW = 8, TP = TH * TW, IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3, NPOS = IH * IW;
    constexpr int CC = 128, HP = CC + 4;                         // channels per chunk, LDS pitch of a halo position
    constexpr int PIECES = NPOS * (CC / 4), PER_THREAD = (PIECES + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int DP = a.cexp + 4;                                   // pitch of a D row: 16 rows spread over all banks for 16-byte reads
    const int KB2 = (a.cout + 15) >> 4, PP = KB2 * 16 + 4;       // NEXT: groups of 16 projected channels, pitch of a P row
    float* D = lds;                                              // [TP][DP]
    float* H = lds + TP * DP;                                    // [NPOS][HP]
    float* Wl = H + NPOS * HP;                                   // [10][cexp]: depthwise taps and bias
    float* P = Wl + 10 * a.cexp;                                 // NEXT: [TP][PP] projected tile (+ residual), slot order, zero padded
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: weight addresses stay scalar)
    const int image = blockIdx.y, level = image / g.batch, frame = image - level * g.batch;
    const LevelGeom lv = g.lv[level];
    const int tiles_x = (lv.Wo + TW - 1) / TW;
    const int split = blockIdx.x % a.nsplit, tile = blockIdx.x / a.nsplit;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * STRIDE - lv.pt, ix0 = ox0 * STRIDE - lv.pl;
    const float* __restrict__ Ein = a.E + (lv.in_off + (long long)frame * lv.H * lv.W) * a.cexp;
    const long long obase = lv.out_off + (long long)frame * lv.Ho * lv.Wo;
    const int n_chunks = (a.cexp + CC - 1) / CC;
    const int j = lane & 15, gq = lane >> 4;                     // MFMA roles: column / row of a tile, k group

    // projection tiles of this wave: NEXT: every workgroup needs the whole projected tile (wave, wave + 8, ...); otherwise
    // the workgroups of a pixel tile split them.  The first tile's weights and residual values are requested before phase 1.
    // The weights come from HBM (last used a frame ago) at ~2 us per dependent load, the chain of a 720-channel projection is
    // 2.5 us: the number of weight pieces a wave has in flight is what phase 2 takes.  PF pieces ride in registers: PF0 of them
    // requested here, the others once phase 1 has released its staging registers; a wave with two tiles (layer 18) walks them
    // as one stream (positions of a tile padded to a multiple of PF, so that a piece's ring slot is a compile-time index).
    constexpr int PF = 16, PF0 = 8;
    const int nt_first = NEXT ? wave : split * 8 + wave, nt_step = NEXT ? 8 : (1 << 20);
    const int KB = a.cexp >> 4, KBP = (KB + PF - 1) / PF * PF;
    const f32x4* __restrict__ wbase = a.Wpr + lane;
    auto wpiece = [&](int nt, int kb) -> f32x4 { return wbase[((size_t)kb * a.n16 + nt) * 64]; };
    f32x4 bq[PF];
#pragma unroll
    for (int u = 0; u < PF0; ++u) bq[u] = wpiece(min(nt_first, a.n16 - 1), min(u, KB - 1));
    long long orow[4];
    bool ovalid[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int p = 4 * gq + t, oy = oy0 + p / TW, ox = ox0 + p % TW;
        ovalid[t] = oy < lv.Ho && ox < lv.Wo;
        orow[t] = obase + (long long)min(oy, lv.Ho - 1) * lv.Wo + min(ox, lv.Wo - 1);
    }
    const float pb_first = a.pr_bias[min(nt_first, a.n16 - 1) * 16 + j];
    // NEXT: this wave's first column tile of the next convolution -- bias and all weight pieces (cout <= 256: at most 16)
    const int per = NEXT ? (a.nx_n16 + a.nsplit - 1) / a.nsplit : 0, xt_first = split * per + wave, xt_end = NEXT ? min((split + 1) * per, a.nx_n16) : 0;
    const size_t xstep = (size_t)a.nx_n16 * 64;
    constexpr int XVN = STRIDE == 2 ? 8 : 16;                    // (the stride-2 blocks have <= 128 output channels and many halo registers)
    f32x4 xv[XVN];
    float xb_first = 0.f;
    if (NEXT) {
        const f32x4* __restrict__ xp = a.Wnx + ((size_t)min(xt_first, a.nx_n16 - 1) * 64 + lane);
#pragma unroll
        for (int kb = 0; kb < XVN; ++kb) if (kb < KB2) xv[kb] = xp[(size_t)kb * xstep];
        xb_first = a.nx_bias[min(xt_first, a.nx_n16 - 1) * 16 + j];
    }
    float rv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.R && nt_first < a.n16 && nt_first * 16 + j < a.cout) {
#pragma unroll
        for (int t = 0; t < 4; ++t) rv[t] = a.R[orow[t] * a.cout + nt_first * 16 + j];
    }
    if (NEXT) {                                                  // padding slots of the last group of 16 (and the pitch padding)
        for (int p = tid; p < TP * 20; p += 512) P[(p / 20) * PP + (KB2 - 1) * 16 + p % 20] = 0.0f;
    }

    // ---- phase 1.  A workgroup is alone on its CU and every wait it executes is exposed, so everything it will read from
    //      memory is requested up front: the halo pieces of ALL channel chunks (registers), the depthwise taps and bias
    //      (through LDS).  The chunk loop then only moves registers to LDS and computes.
    constexpr int MAXCH = 6;                                     // cexp <= 768
    f32x4 stage[MAXCH][PER_THREAD];
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
        if (ch < n_chunks) {
            const int c0 = ch * CC;
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k) {
                const int p = tid + k * 512;
                const int pos = p / (CC / 4), qq = p - pos * (CC / 4);
                const int hy = pos / IW, hx = pos - hy * IW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                const bool ok = p < PIECES && c0 + 4 * qq < a.cexp && iy >= 0 && iy < lv.H && ix >= 0 && ix < lv.W;
                stage[ch][k] = ok ? *(const f32x4*)(Ein + ((long long)iy * lv.W + ix) * a.cexp + c0 + 4 * qq) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    for (int p = tid; p < 10 * (a.cexp >> 2); p += 512) {          // [9 taps + bias][cexp]
        const int row = p / (a.cexp >> 2), c4 = p - row * (a.cexp >> 2);
        *(f32x4*)(Wl + row * a.cexp + 4 * c4) = row < 9 ? *(const f32x4*)(a.Wdw + (size_t)row * a.cexp + 4 * c4) : *(const f32x4*)(a.dw_bias + 4 * c4);
    }
    const int q = tid & 31, px = tid >> 5;                       // compute role: channel quad of the chunk, output pixel of the tile
    const int ty = px / TW, tx = px - ty * TW;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
        if (ch < n_chunks) {                                     // uniform
            const int c0 = ch * CC;
            if (ch) __syncthreads();                             // the previous chunk's halo has been consumed
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k) {
                const int p = tid + k * 512;
                if (p < PIECES) { const int pos = p / (CC / 4), qq = p - pos * (CC / 4); *(f32x4*)(H + pos * HP + 4 * qq) = stage[ch][k]; }
            }
            __syncthreads();
            const int c = c0 + 4 * q;
            if (c < a.cexp) {
                f32x4 acc = *(const f32x4*)(Wl + 9 * a.cexp + c);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 w = *(const f32x4*)(Wl + (ky * 3 + kx) * a.cexp + c);
                        const f32x4 x = *(const f32x4*)(H + ((ty * STRIDE + ky) * IW + tx * STRIDE + kx) * HP + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = fmaf(x[e], w[e], acc[e]);
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) D[px * DP + d_slot16_of_logical(d_logical_of_phys(c + e))] = relu6t(acc[e]);
            }
        }
    }
    __syncthreads();

    // ---- phase 2
#pragma unroll
    for (int u = PF0; u < PF; ++u) bq[u] = wpiece(min(nt_first, a.n16 - 1), min(u, KB - 1));
    // (NEXT: the weights of this wave's SECOND column tile of the next convolution, into the registers phase 1 has released)
    constexpr bool XV2 = false;
    f32x4 xv2[XVN];
    const bool second_xt = XV2 && NEXT && xt_first + 8 < xt_end;
    if (second_xt) {
        const f32x4* __restrict__ xp = a.Wnx + ((size_t)(xt_first + 8) * 64 + lane);
#pragma unroll
        for (int kb = 0; kb < XVN; ++kb) if (kb < KB2) xv2[kb] = xp[(size_t)kb * xstep];
    }
    for (int nt = nt_first; nt < a.n16; nt += nt_step) {
        if (nt != nt_first && a.R && nt * 16 + j < a.cout) {      // (a second tile per wave: only layer 18's 240 columns)
#pragma unroll
            for (int t = 0; t < 4; ++t) rv[t] = a.R[orow[t] * a.cout + nt * 16 + j];
        }
        const float pb = nt == nt_first ? pb_first : a.pr_bias[nt * 16 + j];
        f32x4 acc = {pb, pb, pb, pb};
        const float* __restrict__ ap = D + j * DP + 4 * gq;       // A: row = pixel j of the tile, lane group gq
        const int nt_next = nt + nt_step < a.n16 ? nt + nt_step : -1;
        for (int kb = 0; kb < KBP; kb += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                f32x4 av, bv;
                if (kb + u < KB) {                               // uniform
                    av = *(const f32x4*)(ap + (kb + u) * 16);
                    bv = bq[u];
                }
                // the slot's next piece: PF positions ahead in this tile, or the start of this wave's next tile
                if (kb + u + PF < KB) bq[u] = wpiece(nt, kb + u + PF);
                else if (kb + PF >= KBP && nt_next >= 0 && u < KB) bq[u] = wpiece(nt_next, u);
                if (kb + u < KB) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[t], acc, 0, 0, 0);
                }
            }
        }
        // D fragment: column j, rows (pixels of the tile) 4 gq .. 4 gq + 3
        const int col = nt * 16 + j;
        if (col < a.cout) {
            const int pslot = d_slot16_of_logical(d_logical_of_phys(col));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = acc[t];
                if (a.R) v = v + rv[t];
                if (ovalid[t] && (!NEXT || split == 0)) a.out[orow[t] * a.cout + col] = v;
                if (NEXT) P[(4 * gq + t) * PP + pslot] = v;
            }
        }
    }
    if (!NEXT) return;

    // ---- phase 3: this workgroup's share of the next convolution's column tiles
    __syncthreads();
    for (int xt = xt_first; xt < xt_end; xt += 8) {
        float xb = xb_first;
        if (xt != xt_first) {
            if (XV2) {
#pragma unroll
                for (int kb = 0; kb < XVN; ++kb) xv[kb] = xv2[kb];
            } else {
                const f32x4* __restrict__ xp = a.Wnx + ((size_t)xt * 64 + lane);
#pragma unroll
                for (int kb = 0; kb < XVN; ++kb) if (kb < KB2) xv[kb] = xp[(size_t)kb * xstep];
            }
            xb = a.nx_bias[xt * 16 + j];
        }
        f32x4 acc = {xb, xb, xb, xb};
        const float* __restrict__ ap = P + j * PP + 4 * gq;
#pragma unroll
        for (int kb = 0; kb < XVN; ++kb) {
            if (kb < KB2) {                                      // uniform
                const f32x4 av = *(const f32x4*)(ap + kb * 16);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], xv[kb][t], acc, 0, 0, 0);
            }
        }
        const int col = xt * 16 + j;
        if (col < a.nx_n) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (a.nx_softmax) H[(4 * gq + t) * 68 + col] = acc[t];                 // (the halo buffer is free by now)
                else if (ovalid[t]) a.nx_out[orow[t] * a.nx_ld + col] = a.nx_relu ? relu6t(acc[t]) : acc[t];
            }
        }
    }
    if (a.nx_softmax) {
        // softmax over the nx_n columns of every pixel of the tile, the sum left to right (k_softmax_rows' expressions): one
        // thread per pixel -- 16 short serial chains instead of a launch of its own
        __syncthreads();
        if (tid < TP) {
            const int oy = oy0 + tid / TW, ox = ox0 + tid % TW;
            if (oy < lv.Ho && ox < lv.Wo) {
                float* r = H + tid * 68;
                float mx = r[0];
                for (int k = 1; k < a.nx_n; ++k) mx = fmaxf(mx, r[k]);
                float sum = 0.0f;
                for (int k = 0; k < a.nx_n; ++k) { r[k] = hf_expf_t(r[k] - mx); sum = sum + r[k]; }
                float* o = a.nx_out + (obase + (long long)oy * lv.Wo + ox) * a.nx_ld;
                for (int k = 0; k < a.nx_n; ++k) o[k] = r[k] / sum;
            }
        }
    }
}

static size_t dwproject_lds_bytes(const BlockPack& b, bool next) {
    const size_t npos = b.stride == 1 ? 4 * 10 : 5 * 17;
    return (16 * ((size_t)b.expand + 4) + npos * 132 + 10 * (size_t)b.expand + (next ? 16 * (((size_t)b.cout + 15) / 16 * 16 + 4) : 0)) * sizeof(float);
}

bool dwproject_supported(const BlockPack& b) {
    return b.has_expand && b.pr16.w != nullptr && b.expand % 16 == 0 && b.cout % 8 == 0 && b.cout <= (b.stride == 2 ? 128 : 256) && (b.stride == 1 || b.stride == 2) &&
           b.expand <= 768 && dwproject_lds_bytes(b, true) <= 160 * 1024;
}

hipError_t launch_dwproject(const float* expanded, const BlockPack& b, const float* residual, float* out, const ConvPack16* next,
                            const float* next_bias, float* next_out, int next_relu, int next_softmax, const Geom& g, hipStream_t s) {
    if (!dwproject_supported(b) || (residual && (b.stride != 1 || b.cin != b.cout))) return hipErrorInvalidValue;
    DwProjArgs a;
    a.E = expanded; a.Wdw = b.dw.w; a.dw_bias = b.dw.bias; a.Wpr = (const f32x4*)b.pr16.w; a.pr_bias = b.pr.bias; a.R = residual; a.out = out;
    a.cexp = b.expand; a.cout = b.cout; a.n16 = b.pr16.n16;
    a.Wnx = nullptr; a.nx_bias = nullptr; a.nx_out = nullptr; a.nx_n = a.nx_n16 = a.nx_ld = a.nx_relu = a.nx_softmax = 0;
    if (b.pr.nt_total * 32 < a.n16 * 16) return hipErrorInvalidValue;          // (the shared bias array covers the padded columns)
    if (next) {
        if (next->cin != b.cout || !next->w || !next_bias || !next_out) return hipErrorInvalidValue;
        a.Wnx = (const f32x4*)next->w; a.nx_bias = next_bias; a.nx_out = next_out; a.nx_n = next->n; a.nx_n16 = next->n16; a.nx_ld = next->n;
        a.nx_relu = next_relu; a.nx_softmax = next_softmax;
        if (next_softmax && (next->n > 64 || next->n16 > 8)) return hipErrorInvalidValue;
        a.nsplit = (next->n16 + 15) / 16;                                      // at most two column tiles of the next conv per wave
    } else {
        a.nsplit = (b.pr16.n16 + 7) / 8;
    }
    int maxtiles = 0;
    for (int l = 0; l < g.n_levels; ++l) maxtiles = std::max(maxtiles, ((g.lv[l].Wo + 7) / 8) * ((g.lv[l].Ho + 1) / 2));
    for (int l = 0; l < g.n_levels; ++l)
        if (((g.lv[l].Wo + 7) / 8) * ((g.lv[l].Ho + 1) / 2) != maxtiles) return hipErrorInvalidValue;   // one map size per launch (level 0)
    const size_t lds = dwproject_lds_bytes(b, next != nullptr);
    static std::once_flag attr_once;                                    // > 64 KB of dynamic LDS has to be requested once
    std::call_once(attr_once, []() {
        (void)hipFuncSetAttribute((const void*)k_dwproject<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_dwproject<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_dwproject<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_dwproject<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid((unsigned)(maxtiles * a.nsplit), (unsigned)(g.n_levels * g.batch));
    if (b.stride == 1 && next) hipLaunchKernelGGL((k_dwproject<1, true>), grid, dim3(512), lds, s, a, g);
    else if (b.stride == 1) hipLaunchKernelGGL((k_dwproject<1, false>), grid, dim3(512), lds, s, a, g);
    else if (next) hipLaunchKernelGGL((k_dwproject<2, true>), grid, dim3(512), lds, s, a, g);
    else hipLaunchKernelGGL((k_dwproject<2, false>), grid, dim3(512), lds, s, a, g);
    return hipGetLastError();
}

}  // namespace hfnet
