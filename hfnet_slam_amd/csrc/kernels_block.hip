// kernels_block.hip -- MobileNetV2 inverted-residual blocks (conv_blocks.py:163-312) as single launches for gfx950:
// the expanded tensor never leaves the CU.  Also layer_2 (no expansion conv) and stem + layer_2.
//
// Numerics: as kernels_conv.hip -- every accumulation is the oracle's fma chain (oracle/hfnet_oracle.h): BatchNorm is
// folded into the weights, accumulators start at the folded bias, ReLU6 is one v_med3_f32.
#include "kernels.hpp"

#include <algorithm>
#include <type_traits>

namespace hfnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#include "device_util.hpp"      // sgpr_base / fresh / gvec4_t: scalar-base global loads

__device__ __forceinline__ float relu6f(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }

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
    const f32x4* Wex; const float* ex_bias; int ex_nt_total;     // BatchNorm folded (weights.cpp): accumulators start at the bias
    const float* Wdw; const float* dw_bias;
    const f32x4* Wpr; const float* pr_bias; int pr_nt_total;
    const f32x4* Wex16; const f32x4* Wpr16; int ex_n16, pr_n16;  // ConvPack16 forms (k_block_fused6)
    const void* Wex_bfb;                                           // ... the expansion's with the bias in the spare k slot (odd cin / 8), or null
    const void* Wex_bf; const void* Wpr_bf;                      // split-bf16 forms (launch_repack_bf16x3; k_block_fused8<..., BF = true>), or null
    float* out;
    int cin, cexp, cout, residual, has_expand;
    int level_wgs[HFNET_MAX_LEVELS];   // k_block_fused4: workgroups launched per image of each level (a multiple of 8; exact 1-D grid)
};


// exact 1-D grids over the ragged pyramid batch, order [level][frame][tile]: workgroup b -> (level, frame, tile).  (A 2-D
// grid sized for the largest level launches up to half of its workgroups only to exit; dispatching them costs real time.)
template <int TH, int TW>
__device__ __forceinline__ void decode_tile_grid(const Geom& g, int b, int& level, int& frame, int& tile, int& tiles_x) {
    level = 0;
    int tiles = 0;
    for (;; ++level) {
        tiles_x = (g.lv[level].Wo + TW - 1) / TW;
        tiles = tiles_x * ((g.lv[level].Ho + TH - 1) / TH);
        if (level == g.n_levels - 1 || b < g.batch * tiles) break;
        b -= g.batch * tiles;
    }
    frame = b / tiles;
    tile = b - frame * tiles;
}
template <int TH, int TW>
static long long tile_grid_size(const Geom& g) {
    long long total = 0;
    for (int l = 0; l < g.n_levels; ++l) total += (long long)g.batch * ((g.lv[l].Wo + TW - 1) / TW) * ((g.lv[l].Ho + TH - 1) / TH);
    return total;
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
    int level, frame, tile_id, tiles_x;
    decode_tile_grid<TH, TW>(g, blockIdx.x, level, frame, tile_id, tiles_x);
    const LevelGeom lv = g.lv[level];
    const int tyi = tile_id / tiles_x, txi = tile_id - tyi * tiles_x;
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
    f32x16 pacc[NTO];
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt) {
        const float pb = a.pr_bias[nt * 32 + r];                   // (zero padded to pr_nt_total * 32)
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[nt][i] = pb;
    }
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
    float shpre = 0.f;
    auto fetch_b = [&](int chunk) {
        const unsigned l16 = fresh((unsigned)lane * 16u), r4 = fresh((unsigned)r * 4u);
#pragma unroll
        for (int kq = 0; kq < KQA; ++kq) bpre[kq] = *(gvec4_t)(sgpr_base(a.Wex, (unsigned)(kq * a.ex_nt_total + chunk) * 1024u) + l16);
        shpre = *(gf32_t)(sgpr_base(a.ex_bias, (unsigned)chunk * 128u) + r4);
    };
    auto stage1 = [&](int chunk) {
        if (HAS_EXPAND) {
            if (!PREFETCH_B) fetch_b(chunk);
            f32x4 bfrag[KQA];
#pragma unroll
            for (int kq = 0; kq < KQA; ++kq) bfrag[kq] = bpre[kq];
            f32x16 bias16;
#pragma unroll
            for (int i = 0; i < 16; ++i) bias16[i] = shpre;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                if (m < mt_count) {
                    const int mt = mt_first + m;
                    f32x4 af[KQA];
#pragma unroll
                    for (int kq = 0; kq < KQA; ++kq) {
                        // (out-of-image positions read a valid pixel and are zeroed by zero_border(); padding positions are never read)
                        if (DIET) af[kq] = *(const f32x4*)(aptr[m] + kq * 8);
                        else af[kq] = afrag[m][kq];
                    }
                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][0], bfrag[0][0], bias16, 0, 0, 0);
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
                        for (int i = 0; i < 4; ++i) v[i] = relu6f(acc[4 * q + i]);
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
    stage1(0);
    if (HAS_EXPAND && !interior) zero_border();
    __syncthreads();
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int ch0 = chunk * 32;
        if (PREFETCH_B && chunk + 1 < n_chunks) fetch_b(chunk + 1);
        // depthwise taps / BN of this thread's channel and the projection weights of this chunk
        const int dch = ch0 + dc;
        const bool dact = dch < a.cexp;
        float dwt[9], dsh = 0.f;
        if (dact) {
            const unsigned dc4 = fresh((unsigned)dc * 4u);
#pragma unroll
            for (int t = 0; t < 9; ++t) dwt[t] = *(gf32_t)(sgpr_base(a.Wdw, (unsigned)(t * a.cexp + ch0) * 4u) + dc4);
            dsh = *(gf32_t)(sgpr_base(a.dw_bias, (unsigned)ch0 * 4u) + dc4);
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) dwt[t] = 0.f;
        }
        const int kqc = min(4, (a.cexp - ch0) >> 3);
        f32x4 pfrag[4][NTO];
        auto fetch_p = [&]() {
            const unsigned l16p = fresh((unsigned)lane * 16u);
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
                    pfrag[kq][nt] = kq < kqc ? *(gvec4_t)(sgpr_base(a.Wpr, (unsigned)((chunk * 4 + kq) * a.pr_nt_total + nt) * 1024u) + l16p) : zero4;
        };
        if (out_live && !DIET) fetch_p();
        // ---- stage 2: thread = (channel dc, output row doy), ET -> D
        if (doy < rows_valid) {
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
                float acc = dsh;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc = fmaf(row[ky][ox * STRIDE + kx], dwt[ky * 3 + kx], acc);
                D[(doy * TW + ox) * CEP + dc] = relu6f(acc);    // inactive channel: taps and bias are all 0 -> 0
            }
        }
        __syncthreads();          // D complete, ET free
        // ---- stage 3 of this chunk (reads D) and stage 1 of the next one (writes ET) share this phase
        if (out_live) {
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
        if (chunk + 1 < n_chunks) {
            stage1(chunk + 1);
            if (HAS_EXPAND && !interior) zero_border();
        }
        __syncthreads();          // ET complete, D free
    }
    if (out_live) {
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
                    const unsigned lane_off = (unsigned)(oyb * lv.Wo + oxl) * cout4 + (unsigned)col * 4u;
                    float rv[16];
                    if (a.residual) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int c = (reg & 3) + 8 * (reg >> 2), ry = c / TW, rx = c % TW;
                            const bool ok = full || (oyb + ry < lv.Ho && oxl + rx < lv.Wo);
                            rv[reg] = ok ? *(const float*)((const char*)rbase + lane_off + (unsigned)(ry * lv.Wo + rx) * cout4) : 0.0f;
                        }
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int c = (reg & 3) + 8 * (reg >> 2), ry = c / TW, rx = c % TW;
                        if (full || (oyb + ry < lv.Ho && oxl + rx < lv.Wo)) {
                            float v = pacc[nt][reg];
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
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int op = opl + (reg & 3) + 8 * (reg >> 2);
                        const int oy = oy0 + op / TW, ox = ox0 + op % TW;
                        if (full || (oy < lv.Ho && ox < lv.Wo)) {
                            const int off = (oy * lv.Wo + ox) * a.cout + col;
                            float v = pacc[nt][reg];
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
    const long long total = tile_grid_size<TH, TW>(g);
    if (total <= 0 || total > 0x7fffffffll) return hipErrorInvalidValue;
    dim3 grid((unsigned)total);
    hipLaunchKernelGGL((k_block_fused2<STRIDE, NTO, KQT, HAS_EXPAND, TW>), grid, dim3(256), 0, s, a, g);
    return hipGetLastError();
}

// ---- v4 of the fused block: wave-autonomous tiles, no workgroup barriers.
// Every wave owns a 4 x 8 output tile of one image and runs all three stages on it by itself, per 32-channel chunk of
// the expansion:
//   expand   (MFMA)  the (3 s + 3) x (7 s + 3) halo positions of the tile as M-tiles of 32 positions (s1: 60 -> 2 tiles,
//                    s2: 153 -> 5 tiles), chained from the folded bias, ReLU6, ds_write_b128 into the wave's own LDS slice
//                    (channel-major ET[channel][position])
//   depthwise (VALU) lane = (channel, half of the output rows): the input rows are read once as 16-byte LDS loads, 16
//                    outputs per lane, written back over the same slice as D[pixel][channel] (LDS operations of one wave
//                    execute in order: all reads are issued before the first write)
//   project  (MFMA)  A fragments from D, accumulate into the tile's 32 x (NTO*32) accumulators (k order = expansion channel
//                    order: the oracle's chain)
// f32 MFMA and VALU instructions share one issue port per SIMD (NOTEBOOK.md 4.1), so what counts is that the port never
// idles: v2 synchronises its four waves twice per chunk and its phases are too short to hide their start-up latencies
// (measured: stage times add up exactly, expansion at 42 % of its MFMA rate).  Here nothing couples the 2-4 waves of a
// SIMD, so one wave's LDS round trips, weight fetches and VALU stretches are covered by the others' MFMA chains.  Price:
// the halo of a 32-pixel tile is relatively larger (stride 1: 2.0 instead of 1.5 expansion rows per output pixel).
// A workgroup IS one wave: a multi-wave workgroup gives its wave slots back only when its last wave is done, and with
// four autonomous waves per workgroup the CUs ran at 6.5 of 8 (13 of 16) resident waves on average (SQ_WAVE_CYCLES /
// SQ_BUSY_CU_CYCLES); one-wave workgroups refill every slot the moment it frees (7.2-7.5 of 8): 5-11 % per launch.
// The tile's input fragments stay in registers for all chunks; weights come from L1 / L2 one phase ahead.
template <int S> struct F4Geo {
    static constexpr int TH = 4, TW = 8, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, NPOS = IH * IW, MT_IN = (NPOS + 31) / 32;
    // channel stride of ET in floats: the halo positions rounded up to an odd number of 16-byte pieces (conflict-free
    // 16-byte accesses); the pieces of the last M tile that lie past it are not written.  Stride 2: 156 floats, 19.5 KB per
    // wave, so that eight one-wave workgroups (two per SIMD) fit the CU's LDS -- with 164 it was seven
    static constexpr int EP = ((NPOS + 3) / 4) % 2 == 1 ? (NPOS + 3) / 4 * 4 : (NPOS + 3) / 4 * 4 + 4;
    static constexpr int CEP = 36;                    // pixel stride of D
    static_assert((EP / 4) % 2 == 1, "ET channel stride");
};

// KQO ("k outer"): the expansion walks the input channels in the outer loop with one accumulator per halo M-tile (MT_IN
// independent MFMA chains in flight; the ReLU6 / LDS epilogue of one tile overlaps the last MFMAs of the others) and its
// weights stream through a ring of PF 16-byte pieces requested PF steps ahead instead of being held for the whole chunk
// (KQT * 4 registers, twice with the look-ahead copy).  It is what lets a 96-channel stride-2 block (layer 8: 240
// registers of input fragments) run as a wave-autonomous tile at one wave per SIMD: its input crosses L2 once instead of
// once per expansion chunk.
template <int STRIDE, int NTO, int KQT, bool RES, int OCC, bool KQO = false>
__global__ __launch_bounds__(64, OCC) void k_block_fused4(FusedArgs a, Geom g) {
    using G = F4Geo<STRIDE>;
    constexpr int TH = G::TH, TW = G::TW, IH = G::IH, IW = G::IW, NPOS = G::NPOS, MT_IN = G::MT_IN, EP = G::EP, CEP = G::CEP;
    __shared__ __attribute__((aligned(16))) float ET[32 * EP];
    const int lane = threadIdx.x, half = lane >> 5, r = lane & 31;
    // exact 1-D grid, order [level][frame][workgroup]: an image of level l owns level_wgs[l] consecutive workgroups (its
    // own count rounded up to 8).  (A 2-D grid sized for the largest level launched half of the stride-2 kernels'
    // workgroups only to exit -- dispatching them costs real time.)
    int level = 0, bx = blockIdx.x;
    for (; level < g.n_levels - 1; ++level) {
        const int per = g.batch * a.level_wgs[level];
        if (bx < per) break;
        bx -= per;
    }
    const int frame = bx / a.level_wgs[level];
    bx -= frame * a.level_wgs[level];
    const int image = level * g.batch + frame;
    const LevelGeom lv = g.lv[level];
    const int tiles_x = (lv.Wo + TW - 1) / TW, ntiles = tiles_x * ((lv.Ho + TH - 1) / TH);
    // workgroup b runs on XCD b % 8 (observed; speed only; an image's run starts at a multiple of 8): every XCD gets one
    // contiguous run of THIS image's tiles, so that the halo rows shared by vertical neighbours meet in one L2 and every
    // XCD carries the same load
    const int q = ntiles >> 3, rem = ntiles & 7;
    const int xr = (bx + image) & 7, slot = bx >> 3;           // (the XCDs that take the remainder rotate with the image)
    if (slot >= q + (xr < rem ? 1 : 0)) return;
    const int tile = xr * q + min(xr, rem) + slot;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * STRIDE - lv.pt, ix0 = ox0 * STRIDE - lv.pl;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= lv.H && ix0 + IW <= lv.W;
    const long long in_base = lv.in_off + (long long)frame * lv.H * lv.W;
    const long long out_base = lv.out_off + (long long)frame * lv.Ho * lv.Wo;
    const char* __restrict__ xb = (const char*)(a.X + in_base * a.cin);      // uniform; lane offsets are 32-bit (one image < 4 GB)

    // ---- block input: A fragments of the halo M-tiles (out-of-image / padding positions read a clamped pixel: a row of A
    //      only feeds the same row of the expansion, and those rows are zeroed below or never read)
    f32x4 afrag[MT_IN][KQT];
#pragma unroll
    for (int m = 0; m < MT_IN; ++m) {
        const int pp = m * 32 + r;
        const int hy = pp / IW, hx = pp - hy * IW;
        const int iy = min(max(iy0 + hy, 0), lv.H - 1), ix = min(max(ix0 + hx, 0), lv.W - 1);
        const unsigned off = ((unsigned)(iy * lv.W + ix) * (unsigned)a.cin + (unsigned)(half * 4)) * 4u;
#pragma unroll
        for (int kq = 0; kq < KQT; ++kq) afrag[m][kq] = *(const f32x4*)(xb + off + kq * 32);
    }
    // out-of-image halo positions (the depthwise conv's 'SAME' zero padding): one bit per position, border tiles only
    // (built row by row from the column pattern -- position by position it was 1200 scalar instructions per border tile)
    unsigned long long outside[(NPOS + 63) / 64] = {};
    if (!interior) {
        const unsigned long long full = (1ull << IW) - 1;
        const int clo = min(max(-ix0, 0), IW), chi = min(max(lv.W - ix0, 0), IW);          // columns [clo, chi) are inside
        const unsigned long long cols = (((1ull << clo) - 1) | ~((1ull << chi) - 1)) & full;
#pragma unroll
        for (int hy = 0; hy < IH; ++hy) {
            const unsigned long long bits = (iy0 + hy < 0 || iy0 + hy >= lv.H) ? full : cols;
            const int pos = hy * IW, wd = pos >> 6, sh = pos & 63;
            outside[wd] |= bits << sh;
            if (sh + IW > 64) outside[wd + 1] |= bits >> (64 - sh);
        }
    }
    f32x16 pacc[NTO];
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt) {
        const float pb = a.pr_bias[nt * 32 + r];                   // (zero padded to pr_nt_total * 32)
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[nt][i] = pb;
    }
    const int n_chunks = min(a.ex_nt_total, (a.cexp + 31) >> 5);  // skip all-padding column tiles
    f32x4 bfrag[KQT];
    float ebias;
    // weight loads: uniform (scalar) base + a 32-bit lane offset, so that no vector instruction goes into their addresses
    // (vector ALU instructions and f32 MFMAs share the issue port)
    const unsigned lane16_ = (unsigned)lane * 16u, r4_ = (unsigned)r * 4u;
    auto load_b = [&](int chunk) {
        const unsigned lane16 = fresh(lane16_), r4 = fresh(r4_);
#pragma unroll
        for (int kq = 0; kq < KQT; ++kq)
            bfrag[kq] = *(gvec4_t)(sgpr_base(a.Wex, (unsigned)(kq * a.ex_nt_total + chunk) * 1024u) + lane16);
        ebias = *(gf32_t)(sgpr_base(a.ex_bias, (unsigned)chunk * 128u) + r4);
    };
    constexpr int PF = KQT < 3 ? KQT : 3;                          // KQO: pieces of expansion weights in flight
    f32x4 bq[PF];
    if constexpr (KQO) {
        const unsigned lane16 = fresh(lane16_), r4 = fresh(r4_);
#pragma unroll
        for (int j = 0; j < PF; ++j) bq[j] = *(gvec4_t)(sgpr_base(a.Wex, (unsigned)(j * a.ex_nt_total) * 1024u) + lane16);
        ebias = *(gf32_t)(sgpr_base(a.ex_bias, 0u) + r4);
    } else {
        load_b(0);
    }
    f32x4 bnext[KQT];
    float enext = 0.f;
    const int rh = half;                                           // depthwise role: channel r, output rows 2 rh, 2 rh + 1
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int ch0 = chunk * 32;
        const int kqc = min(4, (a.cexp - ch0) >> 3);               // (channels past cexp are never consumed: clamp, do not zero)
        const unsigned lane16 = fresh(lane16_);
        if constexpr (KQO) {
            const int cn = min(chunk + 1, n_chunks - 1);
            enext = *(gf32_t)(sgpr_base(a.ex_bias, (unsigned)cn * 128u) + fresh(r4_));
        } else {                                                   // next chunk's expansion weights, requested two phases ahead
            const int cn = min(chunk + 1, n_chunks - 1);
            const unsigned r4 = fresh(r4_);
#pragma unroll
            for (int kq = 0; kq < KQT; ++kq)
                bnext[kq] = *(gvec4_t)(sgpr_base(a.Wex, (unsigned)(kq * a.ex_nt_total + cn) * 1024u) + lane16);
            enext = *(gf32_t)(sgpr_base(a.ex_bias, (unsigned)cn * 128u) + r4);
            __builtin_amdgcn_sched_barrier(0);
        }
        // this chunk's projection weights and depthwise taps: requested now, used after the expansion
        // (three waves per SIMD: the projection weights are requested after the depthwise arithmetic, into registers it frees)
        constexpr bool P_LATE = KQO && OCC >= 3;
        f32x4 pfrag[4][NTO];
        auto load_p = [&]() {
            const unsigned l16 = fresh(lane16_);
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
                    pfrag[kq][nt] = *(gvec4_t)(sgpr_base(a.Wpr, (unsigned)((chunk * 4 + min(kq, kqc - 1)) * a.pr_nt_total + nt) * 1024u) + l16);
        };
        if constexpr (!P_LATE) load_p();
        const unsigned dch4 = (unsigned)min(r, a.cexp - 1 - ch0) * 4u;      // (channels past cexp: clamped, never consumed)
        float dwt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) dwt[t] = *(gf32_t)(sgpr_base(a.Wdw, (unsigned)(t * a.cexp + ch0) * 4u) + dch4);
        const float dwb = *(gf32_t)(sgpr_base(a.dw_bias, (unsigned)ch0 * 4u) + dch4);
        // ---- expansion
        if constexpr (KQO) {
            f32x16 acc[MT_IN];
#pragma unroll
            for (int m = 0; m < MT_IN; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][i] = ebias;
            const int cn = min(chunk + 1, n_chunks - 1);
#pragma unroll
            for (int kq = 0; kq < KQT; ++kq) {
                const f32x4 b = bq[kq % PF];
                // the piece PF steps ahead (the next chunk's first pieces at the end: they arrive during the depthwise phase)
                const int kn = kq + PF < KQT ? kq + PF : kq + PF - KQT;
                const int cc = kq + PF < KQT ? chunk : cn;
                const f32x4 bn = *(gvec4_t)(sgpr_base(a.Wex, (unsigned)(kn * a.ex_nt_total + cc) * 1024u) + fresh(lane16_));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < MT_IN; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][kq][t], b[t], acc[m], 0, 0, 0);
                bq[kq % PF] = bn;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < MT_IN; ++m) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu6f(acc[m][4 * q + i]);
                    if (m * 32 + 8 * q + 8 <= EP) *(f32x4*)(ET + r * EP + m * 32 + 8 * q + 4 * half) = v;
                    else if (m * 32 + 8 * q + 4 <= EP) { if (half == 0) *(f32x4*)(ET + r * EP + m * 32 + 8 * q) = v; }
                }
            }
        } else {
            f32x16 bias16;
#pragma unroll
            for (int i = 0; i < 16; ++i) bias16[i] = ebias;
#pragma unroll
            for (int m = 0; m < MT_IN; ++m) {
                f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][0][0], bfrag[0][0], bias16, 0, 0, 0);
#pragma unroll
                for (int kq = 0; kq < KQT; ++kq)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (kq + t > 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][kq][t], bfrag[kq][t], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu6f(acc[4 * q + i]);
                    if (m * 32 + 8 * q + 8 <= EP) *(f32x4*)(ET + r * EP + m * 32 + 8 * q + 4 * half) = v;
                    else if (m * 32 + 8 * q + 4 <= EP) { if (half == 0) *(f32x4*)(ET + r * EP + m * 32 + 8 * q) = v; }
                }
            }
        }
        if (!interior) {
#pragma unroll
            for (int wd = 0; wd < (NPOS + 63) / 64; ++wd) {
                unsigned long long msk = outside[wd];                                  // uniform
                while (msk) {
                    const int pp = wd * 64 + (int)__builtin_ctzll(msk);
                    msk &= msk - 1;
                    if (half == 0) ET[r * EP + pp] = 0.0f;
                }
            }
        }
        asm volatile("" ::: "memory");
        // ---- depthwise: channel r, output rows 2 rh and 2 rh + 1 (input rows 2 rh s .. 2 rh s + s + 2)
        {
            constexpr int NR = STRIDE + 3;                         // input rows for two output rows
            // input rows one at a time: row i feeds output row ro with ky = i - ro * STRIDE, so every output still adds its
            // taps in (ky, kx) order while only ONE row of the channel is held in registers besides the 2 x TW accumulators
            // (all NR rows at once cost 40 / 85 VGPRs more and one occupancy step on the narrow kernels)
            const float* rp = ET + r * EP + rh * 2 * STRIDE * IW;
            float o[2][TW];
#pragma unroll
            for (int ro = 0; ro < 2; ++ro)
#pragma unroll
                for (int ox = 0; ox < TW; ++ox) o[ro][ox] = dwb;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                float row[IW];
                if constexpr (IW % 2 == 0 && (2 * STRIDE * IW) % 2 == 0) {
#pragma unroll
                    for (int x2 = 0; x2 < IW / 2; ++x2) {
                        const float2 v = *(const float2*)(rp + i * IW + x2 * 2);
                        row[2 * x2] = v.x; row[2 * x2 + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int x = 0; x < IW; ++x) row[x] = rp[i * IW + x];
                }
#pragma unroll
                for (int ro = 0; ro < 2; ++ro) {
                    const int ky = i - ro * STRIDE;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int ox = 0; ox < TW; ++ox)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) o[ro][ox] = fmaf(row[ox * STRIDE + kx], dwt[ky * 3 + kx], o[ro][ox]);
                }
            }
#pragma unroll
            for (int ro = 0; ro < 2; ++ro)
#pragma unroll
                for (int ox = 0; ox < TW; ++ox) o[ro][ox] = relu6f(o[ro][ox]);
            asm volatile("" ::: "memory");                         // every ET read is issued before D overwrites the slice
            if constexpr (P_LATE) load_p();
#pragma unroll
            for (int ro = 0; ro < 2; ++ro)
#pragma unroll
                for (int ox = 0; ox < TW; ++ox) ET[((rh * 2 + ro) * TW + ox) * CEP + r] = o[ro][ox];
        }
        asm volatile("" ::: "memory");
        // ---- projection of this chunk's channels
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            if (kq < kqc) {
                const f32x4 av = *(const f32x4*)(ET + r * CEP + kq * 8 + half * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], pfrag[kq][nt][t], pacc[nt], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
        if constexpr (!KQO) {
#pragma unroll
            for (int kq = 0; kq < KQT; ++kq) bfrag[kq] = bnext[kq];
        }
        ebias = enext;
    }
    // ---- output (+ residual): every 32-column tile goes through the LDS slice so that a lane moves 16 consecutive bytes
    float* __restrict__ ob = a.out + out_base * a.cout;                                // uniform
    const float* __restrict__ rb = a.X + in_base * a.cin;                              // residual: same size, cin == cout
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt) {
        if (nt * 32 < a.cout) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) ET[((reg & 3) + 8 * (reg >> 2) + 4 * half) * CEP + r] = pacc[nt][reg];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int piece = lane + 64 * k, px = piece >> 3, c4 = piece & 7;
                const int col = nt * 32 + c4 * 4;
                const int oy = oy0 + px / TW, ox = ox0 + px % TW;
                f32x4 v = *(const f32x4*)(ET + px * CEP + c4 * 4);
                if (col < a.cout && oy < lv.Ho && ox < lv.Wo) {
                    const unsigned off = (unsigned)(oy * lv.Wo + ox) * (unsigned)a.cout + (unsigned)col;
                    if (RES) {
                        const f32x4 rv = *(const f32x4*)(rb + off);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] + rv[j];
                    }
                    *(f32x4*)(ob + off) = v;
                }
            }
            asm volatile("" ::: "memory");
        }
    }
}

template <int STRIDE, int NTO, int KQT, int OCC, bool KQO = false>
static hipError_t launch_block_fused4_t(const FusedArgs& a, const Geom& g, hipStream_t s) {
    using G = F4Geo<STRIDE>;
    if (a.residual && (STRIDE != 1 || a.cin != a.cout)) return hipErrorInvalidValue;
    FusedArgs b = a;
    long long total = 0;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) {
        b.level_wgs[l] = 8;
        if (l >= g.n_levels) continue;
        const int tiles = ((g.lv[l].Wo + G::TW - 1) / G::TW) * ((g.lv[l].Ho + G::TH - 1) / G::TH);
        b.level_wgs[l] = max(8, ((tiles + 7) / 8) * 8);           // multiple of 8: see the XCD mapping in the kernel
        total += (long long)b.level_wgs[l] * g.batch;
    }
    if (total <= 0 || total > 0x7fffffffll) return hipErrorInvalidValue;
    dim3 grid((unsigned)total);
    if (a.residual) hipLaunchKernelGGL((k_block_fused4<STRIDE, NTO, KQT, true, OCC, KQO>), grid, dim3(64), 0, s, b, g);
    else hipLaunchKernelGGL((k_block_fused4<STRIDE, NTO, KQT, false, OCC, KQO>), grid, dim3(64), 0, s, b, g);
    return hipGetLastError();
}

// ---- v8 of the fused block (stride 2): the wave-autonomous tile of v4 with the expanded tensor kept in REGISTERS.
// v4 stages the ReLU6'd expansion through the wave's LDS slice (channel-major ET, 19.5 KB per wave for the 9 x 17 halo of a
// stride-2 tile: LDS, not registers, holds these kernels at two waves per SIMD) only to hand each lane the rows of its channel:
// 20 ds_write_b128 + 85 ds_read_b32 per chunk and one of the two dependent LDS round trips of a chunk.  But the MFMA D fragment
// already has lane = channel; only WHICH halo positions a lane half holds is wrong -- and the A-row -> position assignment is
// free.  Here the depthwise stage is split by output COLUMNS (lane half h computes columns 4h .. 4h+3 of all four rows, which
// read halo columns 8h .. 8h+8), and A row rho = 8 (i >> 2) + 4 h + (i & 3) of M tile m loads the position that half h wants in
// accumulator slot s = 16 m + i:
//     s < 72: halo (s >> 3, 8 h + (s & 7))           -- the half's own 9 x 8 block
//     s >= 72: halo (h ? s - 72 : 8, 16)             -- column 16: rows 0-7 sit with half 1, row 8 with half 0 (153 = 80 + 73)
// so every tap of every output is an accumulator register of the lane itself, except the ninth halo column of the half
// (column 8 for half 0 -- half 1's slots 8 hy; column 16 for half 1 -- its own slots 72 + hy, row 8 from half 0), which
// v_permlane32_swap moves: 20 vector instructions per chunk instead of 105 LDS instructions.  Out-of-image halo positions
// (the depthwise conv's zero padding) cost nothing on interior tiles and one extra MFMA per M tile on border tiles: the chain of
// such an A row starts with fma(1, -1e30, bias) instead of the bias (an in-image row's with fma(0, -1e30, bias) == bias exactly),
// stays there and ReLU6 returns 0.  LDS per wave: the 32 x 36 depthwise tile only (4.6 KB); the chains, the (ky, kx) order of the
// depthwise sums and the projection are v4's: same bits.
// Stride 1 (4 x 8 tile, 6 x 10 halo = 2 M tiles): the same scheme with five own columns per half (half 0: halo columns 0-4 in
// slots 5 hy + k; half 1: columns 5-8 in k = 1..4 and column 9 in k = 0, so that local column c = 1..4 of the half's six-column
// window 4 h .. 4 h + 5 is slot k = c in BOTH halves); local column 0 is (own column 0 | half 0's column 4) and local column 5
// is (half 1's column 5 | own column 9): two v_permlane32_swap per halo row.
template <int STRIDE> struct F8Geo {
    static constexpr int TH = 4, TW = 8, IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3;
    static constexpr int NC = STRIDE == 2 ? 8 : 5;                 // own halo columns of a lane half, per halo row
    static constexpr int MT = STRIDE == 2 ? 5 : 2;                 // halo M tiles (accumulator slots per lane half = 16 MT)
    static constexpr int CEP = 36;
};
// Layer 8 (96 input channels: 240 registers of A fragments) runs at ONE wave per SIMD (OCC = 1) with everything resident: 1064 us per
// 128 frames against the barrier kernel's 1273.  (With the A pieces re-read from L1 / L2 per chunk through a ring of three k-steps it
// fitted two waves per SIMD and took 1389 us: every wave then pulls 60 KB of A per chunk, ~37 B/clk per CU from L2.  Not kept.)
// BF (engine option global_bf16x3, layers 8-14 of calls that take the fused kernels): both 1x1 convolutions on split-bf16 operands (two
// pieces, three products, v_mfma_f32_32x32x16_bf16 -- kernels_conv.hip k_conv_bf16x3 for the scheme and its error bound): the block
// input is split once per tile, the depthwise results once per chunk on their way from LDS into the projection; the D fragment of the
// bf16 MFMA has the f32 one's layout, so the register-resident depthwise stage is unchanged (and exact for its inputs).  NOT the
// oracle's bits: within the tolerance stated in include/hfnet_hip.h.
typedef __bf16 fb16x8 __attribute__((ext_vector_type(8)));
// (pairs: one v_cvt_pk_bf16_f32 rounds two values, the f32 of the hi pieces is a shift / a mask of the packed word: 3 vector instructions per value
//  where the value-by-value form compiles to 4 -- these kernels run with the vector ALU saturated)
typedef __bf16 fb16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fsplit8(const f32x4& v0, const f32x4& v1, fb16x8& hi, fb16x8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = j < 2 ? v0[2 * j] : v1[2 * j - 4], y = j < 2 ? v0[2 * j + 1] : v1[2 * j - 3];
        fb16x2 p;
        p[0] = (__bf16)x; p[1] = (__bf16)y;
        const unsigned pu = __builtin_bit_cast(unsigned, p);
        const float hx = __builtin_bit_cast(float, pu << 16), hy = __builtin_bit_cast(float, pu & 0xffff0000u);
        fb16x2 q;
        q[0] = (__bf16)(x - hx); q[1] = (__bf16)(y - hy);
        hi[2 * j] = p[0]; hi[2 * j + 1] = p[1];
        lo[2 * j] = q[0]; lo[2 * j + 1] = q[1];
    }
}
template <int STRIDE, int NTO, int KQT, bool RES, int OCC, bool BF = false>
__global__ __launch_bounds__(64, OCC) void k_block_fused8(FusedArgs a, Geom g) {
    using G = F8Geo<STRIDE>;
    constexpr int TH = G::TH, TW = G::TW, IH = G::IH, IW = G::IW, NC = G::NC, MT = G::MT, CEP = G::CEP;
    __shared__ __attribute__((aligned(16))) float ET[32 * CEP];
    const int lane = threadIdx.x, half = lane >> 5, r = lane & 31;
    int level = 0, bx = blockIdx.x;
    for (; level < g.n_levels - 1; ++level) {
        const int per = g.batch * a.level_wgs[level];
        if (bx < per) break;
        bx -= per;
    }
    const int frame = bx / a.level_wgs[level];
    bx -= frame * a.level_wgs[level];
    const int image = level * g.batch + frame;
    const LevelGeom lv = g.lv[level];
    const int tiles_x = (lv.Wo + TW - 1) / TW, ntiles = tiles_x * ((lv.Ho + TH - 1) / TH);
    const int q = ntiles >> 3, rem = ntiles & 7;                  // (XCD mapping: as k_block_fused4)
    const int xr = (bx + image) & 7, slot = bx >> 3;
    if (slot >= q + (xr < rem ? 1 : 0)) return;
    const int tile = xr * q + min(xr, rem) + slot;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * STRIDE - lv.pt, ix0 = ox0 * STRIDE - lv.pl;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= lv.H && ix0 + IW <= lv.W;
    const long long in_base = lv.in_off + (long long)frame * lv.H * lv.W;
    const long long out_base = lv.out_off + (long long)frame * lv.Ho * lv.Wo;
    const char* __restrict__ xb = (const char*)(a.X + in_base * a.cin);      // uniform; lane offsets are 32-bit

    // ---- block input: A row r of M tile m is accumulator slot 16 m + ir of lane half hr
    constexpr int KS = (KQT + 1) / 2;                             // BF: 16-k steps of the expansion (the last one may hold 8 channels)
    // BIASK (an odd number of input channel groups: 24 channels): the last step's spare k slot carries a constant 1 against the folded bias in the
    // weights (launch_repack_bf16x3 with_bias), the accumulators start at the inline constant 0 -- no bias tile, no moves (the vector ALU is what
    // bounds these kernels).  The launcher passes the pack with the bias row as Wex_bf.
    constexpr bool BIASK = BF && (KQT % 2 == 1);
    f32x4 afrag[BF ? 1 : MT][BF ? 1 : KQT];
    // 96 input channels (KS = 6): five M tiles of hi / lo fragments are 240 registers, and with the projection's accumulators next to them the
    // accumulator half of the register file overflows (28 fragment registers lived in scratch memory and came back once per chunk).  The LAST
    // M tile's fragments live in the wave's LDS instead (12 KB; a lone wave per SIMD leaves LDS mostly idle) and are read a step ahead.
    constexpr bool ALDS = BF && KS >= 5 && OCC == 1;
    constexpr int MR = ALDS ? MT - 1 : MT;                        // M tiles whose fragments stay in registers
    __shared__ __attribute__((aligned(16))) fb16x8 AL[ALDS ? KS : 1][2][64];
    fb16x8 ah[BF ? MT : 1][BF ? (ALDS ? KS : KS) : 1], al[BF ? MT : 1][BF ? KS : 1];
    float aflag[MT];                                              // 1 where this A row is an out-of-image position (k half 0 only)
    {
        const int hr = (r >> 2) & 1, ir = ((r >> 3) << 2) | (r & 3);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int s = 16 * m + ir;
            int hy, hx;
            if constexpr (STRIDE == 2) {
                hy = s >> 3; hx = 8 * hr + (s & 7);
                if (m == MT - 1 && ir >= 8) { hx = 16; hy = hr ? s - 72 : 8; }
            } else {
                const int sc = min(s, IH * NC - 1);               // (slots 30, 31: padding, never read)
                hy = sc / NC;
                const int k = sc - hy * NC;
                hx = hr ? (k == 0 ? 9 : 4 + k) : k;
            }
            const int iyr = iy0 + hy, ixr = ix0 + hx;
            const bool outside = iyr < 0 || iyr >= lv.H || ixr < 0 || ixr >= lv.W;
            aflag[m] = (outside && half == 0) ? 1.0f : 0.0f;
            const int iy = min(max(iyr, 0), lv.H - 1), ix = min(max(ixr, 0), lv.W - 1);
            if constexpr (BF) {
                // lane (row, half) of a 16-k step holds k = 8 half .. 8 half + 7: the physical slots of channel group 2 ks + half
                const unsigned offp = (unsigned)(iy * lv.W + ix) * (unsigned)a.cin * 4u;
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bool kok = ks * 16 + half * 8 < a.cin;       // (the upper half of a last step of 8 channels: zeros -- BIASK: a one in slot 0)
                    const char* pa = xb + offp + (kok ? (ks * 16 + half * 8) * 4 : 0);
                    const f32x4 v0 = *(const f32x4*)pa, v1 = *(const f32x4*)(pa + 16);
                    const f32x4 one4 = {1.f, 0.f, 0.f, 0.f};
                    fsplit8(kok ? v0 : (BIASK ? one4 : z4), kok ? v1 : z4, ah[m][ks], al[m][ks]);
                    if constexpr (ALDS) { if (m == MT - 1) { AL[ks][0][lane] = ah[m][ks]; AL[ks][1][lane] = al[m][ks]; } }
                }
                if constexpr (KS >= 5) __builtin_amdgcn_sched_barrier(0);   // (one M tile's raw pieces at a time: all of them in flight next to the fragments is 2 x 8 KS MT registers)
            } else {
                const unsigned off = ((unsigned)(iy * lv.W + ix) * (unsigned)a.cin + (unsigned)(half * 4)) * 4u;
#pragma unroll
                for (int kq = 0; kq < KQT; ++kq) afrag[m][kq] = *(const f32x4*)(xb + off + kq * 32);
            }
        }
    }
    const float bneg = half == 0 ? -1.0e30f : 0.0f;
    f32x16 pacc[NTO];
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt) {
        const float pb = a.pr_bias[nt * 32 + r];
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[nt][i] = pb;
    }
    const int n_chunks = min(a.ex_nt_total, (a.cexp + 31) >> 5);
    const unsigned lane16_ = (unsigned)lane * 16u, r4_ = (unsigned)r * 4u;
    constexpr int PF = KQT < 3 ? KQT : 3;                          // pieces of expansion weights in flight
    f32x4 bq[BF ? 1 : PF];
    // BF: the chunk's expansion weights [step][hi | lo], requested for the NEXT chunk right behind this chunk's expansion; from five steps on
    // (72 / 96 input channels) a ring of three steps instead, requested two steps ahead: 24 registers instead of 8 KS
    constexpr bool BRING = BF && KS >= 5;
    fb16x8 bfr[BF ? (BRING ? 3 : KS) : 1][2];
    const fb16x8* __restrict__ wexb = (const fb16x8*)a.Wex_bf;
    const fb16x8* __restrict__ wprb = (const fb16x8*)a.Wpr_bf;
    auto load_bfr = [&](int chunk) {
#pragma unroll
        for (int ks = 0; ks < (BF ? (BRING ? 2 : KS) : 1); ++ks)       // (ring: the chunk's first two steps)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) bfr[ks][hl] = wexb[(((size_t)ks * a.ex_nt_total + chunk) * 2 + hl) * 64 + lane];
    };
    auto load_bfr_step = [&](int chunk, int ks) {                      // ring slot ks % 3
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) bfr[ks % 3][hl] = wexb[(((size_t)ks * a.ex_nt_total + chunk) * 2 + hl) * 64 + lane];
    };
    float ebias;
    {
        const unsigned lane16 = fresh(lane16_), r4 = fresh(r4_);
        if constexpr (BF) load_bfr(0);
        else {
#pragma unroll
            for (int j = 0; j < PF; ++j) bq[j] = *(gvec4_t)(sgpr_base(a.Wex, (unsigned)(j * a.ex_nt_total) * 1024u) + lane16);
        }
        ebias = *(gf32_t)(sgpr_base(a.ex_bias, 0u) + r4);
    }
    auto swp = [](float x, float y, int which) -> float {          // (x.lower, y.lower) [which = 0] / (x.upper, y.upper) [1] as (lower, upper) half
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        return __builtin_bit_cast(float, (unsigned)sw[which]);
    };
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int ch0 = chunk * 32;
        const int kqc = min(4, (a.cexp - ch0) >> 3);
        const int cn = min(chunk + 1, n_chunks - 1);
        const float enext = *(gf32_t)(sgpr_base(a.ex_bias, (unsigned)cn * 128u) + fresh(r4_));
        const unsigned dch4 = (unsigned)min(r, a.cexp - 1 - ch0) * 4u;      // (channels past cexp: clamped, never consumed)
        float dwt[9], dwb;
        auto load_dw = [&]() {
#pragma unroll
            for (int t = 0; t < 9; ++t) dwt[t] = *(gf32_t)(sgpr_base(a.Wdw, (unsigned)(t * a.cexp + ch0) * 4u) + fresh(dch4));
            dwb = *(gf32_t)(sgpr_base(a.dw_bias, (unsigned)ch0 * 4u) + fresh(dch4));
        };
        constexpr bool DW_LATE = OCC == 1 || (BF && KS >= 5);      // (tight on registers: requested behind the expansion, where its registers are free)
        if constexpr (!DW_LATE) load_dw();
        // ---- expansion: MT independent chains, k outer, weights through the ring
        f32x16 acc[MT];
        if constexpr (BF) {
            if constexpr (BRING) {
                // (one wave per SIMD, every register counts: the accumulators start at the bias in place -- 16 registers of a broadcast bias tile less)
                load_bfr_step(chunk, 2);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][i] = BIASK ? 0.0f : ebias;
                    if (!interior) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(aflag[m], bneg, acc[m], 0, 0, 0);
                    const fb16x8 a0 = (ALDS && m == MT - 1) ? AL[0][0][lane] : ah[m][0];
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfr[0][0], acc[m], 0, 0, 0);
                }
            } else {
                f32x16 bias16;
#pragma unroll
                for (int i = 0; i < 16; ++i) bias16[i] = BIASK ? 0.0f : ebias;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    // (border tiles: the chain of an out-of-image row starts at -1e30, as in the f32 form)
                    const f32x16 c0 = interior ? bias16 : __builtin_amdgcn_mfma_f32_32x32x2f32(aflag[m], bneg, bias16, 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m][0], bfr[0][0], c0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                constexpr int dummy = 0; (void)dummy;
                const int slot = BRING ? ks % 3 : ks;
                fb16x8 lh, ll;                                      // (ALDS: the last M tile's fragments of this step)
                if constexpr (ALDS) { lh = AL[ks][0][lane]; ll = AL[ks][1][lane]; }
                if (ks > 0) {
                    if constexpr (BRING) { if (ks + 2 < KS) load_bfr_step(chunk, ks + 2); }
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((ALDS && m == MT - 1) ? lh : ah[m][ks], bfr[slot][0], acc[m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((ALDS && m == MT - 1) ? lh : ah[m][ks], bfr[slot][1], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((ALDS && m == MT - 1) ? ll : al[m][ks], bfr[slot][0], acc[m], 0, 0, 0);
                if constexpr (BRING) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_bfr(cn);                                          // the next chunk's (first) weights arrive during the depthwise / projection
        } else {
            f32x16 bias16;
#pragma unroll
            for (int i = 0; i < 16; ++i) bias16[i] = ebias;
#pragma unroll
            for (int kq = 0; kq < KQT; ++kq) {
                const f32x4 b = bq[kq % PF];
                const int kn = kq + PF < KQT ? kq + PF : kq + PF - KQT;
                const int cc = kq + PF < KQT ? chunk : cn;
                const f32x4 bn = *(gvec4_t)(sgpr_base(a.Wex, (unsigned)(kn * a.ex_nt_total + cc) * 1024u) + fresh(lane16_));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (kq == 0 && t == 0) {
                        if (interior) {
#pragma unroll
                            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][0][0], b[0], bias16, 0, 0, 0);
                        } else {
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                const f32x16 c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aflag[m], bneg, bias16, 0, 0, 0);
                                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][0][0], b[0], c0, 0, 0, 0);
                            }
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m][kq][t], b[t], acc[m], 0, 0, 0);
                    }
                }
                bq[kq % PF] = bn;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (DW_LATE) load_dw();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = relu6f(acc[m][i]);
        // ---- the halo columns of this lane half's window that sit with the other half
        float E[IH], E0[STRIDE == 1 ? IH : 1];
        if constexpr (STRIDE == 2) {
            // E[hy] = (half 0: column 8 = half 1's slot 8 hy | half 1: column 16 = its slot 72 + hy, row 8 from half 0's slot 72)
            const float lo72 = swp(acc[4][8], acc[4][8], 0);       // half 0's halo (8, 16), in both halves
            E[8] = swp(acc[4][0], lo72, 1);
#pragma unroll
            for (int hy = 0; hy < 8; ++hy) E[hy] = swp(acc[(hy * 8) >> 4][(hy * 8) & 15], acc[4][8 + hy], 1);
        } else {
#pragma unroll
            for (int hy = 0; hy < IH; ++hy) {
                const float a0 = acc[(hy * NC) >> 4][(hy * NC) & 15], a1 = acc[(hy * NC + 1) >> 4][(hy * NC + 1) & 15], a4 = acc[(hy * NC + 4) >> 4][(hy * NC + 4) & 15];
                E[hy] = swp(a1, a0, 1);                            // local column 5: (half 1's column 5, own column 9)
                E0[hy] = swp(a0, a4, 0);                           // local column 0: (own column 0, half 0's column 4)
            }
        }
        // this chunk's projection weights: requested here, into registers the expansion has released (PJIT -- one wave per SIMD, every
        // register counts: a k-group at a time, one group ahead of its MFMAs).  BF: [16-k step][column tile][hi | lo], a step at a time.
        constexpr bool PJIT = OCC == 1;
        f32x4 pfrag[BF ? 1 : (PJIT ? 2 : 4)][BF ? 1 : NTO];
        constexpr int PBN = (KS >= 5 && (NTO >= 3 || OCC == 1)) ? 1 : 2;   // (72 -> 432 -> 72, 96 -> 576 -> 48: no registers for a second step's fragments)
        fb16x8 pbf[BF ? PBN : 1][BF ? NTO : 1][2];
        auto load_p = [&](int kq, int slot) {
            const unsigned l16 = fresh(lane16_);
#pragma unroll
            for (int nt = 0; nt < (BF ? 1 : NTO); ++nt)
                pfrag[slot][nt] = *(gvec4_t)(sgpr_base(a.Wpr, (unsigned)((chunk * 4 + min(kq, kqc - 1)) * a.pr_nt_total + nt) * 1024u) + l16);
        };
        auto load_pbf = [&](int s16, int slot) {                  // s16: 16-k step of this chunk (0, 1)
            const int step = min(chunk * 2 + s16, ((a.cexp + 15) >> 4) - 1);
#pragma unroll
            for (int nt = 0; nt < (BF ? NTO : 1); ++nt)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) pbf[slot][nt][hl] = wprb[(((size_t)step * a.pr_nt_total + nt) * 2 + hl) * 64 + lane];
        };
        if constexpr (BF) load_pbf(0, 0);
        else if constexpr (PJIT) load_p(0, 0);
        else {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) load_p(kq, kq);
        }
        // ---- depthwise: channel r, output columns 4 half .. 4 half + 3 of all four rows, taps in (ky, kx) order
#pragma unroll
        for (int oy = 0; oy < TH; ++oy)
#pragma unroll
            for (int ox = 0; ox < 4; ++ox) {
                float v = dwb;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int hy = STRIDE * oy + ky, c = STRIDE * ox + kx;
                        float x;
                        if constexpr (STRIDE == 2) x = c < NC ? acc[(hy * NC + c) >> 4][(hy * NC + c) & 15] : E[hy];
                        else x = c == 0 ? E0[hy] : c < NC ? acc[(hy * NC + c) >> 4][(hy * NC + c) & 15] : E[hy];
                        v = fmaf(x, dwt[ky * 3 + kx], v);
                    }
                ET[(oy * TW + 4 * half + ox) * CEP + r] = relu6f(v);
            }
        asm volatile("" ::: "memory");
        // ---- projection of this chunk's channels
        if constexpr (BF) {
#pragma unroll
            for (int s16 = 0; s16 < 2; ++s16) {
                if (ch0 + s16 * 16 < a.cexp) {                     // (uniform)
                    if (s16 == 0 && PBN == 2) load_pbf(1, 1);
                    if (s16 == 1 && PBN == 1) load_pbf(1, 0);
                    const float* pa = ET + r * CEP + s16 * 16 + half * 8;
                    f32x4 v0 = *(const f32x4*)pa, v1 = *(const f32x4*)(pa + 4);
                    if (a.cexp & 8) {                                  // (uniform; an expansion width with half a 16-k step: the channels past cexp hold clamped garbage -- zeros instead)
                        const bool kok = ch0 + s16 * 16 + half * 8 < a.cexp;
                        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                        v0 = kok ? v0 : z4; v1 = kok ? v1 : z4;
                    }
                    fb16x8 dh, dl;
                    fsplit8(v0, v1, dh, dl);
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, pbf[s16 % PBN][nt][0], pacc[nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, pbf[s16 % PBN][nt][1], pacc[nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl, pbf[s16 % PBN][nt][0], pacc[nt], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            if (kq < kqc) {
                if constexpr (PJIT) { if (kq + 1 < 4) load_p(kq + 1, (kq + 1) & 1); }
                const f32x4 av = *(const f32x4*)(ET + r * CEP + kq * 8 + half * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], pfrag[PJIT ? (kq & 1) : kq][nt][t], pacc[nt], 0, 0, 0);
            }
        }
        }
        asm volatile("" ::: "memory");
        ebias = enext;
    }
    // ---- output (+ residual): as k_block_fused4
    float* __restrict__ ob = a.out + out_base * a.cout;
    const float* __restrict__ rb = a.X + in_base * a.cin;
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt) {
        if (nt * 32 < a.cout) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) ET[((reg & 3) + 8 * (reg >> 2) + 4 * half) * CEP + r] = pacc[nt][reg];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int piece = lane + 64 * k, px = piece >> 3, c4 = piece & 7;
                const int col = nt * 32 + c4 * 4;
                const int oy = oy0 + px / TW, ox = ox0 + px % TW;
                f32x4 v = *(const f32x4*)(ET + px * CEP + c4 * 4);
                if (col < a.cout && oy < lv.Ho && ox < lv.Wo) {
                    const unsigned off = (unsigned)(oy * lv.Wo + ox) * (unsigned)a.cout + (unsigned)col;
                    if (RES) {
                        const f32x4 rv = *(const f32x4*)(rb + off);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] + rv[j];
                    }
                    *(f32x4*)(ob + off) = v;
                }
            }
            asm volatile("" ::: "memory");
        }
    }
}

template <int STRIDE, int NTO, int KQT, int OCC, bool BF = false>
static hipError_t launch_block_fused8_t(const FusedArgs& a, const Geom& g, hipStream_t s) {
    if (a.residual && (STRIDE != 1 || a.cin != a.cout)) return hipErrorInvalidValue;
    FusedArgs b = a;
    long long total = 0;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) {
        b.level_wgs[l] = 8;
        if (l >= g.n_levels) continue;
        const int tiles = ((g.lv[l].Wo + 7) / 8) * ((g.lv[l].Ho + 3) / 4);
        b.level_wgs[l] = max(8, ((tiles + 7) / 8) * 8);
        total += (long long)b.level_wgs[l] * g.batch;
    }
    if (total <= 0 || total > 0x7fffffffll) return hipErrorInvalidValue;
    if (BF && (!a.Wex_bf || !a.Wpr_bf)) return hipErrorInvalidValue;
    if (BF && KQT % 2 == 1) {                                      // the kernel's BIASK form: the expansion pack with the bias row
        if (!a.Wex_bfb) return hipErrorInvalidValue;
        b.Wex_bf = a.Wex_bfb;
    }
    if (a.residual) hipLaunchKernelGGL((k_block_fused8<STRIDE, NTO, KQT, true, OCC, BF>), dim3((unsigned)total), dim3(64), 0, s, b, g);
    else hipLaunchKernelGGL((k_block_fused8<STRIDE, NTO, KQT, false, OCC, BF>), dim3((unsigned)total), dim3(64), 0, s, b, g);
    return hipGetLastError();
}

// ---- v6 of the fused block (stride 1): the wave-autonomous scheme of v4 on v_mfma_f32_16x16x4_f32 with a 6 x 8 output tile.
// What v4 pays for its autonomy is the halo: a 4 x 8 tile's 6 x 10 positions are two 32-row M tiles for one tile of outputs
// (expansion executed 2.0x), and 32-column tiles pad cout = 48 / 72 to 64 / 96.  With 16-row / 16-column tiles a 6 x 8 tile's
// 8 x 10 halo is exactly five M tiles for three tiles of outputs (1.67x) and the projection runs on 48 / 80 columns.  Same
// flop rate per instruction (1024 fma in 8 passes against 2048 in 16), same chains (k ascending from the folded bias: the
// channels of a group of 16 sit in "slot" order, lane group g of MFMA t reads logical channel 4 t + g -- ConvPack16, as
// k_dwproject), same bits.  The block input is in the 32x32x2 kernels' channel order in HBM, so a lane gathers its A values as
// single floats (once per tile, they stay in registers for all chunks).  30 x 47 maps: 30 tiles per frame -- a 64-frame call
// is 1920 wave tiles, one round at two waves per SIMD (v4: 3072, a round and a half).
struct F6Geo {
    static constexpr int TH = 6, TW = 8, IH = 8, IW = 10, NPOS = 80, MT = 5, MO = 3;
    static constexpr int EP = 84;                     // channel stride of ET in floats (21 pieces of 16 bytes: odd)
    static constexpr int CEP = 36;                    // pixel stride of D
};
__device__ __forceinline__ int f6_slot_of_phys(int p) {          // chunk-local device position -> slot of its logical channel (k_dwproject's order)
    const int r = p & 7, l = (p & ~7) | (r < 4 ? 2 * r : 2 * (r - 4) + 1);
    return (l & ~15) | ((l & 3) << 2) | ((l & 15) >> 2);
}

template <int KT, int NT16, bool RES, int OCC>
__global__ __launch_bounds__(64, OCC) void k_block_fused6(FusedArgs a, Geom g) {
    using G = F6Geo;
    constexpr int TH = G::TH, TW = G::TW, IH = G::IH, IW = G::IW, NPOS = G::NPOS, MT = G::MT, MO = G::MO, EP = G::EP, CEP = G::CEP;
    constexpr int KB = (KT + 3) / 4;                              // groups of 16 input channels (the last one may hold 8)
    __shared__ __attribute__((aligned(16))) float ET[32 * EP];
    const int lane = threadIdx.x, j = lane & 15, gq = lane >> 4;
    int level = 0, bx = blockIdx.x;
    for (; level < g.n_levels - 1; ++level) {
        const int per = g.batch * a.level_wgs[level];
        if (bx < per) break;
        bx -= per;
    }
    const int frame = bx / a.level_wgs[level];
    bx -= frame * a.level_wgs[level];
    const int image = level * g.batch + frame;
    const LevelGeom lv = g.lv[level];
    const int tiles_x = (lv.Wo + TW - 1) / TW, ntiles = tiles_x * ((lv.Ho + TH - 1) / TH);
    const int q = ntiles >> 3, rem = ntiles & 7;                  // (XCD mapping: as k_block_fused4)
    const int xr = (bx + image) & 7, slot = bx >> 3;
    if (slot >= q + (xr < rem ? 1 : 0)) return;
    const int tile = xr * q + min(xr, rem) + slot;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 - lv.pt, ix0 = ox0 - lv.pl;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= lv.H && ix0 + IW <= lv.W;
    const long long in_base = lv.in_off + (long long)frame * lv.H * lv.W;
    const long long out_base = lv.out_off + (long long)frame * lv.Ho * lv.Wo;
    const char* __restrict__ xb = (const char*)(a.X + in_base * a.cin);      // uniform; lane offsets are 32-bit

    // ---- block input: MFMA t of group kb wants logical channel 16 kb + 4 t + gq of halo position 16 m + j; in the tensor's
    //      channel order (even channels of a group of 8 first) that is float 16 kb + 8 (t >> 1) + 2 (t & 1) + [4 (gq & 1) + (gq >> 1)]
    float afrag[MT][KT];
    const unsigned lane_c = (unsigned)(4 * (gq & 1) + (gq >> 1)) * 4u;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pp = m * 16 + j;
        const int hy = pp / IW, hx = pp - hy * IW;
        const int iy = min(max(iy0 + hy, 0), lv.H - 1), ix = min(max(ix0 + hx, 0), lv.W - 1);
        const unsigned off = (unsigned)(iy * lv.W + ix) * (unsigned)a.cin * 4u + lane_c;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            afrag[m][kt] = *(const float*)(xb + off + (unsigned)(16 * (kt >> 2) + 8 * ((kt & 3) >> 1) + 2 * (kt & 1)) * 4u);
    }
    unsigned long long outside[2] = {0ull, 0ull};
    if (!interior) {
        const unsigned long long full = (1ull << IW) - 1;
        const int clo = min(max(-ix0, 0), IW), chi = min(max(lv.W - ix0, 0), IW);          // columns [clo, chi) are inside
        const unsigned long long cols = (((1ull << clo) - 1) | ~((1ull << chi) - 1)) & full;
#pragma unroll
        for (int hy = 0; hy < IH; ++hy) {
            const unsigned long long bits = (iy0 + hy < 0 || iy0 + hy >= lv.H) ? full : cols;
            const int pos = hy * IW, wd = pos >> 6, sh = pos & 63;
            outside[wd] |= bits << sh;
            if (sh + IW > 64) outside[wd + 1] |= bits >> (64 - sh);
        }
    }
    f32x4 pacc[MO][NT16];
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) {
        const float pb = a.pr_bias[nt * 16 + j];                   // (zero padded to pr_nt_total * 32 >= NT16 * 16)
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) pacc[mo][nt] = f32x4{pb, pb, pb, pb};
    }
    const int ex_n16 = a.ex_n16, n_chunks = (ex_n16 + 1) >> 1;
    const unsigned lane16_ = (unsigned)lane * 16u, j4_ = (unsigned)j * 4u;
    // expansion weights: one 16-byte piece per (input group, column tile), streamed through a ring PF pieces deep
    constexpr int NP = 2 * KB, PF = NP % 4 == 0 ? 4 : (NP % 3 == 0 ? 3 : (NP % 5 == 0 ? 5 : 2));   // (a divisor of NP: piece s of every chunk lives in slot s % PF)
    static_assert(NP % PF == 0, "ring depth");
    auto ex_piece = [&](int chunk, int s) -> f32x4 {              // s = kb * 2 + nt
        const int nt = min(chunk * 2 + (s & 1), ex_n16 - 1);      // (a last chunk of 16 channels: the second tile repeats the first; never consumed)
        return *(gvec4_t)(sgpr_base(a.Wex16, (unsigned)((s >> 1) * ex_n16 + nt) * 1024u) + fresh(lane16_));
    };
    f32x4 bq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) bq[u] = ex_piece(0, u);
    const int r = lane & 31, rh = lane >> 5;                      // depthwise role: channel r of the chunk, output rows 3 rh .. 3 rh + 2
    const int dslot = f6_slot_of_phys(r);
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int ch0 = chunk * 32;
        const int kbc = min(2, (a.cexp - ch0) >> 4);               // 16-channel groups of this chunk the projection consumes
        const int cn = min(chunk + 1, n_chunks - 1);
        float eb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) eb[nt] = *(gf32_t)(sgpr_base(a.ex_bias, (unsigned)(min(chunk * 2 + nt, ex_n16 - 1) * 16) * 4u) + fresh(j4_));
        const unsigned dch4 = (unsigned)min(r, a.cexp - 1 - ch0) * 4u;      // (channels past cexp: clamped, never consumed)
        float dwt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) dwt[t] = *(gf32_t)(sgpr_base(a.Wdw, (unsigned)(t * a.cexp + ch0) * 4u) + dch4);
        const float dwb = *(gf32_t)(sgpr_base(a.dw_bias, (unsigned)ch0 * 4u) + dch4);
        // ---- expansion: ten independent chains (five halo M tiles x two column tiles), k outer
        {
            f32x4 acc[MT][2];
            const f32x4 ebv[2] = {f32x4{eb[0], eb[0], eb[0], eb[0]}, f32x4{eb[1], eb[1], eb[1], eb[1]}};   // C operand of every chain's first MFMA
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const f32x4 b = bq[s % PF];
                // the piece PF steps ahead (the next chunk's first pieces at the end: they arrive during the depthwise phase)
                bq[s % PF] = s + PF < NP ? ex_piece(chunk, s + PF) : ex_piece(cn, s + PF - NP);
                const int kb = s >> 1, nt = s & 1;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (kb * 4 + t < KT) {
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[m][kb * 4 + t], b[t], (kb == 0 && t == 0) ? ebv[nt] : acc[m][nt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // D fragment: column (channel) nt * 16 + j, rows (halo positions) 16 m + 4 gq .. + 3
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu6f(acc[m][nt][i]);
                    *(f32x4*)(ET + (nt * 16 + j) * EP + m * 16 + 4 * gq) = v;
                }
        }
        if (!interior) {
#pragma unroll
            for (int wd = 0; wd < 2; ++wd) {
                unsigned long long msk = outside[wd];                                  // uniform
                while (msk) {
                    const int pp = wd * 64 + (int)__builtin_ctzll(msk);
                    msk &= msk - 1;
                    if (lane < 32) ET[lane * EP + pp] = 0.0f;
                }
            }
        }
        asm volatile("" ::: "memory");
        // ---- depthwise: channel r, output rows 3 rh .. 3 rh + 2 (input rows 3 rh .. 3 rh + 4), one input row in registers at a time
        f32x4 pfrag[2][NT16];
        {
            const float* rp = ET + r * EP + rh * 3 * IW;
            float o[3][TW];
#pragma unroll
            for (int ro = 0; ro < 3; ++ro)
#pragma unroll
                for (int ox = 0; ox < TW; ++ox) o[ro][ox] = dwb;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                float row[IW];
#pragma unroll
                for (int x2 = 0; x2 < IW / 2; ++x2) {
                    const float2 v = *(const float2*)(rp + i * IW + x2 * 2);
                    row[2 * x2] = v.x; row[2 * x2 + 1] = v.y;
                }
#pragma unroll
                for (int ro = 0; ro < 3; ++ro) {
                    const int ky = i - ro;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int ox = 0; ox < TW; ++ox)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) o[ro][ox] = fmaf(row[ox + kx], dwt[ky * 3 + kx], o[ro][ox]);
                }
            }
#pragma unroll
            for (int ro = 0; ro < 3; ++ro)
#pragma unroll
                for (int ox = 0; ox < TW; ++ox) o[ro][ox] = relu6f(o[ro][ox]);
            asm volatile("" ::: "memory");                         // every ET read is issued before D overwrites the slice
            // this chunk's projection weights: requested here, into registers the expansion has released (requesting them before
            // the depthwise arithmetic, or a deeper ring for the expansion weights, changes nothing: measured)
            {
                const unsigned l16 = fresh(lane16_);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int nt = 0; nt < NT16; ++nt)
                        pfrag[kb][nt] = *(gvec4_t)(sgpr_base(a.Wpr16, (unsigned)((chunk * 2 + min(kb, kbc - 1)) * a.pr_n16 + nt) * 1024u) + l16);
            }
#pragma unroll
            for (int ro = 0; ro < 3; ++ro)
#pragma unroll
                for (int ox = 0; ox < TW; ++ox) ET[((rh * 3 + ro) * TW + ox) * CEP + dslot] = o[ro][ox];
        }
        asm volatile("" ::: "memory");
        // ---- projection of this chunk's channels: A = D[pixel 16 mo + j][slots 16 kb + 4 gq ..]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb < kbc) {
                f32x4 av[MO];
#pragma unroll
                for (int mo = 0; mo < MO; ++mo) av[mo] = *(const f32x4*)(ET + (mo * 16 + j) * CEP + kb * 16 + 4 * gq);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                        for (int nt = 0; nt < NT16; ++nt) pacc[mo][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mo][t], pfrag[kb][nt][t], pacc[mo][nt], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
    }
    // ---- output (+ residual): two column tiles at a time through the LDS slice so that a lane moves 16 consecutive bytes
    float* __restrict__ ob = a.out + out_base * a.cout;                                // uniform
    const float* __restrict__ rb = a.X + in_base * a.cin;                              // residual: same size, cin == cout
#pragma unroll
    for (int np = 0; np < (NT16 + 1) / 2; ++np) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nt = 2 * np + h;
            if (nt < NT16) {
#pragma unroll
                for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ET[(mo * 16 + 4 * gq + i) * CEP + h * 16 + j] = pacc[mo][nt][i];
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int piece = lane + 64 * k, px = piece >> 3, c4 = piece & 7;
            const int col = np * 32 + c4 * 4;
            const int oy = oy0 + px / TW, ox = ox0 + px % TW;
            f32x4 v = *(const f32x4*)(ET + px * CEP + c4 * 4);
            if (col < a.cout && oy < lv.Ho && ox < lv.Wo) {
                const unsigned off = (unsigned)(oy * lv.Wo + ox) * (unsigned)a.cout + (unsigned)col;
                if (RES) {
                    const f32x4 rv = *(const f32x4*)(rb + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] + rv[e];
                }
                *(f32x4*)(ob + off) = v;
            }
        }
        asm volatile("" ::: "memory");
    }
}

template <int KT, int NT16, int OCC>
static hipError_t launch_block_fused6_t(const FusedArgs& a, const Geom& g, hipStream_t s) {
    using G = F6Geo;
    if ((a.residual && a.cin != a.cout) || a.cin != KT * 4 || a.pr_n16 != NT16 || !a.Wex16 || !a.Wpr16) return hipErrorInvalidValue;
    FusedArgs b = a;
    long long total = 0;
    for (int l = 0; l < HFNET_MAX_LEVELS; ++l) {
        b.level_wgs[l] = 8;
        if (l >= g.n_levels) continue;
        const int tiles = ((g.lv[l].Wo + G::TW - 1) / G::TW) * ((g.lv[l].Ho + G::TH - 1) / G::TH);
        b.level_wgs[l] = max(8, ((tiles + 7) / 8) * 8);
        total += (long long)b.level_wgs[l] * g.batch;
    }
    if (total <= 0 || total > 0x7fffffffll) return hipErrorInvalidValue;
    dim3 grid((unsigned)total);
    if (a.residual) hipLaunchKernelGGL((k_block_fused6<KT, NT16, true, OCC>), grid, dim3(64), 0, s, b, g);
    else hipLaunchKernelGGL((k_block_fused6<KT, NT16, false, OCC>), grid, dim3(64), 0, s, b, g);
    return hipGetLastError();
}

// ---- layer_2 (expanded_conv with expansion factor 1: no expand conv, hf_net.py:31-33): depthwise 3x3 + BN + ReLU6 on
// CIN channels, then the 1x1 projection CIN -> COUT + BN, stride 1.  Its tensors are the largest of the network (1/2
// resolution) and its arithmetic the smallest: one thread per output pixel of a 16 x 16 tile on the vector ALUs, the
// fma chain over the CIN depthwise outputs in logical channel order (the oracle's), 16-byte fully coalesced stores.
//
// Shared tail of k_block_noexpand and k_stem_block2: `tile` holds the (T+2)^2 halo tile of the block input, CP floats per
// position (out-of-image positions are zeros: fma(0, w, d) == d, the oracle skips those taps).  The weights are
// wave-uniform scalar loads with compile-time offsets (SGPR fma operands).  Left alone, the scheduler hoists all ~650 of
// them to the top and spills them to VGPR lanes (v_writelane / v_readlane: more VALU work than the convolution itself), so
// they are fetched one step ahead of their use and scheduling barriers keep each batch where it is.
// depthwise 3x3 + folded BN + ReLU6 of one pixel from the LDS halo tile: d[physical channel]
// Row pitch of the halo tile in floats: (T + 2) positions of CIN + 4 floats, rounded up to a multiple of 64.  A 16-byte LDS read is
// serviced in groups of 16 lanes that mix two tile rows (lanes 0-3, 12-15 of one row with lanes 4-11 of the next): with the plain
// pitch 18 * 28 = 504 the two rows' bank patterns collide in two places (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.40 for
// k_stem_block2); a pitch that is 0 mod 64 makes the second row continue the first row's pattern.
template <int CIN, int T>
constexpr int halo_row_pitch() { return ((T + 2) * (CIN + 4) + 63) / 64 * 64; }
template <int CIN, int T>
__device__ __forceinline__ void depthwise_from_tile(const float* tile, int ty, int tx, const float* __restrict__ wd /*[9][CIN] phys, BN folded*/,
                                                    const float* __restrict__ dbias, float (&d)[CIN]) {
    constexpr int CP = CIN + 4, RP = halo_row_pitch<CIN, T>();
#pragma unroll
    for (int c = 0; c < CIN; ++c) d[c] = dbias[c];         // accumulators start at the folded bias
    float wc[CIN], wn[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) wc[c] = wd[c];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) wn[c] = wd[(tap + 1) * CIN + c];
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        const float* xp = tile + (ty + tap / 3) * RP + (tx + tap % 3) * CP;
#pragma unroll
        for (int c4 = 0; c4 < CIN / 4; ++c4) {
            const f32x4 xv = *(const f32x4*)(xp + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[c4 * 4 + j] = fmaf(xv[j], wc[c4 * 4 + j], d[c4 * 4 + j]);
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        if (tap < 8) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) wc[c] = wn[c];
        }
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c) d[c] = relu6f(d[c]);
}

template <int CIN, int COUT, int T>
__device__ __forceinline__ void dw_project_tail(const float* tile, int ty, int tx, const float* __restrict__ wd /*[9][CIN] phys, BN folded*/,
                                                const float* __restrict__ dbias, const float* __restrict__ wp /*[CIN logical][COUT phys], BN folded*/,
                                                const float* __restrict__ pbias, float (&acc)[COUT]) {
    float d[CIN];
    depthwise_from_tile<CIN, T>(tile, ty, tx, wd, dbias, d);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < COUT; ++n) acc[n] = pbias[n];
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
        if (k + 1 < CIN) {
#pragma unroll
            for (int n = 0; n < COUT; ++n) pc[n] = pn[n];
        }
    }
}

// output of a 16 x 16 tile through LDS: a tile row is T pixels x COUT channels = one contiguous 1 KB run of the output
// tensor; written back as consecutive 16-byte pieces per lane instead of four 64-byte-strided stores per thread
template <int COUT, int T>
__device__ __forceinline__ void store_tile_via_lds(float* tile, const float (&acc)[COUT], float* __restrict__ obase, int oy0, int ox0, int Ho, int Wo) {
    constexpr int OP = COUT + 4;                           // 20 words: conflict-free b128
    __syncthreads();                                       // every thread is done reading the input tile
#pragma unroll
    for (int n4 = 0; n4 < COUT / 4; ++n4) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[n4 * 4 + j];
        *(f32x4*)(tile + threadIdx.x * OP + n4 * 4) = v;
    }
    __syncthreads();
    const int cols = min(T, Wo - ox0);                     // valid pixels per tile row
#pragma unroll
    for (int k = 0; k < COUT / 4; ++k) {
        const int q = threadIdx.x + k * 256;               // piece of the tile: row = q / (T * COUT / 4)
        const int row = q / (T * COUT / 4), rem = q - row * (T * COUT / 4);
        const int px = rem / (COUT / 4), part = rem - px * (COUT / 4);
        if (oy0 + row < Ho && px < cols)
            *(f32x4*)(obase + ((long long)(oy0 + row) * Wo + ox0 + px) * COUT + part * 4) = *(const f32x4*)(tile + (row * T + px) * OP + part * 4);
    }
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_block_noexpand(const float* __restrict__ X, float* __restrict__ out, const float* __restrict__ wd,
                                                        const float* __restrict__ dbias, const float* __restrict__ wp,
                                                        const float* __restrict__ pbias, Geom g) {
    // 16x16 output tile per workgroup; the 18x18 input halo tile is staged through LDS with coalesced 96-byte
    // pixel rows, so every input byte crosses HBM ~1.27x instead of up to 9x.
    constexpr int T = 16, SH = T + 2, CP = CIN + 4, RP = halo_row_pitch<CIN, T>();
    __shared__ __attribute__((aligned(16))) float tile[SH * RP];
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
                *(f32x4*)(tp + hy * RP) = ok ? v : zero;
            }
        }
    }
    __syncthreads();
    const int ty = threadIdx.x / T, tx = threadIdx.x - ty * T;
    float acc[COUT];
    dw_project_tail<CIN, COUT, T>(tile, ty, tx, wd, dbias, wp, pbias, acc);
    store_tile_via_lds<COUT, T>(tile, acc, out + (lv.out_off + (long long)frame * lv.Ho * lv.Wo) * COUT, oy0, ox0, lv.Ho, lv.Wo);
}

// ---- stem + layer_2 in one launch: u8 image -> [(x-128)/128, conv 3x3/2 1->CS, BN, ReLU6] -> depthwise 3x3 +
// BN + ReLU6 -> 1x1 CS->COUT + BN.  The half-resolution CS-channel stem tensor (the largest activation of the
// network: 8.7 MB per 752x480 frame, 690 MB per 32-frame step, written by one kernel and read by the next) is produced
// into LDS for an 18x18 halo tile and consumed from there; only the u8 image is read and the COUT-channel layer_2 output
// written.  Vector-ALU kernel (K = 9 / 9 / CS), one thread per output pixel of a 16x16 tile; every fma chain is the
// oracle's.  The stem of the 18 x 18 halo positions comes straight from the u8 image (wave-uniform scalar weights; the
// 324 positions x 2 channel halves are 12 wave-sized work units, three per wave), the rest is dw_project_tail.
template <int CS, int COUT>
__global__ __launch_bounds__(256) void k_stem_block2(ImageSet imgs, const float* __restrict__ stem_w, const float* __restrict__ stem_bias,
                                                     const float* __restrict__ wd, const float* __restrict__ dbias, const float* __restrict__ wp,
                                                     const float* __restrict__ pbias, float* __restrict__ out,
                                                     Geom gs /*image -> stem*/, Geom gb /*stem -> layer_2*/) {
    constexpr int T = 16, SH = T + 2, SP = SH * SH, CP = CS + 4, CH = CS / 2, RP = halo_row_pitch<CS, T>();
    static_assert(CS == 24 && COUT == 16, "written for the 0.75-width network");
    __shared__ __attribute__((aligned(16))) float tile[SH * RP];
    int level, frame, tile_id, tiles_x;
    decode_tile_grid<T, T>(gb, blockIdx.x, level, frame, tile_id, tiles_x);
    const LevelGeom ls = gs.lv[level], lb = gb.lv[level];     // ls: H,W image (cropped), Ho,Wo stem; lb: H,W stem, Ho,Wo out (same size)
    const int tyi = tile_id / tiles_x, txi = tile_id - tyi * tiles_x;
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
                // a row's three bytes as ONE (unaligned) 4-byte load -- the fourth byte is the next pixel of a row that is never
                // the image's last one here -- and (x - 128) / 128 as one fma: x / 128 - 1 is exact, the same value
                const uint8_t* ip = img + (long long)(sy * 2 - ls.pt) * rs + (sx * 2 - ls.pl);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    unsigned wv;
                    __builtin_memcpy(&wv, ip + ky * rs, 4);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) px[ky * 3 + kx] = fmaf((float)((wv >> (8 * kx)) & 0xffu), 0.0078125f, -1.0f);
                }
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
            // weights of four channels at a time as wave-uniform 16-byte scalar loads (36 + 4 SGPRs per group: a second set
            // prefetched ahead does not fit next to the geometry without spilling to VGPR lanes)
            const f32x4* __restrict__ w4 = (const f32x4*)(stem_w + hsel * CH);    // uniform; row t is w4[t * CS / 4 + group]
            const f32x4* __restrict__ sh4 = (const f32x4*)(stem_bias + hsel * CH);
            float* tp = tile + hy * RP + hx * CP + hsel * CH;
#pragma unroll
            for (int grp = 0; grp < CH / 4; ++grp) {
                f32x4 wq[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wq[t] = w4[t * (CS / 4) + grp];
                const f32x4 shq = sh4[grp];
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = shq[j];
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc = fmaf(px[t], wq[t][j], acc);
                    float v = relu6f(acc);
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
    float d[CS];
    depthwise_from_tile<CS, T>(tile, ty, tx, wd, dbias, d);
    // ---- 1x1 projection CS -> COUT on the matrix cores: v_mfma_f32_16x16x4_f32, one M tile per tile row of 16 pixels (a wave
    //      owns four rows), k = the logical channels in ascending order, four per instruction, accumulators starting at the
    //      folded bias: the oracle's fma chain (tools/micro/mfma_order.hip).  The pixel-per-thread depthwise results go
    //      through LDS once, stored so that lane (i, g) of the MFMA finds its A values  k = 4 j + g, j = 0..3  in one
    //      16-byte piece and  k = 16 + 4 j + g, j = 0, 1  in one 8-byte piece of pixel i's row.
    constexpr int DP = CS + 4;                                  // 28 floats per pixel: 16 consecutive rows cover all banks once
    __syncthreads();                                            // every thread is done with the stem tile
    {
        auto phys = [](int k) { return (k & ~7) | ((k & 1) << 2) | ((k & 7) >> 1); };   // slot of logical channel k in a pixel
        float* dp = tile + threadIdx.x * DP;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = d[phys(4 * j + g)];
            *(f32x4*)(dp + 4 * g) = v;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) *(float2*)(dp + 16 + 2 * g) = float2{d[phys(16 + g)], d[phys(20 + g)]};
    }
    const int li = lane & 15, lg = lane >> 4;
    float bw[CS / 4];
#pragma unroll
    for (int j = 0; j < CS / 4; ++j) bw[j] = wp[(4 * j + lg) * COUT + li];
    const float pb = pbias[li];
    f32x4 pacc[4];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // a wave reads only the rows its own lanes wrote
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float* ap = tile + ((wave * 4 + m) * T + li) * DP;
        const f32x4 a0 = *(const f32x4*)(ap + 4 * lg);
        const float2 a1 = *(const float2*)(ap + 16 + 2 * lg);
        f32x4 c = {pb, pb, pb, pb};
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bw[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bw[4], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bw[5], c, 0, 0, 0);
        pacc[m] = c;
    }
    // ---- output: D[pixel 4 g + i of the row][channel lane % 16] -> LDS [pixel][COUT + 4] -> 16-byte pieces of contiguous rows
    constexpr int OP = COUT + 4;
    __syncthreads();                                            // every wave has read its A rows
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[((wave * 4 + m) * T + 4 * lg + i) * OP + li] = pacc[m][i];
    __syncthreads();
    float* obase = out + (lb.out_off + (long long)frame * lb.Ho * lb.Wo) * COUT;
    const int cols = min(T, lb.Wo - ox0);
#pragma unroll
    for (int k = 0; k < COUT / 4; ++k) {
        const int q = threadIdx.x + k * 256;
        const int row = q / (T * COUT / 4), rem = q - row * (T * COUT / 4);
        const int px = rem / (COUT / 4), part = rem - px * (COUT / 4);
        if (oy0 + row < lb.Ho && px < cols)
            *(f32x4*)(obase + ((long long)(oy0 + row) * lb.Wo + ox0 + px) * COUT + part * 4) = *(const f32x4*)(tile + (row * T + px) * OP + part * 4);
    }
}

bool stem_block_fusable(int stem_out, const BlockPack& b) {
    return stem_out == 24 && !b.has_expand && b.stride == 1 && !b.residual && b.cin == 24 && b.cout == 16 && b.pr_logical != nullptr;
}

hipError_t launch_stem_block(const ImageSet& imgs, const float* stem_w, const float* stem_bias, const BlockPack& b, float* out, const Geom& g_stem,
                             const Geom& g_block, hipStream_t s) {
    const long long total = tile_grid_size<16, 16>(g_block);
    if (total <= 0 || total > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_stem_block2<24, 16>), dim3((unsigned)total), dim3(256), 0, s, imgs, stem_w, stem_bias, b.dw.w,
                       b.dw.bias, b.pr_logical, b.pr.bias, out, g_stem, g_block);
    return hipGetLastError();
}

// the shapes with a fused kernel (the 0.75-width network's; anything else runs as expand / depthwise / project launches)
enum FusedKind { FUSED_NONE = 0, FUSED_NOEXPAND, FUSED_V4, FUSED_V2 };
static FusedKind fused_kind(const BlockPack& b, int variant) {
    const int nto = (b.cout + 31) / 32, kq = b.cin / 8, st = b.stride;
    if (b.cin % 8 || b.expand % 8) return FUSED_NONE;
    if (!b.has_expand) return (st == 1 && !b.residual && b.cin == 24 && b.cout == 16 && b.pr_logical) ? FUSED_NOEXPAND : FUSED_NONE;
    if (b.pr.nt_total != nto) return FUSED_NONE;
    const bool v2 = (st == 2 && (kq == 2 || kq == 3) && nto == 1) || (st == 2 && kq == 12 && nto == 2) ||
                    (st == 1 && kq == 3 && nto <= 2) || (st == 1 && kq == 6 && (nto == 2 || nto == 3)) || (st == 1 && kq == 9 && nto == 3);
    const bool v4 = (st == 1 && kq == 3 && nto <= 2) || (st == 1 && kq == 6 && (nto == 2 || nto == 3)) || (st == 1 && kq == 9 && nto == 3) ||
                    (st == 2 && (kq == 2 || kq == 3) && nto == 1);
    if (variant != 2 && v4) return FUSED_V4;
    return v2 ? FUSED_V2 : FUSED_NONE;
}
bool block_fusable(const BlockPack& b, int variant) { return fused_kind(b, variant) != FUSED_NONE; }

bool block_fused_bf16x3_supported(const BlockPack& b) {
    const int nto = (b.cout + 31) / 32, kq = b.cin / 8, st = b.stride;
    return b.has_expand && b.ex_bf && b.pr_bf && b.cin % 8 == 0 && b.pr.nt_total == nto &&
           ((st == 1 && kq == 6 && (nto == 2 || nto == 3)) || (st == 1 && kq == 9 && nto == 3) ||      // layers 7, 9-14
            (st == 2 && (kq == 2 || kq == 3) && nto == 1) || (st == 1 && kq == 3 && nto <= 2) ||        // layers 3, 5 / 4, 6 (scores_bf16x3)
            (st == 2 && kq == 12 && nto == 2) ||                                                         // layer 8: one wave per SIMD, like its f32 form
            (st == 1 && kq == 15 && (nto == 4 || nto == 8)) || (st == 2 && kq == 9 && nto == 4));       // layers 16, 17 / 18 / 15 (no f32 fused form: split-bf16 only)
}

hipError_t launch_block_fused(const float* X, const BlockPack& b, float* out, const Geom& g, int variant, hipStream_t s, int bf16x3) {
    FusedArgs a;
    a.X = X;
    a.Wex_bf = b.ex_bf; a.Wpr_bf = b.pr_bf; a.Wex_bfb = b.ex_bfb;
    a.Wex = (const f32x4*)b.ex.w; a.ex_bias = b.ex.bias; a.ex_nt_total = b.ex.nt_total;
    a.Wdw = b.dw.w; a.dw_bias = b.dw.bias;
    a.Wpr = (const f32x4*)b.pr.w; a.pr_bias = b.pr.bias; a.pr_nt_total = b.pr.nt_total;
    a.Wex16 = (const f32x4*)b.ex16.w; a.Wpr16 = (const f32x4*)b.pr16.w; a.ex_n16 = b.ex16.w ? b.ex16.n16 : 0; a.pr_n16 = b.pr16.w ? b.pr16.n16 : 0;
    a.out = out; a.cin = b.cin; a.cexp = b.expand; a.cout = b.cout; a.residual = b.residual; a.has_expand = b.has_expand;
    const int nto = (b.cout + 31) / 32, kq = b.cin / 8, st = b.stride;
    const long long n_tiles = tile_grid_size<4, 8>(g);             // wave tiles of the k_block_fused4 launch
    const bool occ3 = n_tiles > 2048 || variant == 3;             // (variant 3: the three-waves-per-SIMD instantiations at any size -- tests)
    // A single frame has fewer wave tiles than the chip has SIMDs at two waves each: every wave of k_block_fused4 then runs
    // alone and pays the latency of each of its phases in full, while the barrier-phased kernel puts four waves on a tile
    // (752x480, one frame: layer 5 49 -> 20 us, layers 3-7 together 200 -> 150 us).  Same bits either way.
    FusedKind kind = fused_kind(b, variant);
    const bool small_launch = n_tiles < 2048;
    if (bf16x3 && block_fused_bf16x3_supported(b)) {              // engine option global_bf16x3: tolerance instead of the oracle's bits
        if (st == 1 && kq == 6 && nto == 2) return launch_block_fused8_t<1, 2, 6, 2, true>(a, g, s);
        if (st == 1 && kq == 6 && nto == 3) return launch_block_fused8_t<1, 3, 6, 2, true>(a, g, s);
        if (st == 1 && kq == 9 && nto == 3) return launch_block_fused8_t<1, 3, 9, 2, true>(a, g, s);
        if (st == 2 && kq == 2 && nto == 1) return launch_block_fused8_t<2, 1, 2, 2, true>(a, g, s);
        if (st == 2 && kq == 3 && nto == 1) return launch_block_fused8_t<2, 1, 3, 2, true>(a, g, s);
        if (st == 1 && kq == 3 && nto == 1) return launch_block_fused8_t<1, 1, 3, 3, true>(a, g, s);
        if (st == 1 && kq == 3 && nto == 2) return launch_block_fused8_t<1, 2, 3, 2, true>(a, g, s);
        if (st == 2 && kq == 12 && nto == 2) return launch_block_fused8_t<2, 2, 12, 1, true>(a, g, s);
        if (st == 1 && kq == 15 && nto == 4) return launch_block_fused8_t<1, 4, 15, 1, true>(a, g, s);
        if (st == 1 && kq == 15 && nto == 8) return launch_block_fused8_t<1, 8, 15, 1, true>(a, g, s);
        if (st == 2 && kq == 9 && nto == 4) return launch_block_fused8_t<2, 4, 9, 1, true>(a, g, s);
    }
    if (kind == FUSED_V4 && variant == 4 && small_launch && fused_kind(b, 2) == FUSED_V2) kind = FUSED_V2;
    // v6 (6 x 8 tiles on the 16x16x4 MFMA): what a launch of k_block_fused4's size runs for the stride-1 blocks from layer 6 on
    // (per 64 frames, v4 -> v6: layer 6 331 -> 303 us, 7 966 -> 957, 9-11 94 -> 87, 12 138 -> 98, 13 / 14 245 -> 168); layer 4 (four waves
    // per SIMD in v4, three here: its LDS tile) is slower and stays (variant 7: everywhere, variant 6: at any launch size -- tests)
    // v8 in its stride-1 form beats v6 where the projection has no column padding to lose (layer 7, 48 -> 288 -> 96: 1858 -> 1798 us
    // per 128 frames; everywhere else it is 1-7 % slower than v4 / v6: the lanes' exchange is vector work on a saturated port)
    // (measured per 128 frames, v4 or v6 -> v8: layer 4 1710 -> 1732 us, layer 6 567 -> 602, layer 9 163 -> 165, layer 12 183 -> 188, layer 13
    // 325 -> 347: those instantiations are not kept)
    // (one wave per SIMD for the other v8 shapes, per 128 frames: layer 3 1565 -> 2333 us, layer 5 834 -> 1077, layer 7 1798 -> 1831 -- a lone wave
    //  needs long MFMA chains per chunk (layer 8: 272, layer 7: 96, layer 3: 56 next to ~290 vector instructions) to keep its SIMD busy)
    if (kind == FUSED_V4 && st == 1 && kq == 6 && nto == 3 && (variant == 8 || (variant == 4 && !small_launch))) return launch_block_fused8_t<1, 3, 6, 2>(a, g, s);
    if (kind == FUSED_V4 && st == 1 && a.Wex16 && a.Wpr16 && (variant == 6 || variant == 7 || (variant == 4 && !small_launch))) {
        const int kt = b.cin / 4, n16 = a.pr_n16;
        if (kt == 6 && n16 == 3) return launch_block_fused6_t<6, 3, 2>(a, g, s);
        if (kt == 12 && n16 == 6) return launch_block_fused6_t<12, 6, 2>(a, g, s);
        if (kt == 12 && n16 == 3) return launch_block_fused6_t<12, 3, 2>(a, g, s);
        if (kt == 12 && n16 == 5) return launch_block_fused6_t<12, 5, 2>(a, g, s);
        if (kt == 18 && n16 == 5) return launch_block_fused6_t<18, 5, 2>(a, g, s);
        if (variant == 7 && kt == 6 && n16 == 2) return launch_block_fused6_t<6, 2, 3>(a, g, s);
    }
    // v8 (stride 2, expansion kept in registers): what a launch of k_block_fused4's size runs for the stride-2 blocks (per 128
    // frames, v4 -> v8: layer 3 1803 -> 1599 us, layer 5 954 -> 833); variant 8: at any launch size (tests)
    // layer 8 (96 -> 576 -> 48, stride 2) as a v8 tile at ONE wave per SIMD: the 240 registers of A fragments stay resident in the 512-entry
    // file (the input crosses L2 once instead of once per chunk), five independent expansion chains keep the matrix pipe busy by
    // themselves: 1273 -> 1064 us per 128 frames against the barrier kernel
    if (st == 2 && kq == 12 && nto == 2 && (variant == 8 || (variant == 4 && !small_launch))) return launch_block_fused8_t<2, 2, 12, 1>(a, g, s);
    if (kind == FUSED_V4 && st == 2 && nto == 1 && (variant == 8 || (variant == 4 && !small_launch))) {
        if (kq == 2) return launch_block_fused8_t<2, 1, 2, 2>(a, g, s);
        if (kq == 3) return launch_block_fused8_t<2, 1, 3, 2>(a, g, s);
    }
    switch (kind) {
        case FUSED_NOEXPAND: {
            int maxtiles = 0;
            for (int l = 0; l < g.n_levels; ++l) maxtiles = max(maxtiles, ((g.lv[l].Wo + 15) / 16) * ((g.lv[l].Ho + 15) / 16));
            hipLaunchKernelGGL((k_block_noexpand<24, 16>), dim3(maxtiles, g.n_levels * g.batch), dim3(256), 0, s, a.X, a.out, a.Wdw, a.dw_bias,
                               (const float*)b.pr_logical, a.pr_bias, g);
            return hipGetLastError();
        }
        case FUSED_V4:
            // <stride, 32-column output tiles, cin / 8, waves per SIMD the register budget aims at>
            if (st == 1 && kq == 3 && nto == 1) return launch_block_fused4_t<1, 1, 3, 4>(a, g, s);
            if (st == 1 && kq == 3 && nto == 2) return launch_block_fused4_t<1, 2, 3, 2>(a, g, s);
            // the 30 x 47 layers of a 64-frame call are 3072 tiles: at two waves per SIMD (2048 slots) that is a full round and a half
            // one; at three (KQO: streamed expansion weights, projection weights requested after the expansion) all of them are
            // resident at once: layers 9-11 129 -> 105 us.  Launches that fit two waves per SIMD keep those.  (Layer 12, three column
            // tiles, only fits with 16 spilled registers: 150 -> 128 us by itself, but a kernel with scratch memory that follows
            // one without pays ~40 us at its first launch in the stream, so no spilled variant is kept.)
            if (st == 1 && kq == 6 && nto == 2) return occ3 ? launch_block_fused4_t<1, 2, 6, 3, true>(a, g, s) : launch_block_fused4_t<1, 2, 6, 2>(a, g, s);
            if (st == 1 && kq == 6 && nto == 3) return launch_block_fused4_t<1, 3, 6, 2>(a, g, s);
            // (KQO: 244 registers instead of 256 + 9 spilled: layer 13 304 -> 255 us)
            if (st == 1 && kq == 9 && nto == 3) return launch_block_fused4_t<1, 3, 9, 2, true>(a, g, s);
            if (st == 2 && kq == 2 && nto == 1) return launch_block_fused4_t<2, 1, 2, 2>(a, g, s);
            if (st == 2 && kq == 3 && nto == 1) return launch_block_fused4_t<2, 1, 3, 2>(a, g, s);
            return hipErrorInvalidValue;
        case FUSED_V2:
            // a single frame does not even give every CU one 8 x 16 tile: 8 x 8 tiles (twice the workgroups, half the work each)
            if (small_launch && variant == 4) {
                if (st == 1 && kq == 3 && nto == 1) return launch_block_fused2_t<1, 1, 3, true, 8>(a, g, s);
                if (st == 1 && kq == 3 && nto == 2) return launch_block_fused2_t<1, 2, 3, true, 8>(a, g, s);
                if (st == 1 && kq == 6 && nto == 3) return launch_block_fused2_t<1, 3, 6, true, 8>(a, g, s);
            }
            if (st == 2 && kq == 2 && nto == 1) return launch_block_fused2_t<2, 1, 2, true>(a, g, s);
            if (st == 1 && kq == 3 && nto == 1) return launch_block_fused2_t<1, 1, 3, true>(a, g, s);
            if (st == 2 && kq == 3 && nto == 1) return launch_block_fused2_t<2, 1, 3, true>(a, g, s);
            if (st == 1 && kq == 3 && nto == 2) return launch_block_fused2_t<1, 2, 3, true>(a, g, s);
            if (st == 1 && kq == 6 && nto == 3) return launch_block_fused2_t<1, 3, 6, true>(a, g, s);
            if (st == 1 && kq == 6 && nto == 2) return launch_block_fused2_t<1, 2, 6, true>(a, g, s);
            if (st == 1 && kq == 9 && nto == 3) return launch_block_fused2_t<1, 3, 9, true>(a, g, s);
            if (st == 2 && kq == 12 && nto == 2) return launch_block_fused2_t<2, 2, 12, true>(a, g, s);
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hfnet
