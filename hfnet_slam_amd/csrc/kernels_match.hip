// kernels_match.hip -- brute-force descriptor matching and the keyframe-database scan.
//
//   Matcher::SearchForTriangulation  src/Matcher.cc:845-889   (dot-product GEMM + mutual arg-max)
//   Matcher::SearchByBoW x2           src/Matcher.cc:229-260, 574-618 (cv::BFMatcher L2 crossCheck)
//   Matcher::DescriptorDistance       src/Matcher.cc:1893-1900
//   KeyFrameDatabase scans            src/KeyFrameDatabase.cc:86-104, 178-197
#include "kernels.hpp"

#include <algorithm>
#include <mutex>
#include <cstdlib>
#include <type_traits>

#include <cfloat>

namespace hfnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#include "device_util.hpp"      // sgpr_base / fresh / gvec4_t: the operand pointers of the GEMMs below come out of a descriptor in memory

// =========================================================================== SearchForTriangulation
// S = D1 * D2^T on v_mfma_f32_32x32x2_f32 -- each accumulator is the fused multiply-add chain over k = 0, 1, 2, ... (the
// oracle's order, so the products are the oracle's bits) -- followed by the mutual arg-max of Matcher.cc:860-893.  S is
// never stored: the epilogue of a 128 x 128 workgroup tile reduces every wave's 64 x 64 part to (first maximum, index)
// per row and per column, 8 bytes per row / column and 64-wide tile instead of 256; two small kernels finish the
// reduction over the tiles in ascending order (strict >: the first maximum wins, as in the reference's loops) and do
// the cross-check.
//
// GEMM part: 128 x 128 tile per workgroup, 64 x 64 (2 x 2 MFMA tiles) per wave: every LDS fragment feeds two MFMAs.  K is
// consumed in chunks of 64 staged through LDS with coalesced 256-byte row segments (a lane's own row is 1 KB away from
// its neighbour's, so direct fragment loads thrash L1).  The LDS rows hold the even k of a chunk in their first half and
// the odd k in the second, so a half-wave (which supplies the even resp. odd k of every MFMA step) reads the operands of
// four consecutive steps with one 16-byte read and no select (VALU instructions and f32 MFMAs share the issue pipe).
// Rows are 68 floats apart: 16-byte reads of 16 consecutive rows cover all 64 banks once.
struct TriPart { float v; int idx; };
__host__ __device__ static inline int tri_tiles(int n) { return (n + 63) / 64; }
size_t tri_scratch_floats(int max_rows) { return (size_t)4 * (size_t)max_rows * tri_tiles(max_rows); }   // two directions x 8 bytes

__global__ __launch_bounds__(256) void k_tri_gemm_argmax(const BowPair* __restrict__ pairs, int dim, int max_rows, int only_flagged) {
    const BowPair P = pairs[blockIdx.z];
    if (only_flagged && P.qkey[0] == 0ull) return;            // (after the screened path: only the pairs whose candidate list overflowed)
    const float* __restrict__ d1 = P.q; const float* __restrict__ d2 = P.t;
    const int n1 = P.nq, n2 = P.nt;
    constexpr int LD = 68;
    __shared__ __attribute__((aligned(16))) float As[128 * LD];
    __shared__ __attribute__((aligned(16))) float Bs[128 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    const int row0 = blockIdx.y * 128, col0 = blockIdx.x * 128;
    if (row0 >= n1 || col0 >= n2) return;                     // workgroup-uniform (launches are sized for the largest pair)
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    // staging: 16 threads cover the 256 contiguous bytes of a row chunk; a thread handles rows (tid / 16) * 8 + j
    const int lc = tid & 15, lrow = (tid >> 4) * 8;
    unsigned aoff[8], boff[8];                                // byte offsets of this thread's eight row pieces (sets stay below 4 GB: launch check)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        aoff[j] = ((unsigned)min(row0 + lrow + j, n1 - 1) * (unsigned)dim + (unsigned)lc * 4u) * 4u;
        boff[j] = ((unsigned)min(col0 + lrow + j, n2 - 1) * (unsigned)dim + (unsigned)lc * 4u) * 4u;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    f32x4 sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sa[j] = *(gvec4_t)(sgpr_base(d1, 0) + aoff[j]); sb[j] = *(gvec4_t)(sgpr_base(d2, 0) + boff[j]); }
    for (int k0 = 0; k0 < dim; k0 += 64) {
        __syncthreads();                                     // previous chunk fully consumed
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float* ap = As + (lrow + j) * LD + lc * 2;
            float* bp = Bs + (lrow + j) * LD + lc * 2;
            *(float2*)(ap) = float2{sa[j][0], sa[j][2]}; *(float2*)(ap + 32) = float2{sa[j][1], sa[j][3]};
            *(float2*)(bp) = float2{sb[j][0], sb[j][2]}; *(float2*)(bp + 32) = float2{sb[j][1], sb[j][3]};
        }
        __syncthreads();
        {   // the next chunk's pieces, unconditionally (the last pass re-reads its own chunk: a branch around the loads makes the
            // compiler wait for them and copy them right here, in front of the MFMAs they are meant to hide behind)
            const unsigned kn = (unsigned)min(k0 + 64, dim - 64) * 4u;
            const gbase_t pa = sgpr_base(d1, kn), pb = sgpr_base(d2, kn);
#pragma unroll
            for (int j = 0; j < 8; ++j) { sa[j] = *(gvec4_t)(pa + fresh(aoff[j])); sb[j] = *(gvec4_t)(pb + fresh(boff[j])); }
            __builtin_amdgcn_sched_barrier(0);                // (left alone the scheduler sinks them below the MFMAs: nobody needs them before the next pass)
        }
        const float* ap = As + (wr + r) * LD + half * 32;
        const float* bp = Bs + (wc + r) * LD + half * 32;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const f32x4 a0 = *(const f32x4*)(ap + 4 * m), a1 = *(const f32x4*)(ap + 32 * LD + 4 * m);
            const f32x4 b0 = *(const f32x4*)(bp + 4 * m), b1 = *(const f32x4*)(bp + 32 * LD + 4 * m);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                          // the operand tiles are dead: every wave takes a quarter of the LDS
    if (row0 + wr >= n1 || col0 + wc >= n2) return;           // (wave-uniform, no barrier below) nothing of this 64 x 64 part is inside
    const int nt = tri_tiles(max_rows);
    TriPart* __restrict__ colpart = (TriPart*)P.St;           // [row tile of 64][max_rows]
    TriPart* __restrict__ rowpart = colpart + (size_t)nt * max_rows;   // [max_rows][column tile of 64]
    const int rv = min(64, n1 - row0 - wr), cv = min(64, n2 - col0 - wc);   // valid rows / columns of the part (uniform)
    // ---- column direction: a lane holds, for its column, 32 of the part's 64 rows (ascending with i, reg); the other
    //      half-wave holds the rows in between
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float best = -__builtin_inff();
        int bi = -1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rl = i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                const float v = acc[i][j][reg];
                if (rl < rv && v > best) { best = v; bi = rl; }
            }
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        if (oi >= 0 && (ob > best || bi < 0 || (ob == best && oi < bi))) { best = ob; bi = oi; }
        const int cl = j * 32 + r;
        if (half == 0 && cl < cv) colpart[(size_t)((row0 + wr) >> 6) * max_rows + col0 + wc + cl] = TriPart{best, bi < 0 ? -1 : row0 + wr + bi};
    }
    // ---- row direction: the part goes through LDS once so that a lane owns one row and scans its 64 columns in order
    float* T = As + wave * (64 * LD);                         // (As and Bs are contiguous: 4 x 64 x 68 floats)
    if (wave >= 2) T = Bs + (wave - 2) * (64 * LD);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) T[(i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * LD + j * 32 + r] = acc[i][j][reg];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private slice: program order is enough
    {
        const float* tp = T + lane * LD;
        float best = -__builtin_inff();
        int bj = -1;
#pragma unroll
        for (int c4 = 0; c4 < 16; ++c4) {
            const f32x4 v = *(const f32x4*)(tp + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c4 * 4 + e < cv && v[e] > best) { best = v[e]; bj = c4 * 4 + e; }
        }
        if (lane < rv) rowpart[(size_t)(row0 + wr + lane) * nt + ((col0 + wc) >> 6)] = TriPart{best, bj < 0 ? -1 : col0 + wc + bj};
    }
}

// column j: first row with the largest product above the threshold (Matcher.cc:877-889) -> P.tn (as int); resets the counter
__global__ __launch_bounds__(256) void k_tri_cols(const BowPair* __restrict__ pairs, float threshold, int max_rows, int only_flagged) {
    const BowPair P = pairs[blockIdx.z];
    if (only_flagged && P.qkey[0] == 0ull) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.cnt = 0;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P.nt) return;
    const TriPart* __restrict__ colpart = (const TriPart*)P.St;
    const int nrt = tri_tiles(P.nq);
    float best = threshold;
    int bi = -1;
    for (int t = 0; t < nrt; ++t) {
        const TriPart c = colpart[(size_t)t * max_rows + j];
        if (c.idx >= 0 && c.v > best) { best = c.v; bi = c.idx; }
    }
    ((int*)P.tn)[j] = bi;
}
// row i: first column with the largest product above the threshold, then the cross-check (Matcher.cc:860-893)
__global__ __launch_bounds__(256) void k_tri_rows(const BowPair* __restrict__ pairs, float threshold, int max_rows, int only_flagged) {
    const BowPair P = pairs[blockIdx.z];
    if (only_flagged && P.qkey[0] == 0ull) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.x * 256 >= P.nq) return;                     // workgroup-uniform
    int m = -1;
    if (i < P.nq) {
        const int nt = tri_tiles(max_rows), nct = tri_tiles(P.nt);
        const TriPart* __restrict__ rowpart = (const TriPart*)P.St + (size_t)nt * max_rows + (size_t)i * nt;
        float best = threshold;
        int bj = -1;
        for (int t = 0; t < nct; ++t) {
            const TriPart c = rowpart[t];
            if (c.idx >= 0 && c.v > best) { best = c.v; bj = c.idx; }
        }
        if (bj >= 0 && ((const int*)P.tn)[bj] == i) m = bj;
        P.match[i] = m;
    }
    const unsigned long long hit = __ballot(m >= 0);
    if ((threadIdx.x & 63) == 0 && hit) atomicAdd(P.cnt, __popcll(hit));
}

// =========================================================================== SearchByBoW (BFMatcher)
// cv::BFMatcher(NORM_L2, crossCheck=true) == batchDistance(K=1, crosscheck): every train row picks its
// nearest query (first minimum); a query is matched to the nearest train row that picked it.
// Distances are OpenCV's: sqrt(normL2Sqr), generic 4-way unrolled order  s += v0^2+v1^2+v2^2+v3^2.
// The MFMA dot products St[t][q] only pre-select: every query whose |t|^2+|q|^2-2St lies within a
// rigorous rounding band of the column minimum is re-evaluated in the exact form, in ascending q.
// one launch: |q|^2, |t|^2 (pre-filter only), reset of the per-query keys and of the match counter
// split != null: every row is also left as [k / 4][hi x 4, lo x 4] bf16 pieces (16 bytes per four k: the f32 row's size and
// piece addresses) for the split-bf16 screening GEMM -- see split_bf16 below
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (__bf16)v[i];                                  // (v_cvt_pk_bf16_f32: round to nearest even)
        lo[i] = (__bf16)(v[i] - (float)hi[i]);                 // (the difference is exact in fp32)
    }
}
__global__ __launch_bounds__(256) void k_bow_prep(const BowPair* __restrict__ pairs, int dim, bf16x8* __restrict__ split, int max_rows) {
    const BowPair P = pairs[blockIdx.z];
    const float* __restrict__ q = P.q; const float* __restrict__ t = P.t;
    const int nq = P.nq, nt = P.nt;
    float* __restrict__ qn = P.qn; float* __restrict__ tn = P.tn; unsigned long long* __restrict__ qkey = P.qkey;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.cnt = 0;
    if (i >= nq + nt) return;
    const float* x = i < nq ? q + (long long)i * dim : t + (long long)(i - nq) * dim;
    bf16x8* __restrict__ srow = split ? split + ((long long)(blockIdx.z * 2 + (i < nq ? 0 : 1)) * max_rows + (i < nq ? i : i - nq)) * (dim >> 2) : nullptr;
    float p = 0.0f;
    for (int k = lane * 4; k < dim; k += 256) {
        const f32x4 v = *(const f32x4*)(x + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) p = fmaf(v[j], v[j], p);
        if (srow) {
            bf16x4 h, l;
            split_bf16(v, h, l);
            srow[k >> 2] = bf16x8{h[0], h[1], h[2], h[3], l[0], l[1], l[2], l[3]};
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off, 64);
    if (lane == 0) {
        if (i < nq) { qn[i] = p; qkey[i] = ~0ull; }
        else tn[i - nq] = p;
    }
}

// exact OpenCV L2 between rows a and b (dim == 256): lane g computes group g, lanes then add in order
__device__ __forceinline__ float cv_l2_wave256(const float* a, const float* b, int lane) {
    const f32x4 av = *(const f32x4*)(a + lane * 4), bv = *(const f32x4*)(b + lane * 4);
    const float v0 = av[0] - bv[0], v1 = av[1] - bv[1], v2 = av[2] - bv[2], v3 = av[3] - bv[3];
    const float gsum = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    float s = 0.0f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) s += __shfl(gsum, i, 64);
    return sqrtf(s);
}
// general dim (multiple of 4): groups are strided over the lanes, accumulated in group order
__device__ float cv_l2_wave(const float* a, const float* b, int dim, int lane) {
    float s = 0.0f;
    for (int g0 = 0; g0 < dim / 4; g0 += 64) {
        const int gidx = g0 + lane;
        float gsum = 0.0f;
        if (gidx < dim / 4) {
            const f32x4 av = *(const f32x4*)(a + gidx * 4), bv = *(const f32x4*)(b + gidx * 4);
            const float v0 = av[0] - bv[0], v1 = av[1] - bv[1], v2 = av[2] - bv[2], v3 = av[3] - bv[3];
            gsum = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
        }
        const int cnt = min(64, dim / 4 - g0);
        for (int i = 0; i < cnt; ++i) s += __shfl(gsum, i, 64);
    }
    return sqrtf(s);
}

// ---- SearchForTriangulation, screened (engine option tri_screen_bf16; launches of several pairs).  Only products above
// threshold = 1 - th^2 / 2 can be a row's or a column's best (Matcher.cc:851-889: both searches start from the threshold with a
// strict >), and between descriptors of different scene points they are rare: the GEMM runs on the bf16 matrix pipe
// (k_bow_gemm_cand<true, true>: split operands, three products) and lists every (row, column) whose product CAN exceed the
// threshold (screened value + rigorous bound), k_tri_exact evaluates the listed products as the oracle's fma chains (k
// ascending from 0) and keeps, per row and per column, the largest one above the threshold -- ties to the smaller index: the
// reference's first maximum -- with a 64-bit atomic maximum on (product bits, ~index), k_tri_resolve does the cross-check.
// The list lives in the pair's partial-maxima scratch (P.St); a pair whose list overflows (degenerate sets: most products
// above the threshold) is flagged in P.qkey[0] and goes through the full f32 path afterwards (whose kernels return at once for
// every other pair).  Same matches, bit for bit.
struct TriCand { int i, j; };
struct TriScreen { unsigned int* count; unsigned long long* rowkey; unsigned long long* colkey; TriCand* list; int cap; };
__host__ __device__ static inline int tri_screen_cap(int max_rows) { return 2 * max_rows * (tri_tiles(max_rows) - 1) - 2; }
__device__ __forceinline__ TriScreen tri_screen(float* St, int max_rows) {     // 16 + 16 max_rows + 8 cap bytes == the scratch of one pair
    TriScreen t;
    t.count = (unsigned int*)St;
    t.rowkey = (unsigned long long*)(St + 4);
    t.colkey = t.rowkey + max_rows;
    t.list = (TriCand*)(t.colkey + max_rows);
    t.cap = tri_screen_cap(max_rows);
    return t;
}
__global__ __launch_bounds__(256) void k_tri_init(const BowPair* __restrict__ pairs, int max_rows) {
    const BowPair P = pairs[blockIdx.z];
    const TriScreen ts = tri_screen(P.St, max_rows);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *ts.count = 0u;
    if (i < max_rows) { ts.rowkey[i] = 0ull; ts.colkey[i] = 0ull; }
}
// one thread per listed product: the oracle's chain, then the row's and the column's running best
__global__ __launch_bounds__(64) void k_tri_exact(const BowPair* __restrict__ pairs, int dim, float thr, int max_rows) {
    const BowPair P = pairs[blockIdx.z];
    const TriScreen ts = tri_screen(P.St, max_rows);
    const unsigned n = *ts.count;
    if (n > (unsigned)ts.cap) return;                           // overflow: the full path takes this pair
    const unsigned c = blockIdx.x * 64 + threadIdx.x;          // (one-wave workgroups: the few hundred waves of a launch spread over all CUs)
    if (c >= n) return;
    const TriCand e = ts.list[c];
    const float* __restrict__ a = P.q + (long long)e.i * dim;
    const float* __restrict__ b = P.t + (long long)e.j * dim;
    float acc = 0.0f;
    for (int k0 = 0; k0 < dim; k0 += 64) {                      // 32 row pieces in flight per lane (a lane walks its own two rows: latency, not bandwidth)
        f32x4 av[16], bv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { av[q] = *(const f32x4*)(a + k0 + 4 * q); bv[q] = *(const f32x4*)(b + k0 + 4 * q); }
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fmaf(av[q][u], bv[q][u], acc);
    }
    if (acc > thr) {                                            // (thr > 0 in every use; guarded at launch: positive floats order like their bits)
        const unsigned long long hi = (unsigned long long)__float_as_uint(acc) << 32;
        atomicMax(&ts.rowkey[e.i], hi | (0xffffffffu - (unsigned)e.j));
        atomicMax(&ts.colkey[e.j], hi | (0xffffffffu - (unsigned)e.i));
    }
}
__global__ __launch_bounds__(256) void k_tri_resolve(const BowPair* __restrict__ pairs, int max_rows, int* __restrict__ stat) {
    const BowPair P = pairs[blockIdx.z];
    const TriScreen ts = tri_screen(P.St, max_rows);
    const bool over = *ts.count > (unsigned)ts.cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.qkey[0] = over ? 1ull : 0ull;                         // read by the full path's kernels
        if (stat) { if (over) atomicAdd(stat, 1); atomicAdd(stat + 1, 1); }      // {pairs that overflowed, pairs}: the engine's statistics
    }
    if (over) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.x * 256 >= P.nq) return;                  // workgroup-uniform
    int m = -1;
    if (i < P.nq) {
        const unsigned long long rk = ts.rowkey[i];
        if (rk) {
            const int bj = (int)(0xffffffffu - (unsigned)rk);
            const unsigned long long ck = ts.colkey[bj];       // (non-zero: (i, bj) itself raised it)
            if ((int)(0xffffffffu - (unsigned)ck) == i) m = bj;
        }
        P.match[i] = m;
    }
    const unsigned long long hit = __ballot(m >= 0);
    if ((threadIdx.x & 63) == 0 && hit) atomicAdd(P.cnt, __popcll(hit));
}

// ---- SearchByBoW without the similarity matrix.  S = Q * T^T is computed tile by tile on the matrix cores (128 queries x
// 128 train rows per workgroup, 64 x 64 per wave, operands staged through LDS exactly as gemm_abt_tile128) and never
// stored: with the TRAIN rows along the MFMA columns a lane holds, for its train row, 32 of the wave tile's 64 query
// scores in registers, so the pre-selection of k_bow_train_pass -- "queries whose lower bound of |t|^2 + |q|^2 - 2 S is
// below the smallest upper bound" -- runs in the epilogue against the tile-local smallest upper bound (a superset of the
// final candidates: the global bound can only be smaller).  Per (train row, 64-query tile, half-wave) BOW_SLOTS candidate
// slots {lower bound, query} go to HBM: 1 KB per train row instead of the 4 KB row of S, written once and read once.
// Slot 0 holds the half tile's smallest lower bound.  More candidates than slots in one half tile (descriptors closer than
// the rounding band, e.g. duplicates) are counted in a byte per half tile; k_bow_candidates then evaluates all 32 queries
// of that half tile exactly -- if its smallest lower bound can compete at all.
struct BowCand { unsigned int lo_bits; int q; };
#define BOW_SLOTS 2        // (round 6: 4 -> 2: half the slot traffic; more than two candidates in one half tile: the half tile is evaluated exactly)
// SPLIT: the screening products on the bf16 matrix pipe.  S only pre-selects (every query inside the rounding band of a train
// row's nearest is re-evaluated exactly), so it need not be the f32 chain: each f32 operand is split into two bf16 pieces
// x = hi + lo + e, |e| <= 2^-18 |x| (both round-to-nearest-even), and q.t ~ qh.th + qh.tl + ql.th on v_mfma_f32_32x32x16_bf16 --
// three instructions of 32 cycles per 16 k where the f32 form needs eight of 64 (the f32 MFMA IS the fp32 vector pipe; the bf16
// matrix pipe is otherwise idle in this library and runs beside the vector work of other waves).  The bf16 x bf16 products are
// exact in fp32; what is lost is  ql.tl + e-terms <= 3 * 2^-18 |q||t|  and the fp32 accumulation of 3 dim exact products in
// the unit's own order (<= 2 * 3 dim * 2^-24 |q||t| even if its additions are only faithful): launch_bow_pairs widens the band.
// The rows are split once per pair by k_bow_prep (which reads them anyway for the norms) into 16-byte pieces {hi x 4, lo x 4}
// at the f32 pieces' addresses: this kernel's staging is the f32 form's, minus the k permutation.
// TRI: the same GEMM as the screening pass of SearchForTriangulation (k_tri_* below): instead of candidate slots per train row
// the epilogue appends every (row, column) whose product can exceed the similarity threshold `thr` to the pair's list.
template <bool SPLIT, bool TRI = false>
__global__ __launch_bounds__(256, 2) void k_bow_gemm_cand(const BowPair* __restrict__ pairs, int dim, float band, BowCand* __restrict__ cand,
                                                       unsigned char* __restrict__ overflow, int max_rows, int n_qt, int ct, const bf16x8* __restrict__ split,
                                                       int a_gx, int a_gy, int a_n_pairs, float thr) {
    // A workgroup owns 128 queries and `ct` consecutive 128-row tiles of train rows: the k chunks of all its tiles are ONE
    // software pipeline (the first chunk of the next tile is in flight during the last MFMAs and the epilogue of this one), so
    // the load latency at the start of a tile -- 18 % of a 1000 x 1000 x 256 pair when every tile was its own workgroup -- is
    // paid once per workgroup.  ct = 1 for launches that would not fill the chip otherwise (single pairs).
    // 1-D grid of n_pairs * gx * gy workgroups.  Workgroup b runs on XCD b % 8 (observed; speed only): the workgroups of one
    // pair all go to ONE XCD, so that the re-reads of its rows (every 128-row block is read by all tiles of its row / column)
    // meet in that XCD's L2 -- spread over the eight L2s the split form ran at 7 TB/s of L2 misses and was bound by them
    const int gx = a_gx, gy = a_gy, per_pair = gx * gy, n_pairs = a_n_pairs;
    int pair, rest;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        const int full = n_pairs >> 3;                          // rounds of eight pairs, one per XCD
        if (slot < full * per_pair) { pair = (slot / per_pair) * 8 + xcd; rest = slot - (slot / per_pair) * per_pair; }
        else {                                                  // the last n_pairs % 8 pairs: their workgroups in plain order
            const int q = b - full * per_pair * 8;
            pair = full * 8 + q / per_pair; rest = q - (q / per_pair) * per_pair;
        }
    }
    if (pair >= n_pairs) return;
    pair = __builtin_amdgcn_readfirstlane(pair); rest = __builtin_amdgcn_readfirstlane(rest);      // (uniform by construction: say so)
    const int bx = rest % gx, by = rest / gx;
    const BowPair P = pairs[pair];
    // (split form: the pair's two row blocks of the split array -- same piece offsets as the f32 rows)
    const float* __restrict__ d1 = SPLIT ? (const float*)(split + (long long)(pair * 2) * max_rows * (dim >> 2)) : P.q;
    const float* __restrict__ d2 = SPLIT ? (const float*)(split + (long long)(pair * 2 + 1) * max_rows * (dim >> 2)) : P.t;
    const int n1 = P.nq, n2 = P.nt;
    constexpr int LD = 68;                                    // f32 form: floats per LDS row (see k_tri_gemm_argmax)
    constexpr int LH = 72;                                    // split form: bf16 per LDS row (64 k + 16 bytes: 16-byte reads of 16 consecutive rows cover all banks once)
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SPLIT ? 4 * 128 * LH * 2 : 2 * 128 * LD * 4];
    float* As = (float*)lds_raw; float* Bs = As + 128 * LD;                                        // f32 form
    __bf16* Ah = (__bf16*)lds_raw; __bf16* Al = Ah + 128 * LH; __bf16* Bh = Al + 128 * LH; __bf16* Bl = Bh + 128 * LH;   // split form
    __shared__ float qns[128];                                // |q|^2 of the workgroup's queries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    const int row0 = by * 128, ctile0 = bx * ct;
    if (row0 >= n1 || ctile0 * 128 >= n2) return;             // workgroup-uniform
    const int n_ct = min(ct, (n2 - ctile0 * 128 + 127) >> 7); // column tiles of this workgroup that exist
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    const int lc = tid & 15, lrow = (tid >> 4) * 8;
    unsigned aoff[8];                                         // byte offsets of this thread's eight row pieces (sets stay below 4 GB: launch check)
#pragma unroll
    for (int j = 0; j < 8; ++j) aoff[j] = ((unsigned)min(row0 + lrow + j, n1 - 1) * (unsigned)dim + (unsigned)lc * 4u) * 4u;
    if (tid < 128) qns[tid] = row0 + tid < n1 ? P.qn[row0 + tid] : 0.0f;      // (read after the first barrier of the chunk loop)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    f32x4 sa[8], sb[8];
    const int KC = dim >> 6, n_chunks = n_ct * KC;            // chunks of 64 k per tile / of the workgroup
    auto fetch = [&](int c) {                                 // chunk c of the workgroup's pipeline -> staging registers
        const int tile = c / KC, kc = c - tile * KC;          // uniform
        const unsigned kb = (unsigned)kc * 256u;
        const gbase_t pa = sgpr_base(d1, kb), pb = sgpr_base(d2, kb);
        const int c0 = (ctile0 + tile) * 128 + lrow;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sa[j] = *(gvec4_t)(pa + fresh(aoff[j]));
            sb[j] = *(gvec4_t)(pb + ((unsigned)min(c0 + j, n2 - 1) * (unsigned)dim + (unsigned)lc * 4u) * 4u);
        }
    };
    fetch(0);
    const long long pair_rows = (long long)pair * max_rows;
    const int qt = (row0 + wr) >> 6;                          // 64-query tile index of this wave
    for (int c = 0; c < n_chunks; ++c) {
        __syncthreads();                                     // previous chunk fully consumed
        if constexpr (SPLIT) {
            // rows keep their k order: lane (r, half) of a 32x32x16 MFMA reads k = 8 half .. 8 half + 7 of a step as one 16-byte piece
#pragma unroll
            for (int j = 0; j < 8; ++j) {                     // a staged piece is {hi x 4, lo x 4}
                *(float2*)(Ah + (lrow + j) * LH + lc * 4) = float2{sa[j][0], sa[j][1]}; *(float2*)(Al + (lrow + j) * LH + lc * 4) = float2{sa[j][2], sa[j][3]};
                *(float2*)(Bh + (lrow + j) * LH + lc * 4) = float2{sb[j][0], sb[j][1]}; *(float2*)(Bl + (lrow + j) * LH + lc * 4) = float2{sb[j][2], sb[j][3]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float* ap = As + (lrow + j) * LD + lc * 2;
                float* bp = Bs + (lrow + j) * LD + lc * 2;
                *(float2*)(ap) = float2{sa[j][0], sa[j][2]}; *(float2*)(ap + 32) = float2{sa[j][1], sa[j][3]};
                *(float2*)(bp) = float2{sb[j][0], sb[j][2]}; *(float2*)(bp + 32) = float2{sb[j][1], sb[j][3]};
            }
        }
        __syncthreads();
        // the next chunk's pieces, unconditionally (the last pass re-reads its own chunk: a branch around the loads makes the
        // compiler wait for them and copy them right here, in front of the MFMAs they are meant to hide behind)
        fetch(min(c + 1, n_chunks - 1));
        __builtin_amdgcn_sched_barrier(0);                    // (left alone the scheduler sinks them below the MFMAs: nobody needs them before the next pass)
        if constexpr (SPLIT) {
            const int ao = (wr + r) * LH + half * 8, bo = (wc + r) * LH + half * 8;
#pragma unroll
            for (int m = 0; m < 4; ++m) {                     // four steps of 16 k
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *(const bf16x8*)(Ah + ao + i * 32 * LH + 16 * m); al[i] = *(const bf16x8*)(Al + ao + i * 32 * LH + 16 * m);
                    bh[i] = *(const bf16x8*)(Bh + bo + i * 32 * LH + 16 * m); bl[i] = *(const bf16x8*)(Bl + bo + i * 32 * LH + 16 * m);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const float* ap = As + (wr + r) * LD + half * 32;
            const float* bp = Bs + (wc + r) * LD + half * 32;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const f32x4 a0 = *(const f32x4*)(ap + 4 * m), a1 = *(const f32x4*)(ap + 32 * LD + 4 * m);
                const f32x4 b0 = *(const f32x4*)(bp + 4 * m), b1 = *(const f32x4*)(bp + 32 * LD + 4 * m);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
                }
            }
        }
        const int tile = c / KC;
        if (c - tile * KC != KC - 1) continue;                // (uniform) the tile's last chunk: bounds and candidates per train column
        const int col0 = (ctile0 + tile) * 128;
        if constexpr (TRI) {
            const TriScreen ts = tri_screen(P.St, max_rows);
            // (a pair whose list has overflowed is the full path's: stop counting -- on degenerate sets, where nearly every product
            //  is above the threshold, a million atomics on one counter took 25 ms)
            const bool full = __builtin_nontemporal_load(ts.count) > (unsigned)ts.cap;     // wave-uniform
            if (row0 + wr < n1 && !full) {                    // (wave-uniform)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tj = col0 + wc + j * 32 + r;
                    const float tnj = P.tn[min(tj, n2 - 1)];
                    // hits of this lane's column among the wave tile's 64 rows, as bits (i * 16 + reg): a compare and an OR per product
                    unsigned hits = 0;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int rr = wr + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                            const float ub = fmaf(band, qns[rr] + tnj, acc[i][j][reg]);      // upper bound of the exact product
                            hits |= (ub > thr && row0 + rr < n1 ? 1u : 0u) << (i * 16 + reg);
                        }
                    if (tj >= n2) hits = 0;
                    if (__any(hits != 0)) {                    // (wave-uniform; a handful of hits per wave tile: one atomic per wave and column block)
                        const unsigned n = (unsigned)__popc(hits);
                        unsigned incl = n;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            const unsigned up = (unsigned)__shfl_up((int)incl, d, 64);
                            if (lane >= d) incl += up;
                        }
                        unsigned base = 0;
                        if (lane == 63) base = atomicAdd(ts.count, incl);
                        base = (unsigned)__shfl((int)base, 63, 64) + incl - n;
                        while (hits) {
                            const int b = __builtin_ctz(hits);
                            hits &= hits - 1;
                            const int rr = wr + (b >> 4) * 32 + (b & 3) + 8 * ((b & 15) >> 2) + 4 * half;
                            if (base < (unsigned)ts.cap) ts.list[base] = TriCand{row0 + rr, tj};
                            ++base;
                        }
                    }
                }
            }
        } else
        if (row0 + wr < n1) {                                 // (wave-uniform) a tile past the last query has no slots
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tj = col0 + wc + j * 32 + r;
                const float tnj = P.tn[min(tj, n2 - 1)];
                float lo[2][16];
                float hmin = FLT_MAX;
                if (row0 + 128 <= n1) {                        // (workgroup-uniform) every query row of the tile exists: no selects
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int rr = wr + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                            const float nn = qns[rr] + tnj;
                            const float dd = fmaf(-2.0f, acc[i][j][reg], nn);
                            lo[i][reg] = dd - band * nn;
                            hmin = fminf(hmin, dd + band * nn);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int rr = wr + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;      // query row inside the workgroup tile
                            const bool ok = row0 + rr < n1;
                            const float nn = qns[rr] + tnj;
                            const float dd = fmaf(-2.0f, acc[i][j][reg], nn);
                            lo[i][reg] = ok ? dd - band * nn : __builtin_inff();      // (+inf never passes the <= test below)
                            hmin = fminf(hmin, ok ? dd + band * nn : FLT_MAX);
                        }
                }
                hmin = fminf(hmin, __shfl_xor(hmin, 32, 64));          // over the wave tile's 64 queries of this train column
                // this lane's smallest lower bound first (slot 0), then the other candidates in register order
                float lmin = __builtin_inff();
                int imin = -1;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if (lo[i][reg] < lmin) { lmin = lo[i][reg]; imin = i * 16 + reg; }
                BowCand cs[BOW_SLOTS];
#pragma unroll
                for (int k = 0; k < BOW_SLOTS; ++k) cs[k] = BowCand{0x7f800000u, -1};
                int count = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) count += lo[i][reg] <= hmin ? 1 : 0;
                if (imin >= 0 && lmin <= hmin)
                    cs[0] = BowCand{__float_as_uint(lmin), row0 + wr + (imin >> 4) * 32 + (imin & 3) + 8 * ((imin & 15) >> 2) + 4 * half};
                // more than the nearest query inside the band is rare (near-ties, duplicates): only then the other candidates are
                // put into slots 1.. in register order (a wave-uniform branch around 32 x 12 vector instructions)
                if (__any(count > 1)) {
                    int filled = count > 0 ? 1 : 0;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            if (lo[i][reg] <= hmin && i * 16 + reg != imin) {
                                const int qi = row0 + wr + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
#pragma unroll
                                for (int k = 1; k < BOW_SLOTS; ++k)
                                    if (filled == k) cs[k] = BowCand{__float_as_uint(lo[i][reg]), qi};
                                ++filled;
                            }
                        }
                }
                if (tj < n2) {
                    const long long hs = ((pair_rows + tj) * n_qt + qt) * 2 + half;                  // half-tile index
                    BowCand* dst = cand + hs * BOW_SLOTS;
#pragma unroll
                    for (int k = 0; k < BOW_SLOTS; ++k) dst[k] = cs[k];
                    overflow[hs] = (unsigned char)min(count, 255);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }
}

// SearchForTriangulation for all pairs.  split != null (engine option tri_screen_bf16, launches of several pairs): the screened path
// described at k_tri_init, then the full f32 path for the pairs it flagged; otherwise the full path for every pair.
hipError_t launch_tri_pairs(const BowPair* pairs, int n_pairs, int max_rows, int dim, float threshold, hipStream_t s, void* split_scratch, int* stat) {
    if (n_pairs <= 0 || max_rows <= 0) return hipSuccess;
    if (dim % 64 || (long long)max_rows * dim * 4 >= (1ll << 32)) return hipErrorInvalidValue;   // (32-bit lane offsets inside a descriptor set)
    const int t128 = (max_rows + 127) / 128;
    const bool screened = split_scratch && n_pairs >= 4 && tri_tiles(max_rows) >= 2 && threshold > 0.0f;
    if (screened) {
        // |S~ - S| for the exact chain S: the split loses 1.5 * 2^-18 (|a|^2 + |b|^2), the unit's accumulation of 3 dim exact
        // products at most 3 dim u (|a|^2 + |b|^2) (faithful additions), the chain itself 0.5 dim u (...): 5.9e-5 for dim 256;
        // the bound used is twice that
        const float band = 5e-7f * (float)dim;
        bf16x8* split = (bf16x8*)split_scratch;
        hipLaunchKernelGGL(k_bow_prep, dim3((2 * max_rows + 3) / 4, 1, n_pairs), dim3(256), 0, s, pairs, dim, split, max_rows);
        hipLaunchKernelGGL(k_tri_init, dim3((max_rows + 255) / 256, 1, n_pairs), dim3(256), 0, s, pairs, max_rows);
        int ct = 1;
        while (ct < 4 && ct * 2 <= t128 && (long long)((t128 + 2 * ct - 1) / (2 * ct)) * t128 * n_pairs >= 512) ct *= 2;
        const int gx = (t128 + ct - 1) / ct;
        if ((long long)gx * t128 * n_pairs > 0x7fffffffll) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_bow_gemm_cand<true, true>), dim3((unsigned)(gx * t128 * n_pairs)), dim3(256), 0, s, pairs, dim, band, (BowCand*)nullptr,
                           (unsigned char*)nullptr, max_rows, 0, ct, split, gx, t128, n_pairs, threshold);
        hipLaunchKernelGGL(k_tri_exact, dim3((tri_screen_cap(max_rows) + 63) / 64, 1, n_pairs), dim3(64), 0, s, pairs, dim, threshold, max_rows);
        hipLaunchKernelGGL(k_tri_resolve, dim3((max_rows + 255) / 256, 1, n_pairs), dim3(256), 0, s, pairs, max_rows, stat);
    }
    const int only_flagged = screened ? 1 : 0;
    hipLaunchKernelGGL(k_tri_gemm_argmax, dim3(t128, t128, n_pairs), dim3(256), 0, s, pairs, dim, max_rows, only_flagged);
    hipLaunchKernelGGL(k_tri_cols, dim3((max_rows + 255) / 256, 1, n_pairs), dim3(256), 0, s, pairs, threshold, max_rows, only_flagged);
    hipLaunchKernelGGL(k_tri_rows, dim3((max_rows + 255) / 256, 1, n_pairs), dim3(256), 0, s, pairs, threshold, max_rows, only_flagged);
    return hipGetLastError();
}

// every train row: the candidates of all its query tiles -> exact OpenCV distances -> nearest query (first minimum in
// query order: ties go to the smaller index) -> the per-query key, as k_bow_train_pass.  One wave per train row.
__global__ __launch_bounds__(256) void k_bow_candidates(const BowPair* __restrict__ pairs, int dim, float band, const BowCand* __restrict__ cand,
                                                        const unsigned char* __restrict__ overflow, int max_rows, int n_qt, int* __restrict__ stat) {
    const BowPair P = pairs[blockIdx.z];
    const float* __restrict__ q = P.q; const float* __restrict__ t = P.t;
    const int nq = P.nq, nt = P.nt;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= nt || nq <= 0) return;
    const int lane = threadIdx.x & 63;
    const long long prow = (long long)blockIdx.z * max_rows + j;
    const float* trow = t + (long long)j * dim;
    const float tnj = P.tn[j];
    float bd = FLT_MAX;
    int bi = 0x7fffffff;
    int evals = 0;                                            // (uniform) exact evaluations of this train row: the screen's efficiency (stat)
    auto exact = [&](int qi) {
        ++evals;
        const float d = (dim == 256) ? cv_l2_wave256(trow, q + (long long)qi * dim, lane) : cv_l2_wave(trow, q + (long long)qi * dim, dim, lane);
        if (d < bd || (d == bd && qi < bi)) { bd = d; bi = qi; }
    };
    const int live_halves = ((nq + 63) >> 6) * 2, n_live = live_halves * BOW_SLOTS;      // half tiles / slots this pair has
    const BowCand* __restrict__ slots = cand + prow * n_qt * 2 * BOW_SLOTS;
    const unsigned char* __restrict__ counts = overflow + prow * n_qt * 2;
    // smallest upper bound over all half tiles, from their smallest lower bounds (slot 0): hi = lo + 2 band (|q|^2 + |t|^2),
    // taken generously (x 2.5: rounding of the reconstruction must never shrink the candidate set; a larger bound only adds
    // exact evaluations)
    float umin = FLT_MAX;
    for (int h0 = 0; h0 < live_halves; h0 += 64) {
        const int hh = h0 + lane;
        if (hh < live_halves) {
            const BowCand c = slots[hh * BOW_SLOTS];
            if (c.q >= 0) umin = fminf(umin, __uint_as_float(c.lo_bits) + 2.5f * band * (P.qn[c.q] + tnj));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) umin = fminf(umin, __shfl_xor(umin, off, 64));
    // half tiles with more candidates than slots whose best lower bound can compete: all their 32 queries, exactly
    for (int h0 = 0; h0 < live_halves; h0 += 64) {
        const int hh = h0 + lane;
        bool over = false;
        if (hh < live_halves) over = counts[hh] > BOW_SLOTS && __uint_as_float(slots[hh * BOW_SLOTS].lo_bits) <= umin;
        unsigned long long mask = __ballot(over);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int hb = h0 + b, q0 = (hb >> 1) * 64 + (hb & 1) * 4;            // rows of a half: 4 h + {0..3} + 8 k
            for (int k = 0; k < 8; ++k)
                for (int e = 0; e < 4; ++e) { const int qi = q0 + 8 * k + e; if (qi < nq) exact(qi); }
        }
    }
    for (int s0 = 0; s0 < n_live; s0 += 64) {
        const int sl = s0 + lane;
        BowCand c = {0x7f800000u, -1};
        if (sl < n_live) c = slots[sl];
        unsigned long long mask = __ballot(c.q >= 0 && __uint_as_float(c.lo_bits) <= umin);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            exact(__shfl(c.q, b, 64));
        }
    }
    if (lane == 0 && bi != 0x7fffffff)
        atomicMin(&P.qkey[bi], ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned int)j);
    if (stat && lane == 0 && evals) atomicAdd(stat, evals);
}

// dim == 256: the same selection with the exact distances evaluated in batches.  cv_l2_wave256 ends in a 64-step serial sum
// (OpenCV's accumulation order over the 64 groups of four) that one wave pays per candidate; here a wave owns BOWC_ROWS
// consecutive train rows, every candidate's 64 group sums go into a wave-private LDS row (coalesced row loads, no waiting
// between candidates), and once BOWC_EVALS of them are pending lane e adds up candidate e's row in the same order -- 64
// serial sums side by side.  Per train row the nearest (distance, then query index) is kept by an LDS atomic minimum on the
// packed key.  Same expressions, same order, same bits as k_bow_candidates.
#define BOWC_ROWS 8
#define BOWC_EVALS 32
__global__ __launch_bounds__(256) void k_bow_candidates256(const BowPair* __restrict__ pairs, float band, const BowCand* __restrict__ cand,
                                                           const unsigned char* __restrict__ overflow, int max_rows, int n_qt, int rows_per_wave,
                                                           int* __restrict__ stat) {
    constexpr int dim = 256, GP = 65;
    __shared__ float gs_all[4][BOWC_EVALS * GP];
    __shared__ int meta_q_all[4][BOWC_EVALS], meta_r_all[4][BOWC_EVALS];
    __shared__ unsigned long long rowkey_all[4][BOWC_ROWS];
    const BowPair P = pairs[blockIdx.z];
    const float* __restrict__ q = P.q; const float* __restrict__ t = P.t;
    const int nq = P.nq, nt = P.nt;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j0 = (blockIdx.x * 4 + wave) * rows_per_wave;   // (rows_per_wave <= BOWC_ROWS: fewer for launches of few pairs, which need the waves)
    if (j0 >= nt || nq <= 0) return;                          // (wave-uniform; the LDS below is wave-private: no workgroup barrier anywhere)
    float* gs = gs_all[wave];
    int* meta_q = meta_q_all[wave]; int* meta_r = meta_r_all[wave];
    unsigned long long* rowkey = rowkey_all[wave];
    if (lane < BOWC_ROWS) rowkey[lane] = ~0ull;
    int E = 0;                                                // pending evaluations (uniform)
    int evals = 0;                                            // (uniform) exact evaluations of this wave's train rows: the screen's efficiency (stat)
    auto flush = [&]() {
        evals += E;
        asm volatile("" ::: "memory");                        // (LDS operations of one wave execute in order)
        float s = 0.0f;
        const float* gp = gs + min(lane, BOWC_EVALS - 1) * GP;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) s += gp[i];
        if (lane < E) {
            const float d = sqrtf(s);
            atomicMin(&rowkey[meta_r[lane]], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)meta_q[lane]);
        }
        asm volatile("" ::: "memory");
        E = 0;
    };
    const int live_halves = ((nq + 63) >> 6) * 2, n_live = live_halves * BOW_SLOTS;
    const int rows_here = min(rows_per_wave, nt - j0);
    for (int rl = 0; rl < rows_here; ++rl) {
        const int j = j0 + rl;
        const long long prow = (long long)blockIdx.z * max_rows + j;
        const f32x4 tv = *(const f32x4*)(t + (long long)j * dim + lane * 4);
        const float tnj = P.tn[j];
        auto enqueue = [&](int qi) {
            const f32x4 bv = *(const f32x4*)(q + (long long)qi * dim + lane * 4);
            const float v0 = tv[0] - bv[0], v1 = tv[1] - bv[1], v2 = tv[2] - bv[2], v3 = tv[3] - bv[3];
            gs[E * GP + lane] = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
            if (lane == 0) { meta_q[E] = qi; meta_r[E] = rl; }
            if (++E == BOWC_EVALS) flush();
        };
        const BowCand* __restrict__ slots = cand + prow * n_qt * 2 * BOW_SLOTS;
        const unsigned char* __restrict__ counts = overflow + prow * n_qt * 2;
        float umin = FLT_MAX;                                 // (as k_bow_candidates)
        for (int h0 = 0; h0 < live_halves; h0 += 64) {
            const int hh = h0 + lane;
            if (hh < live_halves) {
                const BowCand c = slots[hh * BOW_SLOTS];
                if (c.q >= 0) umin = fminf(umin, __uint_as_float(c.lo_bits) + 2.5f * band * (P.qn[c.q] + tnj));
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) umin = fminf(umin, __shfl_xor(umin, off, 64));
        for (int h0 = 0; h0 < live_halves; h0 += 64) {
            const int hh = h0 + lane;
            bool over = false;
            if (hh < live_halves) over = counts[hh] > BOW_SLOTS && __uint_as_float(slots[hh * BOW_SLOTS].lo_bits) <= umin;
            unsigned long long mask = __ballot(over);
            while (mask) {
                const int b = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int hb = h0 + b, q0 = (hb >> 1) * 64 + (hb & 1) * 4;
                for (int k = 0; k < 8; ++k)
                    for (int e = 0; e < 4; ++e) { const int qi = q0 + 8 * k + e; if (qi < nq) enqueue(qi); }
            }
        }
        for (int s0 = 0; s0 < n_live; s0 += 64) {
            const int sl = s0 + lane;
            BowCand c = {0x7f800000u, -1};
            if (sl < n_live) c = slots[sl];
            unsigned long long mask = __ballot(c.q >= 0 && __uint_as_float(c.lo_bits) <= umin);
            while (mask) {
                const int b = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                enqueue(__shfl(c.q, b, 64));
            }
        }
    }
    if (E > 0) flush();
    asm volatile("" ::: "memory");
    if (lane < rows_here) {
        const unsigned long long k = rowkey[lane];
        if (k != ~0ull) atomicMin(&P.qkey[(unsigned int)k], (k & 0xffffffff00000000ull) | (unsigned int)(j0 + lane));
    }
    if (stat && lane == 0 && evals) atomicAdd(stat, evals);
}

__global__ __launch_bounds__(256) void k_bow_finalize(const BowPair* __restrict__ pairs, float th_low) {
    const BowPair P = pairs[blockIdx.z];
    const unsigned long long* __restrict__ qkey = P.qkey;
    const int nq = P.nq;
    int32_t* __restrict__ match_q2t = P.match; float* __restrict__ dist = P.dist; int* __restrict__ n_matches = P.cnt;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.x * 256 >= nq) return;                     // whole workgroup beyond this pair's queries
    int m = -1;
    float d = FLT_MAX;
    if (i < nq) {
        const unsigned long long k = qkey[i];
        if (k != ~0ull) {
            d = __uint_as_float((unsigned int)(k >> 32));
            if (d < th_low) m = (int)(unsigned int)k;
        }
        match_q2t[i] = m;
        dist[i] = d;
    }
    // one atomic per wave: the counters of neighbouring pairs share a cache line
    const int cnt = __popcll(__ballot(m >= 0));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_matches, cnt);
}

// fills the per-pair descriptors on the device: row counts may live in device memory (on_device callers never
// bring them to the host), scratch is sliced per pair
__global__ void k_bow_setup(BowPair* __restrict__ pairs, int n_pairs, const float* __restrict__ base, long long set_stride,
                            const int* __restrict__ n_rows, const int* __restrict__ qset, const int* __restrict__ tset, int max_rows, float* St,
                            long long st_stride, float* qn, float* tn, unsigned long long* qkey, int32_t* match, float* dist, int* cnt, long long out_stride) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n_pairs) return;
    BowPair P;
    const int qs = qset[p], ts = tset[p];
    P.q = base + (long long)qs * set_stride; P.t = base + (long long)ts * set_stride;
    P.nq = min(max(n_rows[qs], 0), max_rows); P.nt = min(max(n_rows[ts], 0), max_rows);
    P.St = St + (long long)p * st_stride;
    P.qn = qn + (long long)p * max_rows; P.tn = tn + (long long)p * max_rows; P.qkey = qkey + (long long)p * max_rows;
    P.match = match + (long long)p * out_stride; P.dist = dist + (long long)p * out_stride; P.cnt = cnt + p;
    pairs[p] = P;
}

hipError_t launch_bow_setup(BowPair* pairs, int n_pairs, const float* base, long long set_stride, const int* n_rows, const int* qset,
                            const int* tset, int max_rows, float* St, long long st_stride, float* qn, float* tn, unsigned long long* qkey, int32_t* match,
                            float* dist, int* cnt, long long out_stride, hipStream_t s) {
    hipLaunchKernelGGL(k_bow_setup, dim3((n_pairs + 63) / 64), dim3(64), 0, s, pairs, n_pairs, base, set_stride, n_rows, qset, tset, max_rows, St, st_stride, qn,
                       tn, qkey, match, dist, cnt, out_stride);
    return hipGetLastError();
}

// =========================================================================== row-filtered store matching
// Matcher.cc gathers the rows of a keyframe with (SearchByBoW, :231-246) or without (SearchForTriangulation,
// :808-834) a MapPoint before the brute-force step.  The store keeps one flag byte per row; a filtered side of a
// pair is compacted on the device (order preserved, like the CPU gather), matched as usual and mapped back to
// the original row numbers.  sel >= 0: store slot, unfiltered; sel < 0: compacted set ~sel.
__global__ __launch_bounds__(256) void k_store_compact_map(const unsigned char* __restrict__ flags, const int* __restrict__ store_rows,
                                                           const int* __restrict__ c_slot, const int* __restrict__ c_filter, int max_rows,
                                                           int* __restrict__ map, int* __restrict__ inv, int* __restrict__ c_rows) {
    __shared__ int wave_tot[4];
    const int c = blockIdx.x, slot = c_slot[c];
    const bool want = c_filter[c] == 1;
    const int rows = min(max(store_rows[slot], 0), max_rows);
    const unsigned char* __restrict__ f = flags + (long long)slot * max_rows;
    int* __restrict__ mp = map + (long long)c * max_rows;
    int* __restrict__ iv = inv + (long long)c * max_rows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int running = 0;
    for (int r0 = 0; r0 < max_rows; r0 += 256) {
        const int r = r0 + threadIdx.x;
        const bool keep = r < rows && ((f[min(r, max_rows - 1)] != 0) == want);
        const unsigned long long b = __ballot(keep);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(b);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
        if (r < max_rows) iv[r] = keep ? off + before : -1;
        if (keep) mp[off + before] = r;
        running += total;
    }
    if (threadIdx.x == 0) c_rows[c] = running;
}

__global__ __launch_bounds__(256) void k_store_compact_copy(const float* __restrict__ base, long long set_stride, const int* __restrict__ c_slot,
                                                            const int* __restrict__ map, const int* __restrict__ c_rows, int max_rows, int dim,
                                                            float* __restrict__ comp) {
    const int c = blockIdx.y, k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= c_rows[c]) return;
    const int r = map[(long long)c * max_rows + k];
    const float4* __restrict__ src = (const float4*)(base + (long long)c_slot[c] * set_stride + (long long)r * dim);
    float4* __restrict__ dst = (float4*)(comp + (long long)c * set_stride + (long long)k * dim);
    for (int i = lane; i < dim / 4; i += 64) dst[i] = src[i];
}

__global__ void k_store_setup(BowPair* __restrict__ pairs, int n_pairs, const float* __restrict__ base, const float* __restrict__ comp,
                              long long set_stride, const int* __restrict__ store_rows, const int* __restrict__ c_rows,
                              const int* __restrict__ qsel, const int* __restrict__ tsel, int max_rows, float* St, long long st_stride, float* qn,
                              float* tn, unsigned long long* qkey, int32_t* match, float* dist, int* cnt) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n_pairs) return;
    BowPair P;
    const int qs = qsel[p], ts = tsel[p];
    P.q = qs >= 0 ? base + (long long)qs * set_stride : comp + (long long)(~qs) * set_stride;
    P.t = ts >= 0 ? base + (long long)ts * set_stride : comp + (long long)(~ts) * set_stride;
    P.nq = min(max(qs >= 0 ? store_rows[qs] : c_rows[~qs], 0), max_rows);
    P.nt = min(max(ts >= 0 ? store_rows[ts] : c_rows[~ts], 0), max_rows);
    P.St = St + (long long)p * st_stride;
    P.qn = qn + (long long)p * max_rows; P.tn = tn + (long long)p * max_rows; P.qkey = qkey + (long long)p * max_rows;
    P.match = match + (long long)p * max_rows; P.dist = dist + (long long)p * max_rows; P.cnt = cnt + p;
    pairs[p] = P;
}

// compacted results -> original row numbers of both sides
__global__ __launch_bounds__(256) void k_store_remap(int n_pairs, const int* __restrict__ qsel, const int* __restrict__ tsel,
                                                     const int* __restrict__ c_slot, const int* __restrict__ store_rows,
                                                     const int* __restrict__ map, const int* __restrict__ inv, int max_rows,
                                                     const int32_t* __restrict__ c_match, const float* __restrict__ c_dist,
                                                     int32_t* __restrict__ match, float* __restrict__ dist) {
    const int p = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    if (r >= max_rows) return;
    const int qs = qsel[p], ts = tsel[p];
    const int rows = min(max(store_rows[qs >= 0 ? qs : c_slot[~qs]], 0), max_rows);
    int m = -1;
    float d = FLT_MAX;
    if (r < rows) {
        const int c = qs >= 0 ? r : inv[(long long)(~qs) * max_rows + r];
        if (c >= 0) {
            m = c_match[(long long)p * max_rows + c];
            if (c_dist) d = c_dist[(long long)p * max_rows + c];
            if (m >= 0 && ts < 0) m = map[(long long)(~ts) * max_rows + m];
        }
    }
    match[(long long)p * max_rows + r] = m;
    if (dist) dist[(long long)p * max_rows + r] = d;
}

hipError_t launch_store_compact(const float* base, const unsigned char* flags, long long set_stride, const int* store_rows, int n_compact,
                                const int* c_slot, const int* c_filter, int max_rows, int dim, int* map, int* inv, int* c_rows, float* comp,
                                hipStream_t s) {
    if (n_compact <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_store_compact_map, dim3(n_compact), dim3(256), 0, s, flags, store_rows, c_slot, c_filter, max_rows, map, inv, c_rows);
    hipLaunchKernelGGL(k_store_compact_copy, dim3((max_rows + 3) / 4, n_compact), dim3(256), 0, s, base, set_stride, c_slot, map, c_rows, max_rows,
                       dim, comp);
    return hipGetLastError();
}
hipError_t launch_store_setup(BowPair* pairs, int n_pairs, const float* base, const float* comp, long long set_stride, const int* store_rows,
                              const int* c_rows, const int* qsel, const int* tsel, int max_rows, float* St, long long st_stride, float* qn, float* tn,
                              unsigned long long* qkey, int32_t* match, float* dist, int* cnt, hipStream_t s) {
    hipLaunchKernelGGL(k_store_setup, dim3((n_pairs + 63) / 64), dim3(64), 0, s, pairs, n_pairs, base, comp, set_stride, store_rows, c_rows, qsel,
                       tsel, max_rows, St, st_stride, qn, tn, qkey, match, dist, cnt);
    return hipGetLastError();
}
hipError_t launch_store_remap(int n_pairs, const int* qsel, const int* tsel, const int* c_slot, const int* store_rows, const int* map,
                              const int* inv, int max_rows, const int32_t* c_match, const float* c_dist, int32_t* match, float* dist,
                              hipStream_t s) {
    hipLaunchKernelGGL(k_store_remap, dim3((max_rows + 255) / 256, n_pairs), dim3(256), 0, s, n_pairs, qsel, tsel, c_slot, store_rows, map, inv,
                       max_rows, c_match, c_dist, match, dist);
    return hipGetLastError();
}

// ---- SearchByBoW, launches of many pairs, dim == 256: the screening GEMM as a SWEEP (k_bow_sweep256).
// k_bow_gemm_cand stages both operands of every 128 x 128 tile through LDS in 64-k chunks: per 16-k step a wave reads eight 1 KB fragments for
// twelve MFMAs and the workgroup writes what it reads -- the LDS port is as busy as the matrix pipe (both ~100 % on paper; measured: the matrix
// pipe 0.38 busy), the candidate epilogue (~10 vector instructions per product) runs on top, and two barriers frame every 48 MFMAs.
// Here a wave keeps its 64 queries' fragments -- {hi, lo} x 16 k-steps x two 32-row tiles = 256 registers, the matrix-operand half of the register
// file -- for the whole sweep over the train rows, so only the train side moves: 64 train rows x 16 k x {hi, lo} = 4 KB per step, brought
// into a 128 KB LDS ring by LDS-DMA (one 1 KB piece per wave and step, requested six 4-step chunks ahead) and read by all four waves (four
// fragment reads per twelve MFMAs: a third of the LDS port).  One barrier per 48 MFMAs.  Both sets are laid out in fragment order by
// k_bow_prep_frag ([32-row tile][k-step][hi | lo][lane][8]: 1 KB pieces a wave loads / the DMA moves as they are).
// The accumulators start at -|q|^2 / 2 (rows that do not exist: -3e38), so a product IS the ordering key s' = q.t - |q|^2 / 2 of its train
// column: d^2 = |t|^2 - 2 s'.  The epilogue of a column group runs while the NEXT group's MFMAs are in flight (two accumulator sets), and needs
// ~3.5 vector instructions per product: the register index goes into the five low mantissa bits (one v_and_or), the lane's maximum carries it
// along, and "inside the band of the column's best" is one compare against max - W with
//   W = band (max |q|^2 of the wave's queries + |t|^2) + 2^-16 |max|
// -- at least as wide as k_bow_gemm_cand's per-product band (lo_e <= min_f hi_f  =>  s'_e >= s'_max - max_f band (|q_f|^2 + |t|^2)), the second
// term covers the 2^-18 relative error of the cleared bits on both sides.  A superset of candidates only adds exact evaluations.  The slots
// written are k_bow_gemm_cand's ({lower bound d^2 - band (|q|^2 + |t|^2) of the product, query}, slot 0 = the lane's best; a half tile with more
// candidates than slots: slot 0 = {a lower bound of ALL its products, -1} -- k_bow_candidates* then evaluates its 32 queries exactly).
__global__ __launch_bounds__(256) void k_bow_prep_frag(const BowPair* __restrict__ pairs, bf16x8* __restrict__ frag, int tiles32) {
    // one workgroup = one 32-row tile of one set of one pair: rows -> |.|^2 (pre-filter only), {hi, lo} pieces through LDS -> 32 pieces of 1 KB
    constexpr int PP = 1040;                                   // bytes between staged pieces (1 KB + 16: the 16 k-steps a row's lanes write go to different banks)
    __shared__ __attribute__((aligned(16))) unsigned char st[32 * PP];
    const BowPair P = pairs[blockIdx.z];
    const int which = blockIdx.x >= tiles32 ? 1 : 0, tile = blockIdx.x - which * tiles32;
    const float* __restrict__ x = which ? P.t : P.q;
    const int n = which ? P.nt : P.nq;
    float* __restrict__ norm = which ? P.tn : P.qn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0 && tid == 0) *P.cnt = 0;
    if (tile * 32 >= ((n + 63) & ~63)) return;                 // (uniform; the sweep reads whole 64-row groups: their second tile is written as zeros)
    const int step = lane >> 2, khalf = (lane >> 1) & 1, e0 = (lane & 1) * 4;
    for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr, row = tile * 32 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < n) v = *(const f32x4*)(x + (long long)row * 256 + lane * 4);
        float p = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) p = fmaf(v[j], v[j], p);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off, 64);
        if (lane == 0 && row < n) {
            norm[row] = p;
            if (!which) P.qkey[row] = ~0ull;
        }
        bf16x4 h, l;
        split_bf16(v, h, l);
        *(bf16x4*)(st + (step * 2 + 0) * PP + (khalf * 32 + r) * 16 + e0 * 2) = h;
        *(bf16x4*)(st + (step * 2 + 1) * PP + (khalf * 32 + r) * 16 + e0 * 2) = l;
    }
    __syncthreads();
    bf16x8* __restrict__ dst = frag + ((long long)(blockIdx.z * 2 + which) * tiles32 + tile) * (32 * 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j * 256 + tid] = *(const bf16x8*)(st + (j * 4 + (tid >> 6)) * PP + (tid & 63) * 16);
}

#define BOWS_RING_BYTES 131072
#define BOW_SWEEP_MIN_PAIRS 16                              // fewer pairs: k_bow_gemm_cand (a sweep workgroup loads 256 KB of query fragments before its first MFMA)
#define BOWS_MAX_GROUPS 32                                  // column groups (64 train rows) one workgroup sweeps at most: |t|^2 of 2048 rows in LDS
__global__ __launch_bounds__(256, 1) void k_bow_sweep256(const BowPair* __restrict__ pairs, float band, BowCand* __restrict__ cand, unsigned char* __restrict__ overflow,
                                                         int max_rows, int n_qt, const bf16x8* __restrict__ frag, int tiles32, int groups_per_wg, int a_gx, int a_gy,
                                                         int a_n_pairs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // ring (128 KB) | |q|^2 of the 256 queries | |t|^2 of the swept rows | accumulator start values
    float* qns = (float*)(smem + BOWS_RING_BYTES);
    float* tns = qns + 256;
    // workgroup -> (pair, query block of 256, part of the sweep): the workgroups of a pair on ONE XCD (k_bow_gemm_cand: its rows meet in that L2)
    const int gx = a_gx, gy = a_gy, per_pair = gx * gy, n_pairs = a_n_pairs;
    int pair, rest;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        const int full = n_pairs >> 3;
        if (slot < full * per_pair) { pair = (slot / per_pair) * 8 + xcd; rest = slot - (slot / per_pair) * per_pair; }
        else { const int q = b - full * per_pair * 8; pair = full * 8 + q / per_pair; rest = q - (q / per_pair) * per_pair; }
    }
    if (pair >= n_pairs) return;
    pair = __builtin_amdgcn_readfirstlane(pair); rest = __builtin_amdgcn_readfirstlane(rest);
    const int part = rest % gx, qblock = rest / gx;
    const BowPair P = pairs[pair];
    const int n1 = P.nq, n2 = P.nt;
    const int g_begin = part * groups_per_wg, g_end = min((n2 + 63) >> 6, g_begin + groups_per_wg);
    if (qblock * 256 >= n1 || g_begin >= g_end) return;        // (workgroup-uniform)
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0w = qblock * 256 + wave * 64;
    const bool live = q0w < n1;                                // (a wave without queries still moves its share of the train rows)
    const bf16x8* __restrict__ fq = frag + (long long)(pair * 2) * tiles32 * (32 * 64);
    const bf16x8* __restrict__ ft = frag + (long long)(pair * 2 + 1) * tiles32 * (32 * 64);
    qns[tid] = qblock * 256 + tid < n1 ? P.qn[qblock * 256 + tid] : 0.0f;
    for (int i = tid; i < (g_end - g_begin) * 64; i += 256) tns[i] = g_begin * 64 + i < n2 ? P.tn[g_begin * 64 + i] : 0.0f;
    // this wave's queries: fragments of the two 32-row tiles, all 16 k-steps, hi and lo (a wave without queries loads the last tile: never stored).
    // (Loading them by inline assembly straight into the accumulation half of the register file does NOT work: the register allocator gives every
    //  such output the same scratch tuple and copies it out right behind the request, before any wait the source can place.)
    bf16x8 ah[2][16], al[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bf16x8* pc = fq + ((long long)(min((q0w >> 5) + i, tiles32 - 1) * 16 + s) * 2) * 64 + lane;
            ah[i][s] = pc[0];
            al[i][s] = pc[64];
        }
    __syncthreads();
    // accumulator start per (tile, register) -- the same for every column -- and the band's |q|^2
    // (kept in LDS, [wave][half][tile * 16 + register]: 32 registers would not fit beside the fragments)
    float* ini = tns + groups_per_wg * 64 + (wave * 2 + half) * 32;
    float qmax = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rr = wave * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const bool ok = qblock * 256 + rr < n1;
            if (r == 0) ini[i * 16 + reg] = ok ? -0.5f * qns[rr] : -3.0e38f;
            qmax = fmaxf(qmax, qns[rr]);
        }
    qmax = fmaxf(qmax, __shfl_xor(qmax, 32, 64));
    const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned lane16 = (unsigned)lane * 16u;
    // chunk L of the sweep (4 k-steps of one column group): piece `wave` = (tile wave >> 1, hi | lo = wave & 1) of each step -> ring slot L % 4
    const int n_chunks = (g_end - g_begin) * 4;
    auto request = [&](int L) {
        const int slot = L & 7;                                // (of the index asked for: the slot the last barrier has freed)
        L = min(L, n_chunks - 1);                              // (past the end: the last chunk again, into a slot nobody reads any more: the waits stay constant)
        const int g = g_begin + (L >> 2), c = L & 3;
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            const bf16x8* src = ft + ((long long)((g * 2 + (wave >> 1)) * 16 + c * 4 + ss) * 2 + (wave & 1)) * 64;      // uniform
            const unsigned dst = ring + (unsigned)(((slot * 4 + ss) * 4 + wave) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(src), "v"(lane16), "s"(dst) : "memory", "m0");
        }
    };
    const long long pair_rows = (long long)pair * max_rows;
    const int qt = q0w >> 6;
    // ---- the epilogue of one column tile (32 train rows, this lane's row tj), cut into slices of eight products that go between the MFMAs of
    // one k-step each: eight steps per column tile, sixteen per column group -- the length of the next group's MFMA work.
    // mark (steps 0-3 / 8-11): the register index into the low mantissa bits; the lane's largest and second largest (med3 of {largest, second,
    // new} is the new second).  emit (steps 4-7 / 12-15): the band, the count, the slots -- straight-line: a lane's candidates ARE its largest
    // (and second largest) when there are at most two, a half tile with more is flagged (its 32 queries are evaluated exactly if its bound can
    // compete at all).  Among 64 unrelated queries the two best are inside the band for one train row in twenty: a scan of the 32 registers for
    // "the other candidates" would run in most waves.
    struct Epi { float m, m2, tnj, W, thr; int count; };
    auto lower = [&](float u, float tnj) {                     // the product's lower bound of d^2 (k_bow_gemm_cand's, minus the cleared bits' share)
        const int idx = (int)(__float_as_uint(u) & 31u);
        const int rr = wave * 64 + (idx >> 4) * 32 + (idx & 3) + 8 * ((idx & 15) >> 2) + 4 * half;
        const float nn = qns[rr] + tnj;
        return BowCand{__float_as_uint(fmaf(-2.0f, u, tnj) - fmaf(band, nn, 1.52587890625e-5f * fabsf(u))), qblock * 256 + rr};
    };
    auto epi_slice = [&](f32x16 (&acc)[2][2], int s, int g, Epi& E) {      // s: k-step 0..15 of the group whose MFMAs run meanwhile; g: the group of acc
        const int j = s >> 3, ph = (s >> 2) & 1, k = s & 3, i = k >> 1, r0 = (k & 1) * 8;
        if (!ph) {
            if (k == 0) { E.m = -3.4e38f; E.m2 = -3.4e38f; }
#pragma unroll
            for (int reg = r0; reg < r0 + 8; ++reg) {
                const float u = __uint_as_float((__float_as_uint(acc[i][j][reg]) & ~31u) | (unsigned)(i * 16 + reg));
                acc[i][j][reg] = u;
                E.m2 = __builtin_amdgcn_fmed3f(E.m, E.m2, u);
                E.m = fmaxf(E.m, u);
            }
        } else {
            if (k == 0) {
                E.tnj = tns[(g - g_begin) * 64 + j * 32 + r];
                const float hm = fmaxf(E.m, __shfl_xor(E.m, 32, 64));      // the column's best over the wave's 64 queries
                E.W = fmaf(band, qmax + E.tnj, 1.52587890625e-5f * fabsf(hm));
                E.thr = hm - E.W;
                E.count = 0;
            }
#pragma unroll
            for (int reg = r0; reg < r0 + 8; ++reg) E.count += acc[i][j][reg] >= E.thr ? 1 : 0;
            if (k == 3) {
                const int tj = g * 64 + j * 32 + r;
                BowCand c0 = BowCand{0x7f800000u, -1}, c1 = c0;
                if (E.count > BOW_SLOTS) c0 = BowCand{__float_as_uint(fmaf(-2.0f, E.m, E.tnj) - E.W - E.W), -1};      // below every product's bound of this half tile, no query
                else if (E.count > 0) {
                    c0 = lower(E.m, E.tnj);
                    if (E.count > 1) c1 = lower(E.m2, E.tnj);
                }
                if (tj < n2 && live) {
                    const long long hs = ((pair_rows + tj) * n_qt + qt) * 2 + half;
                    BowCand* dst = cand + hs * BOW_SLOTS;
                    dst[0] = c0; dst[1] = c1;
                    overflow[hs] = (unsigned char)min(E.count, 255);
                }
            }
        }
    };
    // the train fragments of k-step s of the group of parity par (the ring holds two groups), one step ahead of the MFMAs that use them
    bf16x8 bq[2][4];
    auto read_b = [&](int par, int s, bf16x8 (&b)[4]) {
#pragma unroll
        for (int x = 0; x < 4; ++x) b[x] = *(const bf16x8*)(smem + (((par * 16 + s) * 4 + x) * 64 + lane) * 16);
    };
    // the barrier that makes chunk L readable stands in front of the LAST k-step of chunk L - 1 (so that the first fragments of L are read one
    // step ahead like all others): behind it every wave has finished chunk L - 2, whose slot takes chunk L + 6.  My pieces of chunk L have landed
    // when at most the five chunks requested after it (20 pieces) are on their way.  (The epilogue's stores count too and may complete before
    // older loads: they can only make this wait longer than needed, never shorter.)
    auto open_chunk = [&](int L) {
        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        __syncthreads();
        request(L + 6);
    };
    // ---- one column group: MFMAs into `cur`, the epilogue of the group before (in `prv`) between them
    auto group = [&](int g, f32x16 (&cur)[2][2], f32x16 (&prv)[2][2], bool have_prev, int par) {      // par: parity of g - g_begin
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = *(const f32x4*)(ini + i * 16 + q4 * 4);      // (written before the first barrier of the sweep)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cur[i][j][q4 * 4 + e] = v[e];
            }
        Epi E;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if ((s & 3) == 3) open_chunk((g - g_begin) * 4 + (s >> 2) + 1);
            if (s < 15) read_b(par, s + 1, bq[(s + 1) & 1]);
            else read_b(par ^ 1, 0, bq[0]);                    // (the next group's first step; after the last group: stale bytes nobody uses)
            const bf16x8 (&b)[4] = bq[s & 1];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][s], b[j * 2], cur[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][s], b[j * 2 + 1], cur[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i][s], b[j * 2], cur[i][j], 0, 0, 0);
            // (a wave without queries runs the epilogue too -- straight-line code -- and stores nothing)
            if (have_prev) epi_slice(prv, s, g - 1, E);
            // the slice's vector instructions BETWEEN the MFMAs (left alone the scheduler appends them, and with one wave per SIMD nobody else
            // fills the gaps); nothing crosses the end of the step
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x16 accA[2][2], accB[2][2];
#pragma unroll
    for (int L = 0; L < 6; ++L) request(L);
    open_chunk(0);
    read_b(0, 0, bq[0]);
    int g = g_begin;
    group(g, accA, accB, false, 0);
    for (++g; g + 1 < g_end; g += 2) {
        group(g, accB, accA, true, 1);
        group(g + 1, accA, accB, true, 0);
    }
    auto tail = [&](f32x16 (&acc)[2][2], int gl) {
        Epi E;
#pragma unroll
        for (int s = 0; s < 16; ++s) epi_slice(acc, s, gl, E);
    };
    if (g < g_end) {
        group(g, accB, accA, true, 1);
        tail(accB, g);
    } else tail(accA, g - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (no LDS-DMA may land after the workgroup has given its LDS back)
}

// all pairs in four launches (prep, GEMM, train pass, finalize); grids are sized for max_rows, workgroups
// beyond a pair's row counts exit at once
static size_t bow_cand_bytes(int n_pairs, int max_rows) {      // candidate slots + counts, rounded up to 16 bytes
    const size_t n_qt = (size_t)(max_rows + 63) / 64;
    return ((size_t)n_pairs * max_rows * n_qt * 2 * (BOW_SLOTS * sizeof(BowCand) + 1) + 15) & ~(size_t)15;
}
size_t tri_split_offset_bytes(int n_pairs, int max_rows) { return ((size_t)n_pairs * tri_scratch_floats(max_rows) * sizeof(float) + 15) & ~(size_t)15; }
size_t tri_scratch_bytes(int n_pairs, int max_rows, int dim) {   // partial maxima / candidate lists + the split rows of the screened path
    return tri_split_offset_bytes(n_pairs, max_rows) + (size_t)n_pairs * 2 * max_rows * dim * sizeof(float);
}
size_t bow_scratch_bytes(int n_pairs, int max_rows, int dim) {  // + the split rows of both sets of every pair (split-bf16 screening; whole 64-row groups: k_bow_prep_frag)
    return bow_cand_bytes(n_pairs, max_rows) + (size_t)n_pairs * 2 * ((max_rows + 63) & ~63) * dim * sizeof(float);
}

hipError_t launch_bow_pairs(const BowPair* pairs, int n_pairs, int max_rows, int dim, float th_low, void* scratch, hipStream_t s, int screen_bf16, int* stat) {
    if (n_pairs <= 0 || max_rows <= 0) return hipSuccess;
    if (dim % 64 || (long long)max_rows * dim * 4 >= (1ll << 32)) return hipErrorInvalidValue;   // (32-bit lane offsets inside a descriptor set)
    // G = |q|^2 + |t|^2 - 2 q.t (norms and the MFMA chain in fp32) against the exactly evaluated form X (OpenCV's order): with
    // u = 2^-24 and g = dim u (the bound of a dim-term fp32 sum),  |G - d^2| <= (2 g + 2 u)(|q|^2 + |t|^2)  (two norms, the dot
    // product bounded by half the norms, two roundings) and  |X - d^2| <= g d^2 <= 2 g (|q|^2 + |t|^2):  together
    // 4.1 dim u (|q|^2 + |t|^2) = 6.2e-5 for dim 256.  band = 5e-7 dim (1.28e-4 for dim 256) is twice that worst case; measured
    // differences are two orders of magnitude smaller.  (Round 1 used 1.1e-3: every query within 8e-3 of a row's nearest was
    // re-evaluated exactly -- three per train row on descriptors of one scene instead of one.)
    // Split-bf16 screening (k_bow_gemm_cand<true>): the dot product additionally loses 3 * 2^-18 |q||t| (dropped lo.lo and
    // the splitting remainders) and is accumulated from 3 dim exact products in the matrix unit's own order: with additions that
    // are at least faithful  |S~ - q.t| <= (1.5 * 2^-18 + 3 dim u)(|q|^2 + |t|^2),  twice that in G, plus the norms and X as
    // above: 1.5e-4 for dim 256.  band = 1.25e-6 dim (3.2e-4) is again twice the worst case (tools/dev/match_band.py, a numpy emulation:
    // the largest difference seen is 100 times smaller); the candidates stay one per train row on descriptors of one scene.
    const float band = (screen_bf16 ? 1.25e-6f : 5e-7f) * (float)dim;
    const int n_qt = (max_rows + 63) / 64;
    BowCand* cand = (BowCand*)scratch;
    unsigned char* overflow = (unsigned char*)scratch + (size_t)n_pairs * max_rows * n_qt * 2 * BOW_SLOTS * sizeof(BowCand);   // candidate counts per half tile
    bf16x8* split = screen_bf16 ? (bf16x8*)((unsigned char*)scratch + bow_cand_bytes(n_pairs, max_rows)) : nullptr;
    const int t64 = (max_rows + 63) / 64;
    if (screen_bf16 && dim == 256 && n_pairs >= BOW_SWEEP_MIN_PAIRS && t64 <= 4 * BOWS_MAX_GROUPS) {
        // the sweep (k_bow_sweep256): query blocks of 256 x parts of the train rows, split only as far as the chip needs workgroups
        const int tiles32 = t64 * 2, gy = (max_rows + 255) / 256;
        int gx = 1;
        while ((long long)n_pairs * gy * gx < 256 && (t64 + 2 * gx - 1) / (2 * gx) >= 4) gx *= 2;
        while ((t64 + gx - 1) / gx > BOWS_MAX_GROUPS) gx *= 2;
        const int gpw = (t64 + gx - 1) / gx;
        if ((long long)gx * gy * n_pairs > 0x7fffffffll) return hipErrorInvalidValue;
        const size_t lds = BOWS_RING_BYTES + (256 + (size_t)gpw * 64 + 256) * sizeof(float);
        static std::once_flag attr_once;                        // > 64 KB of dynamic LDS has to be requested once
        std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)k_bow_sweep256, hipFuncAttributeMaxDynamicSharedMemorySize, BOWS_RING_BYTES + (512 + BOWS_MAX_GROUPS * 64) * 4); });
        hipLaunchKernelGGL(k_bow_prep_frag, dim3(2 * tiles32, 1, n_pairs), dim3(256), 0, s, pairs, split, tiles32);
        hipLaunchKernelGGL(k_bow_sweep256, dim3((unsigned)(gx * gy * n_pairs)), dim3(256), lds, s, pairs, band, cand, overflow, max_rows, n_qt, split, tiles32, gpw, gx, gy, n_pairs);
        int rpw = BOWC_ROWS;
        while (rpw > 1 && (long long)((max_rows + 4 * rpw - 1) / (4 * rpw)) * n_pairs < 512) rpw >>= 1;
        hipLaunchKernelGGL(k_bow_candidates256, dim3((max_rows + 4 * rpw - 1) / (4 * rpw), 1, n_pairs), dim3(256), 0, s, pairs, band, cand, overflow, max_rows, n_qt, rpw, stat);
        hipLaunchKernelGGL(k_bow_finalize, dim3((max_rows + 255) / 256, 1, n_pairs), dim3(256), 0, s, pairs, th_low);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_bow_prep, dim3((2 * max_rows + 3) / 4, 1, n_pairs), dim3(256), 0, s, pairs, dim, split, max_rows);
    // column tiles per workgroup: as many (up to 4) as still leave the launch two workgroups per CU
    const int t128 = (max_rows + 127) / 128;
    int ct = 1;
    while (ct < 4 && ct * 2 <= t128 && (long long)((t128 + 2 * ct - 1) / (2 * ct)) * t128 * n_pairs >= 512) ct *= 2;
    const int gx = (t128 + ct - 1) / ct;
    if ((long long)gx * t128 * n_pairs > 0x7fffffffll) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(gx * t128 * n_pairs));
    if (screen_bf16)
        hipLaunchKernelGGL(k_bow_gemm_cand<true>, grid, dim3(256), 0, s, pairs, dim, band, cand, overflow, max_rows, n_qt, ct, split, gx, t128, n_pairs, 0.0f);
    else
        hipLaunchKernelGGL(k_bow_gemm_cand<false>, grid, dim3(256), 0, s, pairs, dim, band, cand, overflow, max_rows, n_qt, ct, nullptr, gx, t128, n_pairs, 0.0f);
    if (dim == 256) {
        int rpw = BOWC_ROWS;                                  // train rows per wave: 8 when the launch has waves to spare (measured: 32 pairs of 1000 rows 16 / 8 / 4 / 2 -> 30 / 20 / 23 / 25 us)
        while (rpw > 1 && (long long)((max_rows + 4 * rpw - 1) / (4 * rpw)) * n_pairs < 512) rpw >>= 1;
        hipLaunchKernelGGL(k_bow_candidates256, dim3((max_rows + 4 * rpw - 1) / (4 * rpw), 1, n_pairs), dim3(256), 0, s, pairs, band, cand, overflow, max_rows, n_qt, rpw, stat);
    }
    else
        hipLaunchKernelGGL(k_bow_candidates, dim3((max_rows + 3) / 4, 1, n_pairs), dim3(256), 0, s, pairs, dim, band, cand, overflow, max_rows, n_qt, stat);
    hipLaunchKernelGGL(k_bow_finalize, dim3((max_rows + 255) / 256, 1, n_pairs), dim3(256), 0, s, pairs, th_low);
    return hipGetLastError();
}

// =========================================================================== DescriptorDistance
// (des1 - des2).norm() in tree256 order; one wave
__device__ __forceinline__ float tree256_wave4(f32x4 p) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = p[j] + __shfl_xor(p[j], off, 64);
    }
    const float a = p[0] + p[2], b = p[1] + p[3];
    return a + b;
}
__device__ __forceinline__ float sumsq_diff_tree256_wave(const float* a, const float* b, int dim, int lane) {
    f32x4 p = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < dim; k += 256) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = k + lane * 4 + c;
            if (i < dim) { const float d = a[i] - b[i]; p[c] = fmaf(d, d, p[c]); }
        }
    }
    return tree256_wave4(p);
}
__global__ __launch_bounds__(64) void k_descriptor_distance(const float* a, const float* b, int dim, float* out) {
    const float ss = sumsq_diff_tree256_wave(a, b, dim, threadIdx.x);
    if (threadIdx.x == 0) *out = sqrtf(ss);
}
hipError_t launch_descriptor_distance(const float* a, const float* b, int dim, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_descriptor_distance, dim3(1), dim3(64), 0, s, a, b, dim, out);
    return hipGetLastError();
}

// =========================================================================== windowed matchers: the candidate loop
// Matcher.cc:74-110 (SearchByProjection) and its siblings: one wave per query descriptor walks the query's candidate list
// in order; every distance is Matcher::DescriptorDistance in tree256 order (the query row stays in registers), best and
// second best with their pyramid levels follow the reference's strict-< update rule.  The candidate lists (frame grid /
// map geometry) and the threshold / ratio / ownership tests around the loop stay on the CPU side.
__global__ __launch_bounds__(256) void k_match_candidates(const float* __restrict__ query, int nq, const float* __restrict__ train,
                                                          const int* __restrict__ train_level, int dim, const int* __restrict__ cand_offsets,
                                                          const int* __restrict__ cand_index, int* __restrict__ best_idx, float* __restrict__ best_dist,
                                                          int* __restrict__ best_level, float* __restrict__ second_dist, int* __restrict__ second_level) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nq) return;
    const float* qrow = query + (long long)i * dim;
    float bd = FLT_MAX, bd2 = FLT_MAX;
    int bl = -1, bl2 = -1, bi = -1;
    const int c0 = cand_offsets[i], c1 = cand_offsets[i + 1];
    for (int c = c0; c < c1; ++c) {
        const int idx = cand_index[c];
        const float dist = sqrtf(sumsq_diff_tree256_wave(qrow, train + (long long)idx * dim, dim, lane));
        const int level = train_level ? train_level[idx] : 0;
        if (dist < bd) { bd2 = bd; bl2 = bl; bd = dist; bl = level; bi = idx; }
        else if (dist < bd2) { bl2 = level; bd2 = dist; }
    }
    if (lane == 0) { best_idx[i] = bi; best_dist[i] = bd; best_level[i] = bl; second_dist[i] = bd2; second_level[i] = bl2; }
}

hipError_t launch_match_candidates(const float* query, int nq, const float* train, const int* train_level, int dim, const int* cand_offsets,
                                   const int* cand_index, int* best_idx, float* best_dist, int* best_level, float* second_dist, int* second_level,
                                   hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_match_candidates, dim3((nq + 3) / 4), dim3(256), 0, s, query, nq, train, train_level, dim, cand_offsets, cand_index, best_idx,
                       best_dist, best_level, second_dist, second_level);
    return hipGetLastError();
}

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:366-400), one workgroup per map point: all pairwise
// DescriptorDistance of its n <= DISTINCT_MAX observation descriptors into LDS (a wave per pair), then one thread per row
// picks the row's median -- element (int)(0.5 (n - 1)) of the sorted row, found by rank counting -- and thread 0 the first row
// with the smallest median.
#define DISTINCT_MAX 96
__global__ __launch_bounds__(256) void k_distinctive(const float* __restrict__ desc, const int* __restrict__ set_offsets, int dim, int* __restrict__ best) {
    __shared__ float dist[DISTINCT_MAX * DISTINCT_MAX];
    __shared__ float med[DISTINCT_MAX];
    const int s = blockIdx.x, o = set_offsets[s], n = set_offsets[s + 1] - o;
    if (n <= 0) { if (threadIdx.x == 0) best[s] = -1; return; }
    if (n > DISTINCT_MAX) { if (threadIdx.x == 0) best[s] = -2; return; }          // (the host entry point rejects such sets up front)
    const float* d = desc + (long long)o * dim;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave; p < n * n; p += 4) {                                        // p = i * n + j, upper triangle only
        const int i = p / n, j = p - i * n;
        if (j < i) continue;                                                       // wave-uniform
        const float v = j == i ? 0.0f : sqrtf(sumsq_diff_tree256_wave(d + (long long)i * dim, d + (long long)j * dim, dim, lane));
        if (lane == 0) { dist[i * n + j] = v; dist[j * n + i] = v; }
    }
    __syncthreads();
    const int k = (int)(0.5 * (n - 1));
    for (int i = threadIdx.x; i < n; i += 256) {
        // the k-th smallest of row i: the value v with (#smaller) <= k < (#smaller + #equal)
        float m = 0.0f;
        for (int a = 0; a < n; ++a) {
            const float v = dist[i * n + a];
            int less = 0, equal = 0;
            for (int b = 0; b < n; ++b) { const float w = dist[i * n + b]; less += w < v; equal += w == v; }
            if (less <= k && k < less + equal) { m = v; break; }
        }
        med[i] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bm = FLT_MAX;
        int bi = 0;
        for (int i = 0; i < n; ++i) if (med[i] < bm) { bm = med[i]; bi = i; }
        best[s] = bi;
    }
}

int distinctive_max_rows() { return DISTINCT_MAX; }
hipError_t launch_distinctive(const float* desc, const int* set_offsets, int n_sets, int dim, int* best, hipStream_t s) {
    if (n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_distinctive, dim3(n_sets), dim3(256), 0, s, desc, set_offsets, dim, best);
    return hipGetLastError();
}

// =========================================================================== keyframe database scan
// score = max(0, 1 - ||q - d||) for every occupied slot; one wave per slot, 16-byte coalesced loads
// (HBM-bound: dim*4 bytes per slot); the best score is tracked with an atomic max on the float bits.
__global__ __launch_bounds__(256) void k_db_scores(const float* __restrict__ q, const float* __restrict__ db, const unsigned char* __restrict__ occupied,
                                                   int n, int dim, float* __restrict__ scores, unsigned int* __restrict__ best_partial) {
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    float best = 0.0f;
    for (int i = wid; i < n; i += gridDim.x * 4) {
        if (!occupied[i]) { if (lane == 0) scores[i] = -1.0f; continue; }
        const float* d = db + (long long)i * dim + lane * 4;
        const float* qp = q + lane * 4;
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < dim; k0 += 2048) {                       // 8 row loads in flight per lane
            f32x4 dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) dv[u] = k0 + u * 256 < dim ? *(const f32x4*)(d + k0 + u * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u * 256 < dim) {
                    const f32x4 qv = *(const f32x4*)(qp + k0 + u * 256);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float df = qv[c] - dv[u][c]; p[c] = fmaf(df, df, p[c]); }
                }
            }
        }
        const float ss = tree256_wave4(p);
        const float sc = 1 - sqrtf(ss);
        const float score = sc > 0.f ? sc : 0.f;
        if (lane == 0) scores[i] = score;
        best = fmaxf(best, score);
    }
    // no atomics: 10 000 atomic maxima on one word cost more than the scan; the filter kernel reduces the partials
    if (lane == 0) best_partial[wid] = __float_as_uint(best);
}

// Several queries per pass over the database (loop-closure stress, BASELINE config 5): a wave keeps one
// database row in registers (dim <= 4096: 64 floats per lane) and scores it against a tile of DBQ queries staged
// in LDS, so the row crosses HBM once per DBQ queries.  Per (query, row) the arithmetic and its order are those
// of k_db_scores -- the same bits.  The best score of a query is kept per wave and published with one atomic.
#define DBQ 8
__global__ __launch_bounds__(256) void k_db_scores_batch(const float* __restrict__ q, int n_queries, const float* __restrict__ db,
                                                         const unsigned char* __restrict__ occupied, int n, int dim,
                                                         float* __restrict__ scores, unsigned int* __restrict__ best_bits) {
    extern __shared__ __attribute__((aligned(16))) float qs[];         // [DBQ][dim]
    const int q0 = blockIdx.y * DBQ, nq = min(DBQ, n_queries - q0);
    for (int i = threadIdx.x * 4; i < nq * dim; i += 1024) *(f32x4*)(qs + i) = *(const f32x4*)(q + (long long)q0 * dim + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = dim >> 8;                                       // 256 floats per chunk, <= 16
    float best[DBQ];
#pragma unroll
    for (int j = 0; j < DBQ; ++j) best[j] = 0.0f;
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        if (!occupied[i]) {
            if (lane < nq) scores[(long long)(q0 + lane) * n + i] = -1.0f;
            continue;
        }
        const float* d = db + (long long)i * dim + lane * 4;
        f32x4 row[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) row[k] = k < chunks ? *(const f32x4*)(d + k * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < DBQ; ++j) {
            if (j < nq) {
                const float* qp = qs + j * dim + lane * 4;
                f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < chunks) {
                        const f32x4 qv = *(const f32x4*)(qp + k * 256);
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const float df = qv[c] - row[k][c]; p[c] = fmaf(df, df, p[c]); }
                    }
                }
                const float ss = tree256_wave4(p);
                const float sc = 1 - sqrtf(ss);
                const float score = sc > 0.f ? sc : 0.f;
                if (lane == 0) scores[(long long)(q0 + j) * n + i] = score;
                best[j] = fmaxf(best[j], score);
            }
        }
    }
    if (lane == 0) {
        const int wid = blockIdx.x * 4 + wave, parts = gridDim.x * 4;
#pragma unroll
        for (int j = 0; j < DBQ; ++j)
            if (j < nq) best_bits[(long long)(q0 + j) * parts + wid] = __float_as_uint(best[j]);
    }
}

// ---- many queries at once (loop-closure bursts, BASELINE config 5): screen on the integer matrix pipe, decide with the exact chain.
// The place-recognition score is max(0, 1 - ||q - d||) (KeyFrameDatabase.cc:93): EXACTLY 0 for every keyframe at distance >= 1 from
// the query -- for descriptors of different places, nearly all of them.  So the batched query needs the exact chain of k_db_scores
// only for the slots that can be closer than 1, and a crude product is enough to find those.  The crude product is an INTEGER one: every
// vector x is kept as 8-bit steps of its own scale,
//   x_i = s a_i + r_i,   s = max|x| / 127,   a_i = rint(x_i / s) in [-127, 127],   |r_i| <= s / 2 (1 + 1e-4: the fp32 scaling),
// and for a query (s_q, a) and a database row (s_d, b)
//   q.d = s_q s_d sum a_i b_i  +  sum s_q a_i r'_i  +  sum r_i s_d b_i  +  sum r_i r'_i
//   |q.d - s_q s_d sum a_i b_i| <= (s_d / 2) A1 + (s_q / 2) B1 + dim s_q s_d / 4 =: err,   A1 = s_q sum|a_i|,  B1 = s_d sum|b_i|
// (sum|a_i|, sum|b_i| and the scales are stored with the squared norms: k_db_rowstat, k_db_quant).  sum a_i b_i is EXACT in int32 (<= 4096 * 127^2 per 4096 elements;
// descriptor lengths up to 2^17 fit), so nothing depends on the order the matrix unit or the k-parts add in.  With
//   d2~ = |q|^2 + |d|^2 - 2 s_q s_d sum a_i b_i
// a slot with  d2~ >= 1 + 2.002 err + 1e-4 (|q|^2 + |d|^2)  has a true squared distance >= 1 + 5e-5 (|q|^2 + |d|^2) (the last term covers the
// fp32 norms, <= 3e-6 relative in tree256 order, and the three roundings of the scaled product): the exact chain (relative error <= 2e-6) returns a
// distance >= 1 and the score 0 -- which is what is written for it.  Every other occupied slot is re-scored with the exact chain (anything not
// finite fails the comparison and goes there too).  All outputs of hfnet_db_query_batch are therefore the exact scan's bits: scores of EVERY
// slot, best, candidates.  For unit vectors of 4096 roughly Gaussian components: s ~ 5e-4, A1 ~ 51, err ~ 0.027: slots beyond d2 = 1.055 are
// ruled out (the bf16 screen of rounds 4-6a: 1.018, at twice the bytes).
// History, 64 queries against 10 000 keyframes of 4096 (rocprofv3 kernel times): round 2-3's f32 MFMA form (S on v_mfma_f32_32x32x2_f32, two
// exact re-scoring passes) 89 us; the bf16 screen k_db_screen (k_db_gemm's structure: 128 rows x 128 queries x a quarter of k per workgroup,
// both operands through LDS in 64-k chunks, two barriers per chunk) 18.3 us + 7.1 us for the decision kernel -- 24 KB in flight per workgroup,
// a third of every workgroup's fetches the QUERIES again; the same product as a sweep over a bf16 copy in fragment order (the queries' k-part
// resident in LDS, the rows straight into registers) 17.4 us: 3.5 us of launch / prologue / epilogue + the copy's 82 MB at 5.9 TB/s, which is
// what the memory system gives (half the descriptor length: 10.5 us).  So the bytes had to go: one byte per element, 10.5 + 8.8 us.  Then the
// second kernel: a dependent launch costs 4.7 us before its first instruction (k_db_decide with an empty body), more than the decision's own
// 4 us -- the sums have to be complete inside ONE kernel, i.e. a workgroup needs all of k for its rows, i.e. ALL the queries' fragments (256 KB
// for 64 queries; the LDS is 160 KB).  Through LDS in two k-parts, one after the other: 19.7 us (each refill is a barrier and 128 KB from L2
// with the memory pipe running dry behind it).  In REGISTERS -- one wave per SIMD, 512 registers, a wave holds every query's fragments for its
// quarter of k: 15.7 us (k_db_sweep below).  What this form costs is the broadcast: every CU fetches the same 256 KB, 64 MB of L2 reads at
// ~6 TB/s (s_memtime: 5.5 us from a workgroup's start to its first MFMAs; 32 queries: 11.4 us); and the 57 of 313 tiles that are some
// workgroup's second.
// The copy is kept in the matrix unit's FRAGMENT ORDER -- [32-row tile][32-k step][lane = row & 31 | k-half << 5][16 x i8]: the 1 KB a wave's
// 64 lanes hand to one v_mfma_i32_32x32x32_i8 is 1 KB of consecutive memory.  Queries are the M side, database rows the N side of the
// product: a lane ends up with ONE row's sums, 32 consecutive slots of a query per store.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
struct DbRowStat { float norm, scale; int l1; float pad; };   // |x|^2 (tree256 order, as k_sumsq_rows), s, sum|a_i|
// a row's statistics: one wave per row, sixteen 1 KB requests in flight
__global__ __launch_bounds__(256) void k_db_rowstat(const float* __restrict__ x, int n_rows, int dim, DbRowStat* __restrict__ stat) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const float* __restrict__ xr = x + (long long)row * dim;
    f32x4 p = {0.f, 0.f, 0.f, 0.f};
    float m = 0.0f;
#pragma unroll 1
    for (int kseg = 0; kseg < dim; kseg += 4096) {
        const int seg = min(4096, dim - kseg);
        f32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {                         // (no branch around a request: every join would wait for all of them)
            const f32x4 t = *(const f32x4*)(xr + kseg + (j * 256 < seg ? j * 256 : 0) + lane * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[j][c] = j * 256 < seg ? t[c] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)                           // (zeros beyond the row's end change neither the chain's value nor the maximum)
#pragma unroll
            for (int c = 0; c < 4; ++c) { p[c] = fmaf(v[j][c], v[j][c], p[c]); m = fmaxf(m, fabsf(v[j][c])); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const float ss = tree256_wave4(p);
    if (lane == 0) stat[row] = DbRowStat{ss, m / 127.0f, 0, 0.0f};
}
// both of the above for a FEW rows (a call's queries, the tiles of added slots) in one launch -- a dependent launch costs 4.7 us before its first
// instruction: one workgroup per row of whole 32-row tiles (rows >= n_rows: zeros), a thread = one 16-byte piece of the row's steps; maximum,
// |x|^2 and sum|a_i| over the workgroup.  (|x|^2 in another order than k_db_rowstat's: the bound's 1e-4 (|q|^2 + |d|^2) does not care.)
__global__ __launch_bounds__(256) void k_db_prep_rows(const float* __restrict__ x, int n_rows, int dim, DbRowStat* __restrict__ stat, i32x4* __restrict__ frag) {
    __shared__ float s_m[4], s_ss[4];
    __shared__ int s_l1[4];
    const int row = blockIdx.x, tile = row >> 5, r = row & 31, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = dim >> 5, k = tid * 16;                    // (dim <= 4096: one piece per thread)
    const bool mine = k < dim;
    i32x4* __restrict__ dst = frag + ((long long)tile * ks + (k >> 5)) * 64 + ((k >> 4) & 1) * 32 + r;
    if (row >= n_rows) {                                       // (workgroup-uniform)
        if (mine) *dst = i32x4{0, 0, 0, 0};
        return;
    }
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = mine ? *(const f32x4*)(x + (long long)row * dim + k + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    float m = 0.0f, ss = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) { m = fmaxf(m, fabsf(v[j][c])); ss = fmaf(v[j][c], v[j][c], ss); }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { m = fmaxf(m, __shfl_xor(m, off, 64)); ss += __shfl_xor(ss, off, 64); }
    if (lane == 0) { s_m[wave] = m; s_ss[wave] = ss; }
    __syncthreads();
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    ss = (s_ss[0] + s_ss[1]) + (s_ss[2] + s_ss[3]);
    const float sc = m / 127.0f, inv = sc > 0.0f ? 1.0f / sc : 0.0f;      // (k_db_quant's)
    float l1 = 0.0f;
    i32x4 w = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float a = rintf(v[j][c] * inv);              // within [-127, 127]; NaN -> 0 (k_db_quant)
            l1 += fabsf(a);
            w[j] |= (int)(((unsigned)(int)a & 255u) << (8 * c));
        }
    if (mine) *dst = w;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) l1 += __shfl_xor(l1, off, 64);
    if (lane == 0) s_l1[wave] = (int)l1;
    __syncthreads();
    if (tid == 0) stat[row] = DbRowStat{ss, sc, (s_l1[0] + s_l1[1]) + (s_l1[2] + s_l1[3]), 0.0f};
}

// the 8-bit steps of 32 rows x 512 elements (16 pieces of 1 KB) per workgroup, a wave: eight rows, all their requests in flight; sum|a_i| by
// integer atomics (exact in any order).  Rows >= n_rows of the last tile: zeros.  (One workgroup per 32-row tile was 39 us for a call's 64
// queries: a single CU's vector ALU walking 131 072 elements.)
__global__ __launch_bounds__(256) void k_db_quant(const float* __restrict__ x, int n_rows, int dim, DbRowStat* __restrict__ stat, i32x4* __restrict__ frag) {
    constexpr int PP = 1040;                                   // bytes between staged pieces (1 KB + 16: the eight k-steps a row's lanes write go to different banks)
    __shared__ __attribute__((aligned(16))) unsigned char st[16 * PP];
    const int tile = blockIdx.x, k0 = blockIdx.y * 512, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_half = min(2, (dim - k0) >> 8);               // (dim is a multiple of 256: the last chunk may be half)
    f32x4 v[8][2];
    float inv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int row = tile * 32 + wave * 8 + rr;
        const float* __restrict__ xr = x + (long long)min(row, n_rows - 1) * dim + k0 + lane * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) v[rr][h] = *(const f32x4*)(xr + (h < n_half ? h * 256 : 0));
        const float sc = stat[min(row, n_rows - 1)].scale;
        inv[rr] = row < n_rows && sc > 0.0f ? 1.0f / sc : 0.0f;     // (rows that do not exist: zeros)
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr, row = tile * 32 + r;
        float l1 = 0.0f;                                       // (<= 8 x 127 per lane: exact)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned w = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // |x| <= 127 s (1 + 2^-23), the reciprocal and the product add 2^-23: the nearest integer is within [-127, 127]; NaN -> 0
                const float a = rintf(v[rr][h][c] * inv[rr]);
                l1 += h < n_half ? fabsf(a) : 0.0f;            // (a half chunk's second half was read from the first half's address: not counted)
                w |= ((unsigned)(int)a & 255u) << (8 * c);
            }
            // element 256 h + 4 lane + c of the chunk: step 8 h + (lane >> 3), k-half (lane >> 2) & 1, byte 4 (lane & 3) + c
            if (h < n_half) *(unsigned*)(st + (8 * h + (lane >> 3)) * PP + ((((lane >> 2) & 1) * 32 + r) * 16) + (lane & 3) * 4) = w;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) l1 += __shfl_xor(l1, off, 64);
        if (lane == 0 && row < n_rows) atomicAdd(&stat[row].l1, (int)l1);
    }
    __syncthreads();
    i32x4* __restrict__ dst = frag + ((long long)tile * (dim >> 5) + (k0 >> 5)) * 64;
    for (int pc = wave; pc < 8 * n_half; pc += 4) dst[pc * 64 + lane] = *(const i32x4*)(st + pc * PP + lane * 16);
}

// exact score of one (query, slot): ||q - d|| in tree256 order, the chain of k_db_scores
template <int DEPTH>
__device__ __forceinline__ float db_exact_u(const float* __restrict__ q, const float* __restrict__ d, int dim, int lane) {
    f32x4 p = {0.f, 0.f, 0.f, 0.f};
    // (DEPTH steps' loads in flight: the handful of pairs a batch re-scores decide how long the kernel's last waves run -- one wave walking 32 KB with
    //  a memory round trip per step was 8 us; the chain per (lane, component) still runs k ascending)
    int k0 = 0;
    for (; k0 + DEPTH * 256 <= dim; k0 += DEPTH * 256) {
        f32x4 dv[DEPTH], qv[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) { dv[j] = *(const f32x4*)(d + k0 + j * 256 + lane * 4); qv[j] = *(const f32x4*)(q + k0 + j * 256 + lane * 4); }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float df = qv[j][c] - dv[j][c]; p[c] = fmaf(df, df, p[c]); }
    }
    for (; k0 < dim; k0 += 256) {                              // per (lane, component): one chain, k ascending -- k_db_scores' order
        const f32x4 dv = *(const f32x4*)(d + k0 + lane * 4), qv = *(const f32x4*)(q + k0 + lane * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float df = qv[c] - dv[c]; p[c] = fmaf(df, df, p[c]); }
    }
    return 1 - sqrtf(tree256_wave4(p));
}

#define DBS_WAVES 4
// QT: query tiles of 32 (1, 2); FULL: dim == 4096 (thirty-two 32-k steps per wave; other lengths: the same code with every step guarded).
// One workgroup per CU, one wave per SIMD -- 512 registers each: a wave keeps the fragments of ALL the launch's queries for ITS QUARTER of k
// in registers (QT x 32 steps x 4 = 256 for 64 queries: the matrix-operand half of the file), so the queries cost no LDS traffic, no k-parts
// and no refill; the four waves walk the workgroup's tiles together, each streaming its quarter of a tile's rows (32 KB) straight from memory
// into a register ring (the next tile's piece requested as a piece is consumed), and meet once per tile: 16 QT accumulator registers per wave
// through LDS, one barrier, then wave w decides the pairs of registers 4 w .. 4 w + 3 of every query tile -- bound test, exact chain for what
// is left, scores, maxima.  The queries' fragments are requested step by step (all tiles of a step together), so the first tile's MFMAs start
// when the first step has arrived, not the last: every CU fetching the same 256 KB is 4 us of L2 time (8 MB per XCD), the one cost of this
// form -- measured (s_memtime) 10 us from the kernel's start to the end of the first tile's MFMAs when they waited for all of it.
#define DBS_DEFER 512                                         // pairs a workgroup keeps for the exact chain behind its sweep
template <int QT, bool FULL>
__global__ __launch_bounds__(DBS_WAVES * 64, 1) void k_db_sweep(const i32x4* __restrict__ qfrag, int n_queries, int q0, const i32x4* __restrict__ dbfrag, int n, int dim,
                                                                const float* __restrict__ q, const float* __restrict__ db,
                                                                const DbRowStat* __restrict__ qstat, const DbRowStat* __restrict__ dstat,
                                                                const unsigned char* __restrict__ occupied, float* __restrict__ scores,
                                                                unsigned int* __restrict__ best_bits, int n_partials, int* __restrict__ stat) {
    __shared__ __attribute__((aligned(16))) int s_acc[2][DBS_WAVES][QT * 16][64];     // [buffer][wave][16 qt + reg][lane]
    __shared__ DbRowStat s_q[QT * 32];
    __shared__ int s_list[DBS_WAVES][QT * 256];
    __shared__ float s_res[DBS_WAVES][QT * 256];
    __shared__ int s_cnt[DBS_WAVES];
    __shared__ int s_defer[DBS_DEFER];                         // tile << 11 | wave << 9 | (4 qt + r4) << 6 | lane; -1: none
    __shared__ int s_dn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = dim >> 5, spw = FULL ? 32 : ks >> 2, n_tiles = (n + 31) >> 5;      // (dim is a multiple of 256: ks of eight)
    const int G = gridDim.x;
    if (tid < QT * 32) s_q[tid] = qstat[min(q0 + tid, n_queries - 1)];
    for (int i = tid; i < DBS_DEFER; i += DBS_WAVES * 64) s_defer[i] = -1;
    if (tid == 0) s_dn = 0;
    int t = blockIdx.x;                                        // (the grid has at most n_tiles workgroups)
    // the ring: this wave's quarter of tile t; the queries' fragments step by step (in this order: the waits of the loop below count requests)
    i32x4 b[32];
    i32x4 fq[QT][32];
    const i32x4* dq = dbfrag + (long long)wave * spw * 64 + lane;      // (+ t ks 64: tile t)
    const i32x4* qq = qfrag + ((long long)(q0 >> 5) * ks + wave * spw) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (FULL || j < spw) {
            b[j] = dq[((long long)t * ks + j) * 64];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) fq[qt][j] = qq[((long long)qt * ks + j) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                           // s_q, s_defer
    // (opaque copies: what the stores' addresses are computed from must not look loop invariant)
    int n_o = n, np_o = n_partials, nq_o = n_queries, q0_o = q0;
    for (int buf = 0; t < n_tiles; t += G, buf ^= 1) {
        asm volatile("" : "+s"(n_o), "+s"(np_o), "+s"(nq_o), "+s"(q0_o));
        const int slot = t * 32 + (lane & 31), half = lane >> 5;
        const unsigned char occ_b = occupied[min(slot, n_o - 1)];      // (requested now, used behind the barrier)
        const DbRowStat ds = dstat[min(slot, n_o - 1)];
        i32x16 acc[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[qt][i] = 0;
        const int tn = t + G;
        const bool more = tn < n_tiles;                        // (uniform)
        const i32x4* nx = dq + (long long)(more ? tn : t) * ks * 64;
        auto batch = [&](auto pf) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (FULL || j < spw) {
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) acc[qt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fq[qt][j], b[j], acc[qt], 0, 0, 0);
                    if (decltype(pf)::value) b[j] = nx[(long long)j * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (more) batch(std::true_type{}); else batch(std::false_type{});
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) s_acc[buf][wave][qt * 16 + reg][lane] = acc[qt][reg];
        __syncthreads();
        // ---- the tile's sums: lane (row = lane & 31, half = lane >> 5), register reg of tile qt: query qt * 32 + (reg & 3) + 8 (reg >> 2) + 4 half;
        // wave w decides registers 4 w + r4.  The pairs the bound cannot rule out go to the wave's list ((4 qt + r4) << 6 | lane); a few of them
        // (the usual case: a keyframe's handful of true neighbours) are left for the end of the sweep -- the exact chain is two memory round
        // trips during which the other three waves would stand at the next barrier --, more are walked here (one inlined copy), a second pass
        // picks the results up.
        // (one wave per SIMD: nothing hides an LDS round trip but the wave's own independent requests -- all sums and all query statistics are
        //  read before the first is used, the list is appended to afterwards, the maxima are reduced level by level for all pairs together:
        //  written pair by pair the decision was 4 us per tile, a hundred dependent LDS round trips)
        if (lane == 0) s_cnt[wave] = 0;
        const bool occ = slot < n_o && occ_b;
        int dot[QT * 4];
        DbRowStat qsv[QT * 4];
#pragma unroll
        for (int p = 0; p < QT * 4; ++p) {
            const int qt = p >> 2, r4 = p & 3;
            int w4[DBS_WAVES];
#pragma unroll
            for (int w = 0; w < DBS_WAVES; ++w) w4[w] = s_acc[buf][w][qt * 16 + 4 * wave + r4][lane];
            dot[p] = (w4[0] + w4[1]) + (w4[2] + w4[3]);
            qsv[p] = s_q[qt * 32 + r4 + 8 * wave + 4 * half];
        }
        unsigned mine = 0;
#pragma unroll
        for (int p = 0; p < QT * 4; ++p) {
            const int c = (p >> 2) * 32 + (p & 3) + 8 * wave + 4 * half;
            const DbRowStat qs = qsv[p];
            const float tt = qs.norm + ds.norm;
            const float err = ds.scale * qs.scale * (0.5f * ((float)qs.l1 + (float)ds.l1) + 0.25f * (float)dim);      // (s_d / 2) A1 + (s_q / 2) B1 + dim s_q s_d / 4
            const float d2 = fmaf(-2.0f, qs.scale * ds.scale * (float)dot[p], tt);
            // (anything not finite fails the comparison and goes to the exact chain too)
            if (occ && q0_o + c < nq_o && !(d2 >= 1.0f + 2.002f * err + 1e-4f * tt)) mine |= 1u << p;
        }
        if (__ballot(mine != 0)) {                             // (wave-uniform; rare)
#pragma unroll
            for (int p = 0; p < QT * 4; ++p)
                if (mine >> p & 1) s_list[wave][atomicAdd(&s_cnt[wave], 1)] = p << 6 | lane;
        }
        const int cnt = s_cnt[wave];                           // (LDS operations of one wave execute in order)
        bool deferred = false;
        if (cnt) {                                             // (wave-uniform; rare)
            if (stat && lane == 0) atomicAdd(stat, cnt);
            if (cnt <= 4) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_dn, cnt);
                base = __builtin_amdgcn_readfirstlane(base);
                deferred = base + cnt <= DBS_DEFER;            // (a full list: its last slots stay -1)
                if (deferred && lane < cnt) s_defer[base + lane] = t << 11 | wave << 9 | s_list[wave][lane];
            }
            if (!deferred)
                for (int k = 0; k < cnt; ++k) {
                    const int e = s_list[wave][k], el = e & 63;
                    const int e_c = (e >> 8) * 32 + ((e >> 6) & 3) + 8 * wave + 4 * (el >> 5), e_slot = t * 32 + (el & 31);
                    const float ex = db_exact_u<4>(q + (long long)(q0_o + e_c) * dim, db + (long long)e_slot * dim, dim, lane);
                    if (lane == 0) s_res[wave][e] = ex;
                }
        }
        float best[QT * 4];
#pragma unroll
        for (int p = 0; p < QT * 4; ++p) {
            const int qi = q0_o + (p >> 2) * 32 + (p & 3) + 8 * wave + 4 * half;
            float u = occ ? 0.0f : -1.0f;                      // (-1: empty slot)
            if ((mine >> p & 1) && !deferred) u = fmaxf(s_res[wave][p << 6 | lane], 0.0f);      // (a deferred pair: 0 for now)
            if (slot < n_o && qi < nq_o) scores[(long long)qi * n_o + slot] = u;
            best[p] = fmaxf(u, 0.0f);
        }
        // the maximum of a half wave's 32 rows by data-parallel primitives (a max does not care about the order): within quads, within rows of
        // 16 (half mirror, mirror), then lane 15 / 47 into the row above -- lanes 16-31 / 48-63 hold it; no LDS round trips
#pragma unroll
        for (int p = 0; p < QT * 4; ++p) {
            int v = (int)__float_as_uint(best[p]);             // (scores >= 0: the bits order like the floats)
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));      // quad_perm [1, 0, 3, 2]
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));      // quad_perm [2, 3, 0, 1]
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));     // row_half_mirror
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));     // row_mirror
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));     // row_bcast15 into rows 1 and 3
            const int qi = q0_o + (p >> 2) * 32 + (p & 3) + 8 * wave + 4 * half;
            if ((lane & 31) == 31 && qi < nq_o) best_bits[(long long)qi * np_o + t] = (unsigned)v;
        }
    }
    // ---- the deferred pairs: every wave takes its share; a score >= 0 as an unsigned integer orders like the float, so the tile's maximum is
    // raised by an atomic (its store above is behind the barrier)
    __syncthreads();
    const int nd = min(s_dn, DBS_DEFER);
    for (int k = wave; k < nd; k += DBS_WAVES) {
        const int e = s_defer[k];
        if (e < 0) continue;                                   // (uniform)
        const int et = e >> 11, ew = (e >> 9) & 3, eqt = (e >> 8) & 1, er4 = (e >> 6) & 3, el = e & 63;
        const int qi = q0 + eqt * 32 + er4 + 8 * ew + 4 * (el >> 5), e_slot = et * 32 + (el & 31);
        const float u = fmaxf(db_exact_u<16>(q + (long long)qi * dim, db + (long long)e_slot * dim, dim, lane), 0.0f);
        if (lane == 0) {
            scores[(long long)qi * n + e_slot] = u;
            atomicMax(best_bits + (long long)qi * n_partials + et, __float_as_uint(u));
        }
    }
}

hipError_t launch_db_prep_hi(const float* x, int n_rows, int dim, float* stat, void* hi, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    if (dim % 256) return hipErrorInvalidValue;
    if (n_rows <= 256 && dim <= 4096) {                                  // a call's queries, the tiles of a few added slots: one launch
        hipLaunchKernelGGL(k_db_prep_rows, dim3((n_rows + 31) & ~31), dim3(256), 0, s, x, n_rows, dim, (DbRowStat*)stat, (i32x4*)hi);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_db_rowstat, dim3((n_rows + 3) / 4), dim3(256), 0, s, x, n_rows, dim, (DbRowStat*)stat);
    hipLaunchKernelGGL(k_db_quant, dim3((n_rows + 31) / 32, (dim + 511) / 512), dim3(256), 0, s, x, n_rows, dim, (DbRowStat*)stat, (i32x4*)hi);
    return hipGetLastError();
}

int db_gemm_partials(int n) { return (n + 31) / 32; }      // one maximum per 32-row tile and query
bool db_screen_supported(int dim) { return dim > 0 && dim % 256 == 0 && dim <= 4096; }      // (a wave's quarter of k: <= thirty-two 32-k steps of fragments in registers)
size_t db_hi_bytes(int n_rows, int dim) { return (size_t)((n_rows + 31) & ~31) * (size_t)dim; }     // whole 32-row tiles (k_db_quant's unit), one byte per element
size_t db_stat_floats(int n_rows) { return (size_t)4 * (size_t)n_rows; }

// scores of up to 64 queries q0 .. of n_queries against the n slots of the database (see above), one launch: q / db: f32 rows, qh / dbh: their
// 8-bit copies in fragment order, qstat / dstat: DbRowStat per row (launch_db_prep_hi); best_partial: [n_queries][db_gemm_partials(n)]
hipError_t launch_db_sweep(const float* q, const void* qh, int n_queries, int q0, const float* qstat, const float* db, const void* dbh, const float* dstat,
                           const unsigned char* occupied, int n, int dim, float* scores, unsigned int* best_partial, hipStream_t s, int* stat) {
    if (n <= 0 || q0 >= n_queries) return hipSuccess;
    if (!db_screen_supported(dim)) return hipErrorInvalidValue;
    const int n_tiles = (n + 31) / 32;
    const int nq = std::min(64, n_queries - q0);
    const int qt = nq <= 32 ? 1 : 2;
    // a workgroup per CU: tiles blockIdx.x, + 256, ...  (Every workgroup the same number of tiles -- 313 tiles as 157 workgroups of two instead of 256
    // of which 57 take a second, and 100 fewer fetches of the queries' fragments -- measured SLOWER on the same box: 16.4 against 15.7 us; 200 workgroups: 15.7, 128: 18.5.)
    const dim3 grid((unsigned)std::min(256, n_tiles)), block(DBS_WAVES * 64);
    const i32x4* qf = (const i32x4*)qh; const i32x4* df = (const i32x4*)dbh;
    const DbRowStat* qs = (const DbRowStat*)qstat; const DbRowStat* ds = (const DbRowStat*)dstat;
    const int np = db_gemm_partials(n);
#define DBS_LAUNCH(QT_, FULL_) hipLaunchKernelGGL((k_db_sweep<QT_, FULL_>), grid, block, 0, s, qf, n_queries, q0, df, n, dim, q, db, qs, ds, occupied, scores, best_partial, np, stat)
    if (dim == 4096) { if (qt == 1) DBS_LAUNCH(1, true); else DBS_LAUNCH(2, true); }
    else { if (qt == 1) DBS_LAUNCH(1, false); else DBS_LAUNCH(2, false); }
#undef DBS_LAUNCH
    return hipGetLastError();
}

// keep slots with score > 0.8*best (mode 0) / > max(0.5, 0.8*best) (mode 1), ascending slot order
__global__ __launch_bounds__(1024) void k_db_filter(const float* __restrict__ scores, int n, int mode, const unsigned int* __restrict__ best_bits,
                                                    int n_partials, int32_t* __restrict__ cand_slot, float* __restrict__ cand_score,
                                                    int* __restrict__ n_cand, float* __restrict__ best_out) {
    __shared__ int wsum[16];
    // one workgroup per query (blockIdx.x): rows of n scores / candidates
    scores += (long long)blockIdx.x * n; cand_slot += (long long)blockIdx.x * n; cand_score += (long long)blockIdx.x * n;
    best_bits += (long long)blockIdx.x * n_partials; n_cand += blockIdx.x; best_out += blockIdx.x;
    __shared__ float wbest[16];
    float bl = 0.0f;                                          // scores are >= 0
    for (int i = threadIdx.x; i < n_partials; i += 1024) bl = fmaxf(bl, __uint_as_float(best_bits[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bl = fmaxf(bl, __shfl_xor(bl, off, 64));
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = bl;
    __syncthreads();
    float best = wbest[0];
    for (int w = 1; w < 16; ++w) best = fmaxf(best, wbest[w]);
    float min_score = best * 0.8f;
    if (mode == 1) min_score = fmaxf(0.5f, min_score);
    // ascending slot order without a barrier per 1024 slots: wave w owns the contiguous range [w, w + 1) per_wave, counts its survivors (ballots, the
    // loads of four 64-slot pieces in flight), one barrier hands every wave its offset, and a second pass over the same range (L2 hits) writes them out
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_wave = (((n + 15) >> 4) + 63) & ~63;
    const int w0 = wave * per_wave, w1 = min(n, w0 + per_wave);
    int cnt = 0;                                              // (wave-uniform)
    for (int i0 = w0; i0 < w1; i0 += 256) {
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; sc[u] = i < w1 ? scores[i] : -1.0f; }      // (-1 never passes: min_score >= 0)
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt += __popcll(__ballot(sc[u] > min_score));
    }
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    int o = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wave) o += c; tot += c; }
    for (int i0 = w0; i0 < w1 && cnt > 0; i0 += 256) {
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; sc[u] = i < w1 ? scores[i] : -1.0f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool keep = sc[u] > min_score;
            const unsigned long long mask = __ballot(keep);
            if (keep) { const int at = o + __popcll(mask & ((1ull << lane) - 1ull)); cand_slot[at] = i0 + u * 64 + lane; cand_score[at] = sc[u]; }
            o += __popcll(mask);
        }
    }
    if (tid == 0) { *n_cand = tot; *best_out = best; }
}

// workgroups of the scan kernels (4 waves each); the per-wave best scores are the filter kernel's partials
int db_scan_workgroups(int n) { const int w = (n + 3) / 4; return w < 1024 ? (w > 0 ? w : 1) : 1024; }

hipError_t launch_db_scores(const float* q, const float* db, const unsigned char* occupied, int n, int dim, float* scores,
                            unsigned int* best_partial, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (dim % 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_db_scores, dim3(db_scan_workgroups(n)), dim3(256), 0, s, q, db, occupied, n, dim, scores, best_partial);
    return hipGetLastError();
}
hipError_t launch_db_scores_batch(const float* q, int n_queries, const float* db, const unsigned char* occupied, int n, int dim, float* scores,
                                  unsigned int* best_partial, hipStream_t s) {
    if (n <= 0 || n_queries <= 0) return hipSuccess;
    if (dim % 256 || dim > 4096) return hipErrorInvalidValue;
    const size_t lds = (size_t)DBQ * dim * sizeof(float);
    static std::once_flag attr_once;                                    // > 64 KB of dynamic LDS has to be requested once
    std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)k_db_scores_batch, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); });
    const int w = db_scan_workgroups(n), wgs = w < 256 ? w : 256;      // one 128 KB query tile per CU
    hipLaunchKernelGGL(k_db_scores_batch, dim3(wgs, (n_queries + DBQ - 1) / DBQ), dim3(256), lds, s, q, n_queries, db, occupied, n, dim, scores, best_partial);
    return hipGetLastError();
}
int db_batch_workgroups(int n) { const int w = db_scan_workgroups(n); return w < 256 ? w : 256; }
hipError_t launch_db_filter(const float* scores, int n, int mode, const unsigned int* best_partial, int n_partials, int32_t* cand_slot,
                            float* cand_score, int* n_cand, float* best, int n_queries, hipStream_t s) {
    if (n_queries <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_db_filter, dim3(n_queries), dim3(1024), 0, s, scores, n, mode, best_partial, n_partials, cand_slot, cand_score, n_cand, best);
    return hipGetLastError();
}

}  // namespace hfnet
