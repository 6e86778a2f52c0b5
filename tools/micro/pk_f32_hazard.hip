// pk_f32_hazard.hip -- NOTEBOOK.md R4.8 in its smallest form: a victim kernel whose waves do nothing but v_pk_mul_f32 (written as inline assembly) and
// compare both halves with plain v_mul_f32 of the same operands, lane by lane, on stream A; on stream B either nothing, a synthetic bf16-MFMA
// kernel, or the real split-bf16 fused block of layer 13 (kernels_block.hip), launched back to back.  Reports mismatches per 16-lane group.
//   build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -w -I hfnet_slam_amd/csrc -I include -x hip \
//            tools/micro/pk_f32_hazard.hip hfnet_slam_amd/csrc/kernels_block.hip hfnet_slam_amd/csrc/kernels_conv.hip -o tools/micro/pk_f32_hazard
//   run (GPU box): tools/micro/pk_f32_hazard <neighbour 0 none | 1 synthetic | 2 split-bf16 fused block | 3 exact fused block | 4 synthetic + bf16 split | 5 + 224 registers> [iterations] [0 packed with op_sel | 1 two plain multiplies | 2 packed without op_sel | 10 + n: s_nop n in between | 30 a v_nop in between | 31 loaded pair as src0 | 32 no load | 33 what the wrong value is]
#include "../../hfnet_slam_amd/csrc/kernels.hpp"
#include <cstdlib>
#include <vector>
using namespace hfnet;
namespace hfnet { void set_error(const char*, ...) {} const char* get_error() { return ""; } }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_victim(const float* src, unsigned* bad /* [4] per 16-lane group */, int rounds, int plain) {
    const int lane = threadIdx.x & 63;
    const f32x2 b = {1.0009765625f, 0.9990234375f};
    unsigned mism = 0, other = 0;
    for (int r = 0; r < rounds; ++r) {
        // as in k_sample: the operands arrive by a load, the packed multiply is the first instruction behind the s_waitcnt
        const float* ptr = src + (((blockIdx.x * 7 + r * 13) & 2047) * 2);      // (wave-uniform address: every lane loads the same pair)
        f32x2 a, p; float lo, hi;
        if (plain >= 10 && plain < 30) {      // wait states between the s_waitcnt and the packed multiply: how many make it right?
#define PKW(N) asm volatile("global_load_dwordx2 %0, %2, off\n\ts_waitcnt vmcnt(0)\n\ts_nop " #N "\n\tv_pk_mul_f32 %1, %3, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(a), "=&v"(p) : "v"(ptr), "v"(b) : "memory")
            switch (plain) { case 10: PKW(0); break; case 11: PKW(1); break; case 12: PKW(2); break; case 13: PKW(3); break; case 15: PKW(5); break; default: PKW(7); break; }
#undef PKW
        } else if (plain == 30)                 // an independent vector instruction in between instead of an s_nop
            asm volatile("global_load_dwordx2 %0, %2, off\n\ts_waitcnt vmcnt(0)\n\tv_nop\n\tv_pk_mul_f32 %1, %3, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(a), "=&v"(p) : "v"(ptr), "v"(b) : "memory");
        else if (plain == 31)                   // the loaded pair as the FIRST source
            asm volatile("global_load_dwordx2 %0, %2, off\n\ts_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %1, %0, %3 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(a), "=&v"(p) : "v"(ptr), "v"(b) : "memory");
        else if (plain == 32) {                 // no load: the pair comes from vector instructions
            a[0] = src[(r * 5 + lane) & 4095] + 1.0f; a[1] = src[(r * 3 + lane + 11) & 4095] - 1.0f;
            asm volatile("s_nop 4\n\tv_pk_mul_f32 %0, %2, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(p) : "v"(a), "v"(b));
        } else if (plain == 0 || plain == 33)
            asm volatile("global_load_dwordx2 %0, %2, off\n\ts_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %1, %3, %0 op_sel:[0,1] op_sel_hi:[1,0]"
                         : "=&v"(a), "=&v"(p) : "v"(ptr), "v"(b) : "memory");
        else if (plain == 1)      // the same position taken by two plain multiplies
        {   float a0, a1, p0, p1;
            asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:4\n\ts_waitcnt vmcnt(0)\n\tv_mul_f32 %2, %5, %1\n\tv_mul_f32 %3, %6, %0"
                         : "=&v"(a0), "=&v"(a1), "=&v"(p0), "=&v"(p1) : "v"(ptr), "v"(b[0]), "v"(b[1]) : "memory");
            a[0] = a0; a[1] = a1; p[0] = p0; p[1] = p1; }
        else                       // the packed multiply without operand selection (p.lo = b.lo * a.lo, p.hi = b.hi * a.hi), results swapped below
            asm volatile("global_load_dwordx2 %0, %2, off\n\ts_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %1, %3, %0"
                         : "=&v"(a), "=&v"(p) : "v"(ptr), "v"(b) : "memory");
        if (plain == 2) { float q0, q1; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(b[0]), "v"(a[0])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(b[1]), "v"(a[1]));
                          mism += (__float_as_int(p[0]) != __float_as_int(q0)) + (__float_as_int(p[1]) != __float_as_int(q1)); continue; }
        // (op_sel:[0,1] op_sel_hi:[1,0]: p.lo = b.lo * a.hi, p.hi = b.hi * a.lo -- the operand selection the compiler produced in k_sample)
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(b[0]), "v"(a[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(b[1]), "v"(a[0]));
        if (plain == 33) {                      // what IS the wrong value?  count the halves that equal the product WITHOUT the operand selection
            float ulo, uhi;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ulo) : "v"(b[0]), "v"(a[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(uhi) : "v"(b[1]), "v"(a[1]));
            if (__float_as_int(p[0]) != __float_as_int(lo)) { if (__float_as_int(p[0]) == __float_as_int(ulo)) ++mism; else ++other; }
            if (__float_as_int(p[1]) != __float_as_int(hi)) { if (__float_as_int(p[1]) == __float_as_int(uhi)) ++mism; else ++other; }
            continue;
        }
        mism += (__float_as_int(p[0]) != __float_as_int(lo)) + (__float_as_int(p[1]) != __float_as_int(hi));
    }
    if (mism) atomicAdd(&bad[lane >> 4], mism);
    if (other) atomicAdd(&bad[4 + (lane >> 4)], other);
}
template <int KIND>    // 0: bf16 MFMA + a little LDS; 1: + the split of f32 values into bf16 pieces per step (v_cvt_pk_bf16_f32, subtract, convert again); 2: + 224 accumulator registers
__global__ __launch_bounds__(64, 2) void k_busy(const float* src, float* dst, int steps) {
    constexpr int NA = KIND == 2 ? 14 : 12;
    __shared__ float tile[32 * 36];
    const int lane = threadIdx.x;
    f32x16 acc[NA];
    for (int m = 0; m < NA; ++m) for (int i = 0; i < 16; ++i) acc[m][i] = (float)(m + i);
    bf16x8 a, b, al;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(src[(blockIdx.x * 64 + lane + e) & 4095]); b[e] = (__bf16)(0.001f * (float)(e + 1)); al[e] = a[e]; }
    for (int i = 0; i < 36; ++i) tile[(lane & 31) * 36 + i] = src[(lane * 36 + i) & 4095];
    for (int s = 0; s < steps; ++s) {
        if (KIND >= 1) {
            const float* pa = tile + (lane & 31) * 36 + (s & 3) * 8;
            for (int e = 0; e < 8; ++e) { const float v = pa[e]; a[e] = (__bf16)v; al[e] = (__bf16)(v - (float)a[e]); }
        }
        for (int m = 0; m < NA; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
        if (KIND >= 1) for (int m = 0; m < NA; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc[m], 0, 0, 0);
        tile[(s & 31) * 36 + (lane & 31)] = acc[s % NA][0] * 1e-6f;
        asm volatile("" ::: "memory");
        if (KIND == 0) a[0] = (__bf16)tile[((s + 1) & 31) * 36 + (lane & 31)];
    }
    float sum = 0.f;
    for (int m = 0; m < NA; ++m) for (int i = 0; i < 16; ++i) sum += acc[m][i];
    dst[blockIdx.x * 64 + lane] = sum;
}
static float* dev_rand(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
    const int neighbour = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 500, plain = argc > 3 ? atoi(argv[3]) : 0;
    float* src = dev_rand(4096, 4.0f); float* bdst; unsigned* bad;
    CK(hipMalloc(&bdst, 4096 * 64 * 4)); CK(hipMalloc(&bad, 32)); CK(hipMemset(bad, 0, 32));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    // layer 13: 72 -> 432 -> 72, 4 frames of 30 x 47
    BlockPack b{}; b.cin = 72; b.expand = 432; b.stride = 1; b.cout = 72; b.residual = 1; b.has_expand = 1;
    b.ex.taps = 1; b.ex.cin = 72; b.ex.n = 432; b.ex.nt_total = 14; b.ex.w = dev_rand((size_t)9 * 14 * 256, 0.2f); b.ex.bias = dev_rand(14 * 32, 0.2f);
    b.dw.c = 432; b.dw.w = dev_rand(9 * 432, 0.3f); b.dw.bias = dev_rand(432, 0.2f);
    b.pr.taps = 1; b.pr.cin = 432; b.pr.n = 72; b.pr.nt_total = 3; b.pr.w = dev_rand((size_t)54 * 3 * 256, 0.1f); b.pr.bias = dev_rand(3 * 32, 0.2f);
    b.ex16.cin = 72; b.ex16.n = 432; b.ex16.n16 = 27; b.ex16.w = dev_rand((size_t)5 * 27 * 256, 0.2f);
    b.pr16.cin = 432; b.pr16.n = 72; b.pr16.n16 = 5; b.pr16.w = dev_rand((size_t)27 * 5 * 256, 0.1f);
    void *e, *p; CK(hipMalloc(&e, bf16x3_pack_bytes(b.ex))); CK(hipMalloc(&p, bf16x3_pack_bytes(b.pr)));
    CK(launch_repack_bf16x3(b.ex, e, sb)); CK(launch_repack_bf16x3(b.pr, p, sb)); b.ex_bf = e; b.pr_bf = p;
    Geom g{}; g.n_levels = 1; g.batch = 4; g.lv[0].H = 30; g.lv[0].W = 47; g.lv[0].Ho = 30; g.lv[0].Wo = 47; g.lv[0].pt = 1; g.lv[0].pl = 1;
    float* X = dev_rand((size_t)4 * 30 * 47 * 72, 2.0f); float* Y; CK(hipMalloc(&Y, (size_t)4 * 30 * 47 * 72 * 4));
    CK(hipDeviceSynchronize());
    for (int it = 0; it < iters; ++it) {
        for (int q = 0; q < 12; ++q) {
            if (neighbour == 1) hipLaunchKernelGGL(k_busy<0>, dim3(192), dim3(64), 0, sb, src, bdst, 60);
            if (neighbour == 4) hipLaunchKernelGGL(k_busy<1>, dim3(192), dim3(64), 0, sb, src, bdst, 40);
            if (neighbour == 5) hipLaunchKernelGGL(k_busy<2>, dim3(192), dim3(64), 0, sb, src, bdst, 40);
            if (neighbour == 2) CK(launch_block_fused(X, b, Y, g, 4, sb, 1));
            if (neighbour == 3) CK(launch_block_fused(X, b, Y, g, 4, sb, 0));
        }
        hipLaunchKernelGGL(k_victim, dim3(1024), dim3(256), 0, sa, src, bad, 400, plain);
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
    }
    unsigned h[8]; CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
    if (plain == 33) printf("   wrong halves in lanes 48-63 that equal the product WITHOUT operand selection: %u; that are something else: %u\n", h[3], h[7]);
    const char* names[8] = {"nothing", "a synthetic bf16-MFMA kernel", "the split-bf16 fused block (layer 13)", "the exact fused block (layer 13)", "a synthetic bf16-MFMA kernel that splits f32 into bf16 pieces", "the same with 224 accumulator registers", "", ""};
    printf("%s right behind the load's s_waitcnt, checked against v_mul_f32 later; %d launches of 262144 lanes x 400 products beside %s: mismatching halves in lanes 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u\n",
           plain == 1 ? "two v_mul_f32" : plain == 2 ? "v_pk_mul_f32 (no op_sel)" : plain >= 10 && plain < 30 ? "s_nop (mode - 10), then v_pk_mul_f32 with op_sel," : plain == 30 ? "v_nop, then v_pk_mul_f32 with op_sel," : plain == 31 ? "v_pk_mul_f32 with op_sel, loaded pair as src0," : plain == 32 ? "v_pk_mul_f32 with op_sel on a pair computed by vector instructions (no load)," : plain == 33 ? "v_pk_mul_f32 with op_sel " : "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", iters, names[neighbour & 7], h[0], h[1], h[2], h[3]);
    return 0;
}
