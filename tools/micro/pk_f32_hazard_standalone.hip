// pk_f32_hazard_standalone.hip -- gfx950 (MI355X): v_pk_mul_f32 with cross-half selection on its SECOND source returns wrong values in lanes 48-63 while
// another wave on the SIMD converts f32 to packed bf16 (v_cvt_pk_bf16_f32) between bf16 MFMAs.  No other file needed.  (NOTEBOOK.md R4.8)
//   build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/micro/pk_f32_hazard_standalone.hip -o tools/micro/pk_f32_hazard_standalone
//          (-fno-slp-vectorize shapes the NEIGHBOUR: 16 v_cvt_pk_bf16_f32 + 8 v_sub_f32 per step; packed by SLP -- 10 + v_pk_add_f32 -- it does not trigger)
//   run:   tools/micro/pk_f32_hazard_standalone      measured (ROCm 7.2, MI355X): 0 / 0 / 0 / 0 in fourteen lines (incl. v_pk_add_f32 op_sel_hi:[1,0] and v_cvt_pk_bf16_f32 as victims), 0 / 0 / 0 / 190 000-295 000 for "second source
//          crossed" beside "bf16 MFMAs + v_cvt_pk_bf16_f32 splits"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FORM>    // 3: v_pk_add_f32 op_sel_hi:[1,0]; 4: v_cvt_pk_bf16_f32; v_pk_mul_f32 -- 0: second source crossed (op_sel:[0,1] op_sel_hi:[1,0]); 1: the same products with the FIRST source crossed; 2: no selection
__global__ __launch_bounds__(256) void k_victim(const float* src, unsigned* bad /* per 16-lane group */, int rounds) {
    const int lane = threadIdx.x & 63;
    const f32x2 b = {1.0009765625f, 0.9990234375f};
    unsigned mism = 0;
    for (int r = 0; r < rounds; ++r) {
        f32x2 a = {src[(r * 5 + lane + blockIdx.x) & 4095] + 1.0f, src[(r * 3 + lane + 11) & 4095] - 1.0f}, p;
        float lo, hi;
        if (FORM == 0) asm volatile("s_nop 4\n\tv_pk_mul_f32 %0, %2, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(p) : "v"(a), "v"(b));     // p.lo = b.lo * a.hi, p.hi = b.hi * a.lo
        if (FORM == 1) asm volatile("s_nop 4\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(p) : "v"(a), "v"(b));     // p.lo = a.hi * b.lo, p.hi = a.lo * b.hi
        if (FORM == 2) asm volatile("s_nop 4\n\tv_pk_mul_f32 %0, %2, %1" : "=&v"(p) : "v"(a), "v"(b));                                 // p.lo = b.lo * a.lo, p.hi = b.hi * a.hi
        if (FORM == 3) {      // v_pk_add_f32 with the second source's low half used twice (what clang emits for "pair + scalar")
            asm volatile("s_nop 4\n\tv_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(p) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(a[0]), "v"(b[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(a[1]), "v"(b[0]));
            mism += (__float_as_int(p[0]) != __float_as_int(lo)) + (__float_as_int(p[1]) != __float_as_int(hi));
            continue;
        }
        if (FORM == 4) {      // the packed conversion itself as the victim, against round-to-nearest-even done with integer instructions
            unsigned pk;
            asm volatile("s_nop 4\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=&v"(pk) : "v"(a[0]), "v"(a[1]));
            auto rne = [](float x) { unsigned u = __float_as_uint(x); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
            mism += ((pk & 0xffffu) != rne(a[0])) + ((pk >> 16) != rne(a[1]));
            continue;
        }
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(b[0]), "v"(FORM == 2 ? a[0] : a[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(b[1]), "v"(FORM == 2 ? a[1] : a[0]));
        mism += (__float_as_int(p[0]) != __float_as_int(lo)) + (__float_as_int(p[1]) != __float_as_int(hi));
    }
    if (mism) atomicAdd(&bad[lane >> 4], mism);
}
template <bool SPLIT>  // one-wave workgroups, bf16 MFMA back to back; SPLIT: every step splits eight f32 values into two bf16 pieces first (v_cvt_pk_bf16_f32)
__global__ __launch_bounds__(64, 2) void k_neighbour(const float* src, float* dst, int steps) {
    __shared__ float tile[32 * 36];
    const int lane = threadIdx.x;
    f32x16 acc[12];
    for (int m = 0; m < 12; ++m) for (int i = 0; i < 16; ++i) acc[m][i] = (float)(m + i);
    bf16x8 a, b, al;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(src[(blockIdx.x * 64 + lane + e) & 4095]); b[e] = (__bf16)(0.001f * (float)(e + 1)); al[e] = a[e]; }
    for (int i = 0; i < 36; ++i) tile[(lane & 31) * 36 + i] = src[(lane * 36 + i) & 4095];
    for (int s = 0; s < steps; ++s) {
        if (SPLIT) {
            const float* pa = tile + (lane & 31) * 36 + (s & 3) * 8;
            for (int e = 0; e < 8; ++e) { const float v = pa[e]; a[e] = (__bf16)v; al[e] = (__bf16)(v - (float)a[e]); }
        }
        for (int m = 0; m < 12; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
        if (SPLIT) for (int m = 0; m < 12; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc[m], 0, 0, 0);
        tile[(s & 31) * 36 + (lane & 31)] = acc[s % 12][0] * 1e-6f;
        asm volatile("" ::: "memory");
        if (!SPLIT) a[0] = (__bf16)tile[((s + 1) & 31) * 36 + (lane & 31)];
    }
    float sum = 0.f;
    for (int m = 0; m < 12; ++m) for (int i = 0; i < 16; ++i) sum += acc[m][i];
    dst[blockIdx.x * 64 + lane] = sum;
}
int main() {
    std::vector<float> h(4096);
    for (float& v : h) v = 4.0f * ((float)rand() / RAND_MAX - 0.5f);
    float *src, *dst; unsigned* bad;
    hipMalloc(&src, 4096 * 4); hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice); hipMalloc(&dst, 192 * 64 * 4); hipMalloc(&bad, 16);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    const char* forms[5] = {"v_pk_mul_f32, second source crossed", "v_pk_mul_f32, first source crossed ", "v_pk_mul_f32, no selection         ", "v_pk_add_f32 op_sel_hi:[1,0]       ", "v_cvt_pk_bf16_f32                  "};
    for (int neighbour = 0; neighbour < 3; ++neighbour)
        for (int form = 0; form < 5; ++form) {
            hipMemset(bad, 0, 16); hipDeviceSynchronize();
            for (int it = 0; it < 200; ++it) {
                for (int q = 0; q < 12; ++q) {
                    if (neighbour == 1) hipLaunchKernelGGL(k_neighbour<false>, dim3(192), dim3(64), 0, sb, src, dst, 40);
                    if (neighbour == 2) hipLaunchKernelGGL(k_neighbour<true>, dim3(192), dim3(64), 0, sb, src, dst, 40);
                }
                if (form == 0) hipLaunchKernelGGL(k_victim<0>, dim3(1024), dim3(256), 0, sa, src, bad, 400);
                if (form == 1) hipLaunchKernelGGL(k_victim<1>, dim3(1024), dim3(256), 0, sa, src, bad, 400);
                if (form == 2) hipLaunchKernelGGL(k_victim<2>, dim3(1024), dim3(256), 0, sa, src, bad, 400);
                if (form == 3) hipLaunchKernelGGL(k_victim<3>, dim3(1024), dim3(256), 0, sa, src, bad, 400);
                if (form == 4) hipLaunchKernelGGL(k_victim<4>, dim3(1024), dim3(256), 0, sa, src, bad, 400);
                hipStreamSynchronize(sa); hipStreamSynchronize(sb);
            }
            unsigned c[4]; hipMemcpy(c, bad, 16, hipMemcpyDeviceToHost);
            printf("%s, beside %-44s: wrong halves in lanes 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u\n", forms[form],
                   neighbour == 0 ? "nothing" : neighbour == 1 ? "bf16 MFMAs" : "bf16 MFMAs + v_cvt_pk_bf16_f32 splits", c[0], c[1], c[2], c[3]);
        }
    return 0;
}
