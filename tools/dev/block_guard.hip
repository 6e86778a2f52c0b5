// block_guard.hip -- development harness: does a fused block launch write anywhere but its output?  Everything the launch touches lives in
// ONE arena filled with a pattern; after the launch every byte outside the output tensor must still hold the pattern (and the output
// tensor must be completely written).   build: BB_MAIN=tools/dev/block_guard.hip BB_OUT=block_guard BB_DEFS=hfnet_slam_amd/csrc/kernels_conv.hip bash tools/dev/build_block_bench.sh
//   run (GPU box): tools/dev/block_guard <layer 3..14> <frames> <variant> <bf16x3 0|1>
#include "../../hfnet_slam_amd/csrc/kernels.hpp"
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace hfnet;
namespace hfnet { void set_error(const char*, ...) {} const char* get_error() { return ""; } }
static std::vector<unsigned> H;                       // host image of the arena (words)
static size_t top = 0;
static const unsigned PAT = 0x7fc0dead;               // (a NaN: an output element left unwritten shows)
static size_t carve(size_t n_floats, float scale) {   // returns the word offset; scale 0: leave the pattern
    top = (top + 1023) & ~(size_t)1023;
    const size_t off = top;
    top += n_floats;
    if (top > H.size()) { printf("arena too small\n"); exit(2); }
    if (scale != 0.f) for (size_t i = 0; i < n_floats; ++i) { const float v = scale * ((float)rand() / RAND_MAX - 0.5f); memcpy(&H[off + i], &v, 4); }
    return off;
}
int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 13, frames = argc > 2 ? atoi(argv[2]) : 4, variant = argc > 3 ? atoi(argv[3]) : 4, bf = argc > 4 ? atoi(argv[4]) : 1;
    const int st[19] = {0, 2, 1, 2, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1};
    const int co[19] = {0, 24, 16, 24, 24, 24, 48, 96, 48, 48, 48, 48, 72, 72, 72, 120, 120, 120, 240};
    const int lw[4] = {752, 624, 520, 432}, lh[4] = {480, 400, 328, 272};
    H.assign((size_t)96 << 20, PAT);                  // 384 MB
    top = (size_t)8 << 20;                            // 32 MB of pattern in front of everything
    BlockPack b{};
    b.cin = co[L - 1]; b.expand = b.cin * 6; b.stride = st[L]; b.cout = co[L]; b.residual = b.stride == 1 && b.cin == b.cout; b.has_expand = 1;
    b.ex.taps = 1; b.ex.cin = b.cin; b.ex.n = b.expand; b.ex.nt_total = (b.expand + 31) / 32;
    b.pr.taps = 1; b.pr.cin = b.expand; b.pr.n = b.cout; b.pr.nt_total = (b.cout + 31) / 32;
    b.dw.c = b.expand;
    const size_t o_exw = carve((size_t)b.cin / 8 * b.ex.nt_total * 256, 0.2f), o_exb = carve(b.ex.nt_total * 32, 0.2f);
    const size_t o_dww = carve(9 * b.expand, 0.3f), o_dwb = carve(b.expand, 0.2f);
    const size_t o_prw = carve((size_t)b.expand / 8 * b.pr.nt_total * 256, 0.1f), o_prb = carve(b.pr.nt_total * 32, 0.2f);
    const size_t n_exbf = bf16x3_pack_bytes(b.ex) / 4, n_prbf = bf16x3_pack_bytes(b.pr) / 4;
    const size_t o_exbf = carve(n_exbf, 0.f), o_prbf = carve(n_prbf, 0.f);
    Geom g{};
    g.n_levels = L <= 7 ? 4 : 1; g.batch = frames;
    long long in_off = 0, out_off = 0;
    for (int l = 0; l < g.n_levels; ++l) {
        int h = lh[l], w = lw[l];
        for (int k = 1; k < L; ++k) { h = same_out(h, st[k]); w = same_out(w, st[k]); }
        LevelGeom& v = g.lv[l];
        v.H = h; v.W = w; v.Ho = same_out(h, b.stride); v.Wo = same_out(w, b.stride);
        v.pt = same_pad_before(h, 3, b.stride); v.pl = same_pad_before(w, 3, b.stride);
        v.in_off = in_off; v.out_off = out_off;
        in_off += (long long)frames * h * w; out_off += (long long)frames * v.Ho * v.Wo;
    }
    const size_t n_x = (size_t)in_off * b.cin, n_y = (size_t)out_off * b.cout;
    const size_t o_x = carve(n_x, 2.0f), o_y = carve(n_y, 0.f);
    top += (size_t)8 << 20;
    unsigned* D; hipMalloc(&D, H.size() * 4); hipMemcpy(D, H.data(), H.size() * 4, hipMemcpyHostToDevice);
    float* F = (float*)D;
    b.ex.w = F + o_exw; b.ex.bias = F + o_exb; b.dw.w = F + o_dww; b.dw.bias = F + o_dwb; b.pr.w = F + o_prw; b.pr.bias = F + o_prb;
    hipStream_t s; hipStreamCreate(&s);
    if (bf) {
        b.ex_bf = F + o_exbf; b.pr_bf = F + o_prbf;
        if (launch_repack_bf16x3(b.ex, (void*)b.ex_bf, s) != hipSuccess || launch_repack_bf16x3(b.pr, (void*)b.pr_bf, s) != hipSuccess) { printf("repack failed\n"); return 1; }
        hipStreamSynchronize(s);
        hipMemcpy(H.data() + o_exbf, D + o_exbf, n_exbf * 4, hipMemcpyDeviceToHost); hipMemcpy(H.data() + o_prbf, D + o_prbf, n_prbf * 4, hipMemcpyDeviceToHost);
        if (!block_fused_bf16x3_supported(b)) { printf("layer %d: no split-bf16 fused form\n", L); return 1; }
    }
    if (!block_fusable(b, variant)) { printf("layer %d: no fused kernel for variant %d\n", L, variant); return 1; }
    const hipError_t er = launch_block_fused(F + o_x, b, F + o_y, g, variant, s, bf);
    const hipError_t es = hipStreamSynchronize(s);
    if (er != hipSuccess || es != hipSuccess) { printf("launch failed: %s / %s\n", hipGetErrorString(er), hipGetErrorString(es)); return 1; }
    std::vector<unsigned> R(H.size());
    hipMemcpy(R.data(), D, R.size() * 4, hipMemcpyDeviceToHost);
    size_t stray = 0, first = 0, unwritten = 0;
    for (size_t i = 0; i < R.size(); ++i) {
        if (i >= o_y && i < o_y + n_y) { if (R[i] == PAT) ++unwritten; continue; }
        if (R[i] != H[i]) { if (!stray) first = i; ++stray; }
    }
    printf("L%02d frames %d variant %d bf16x3 %d: %zu words changed outside the output (first at word %zu; output = [%zu, %zu), input = [%zu, %zu)), %zu output words unwritten\n",
           L, frames, variant, bf, stray, first, o_y, o_y + n_y, o_x, o_x + n_x, unwritten);
    return stray || unwritten ? 1 : 0;
}
