// block_bench.hip -- development harness: time one fused inverted-residual block launch (kernels_block.hip) on synthetic data.
//   build: bash tools/dev/build_block_bench.sh      run (GPU box): tools/dev/block_bench <layer 3..14> <frames> <variant> [reps]
// Timing only (random weights / activations); parity is the business of tests/.
#include "../../hfnet_slam_amd/csrc/kernels.hpp"
#include <cstdlib>
#include <vector>
using namespace hfnet;
namespace hfnet { void set_error(const char*, ...) {} const char* get_error() { return ""; } }
static float* dev_rand(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 8, frames = argc > 2 ? atoi(argv[2]) : 64, variant = argc > 3 ? atoi(argv[3]) : 4, reps = argc > 4 ? atoi(argv[4]) : 10;
    // (stride, cout) of layers 2..18 at depth multiplier 0.75; cin = previous cout, expansion 6x
    const int st[19] = {0, 2, 1, 2, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1};
    const int co[19] = {0, 24, 16, 24, 24, 24, 48, 96, 48, 48, 48, 48, 72, 72, 72, 120, 120, 120, 240};
    const int lw[4] = {752, 624, 520, 432}, lh[4] = {480, 400, 328, 272};
    BlockPack b{};
    b.cin = co[L - 1]; b.expand = b.cin * 6; b.stride = st[L]; b.cout = co[L]; b.residual = b.stride == 1 && b.cin == b.cout; b.has_expand = 1;
    b.ex.taps = 1; b.ex.cin = b.cin; b.ex.n = b.expand; b.ex.nt_total = (b.expand + 31) / 32;
    b.ex.w = dev_rand((size_t)b.cin / 8 * b.ex.nt_total * 256, 0.2f); b.ex.bias = dev_rand(b.ex.nt_total * 32, 0.2f);
    b.dw.c = b.expand; b.dw.w = dev_rand(9 * b.expand, 0.3f); b.dw.bias = dev_rand(b.expand, 0.2f);
    b.pr.taps = 1; b.pr.cin = b.expand; b.pr.n = b.cout; b.pr.nt_total = (b.cout + 31) / 32;
    b.pr.w = dev_rand((size_t)b.expand / 8 * b.pr.nt_total * 256, 0.1f); b.pr.bias = dev_rand(b.pr.nt_total * 32, 0.2f);
    // (ConvPack16 forms for k_block_fused6: [ceil(cin / 16)][n16][64][4])
    b.ex16.cin = b.cin; b.ex16.n = b.expand; b.ex16.n16 = (b.expand + 15) / 16;
    b.ex16.w = dev_rand((size_t)((b.cin + 15) / 16) * b.ex16.n16 * 256, 0.2f);
    b.pr16.cin = b.expand; b.pr16.n = b.cout; b.pr16.n16 = (b.cout + 15) / 16;
    b.pr16.w = dev_rand((size_t)((b.expand + 15) / 16) * b.pr16.n16 * 256, 0.1f);
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
    float* X = dev_rand((size_t)in_off * b.cin, 2.0f);
    float* Y; hipMalloc(&Y, (size_t)out_off * b.cout * 4);
    if (!block_fusable(b, variant)) { printf("layer %d: no fused kernel for variant %d\n", L, variant); return 1; }
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) launch_block_fused(X, b, Y, g, variant, s);   // (long enough for the clocks to settle where a running pipeline holds them: three launches after idle measured 10-15 % off)
    hipStreamSynchronize(s);
    float best = 1e30f, sum = 0.f;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0, s);
        hipError_t er = launch_block_fused(X, b, Y, g, variant, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        if (er != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(er)); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    const double px_in = (double)in_off, px_out = (double)out_off;
    const double flop = 2.0 * px_in * b.cin * b.expand + 18.0 * px_out * b.expand + 2.0 * px_out * b.expand * b.cout;
    printf("L%02d variant %d frames %d: avg %.1f us  min %.1f us  alg %.1f TFLOP/s = %.3f of 157.3\n", L, variant, frames, sum / reps * 1e3, best * 1e3,
           flop / (sum / reps * 1e-3) / 1e12, flop / (sum / reps * 1e-3) / 1e12 / 157.3);
    return 0;
}
