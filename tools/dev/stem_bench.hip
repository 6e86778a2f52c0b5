// stem_bench.hip -- development harness: time the stem + layer_2 launch (k_stem_block2, kernels_block.hip) on synthetic data.
//   build: BB_MAIN=tools/dev/stem_bench.hip BB_OUT=stem_bench bash tools/dev/build_block_bench.sh     run (GPU box): tools/dev/stem_bench <frames> [reps]
// Timing only (random weights / images); parity is the business of tests/.
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
    const int frames = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 10;
    const int lw[4] = {752, 624, 520, 432}, lh[4] = {480, 400, 328, 272};      // cropped level sizes (multiples of 8)
    BlockPack b{};
    b.cin = 24; b.expand = 24; b.stride = 1; b.cout = 16; b.residual = false; b.has_expand = 0;
    b.dw.c = 24; b.dw.w = dev_rand(9 * 24, 0.3f); b.dw.bias = dev_rand(24, 0.2f);
    b.pr.taps = 1; b.pr.cin = 24; b.pr.n = 16; b.pr.nt_total = 1; b.pr.bias = dev_rand(32, 0.2f);
    b.pr_logical = dev_rand(24 * 16, 0.2f);
    float* stem_w = dev_rand(9 * 24, 0.3f);
    float* stem_b = dev_rand(24, 0.2f);
    Geom gs{}, gb{};
    gs.n_levels = gb.n_levels = 4; gs.batch = gb.batch = frames;
    ImageSet imgs{};
    long long out_off = 0;
    for (int l = 0; l < 4; ++l) {
        const int h = lh[l], w = lw[l], ho = same_out(h, 2), wo = same_out(w, 2);
        gs.lv[l] = LevelGeom{h, w, ho, wo, same_pad_before(h, 3, 2), same_pad_before(w, 3, 2), 0, 0};
        gb.lv[l] = LevelGeom{ho, wo, ho, wo, 1, 1, 0, out_off};
        out_off += (long long)frames * ho * wo;
        std::vector<uint8_t> hi((size_t)frames * h * w);
        for (auto& v : hi) v = (uint8_t)(rand() & 255);
        uint8_t* d; hipMalloc(&d, hi.size() + 64); hipMemcpy(d, hi.data(), hi.size(), hipMemcpyHostToDevice);
        imgs.ptr[l] = d; imgs.row_stride[l] = w; imgs.frame_stride[l] = (long long)h * w;
    }
    float* Y; hipMalloc(&Y, (size_t)out_off * 16 * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 400; ++i) launch_stem_block(imgs, stem_w, stem_b, b, Y, gs, gb, s);   // (long enough for the clocks to settle where a running pipeline holds them)
    hipStreamSynchronize(s);
    float best = 1e30f, sum = 0.f;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0, s);
        hipError_t er = launch_stem_block(imgs, stem_w, stem_b, b, Y, gs, gb, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        if (er != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(er)); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    printf("stem_block %d frames: best %.1f us  mean %.1f us\n", frames, best * 1e3, sum / reps * 1e3);
    return 0;
}
