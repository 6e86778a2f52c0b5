// conv_guard.hip -- development harness: does a split-bf16 1x1 convolution launch (k_conv_bf16x3<false>) write anywhere but its output rows?
// Same scheme as block_guard.hip: one arena filled with a pattern, everything the launch touches inside it.
//   build: BB_MAIN=tools/dev/conv_guard.hip BB_OUT=conv_guard BB_SRC=hfnet_slam_amd/csrc/kernels_conv.hip bash tools/dev/build_block_bench.sh
//   run (GPU box): tools/dev/conv_guard <cin> <n> <rows> <relu6> <residual>
#include "../../hfnet_slam_amd/csrc/kernels.hpp"
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace hfnet;
namespace hfnet { void set_error(const char*, ...) {} const char* get_error() { return ""; } }
static std::vector<unsigned> H;
static size_t top = 0;
static const unsigned PAT = 0x7fc0dead;
static size_t carve(size_t n_floats, float scale) {
    top = (top + 1023) & ~(size_t)1023;
    const size_t off = top;
    top += n_floats;
    if (top > H.size()) { printf("arena too small\n"); exit(2); }
    if (scale != 0.f) for (size_t i = 0; i < n_floats; ++i) { const float v = scale * ((float)rand() / RAND_MAX - 0.5f); memcpy(&H[off + i], &v, 4); }
    return off;
}
int main(int argc, char** argv) {
    const int cin = argc > 1 ? atoi(argv[1]) : 72, n = argc > 2 ? atoi(argv[2]) : 432, relu6 = argc > 4 ? atoi(argv[4]) : 1, res = argc > 5 ? atoi(argv[5]) : 0;
    const long long P = argc > 3 ? atoll(argv[3]) : 5640;
    H.assign((size_t)64 << 20, PAT);
    top = (size_t)8 << 20;
    ConvPack cp;
    cp.taps = 1; cp.cin = cin; cp.n = n; cp.nt_total = (n + 31) / 32;
    const size_t o_w = carve((size_t)cin / 8 * cp.nt_total * 256, 0.2f), o_b = carve(cp.nt_total * 32, 0.2f);
    const size_t n_bf = bf16x3_pack_bytes(cp) / 4, o_bf = carve(n_bf, 0.f);
    const size_t o_a = carve((size_t)P * cin, 2.0f), o_r = carve((size_t)P * n, 1.0f), n_y = (size_t)P * n, o_y = carve(n_y, 0.f);
    unsigned* D; hipMalloc(&D, H.size() * 4); hipMemcpy(D, H.data(), H.size() * 4, hipMemcpyHostToDevice);
    float* F = (float*)D;
    cp.w = F + o_w; cp.bias = F + o_b;
    hipStream_t s; hipStreamCreate(&s);
    if (launch_repack_bf16x3(cp, F + o_bf, s) != hipSuccess) { printf("repack failed\n"); return 1; }
    hipStreamSynchronize(s);
    hipMemcpy(H.data() + o_bf, D + o_bf, n_bf * 4, hipMemcpyDeviceToHost);
    const hipError_t er = launch_pointwise_bf16x3(F + o_a, cp, F + o_bf, res ? F + o_r : nullptr, F + o_y, P, relu6, s, nullptr, 0, 0);
    const hipError_t es = hipStreamSynchronize(s);
    if (er != hipSuccess || es != hipSuccess) { printf("launch failed: %s / %s\n", hipGetErrorString(er), hipGetErrorString(es)); return 1; }
    std::vector<unsigned> R(H.size());
    hipMemcpy(R.data(), D, R.size() * 4, hipMemcpyDeviceToHost);
    size_t stray = 0, first = 0, unwritten = 0;
    for (size_t i = 0; i < R.size(); ++i) {
        if (i >= o_y && i < o_y + n_y) { if (R[i] == PAT) ++unwritten; continue; }
        if (R[i] != H[i]) { if (!stray) first = i; ++stray; }
    }
    printf("1x1 %d -> %d, %lld rows, relu6 %d residual %d: %zu words changed outside the output (first at word %zu; output = [%zu, %zu)), %zu output words unwritten\n",
           cin, n, P, relu6, res, stray, first, o_y, o_y + n_y, unwritten);
    return stray || unwritten ? 1 : 0;
}
