// common.hpp -- shared declarations of the HIP library (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hfnet_hip.h"

namespace hfnet {

void set_error(const char* fmt, ...);
const char* get_error();
// every extern "C" entry point is a function-try-block that ends in this (called inside the handler): the ABI's "nothing here throws"
// (include/hfnet_hip.h) -- a std::bad_alloc or any other C++ exception becomes HFNET_ERR_INTERNAL with its text in hfnet_last_error()
int api_exception() noexcept;

#define HF_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            ::hfnet::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return HFNET_ERR_DEVICE;                                                              \
        }                                                                                         \
    } while (0)

// every device allocation of the library (devmem.cpp): hipMalloc / hipFree, or -- HFNET_GUARD_ALLOC / HFNET_GUARD_FILL in the
// environment, diagnostics -- page-guarded and / or poisoned allocations that make an out-of-bounds access or a read of
// never-written memory fail in the first test that has it
hipError_t dev_malloc(void** out, size_t bytes);
hipError_t dev_free(void* p);
int dev_guard_mode();
// blocking copies between host memory of any kind and the device through a process-wide pinned block (set-up paths; devmem.cpp)
hipError_t copy_h2d_blocking(void* dst_dev, const void* src_host, size_t bytes);
hipError_t copy_d2h_blocking(void* dst_host, const void* src_dev, size_t bytes);
template <class T> inline hipError_t dev_malloc(T** out, size_t bytes) { return dev_malloc((void**)out, bytes); }

#define HF_TRY(expr)                      \
    do {                                  \
        int s__ = (expr);                 \
        if (s__ != HFNET_OK) return s__;  \
    } while (0)

// ---- device channel layout ------------------------------------------------------------------
// Activations that feed an MFMA convolution are stored with the channels of every group of 8
// permuted: physical slot p of a group holds logical channel {0,2,4,6,1,3,5,7}[p].  A lane of
// v_mfma_f32_32x32x2_f32 then loads 4 consecutive floats (lanes 0-31: slots 0-3, lanes 32-63:
// slots 4-7) and four back-to-back MFMAs consume logical channels (0,1), (2,3), (4,5), (6,7) --
// the accumulation order of the oracle -- without any in-register shuffling.
inline int logical_of_phys(int p) { const int r = p & 7; return (p & ~7) | (r < 4 ? 2 * r : 2 * (r - 4) + 1); }
inline int phys_of_logical(int i) { return (i & ~7) | ((i & 1) << 2) | ((i & 7) >> 1); }

inline int same_out(int in, int stride) { return (in + stride - 1) / stride; }
inline int same_pad_before(int in, int k, int stride) {
    const int o = same_out(in, stride);
    int total = (o - 1) * stride + k - in;
    if (total < 0) total = 0;
    return total / 2;
}
inline int cv_round(float v) { return (int)lrintf(v); }

// ---- geometry handed to spatial kernels (by value) --------------------------------------------
struct LevelGeom {
    int H, W;            // input rows / cols
    int Ho, Wo;          // output rows / cols
    int pt, pl;          // padding before (top / left)
    long long in_off;    // first input pixel of this level's frame 0 in the layer's input tensor
    long long out_off;   // first output pixel of this level's frame 0 in the output tensor
};
struct Geom {
    int n_levels;
    int batch;           // frames per level; image id = level * batch + frame
    LevelGeom lv[HFNET_MAX_LEVELS];
};

// u8 source images of every level (pyramid): frame f of level l starts at ptr[l] + f * frame_stride[l]
struct ImageSet {
    const uint8_t* ptr[HFNET_MAX_LEVELS];
    long long frame_stride[HFNET_MAX_LEVELS];
    int row_stride[HFNET_MAX_LEVELS];
};

// ---- weights ------------------------------------------------------------------------------------
struct HostTensor { std::string name; int ndim; int dims[4]; const float* data; };

struct WeightFile {
    std::vector<unsigned char> blob;
    std::vector<HostTensor> tensors;
    int load(const char* path);
    const HostTensor* find(const std::string& name) const;
};

// one convolution packed for the MFMA kernels.  Inference BatchNorm is folded on the host exactly as the oracle folds it
// (oracle/hfnet_oracle.c fold_bn): w[..., c] *= scale[c], bias[c] = shift[c]; accumulators start at bias[c].
struct ConvPack {
    int taps = 1;        // 1 (1x1) or 9 (3x3)
    int cin = 0;         // channels per tap (multiple of 8)
    int n = 0;           // valid output columns
    int nt_total = 0;    // 32-wide column tiles (padded to a multiple of nt_per_block)
    int nt_per_block = 1;
    float* w = nullptr;      // device: [taps*cin/8][nt_total][64][4]
    float* bias = nullptr;   // device: [nt_total*32], zero padded
};
// x @ W + b (layers.py:99-107) for v_mfma_f32_16x16x4_f32: 16 frames x 16 outputs per wave, one accumulator chain over
// i = 0 .. n_in-1 from b[j] (the oracle's order).  MFMA number m of the chain consumes inputs 4m .. 4m+3, lane (g = lane / 16)
// supplies input 4m + g; a lane loads 16 bytes = its operands of four consecutive MFMAs, so both the activations and the
// weights are stored with the inputs of every group of 16 in slot order: slot 4 g + t holds logical input 4 t + g.
struct FcPack {
    int n_in = 0, n_out = 0;
    float* w = nullptr;      // device: [n_in/16][n_out/16][64 lanes][4]
    float* bias = nullptr;   // device: [n_out]
};
inline int fc_slot_of_logical(int i) { const int r = i & 15; return (i & ~15) | ((r & 3) << 2) | (r >> 2); }
struct DwPack { int c = 0; float* w = nullptr; float* bias = nullptr; };  // [9][C] phys (BN folded)
// a 1x1 convolution packed as v_mfma_f32_16x16x4_f32 B fragments for the single-frame kernels (kernels_tail.hip): MFMA m of
// the chain consumes LOGICAL input channels 4m .. 4m+3 (lane / 16 picks one), a lane's 16 bytes are its operands of four
// consecutive MFMAs, 16-wide column tiles in the PHYSICAL column order of the output tensor.  Same folded values as the
// ConvPack of the layer (its bias array is shared).
struct ConvPack16 { int cin = 0, n = 0, n16 = 0; float* w = nullptr; };   // device: [ceil(cin/16)][n16][64 lanes][4], zero padded
struct BlockPack {
    int cin, expand, stride, cout, residual, has_expand;
    ConvPack ex; DwPack dw; ConvPack pr;
    ConvPack16 pr16, ex16;         // projection / expansion for k_dwproject (layers 8-18 / 9-18)
    float* pr_logical = nullptr;   // projection weights [k logical][n physical] for the vector-ALU layer_2 kernel
    void* ex_bf = nullptr;         // expansion / projection weights split into bf16 pieces (engine options scores_bf16x3: layers 3-7,
    void* pr_bf = nullptr;         // global_bf16x3: layers 8-18)
    void* ex_bfb = nullptr;        // ... the expansion's with the folded bias in the spare k slot (input widths of an odd number of channel groups: 24)
};

struct DeviceWeights {
    int stem_out = 0, c_local = 0, c_global = 0, n_clusters = 0, global_dim = 0, det_hidden = 0;
    float* stem_w = nullptr;      // [9][stem_out] phys order (BN folded)
    float* stem_bias = nullptr;
    BlockPack blocks[17];
    ConvPack desc1, desc2, det1, det2, memb;
    void* desc1_bf = nullptr;     // the descriptor head's weights split into bf16 pieces (launch_repack_bf16x3; engine option desc_bf16x3),
    void* desc2_bf = nullptr;     // null where the shapes do not fit the split-bf16 kernels
    void* det1_bf = nullptr;      // the detector head's (engine option scores_bf16x3)
    void* det2_bf = nullptr;
    void* memb_bf = nullptr;      // the NetVLAD memberships conv's (engine option global_bf16x3)
    ConvPack16 memb16;            // the memberships conv as the "next 1x1" of layer 18's k_dwproject
    float* clusters = nullptr;    // [K][D] logical
    FcPack fc;                    // dimensionality reduction 7680 -> 4096
    void* fc_bf = nullptr;        // its weights as split-bf16 pieces in the activations' memory order (launch_repack_fc_bf16x3; engine option global_bf16x3)
    std::vector<void*> allocations;
    int build(const WeightFile& wf);
    void release();
};

// ---- profiling ------------------------------------------------------------------------------------
struct Profiler {
    bool enabled = false;
    bool active = false;       // the last begin() recorded an event pair
    std::string filter;        // empty = all kernels
    struct Rec { hipEvent_t a, b; int id; };
    std::vector<std::string> names;
    std::vector<int> launches;
    std::vector<double> total_ms;
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    int id_of(const char* name);
    void begin(const char* name, hipStream_t s);
    void end(hipStream_t s);
    void flush();
    void reset();
};

}  // namespace hfnet
