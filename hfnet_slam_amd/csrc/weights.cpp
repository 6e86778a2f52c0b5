// weights.cpp -- "HFNETW1" container reader, BatchNorm folding and packing of every convolution
// into the layout the gfx950 kernels consume.  Host code only.
//
// What the reference does at this point: parse HF-Net.onnx / load the SavedModel and let the
// runtime lay weights out (src/Extractors/HFNetRTModel.cc:208-254, HFNetTFModelV2.cc:180-202).
// Tensor names / layouts: hfnet_slam_amd/weights.py.
#include "common.hpp"
#include "kernels.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <new>
#include <stdexcept>

namespace hfnet {

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }
int api_exception() noexcept {
    try { throw; }                                            // (called inside a handler: the exception in flight)
    catch (const std::bad_alloc&) { set_error("out of host memory"); }
    catch (const std::exception& e) { set_error("internal error: %s", e.what()); }
    catch (...) { set_error("internal error: unknown C++ exception"); }
    return HFNET_ERR_INTERNAL;
}

// ------------------------------------------------------------------------------------ container
#pragma pack(push, 1)
struct RawEntry { char name[96]; uint32_t ndim; uint32_t dims[4]; uint32_t pad; uint64_t offset; uint64_t nbytes; };
#pragma pack(pop)
static_assert(sizeof(RawEntry) == 136, "container entry layout");

int WeightFile::load(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { set_error("cannot open weight container '%s'", path ? path : "(null)"); return HFNET_ERR_IO; }
    const std::streamsize sz = f.tellg();
    if (sz < 16 || sz > ((std::streamsize)1 << 36)) {         // (a directory opens and reports -1 or LONG_MAX here)
        set_error("'%s' is not an HFNETW1 container (size %lld)", path, (long long)sz); return HFNET_ERR_IO; }
    f.seekg(0);
    blob.resize((size_t)sz);
    if (!f.read((char*)blob.data(), sz)) { set_error("short read on '%s'", path); return HFNET_ERR_IO; }
    if (sz < 16 || std::memcmp(blob.data(), "HFNETW1\0", 8) != 0) { set_error("'%s' is not an HFNETW1 container", path); return HFNET_ERR_IO; }
    uint32_t n;
    std::memcpy(&n, blob.data() + 8, 4);
    if (16 + (size_t)n * sizeof(RawEntry) > blob.size()) { set_error("'%s': truncated table", path); return HFNET_ERR_IO; }
    tensors.clear();
    for (uint32_t i = 0; i < n; ++i) {
        RawEntry e;
        std::memcpy(&e, blob.data() + 16 + (size_t)i * sizeof(RawEntry), sizeof e);
        const size_t table_end = 16 + (size_t)n * sizeof(RawEntry);
        if (e.ndim > 4 || e.offset < table_end || e.offset > blob.size() || e.nbytes > blob.size() - e.offset || (e.offset & 3)) {
            set_error("'%s': bad entry %u", path, i); return HFNET_ERR_IO; }
        HostTensor t;
        e.name[95] = 0;
        t.name = e.name;
        t.ndim = (int)e.ndim;
        size_t count = 1;
        for (int d = 0; d < 4; ++d) { t.dims[d] = (int)e.dims[d]; if (d < t.ndim) count *= e.dims[d]; }
        if (count * 4 != e.nbytes) { set_error("'%s': entry %s size mismatch", path, e.name); return HFNET_ERR_IO; }
        t.data = (const float*)(blob.data() + e.offset);
        tensors.push_back(t);
    }
    return HFNET_OK;
}

const HostTensor* WeightFile::find(const std::string& name) const {
    for (const auto& t : tensors) if (t.name == name) return &t;
    return nullptr;
}

// ------------------------------------------------------------------------------------ packing
namespace {

struct Folded { std::vector<float> scale, shift; };

// slim.batch_norm (inference): y = x * scale + shift; the same float expressions as the oracle
// (oracle/hfnet_oracle.c fold_bn): scale = gamma / sqrt(var + 1e-3), shift = beta - mean*scale.  The caller folds the
// scale into the weights (w * scale, one f32 rounding) and starts the accumulators at the shift.  gamma may be absent for
// ONE scope only: slim.batch_norm defaults to scale=False and the NetVLAD memberships conv is built outside the mobilenet
// arg_scope (hfnet/models/utils/layers.py:71-76), so a real checkpoint has no gamma there.  Anywhere else a missing gamma
// means a truncated or mis-scoped container and is an error (it would silently load as 1).
bool gamma_optional(const std::string& scope) { return scope == "global_head/vlad/memberships"; }

int fold_bn(const WeightFile& wf, const std::string& scope, int c, Folded& out) {
    const HostTensor* g = wf.find(scope + "/BatchNorm/gamma");
    if (!g && !gamma_optional(scope)) { set_error("weights: '%s/BatchNorm/gamma' missing", scope.c_str()); return HFNET_ERR_IO; }
    const HostTensor* b = wf.find(scope + "/BatchNorm/beta");
    const HostTensor* m = wf.find(scope + "/BatchNorm/moving_mean");
    const HostTensor* v = wf.find(scope + "/BatchNorm/moving_variance");
    if (!b || !m || !v || b->dims[0] != c || m->dims[0] != c || v->dims[0] != c || (g && g->dims[0] != c)) {
        set_error("weights: BatchNorm of '%s' missing or mis-sized", scope.c_str()); return HFNET_ERR_IO; }
    out.scale.resize(c);
    out.shift.resize(c);
    for (int i = 0; i < c; ++i) {
        const float s = (g ? g->data[i] : 1.0f) / sqrtf(v->data[i] + 1e-3f);
        const float ms = m->data[i] * s;
        out.scale[i] = s;
        out.shift[i] = b->data[i] - ms;
    }
    return HFNET_OK;
}

int choose_nt(int tiles) {
    int best = 1;
    double best_cost = 1e30;
    for (int nt = 8; nt >= 1; --nt) {
        const int padded = (tiles + nt - 1) / nt * nt;
        const double cost = padded * (1.0 + 0.5 / nt);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = nt; }
    }
    return best;
}

}  // namespace

static int upload(DeviceWeights& dw, const std::vector<float>& h, float** out) {
    void* p = nullptr;
    HF_HIP(dev_malloc(&p, h.size() * sizeof(float)));
    dw.allocations.push_back(p);
    HF_HIP(copy_h2d_blocking(p, h.data(), h.size() * sizeof(float)));
    *out = (float*)p;
    return HFNET_OK;
}

// W: HWIO [taps][cin][n] in logical channel order.  Input channels are consumed in the physical
// order of common.hpp; output columns are emitted in physical order when `out_phys`.
static int pack_conv(DeviceWeights& dw, const float* W, int taps, int cin, int n, const float* scale_l,
                     const float* shift_l, bool out_phys, ConvPack& cp, ConvPack16* p16 = nullptr) {
    if (cin % 8 != 0 || (out_phys && n % 8 != 0)) { set_error("pack_conv: channel count not a multiple of 8 (cin=%d n=%d)", cin, n); return HFNET_ERR_IO; }
    cp.taps = taps; cp.cin = cin; cp.n = n;
    const int tiles = (n + 31) / 32;
    cp.nt_per_block = choose_nt(tiles);
    cp.nt_total = (tiles + cp.nt_per_block - 1) / cp.nt_per_block * cp.nt_per_block;
    const int KQ = taps * cin / 8;
    std::vector<float> w((size_t)KQ * cp.nt_total * 64 * 4, 0.f), sh((size_t)cp.nt_total * 32, 0.f);
    for (int kq = 0; kq < KQ; ++kq) {
        const int tap = (kq * 8) / cin, c0 = (kq * 8) % cin;
        for (int nt = 0; nt < cp.nt_total; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int half = lane >> 5, j = nt * 32 + (lane & 31);
                if (j >= n) continue;
                const int nl = out_phys ? logical_of_phys(j) : j;
                for (int t = 0; t < 4; ++t) {
                    const int cl = c0 + 2 * t + half;   // logical input channel of MFMA t, k = half
                    w[(((size_t)kq * cp.nt_total + nt) * 64 + lane) * 4 + t] = W[((size_t)tap * cin + cl) * n + nl] * scale_l[nl];
                }
            }
    }
    for (int j = 0; j < n; ++j) {
        const int nl = out_phys ? logical_of_phys(j) : j;
        sh[j] = shift_l[nl];
    }
    HF_TRY(upload(dw, w, &cp.w));
    HF_TRY(upload(dw, sh, &cp.bias));
    if (p16 && taps == 1) {
        // the same folded values as 16x16x4 B fragments (ConvPack16, common.hpp); an input width that is not a multiple of 16
        // is padded with zero weights (the kernels pad the activations with zeros: fma(0, 0, acc) == acc)
        const int n16 = (n + 15) / 16, kb16 = (cin + 15) / 16;
        std::vector<float> w16((size_t)kb16 * n16 * 64 * 4, 0.f);
        for (int kb = 0; kb < kb16; ++kb)
            for (int nt = 0; nt < n16; ++nt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = nt * 16 + (lane & 15), g = lane >> 4;
                    if (j >= n) continue;
                    const int nl = out_phys ? logical_of_phys(j) : j;
                    for (int t = 0; t < 4; ++t) {
                        const int k = kb * 16 + 4 * t + g;
                        if (k < cin) w16[(((size_t)kb * n16 + nt) * 64 + lane) * 4 + t] = W[(size_t)k * n + nl] * scale_l[nl];
                    }
                }
        p16->cin = cin; p16->n = n; p16->n16 = n16;
        HF_TRY(upload(dw, w16, &p16->w));
    }
    return HFNET_OK;
}

static int pack_conv_bn(DeviceWeights& dw, const WeightFile& wf, const std::string& scope, bool out_phys, ConvPack& cp, int* n_out,
                        ConvPack16* p16 = nullptr) {
    const HostTensor* w = wf.find(scope + "/weights");
    if (!w || w->ndim != 4) { set_error("weights: '%s/weights' missing", scope.c_str()); return HFNET_ERR_IO; }
    const int taps = w->dims[0] * w->dims[1], cin = w->dims[2], n = w->dims[3];
    Folded f;
    HF_TRY(fold_bn(wf, scope, n, f));
    if (n_out) *n_out = n;
    return pack_conv(dw, w->data, taps, cin, n, f.scale.data(), f.shift.data(), out_phys, cp, p16);
}

// 1x1 conv with biases and no normaliser (hf_net.py:66-72): scale 1 (w * 1.0f is exact), accumulators start at b
static int pack_conv_bias(DeviceWeights& dw, const WeightFile& wf, const std::string& scope, ConvPack& cp, int* n_out) {
    const HostTensor* w = wf.find(scope + "/weights");
    const HostTensor* b = wf.find(scope + "/biases");
    if (!w || !b || w->ndim != 4) { set_error("weights: '%s' missing", scope.c_str()); return HFNET_ERR_IO; }
    const int n = w->dims[3];
    std::vector<float> ones((size_t)n, 1.0f);
    if (n_out) *n_out = n;
    return pack_conv(dw, w->data, w->dims[0] * w->dims[1], w->dims[2], n, ones.data(), b->data, false, cp);
}

static int pack_dw(DeviceWeights& dw, const WeightFile& wf, const std::string& scope, DwPack& dp) {
    const HostTensor* w = wf.find(scope + "/depthwise_weights");
    if (!w || w->ndim != 4 || w->dims[0] != 3 || w->dims[1] != 3) { set_error("weights: '%s/depthwise_weights' missing", scope.c_str()); return HFNET_ERR_IO; }
    const int c = w->dims[2];
    if (c % 8) { set_error("depthwise '%s': %d channels not a multiple of 8", scope.c_str(), c); return HFNET_ERR_IO; }
    Folded f;
    HF_TRY(fold_bn(wf, scope, c, f));
    std::vector<float> wp((size_t)9 * c), sh(c);
    for (int p = 0; p < c; ++p) {
        const int l = logical_of_phys(p);
        for (int t = 0; t < 9; ++t) wp[(size_t)t * c + p] = w->data[(size_t)t * c + l] * f.scale[l];
        sh[p] = f.shift[l];
    }
    dp.c = c;
    HF_TRY(upload(dw, wp, &dp.w));
    HF_TRY(upload(dw, sh, &dp.bias));
    return HFNET_OK;
}

static const int kStrides[17] = {1, 2, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1};  // hf_net.py:31-50

int DeviceWeights::build(const WeightFile& wf) {
    {   // stem: 3x3 s2, 1 -> stem_out (hf_net.py:30)
        const HostTensor* w = wf.find("MobilenetV2/Conv/weights");
        if (!w || w->ndim != 4 || w->dims[0] != 3 || w->dims[1] != 3 || w->dims[2] != 1) { set_error("weights: stem conv missing"); return HFNET_ERR_IO; }
        stem_out = w->dims[3];
        if (stem_out % 8 || stem_out > 64) { set_error("stem width %d unsupported", stem_out); return HFNET_ERR_IO; }
        Folded f;
        HF_TRY(fold_bn(wf, "MobilenetV2/Conv", stem_out, f));
        std::vector<float> wp((size_t)9 * stem_out), sh(stem_out);
        for (int p = 0; p < stem_out; ++p) {
            const int l = logical_of_phys(p);
            for (int t = 0; t < 9; ++t) wp[(size_t)t * stem_out + p] = w->data[(size_t)t * stem_out + l] * f.scale[l];
            sh[p] = f.shift[l];
        }
        HF_TRY(upload(*this, wp, &stem_w));
        HF_TRY(upload(*this, sh, &stem_bias));
    }
    int cin = stem_out;
    for (int i = 0; i < 17; ++i) {
        BlockPack& b = blocks[i];
        const std::string scope = i == 0 ? std::string("MobilenetV2/expanded_conv") : "MobilenetV2/expanded_conv_" + std::to_string(i);
        b.cin = cin;
        b.stride = kStrides[i];
        b.has_expand = wf.find(scope + "/expand/weights") != nullptr;
        b.expand = cin;
        if (b.has_expand) HF_TRY(pack_conv_bn(*this, wf, scope + "/expand", true, b.ex, &b.expand, i >= 2 ? &b.ex16 : nullptr));   // (k_dwproject: layers 9-18; k_block_fused6)
        HF_TRY(pack_dw(*this, wf, scope + "/depthwise", b.dw));
        if (b.dw.c != b.expand) { set_error("block %d: depthwise width %d != expansion %d", i, b.dw.c, b.expand); return HFNET_ERR_IO; }
        HF_TRY(pack_conv_bn(*this, wf, scope + "/project", true, b.pr, &b.cout, i >= 2 ? &b.pr16 : nullptr));   // (k_dwproject: layers 8-18; k_block_fused6)
        if (b.pr.cin != b.expand || (b.has_expand && b.ex.cin != cin)) { set_error("block %d: channel mismatch", i); return HFNET_ERR_IO; }
        b.residual = (b.stride == 1 && b.cin == b.cout);   // conv_blocks.py:304-311
        if (!b.has_expand) {
            const HostTensor* pw = wf.find(scope + "/project/weights");
            Folded pf;
            HF_TRY(fold_bn(wf, scope + "/project", b.cout, pf));
            std::vector<float> wl((size_t)b.expand * b.cout);
            for (int k = 0; k < b.expand; ++k)
                for (int n = 0; n < b.cout; ++n) {
                    const int nl = logical_of_phys(n);
                    wl[(size_t)k * b.cout + n] = pw->data[(size_t)k * b.cout + nl] * pf.scale[nl];
                }
            HF_TRY(upload(*this, wl, &b.pr_logical));
        }
        cin = b.cout;
    }
    c_local = blocks[5].cout;    // layer_7
    c_global = blocks[16].cout;  // layer_18
    int n = 0;
    HF_TRY(pack_conv_bn(*this, wf, "local_head/descriptor/Conv", true, desc1, &n));
    if (n != HFNET_DESC_DIM || desc1.cin != c_local) { set_error("descriptor head shape mismatch"); return HFNET_ERR_IO; }
    HF_TRY(pack_conv_bias(*this, wf, "local_head/descriptor/Conv_1", desc2, &n));
    if (n != HFNET_DESC_DIM) { set_error("descriptor dim %d != 256", n); return HFNET_ERR_IO; }
    HF_TRY(pack_conv_bn(*this, wf, "local_head/detector/Conv", true, det1, &det_hidden));
    HF_TRY(pack_conv_bias(*this, wf, "local_head/detector/Conv_1", det2, &n));
    if (n != 65 || det2.cin != det_hidden) { set_error("detector head shape mismatch"); return HFNET_ERR_IO; }
    {   // engine options desc_bf16x3 / global_bf16x3 / scores_bf16x3: the same folded weights as bf16 hi / lo pieces
        auto split = [&](const ConvPack& cp, void** out) -> int {
            *out = nullptr;
            if (!cp.w || !bf16x3_supported(cp)) return HFNET_OK;
            void* p = nullptr;
            HF_HIP(dev_malloc(&p, bf16x3_pack_bytes(cp)));
            allocations.push_back(p);
            HF_HIP(launch_repack_bf16x3(cp, p, nullptr));
            *out = p;
            return HFNET_OK;
        };
        HF_TRY(split(desc1, &desc1_bf)); HF_TRY(split(desc2, &desc2_bf));
        if (!desc1_bf || !desc2_bf) desc1_bf = desc2_bf = nullptr;
        HF_TRY(split(det1, &det1_bf)); HF_TRY(split(det2, &det2_bf));
        for (int i = 1; i < 17; ++i) {                              // layers 3-18 (3-7: scores_bf16x3, 8-18: global_bf16x3)
            if (blocks[i].has_expand) HF_TRY(split(blocks[i].ex, &blocks[i].ex_bf));
            if (blocks[i].has_expand && blocks[i].ex_bf && (blocks[i].ex.cin / 8) % 2 == 1) {
                void* p = nullptr;
                HF_HIP(dev_malloc(&p, bf16x3_pack_bytes(blocks[i].ex)));
                allocations.push_back(p);
                HF_HIP(launch_repack_bf16x3(blocks[i].ex, p, nullptr, 1));
                blocks[i].ex_bfb = p;
            }
            HF_TRY(split(blocks[i].pr, &blocks[i].pr_bf));
        }
        HF_HIP(hipStreamSynchronize(nullptr));
    }
    HF_TRY(pack_conv_bn(*this, wf, "global_head/vlad/memberships", false, memb, &n_clusters, &memb16));
    if (memb.cin != c_global) { set_error("memberships conv width mismatch"); return HFNET_ERR_IO; }
    if (memb.w && bf16x3_supported(memb)) {
        HF_HIP(dev_malloc(&memb_bf, bf16x3_pack_bytes(memb)));
        allocations.push_back(memb_bf);
        HF_HIP(launch_repack_bf16x3(memb, memb_bf, nullptr));
        HF_HIP(hipStreamSynchronize(nullptr));
    }
    const HostTensor* cl = wf.find("global_head/vlad/clusters");
    const HostTensor* fw = wf.find("global_head/dimensionality_reduction/weights");
    const HostTensor* fb = wf.find("global_head/dimensionality_reduction/biases");
    if (!cl || !fw || !fb || cl->dims[0] != n_clusters || cl->dims[1] != c_global || fw->dims[0] != n_clusters * c_global) {
        set_error("weights: global head tensors missing or mis-sized"); return HFNET_ERR_IO; }
    global_dim = fw->dims[1];
    {
        std::vector<float> c(cl->data, cl->data + (size_t)n_clusters * c_global);
        HF_TRY(upload(*this, c, &clusters));
        // x @ W + b (layers.py:99-107) as one MFMA GEMM over the frames of a batch (FcPack, common.hpp): the oracle's chain
        // b[j] + sum_i x[i] * W[i][j], i ascending
        const int N = n_clusters * c_global, G = global_dim;
        if (N % 16 || G % 16) { set_error("weights: FC %d -> %d: both must be multiples of 16", N, G); return HFNET_ERR_IO; }
        std::vector<float> t((size_t)N * G);
        for (int kg = 0; kg < N / 16; ++kg)
            for (int ct = 0; ct < G / 16; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int tt = 0; tt < 4; ++tt)      // MFMA 4 kg + tt of the chain, k slot lane / 16, column lane % 16
                        t[((((size_t)kg * (G / 16) + ct) * 64) + lane) * 4 + tt] = fw->data[(size_t)(kg * 16 + 4 * tt + lane / 16) * G + ct * 16 + lane % 16];
        fc.n_in = N; fc.n_out = G;
        HF_TRY(upload(*this, t, &fc.w));
        std::vector<float> b(fb->data, fb->data + G);
        HF_TRY(upload(*this, b, &fc.bias));
        if (fc_bf16x3_supported(fc)) {
            HF_HIP(dev_malloc(&fc_bf, fc_bf16x3_pack_bytes(fc)));
            allocations.push_back(fc_bf);
            HF_HIP(launch_repack_fc_bf16x3(fc, fc_bf, nullptr));
            HF_HIP(hipStreamSynchronize(nullptr));
        }
    }
    return HFNET_OK;
}

void DeviceWeights::release() {
    for (void* p : allocations) (void)dev_free(p);
    allocations.clear();
}

// ------------------------------------------------------------------------------------ profiler
int Profiler::id_of(const char* name) {
    for (size_t i = 0; i < names.size(); ++i) if (names[i] == name) return (int)i;
    names.emplace_back(name);
    launches.push_back(0);
    total_ms.push_back(0.0);
    return (int)names.size() - 1;
}
void Profiler::begin(const char* name, hipStream_t s) {
    active = false;
    if (!enabled) return;
    if (!filter.empty() && filter != name) return;
    active = true;
    Rec r;
    r.id = id_of(name);
    auto take = [&]() { hipEvent_t e; if (!pool.empty()) { e = pool.back(); pool.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
    r.a = take();
    r.b = take();
    (void)hipEventRecord(r.a, s);
    pending.push_back(r);
}
void Profiler::end(hipStream_t s) {
    if (!enabled || !active || pending.empty()) return;
    (void)hipEventRecord(pending.back().b, s);
}
void Profiler::flush() {
    for (auto& r : pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            launches[r.id] += 1;
            total_ms[r.id] += ms;
        }
        pool.push_back(r.a);
        pool.push_back(r.b);
    }
    pending.clear();
}
void Profiler::reset() {
    flush();
    names.clear();
    launches.clear();
    total_ms.clear();
}

}  // namespace hfnet
