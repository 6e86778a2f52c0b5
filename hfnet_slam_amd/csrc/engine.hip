// engine.hip -- network plan / forward pass over a ragged [level][frame] batch, and the objects
// behind the C ABI.  Host code; every kernel lives in kernels_*.hip.
//
// Reference call stack this replaces:
//   HFextractor::operator() -> ComputePyramid -> per level BaseModel::Detect
//     -> Mat2Tensor, session Run / executeV2, GetLocalFeaturesFromTensor   (src/Extractors/*.cc)
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace hfnet {

int* Options::find(const char* name) {
    if (!name) return nullptr;
    const struct { const char* n; int* p; } tab[] = {{"fuse_blocks", &fuse_blocks}, {"fuse_max_layer", &fuse_max_layer}, {"fused_variant", &fused_variant},
                                                       {"fuse_stem", &fuse_stem}, {"dense_desc", &dense_desc}, {"two_streams", &two_streams},
                                                       {"graph", &graph}, {"pinned_frames", &pinned_frames}, {"db_gemm_min_queries", &db_gemm_min_queries},
                                                       {"conv_wlds", &conv_wlds}, {"fuse_min_wgs", &fuse_min_wgs}, {"copy_threads", &copy_threads}, {"tail_fuse", &tail_fuse}, {"dedupe_taps", &dedupe_taps}, {"pyramid_fuse", &pyramid_fuse}, {"interleave", &interleave}, {"host_global", &host_global}, {"det_fuse", &det_fuse}, {"match_screen_bf16", &match_screen_bf16}, {"tri_screen_bf16", &tri_screen_bf16}, {"desc_bf16x3", &desc_bf16x3}, {"global_bf16x3", &global_bf16x3}};
    for (const auto& t : tab) if (std::strcmp(t.n, name) == 0) return t.p;
    return nullptr;
}

// ------------------------------------------------------------------------------------ memory
int DevMem::ensure(size_t n) {
    if (n <= bytes) return HFNET_OK;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    HF_HIP(hipMalloc(&p, n));
    bytes = n;
    return HFNET_OK;
}
void DevMem::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

Engine::~Engine() {
    (void)hipSetDevice(device);
    prof.flush();
    for (auto ev : prof.pool) (void)hipEventDestroy(ev);
    if (stream) (void)hipStreamSynchronize(stream);          // (the statistics copy of a screened SearchForTriangulation may still be in flight)
    for (DevMem* m : {&m_a, &m_b, &m_s, &m_qn, &m_tn, &m_key, &m_i0, &m_i1, &m_f0, &m_f1, &m_cnt, &m_pairs, &m_tri_stat}) m->release();
    w.release();
    if (h_res) (void)hipHostFree(h_res);
    if (h_tri_stat) (void)hipHostFree(h_tri_stat);
    if (ev_extract) (void)hipEventDestroy(ev_extract);
    if (ev_match) (void)hipEventDestroy(ev_match);
    if (ev_tri_stat) (void)hipEventDestroy(ev_tri_stat);
    if (stream) (void)hipStreamDestroy(stream);
}

bool Engine::pinned_results(size_t bytes) {
    constexpr size_t cap = 1 << 20;
    if (bytes > cap) return false;
    if (!h_res) { void* p = nullptr; if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; } h_res = (unsigned char*)p; }
    return true;
}

hipError_t Engine::note_extract(hipStream_t net_stream) {
    std::lock_guard<std::mutex> lk(ev_mu);
    if (!ev_extract) { hipError_t r = hipEventCreateWithFlags(&ev_extract, hipEventDisableTiming); if (r != hipSuccess) return r; }
    // extractions may come from several extractors (streams): chain them, so that waiting for the latest record
    // implies every earlier one
    if (ev_extract_set) { hipError_t r = hipStreamWaitEvent(net_stream, ev_extract, 0); if (r != hipSuccess) return r; }
    ev_extract_set = true;
    return hipEventRecord(ev_extract, net_stream);
}
hipError_t Engine::wait_extract() {
    std::lock_guard<std::mutex> lk(ev_mu);
    return ev_extract_set ? hipStreamWaitEvent(stream, ev_extract, 0) : hipSuccess;
}
hipError_t Engine::wait_fence(hipStream_t net_stream) {
    std::lock_guard<std::mutex> lk(ev_mu);
    return ev_match_set ? hipStreamWaitEvent(net_stream, ev_match, 0) : hipSuccess;
}

// one polite spin iteration of the host waits on the pinned flags (the pause intrinsic is x86-only)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

template <class T>
static int dalloc(std::vector<void*>& allocs, T** out, size_t count) {
    void* p = nullptr;
    HF_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    allocs.push_back(p);
    *out = (T*)p;
    return HFNET_OK;
}

#define HF_LAUNCH(eng, strm, name, call)                                                   \
    do {                                                                                   \
        hipError_t er__;                                                                   \
        if ((eng)->prof.enabled) {                                                         \
            std::lock_guard<std::mutex> lk__((eng)->prof_mu);                              \
            (eng)->prof.begin(name, strm);                                                 \
            er__ = (call);                                                                 \
            (eng)->prof.end(strm);                                                         \
        } else {                                                                           \
            er__ = (call);                                                                 \
        }                                                                                  \
        if (er__ != hipSuccess) {                                                          \
            set_error("launch %s failed: %s", name, hipGetErrorString(er__));              \
            return HFNET_ERR_DEVICE;                                                       \
        }                                                                                  \
    } while (0)

// ------------------------------------------------------------------------------------ Net
static int layer_channels(const DeviceWeights& w, int layer) { return layer == 1 ? w.stem_out : w.blocks[layer - 2].cout; }

static void compute_offsets(Net& n, int batch) {
    const NetConfig& c = n.cfg;
    for (int L = 1; L <= 18; ++L) {
        long long off = 0;
        const int nl = (L <= 7) ? c.n_levels : 1;
        for (int l = 0; l < HFNET_MAX_LEVELS + 1; ++l) n.pix[L][l] = 0;
        for (int l = 0; l < nl; ++l) { n.pix[L][l] = off; off += (long long)batch * n.lp[l].h[L] * n.lp[l].w[L]; }
        for (int l = nl; l <= HFNET_MAX_LEVELS; ++l) n.pix[L][l] = off;
    }
    long long oi = 0, oc = 0;
    for (int l = 0; l < c.n_levels; ++l) {
        n.pix_img[l] = oi; oi += (long long)batch * n.lp[l].Hc * n.lp[l].Wc;
        n.pix_cell[l] = oc; oc += (long long)batch * n.lp[l].h[7] * n.lp[l].w[7];
    }
    for (int l = c.n_levels; l <= HFNET_MAX_LEVELS; ++l) { n.pix_img[l] = oi; n.pix_cell[l] = oc; }
}

int Net::build(Engine* eng, const NetConfig& c) {
    e = eng;
    cfg = c;
    // A/B and diagnostics switches of the engine (hfnet_engine_set_option), fixed for the lifetime of this network
    fuse_blocks = e->opt.fuse_blocks; fuse_max_layer = e->opt.fuse_max_layer; fused_variant = e->opt.fused_variant; fuse_min_wgs = e->opt.fuse_min_wgs; tail_fuse = e->opt.tail_fuse; dedupe_taps = e->opt.dedupe_taps; interleave = e->opt.interleave; det_fuse = e->opt.det_fuse; desc_bf16x3 = e->opt.desc_bf16x3; global_bf16x3 = e->opt.global_bf16x3;
    force_dense = e->opt.dense_desc; fuse_stem = e->opt.fuse_stem; conv_wlds = e->opt.conv_wlds;
    const DeviceWeights& w = e->w;
    if (c.n_levels < 1 || c.n_levels > HFNET_MAX_LEVELS || c.batch < 1) { set_error("net: bad level / batch count"); return HFNET_ERR_INVALID_ARG; }
    if (c.from_intermediate && (c.n_levels != 1 || !c.global)) { set_error("net: intermediate input needs one level and the global head"); return HFNET_ERR_INVALID_ARG; }
    for (int l = 0; l < c.n_levels; ++l) {
        LevelPlan& p = lp[l];
        p = LevelPlan();
        p.W = c.width[l]; p.H = c.height[l];
        int first = 1;
        if (c.from_intermediate) {
            p.h[7] = p.H; p.w[7] = p.W; first = 8;
        } else {
            p.Hc = p.H / 8 * 8; p.Wc = p.W / 8 * 8;
            if (p.Hc < 8 || p.Wc < 8) { set_error("net: level %d image %dx%d too small", l, p.W, p.H); return HFNET_ERR_SHAPE; }
        }
        for (int L = first; L <= 18; ++L) {
            const int stride = L == 1 ? 2 : w.blocks[L - 2].stride;
            const int ih = L == 1 ? p.Hc : p.h[L - 1], iw = L == 1 ? p.Wc : p.w[L - 1];
            p.h[L] = same_out(ih, stride); p.w[L] = same_out(iw, stride);
            p.pt[L] = same_pad_before(ih, 3, stride); p.pl[L] = same_pad_before(iw, 3, stride);
        }
        if (!c.from_intermediate && (p.h[7] != p.Hc / 8 || p.w[7] != p.Wc / 8)) { set_error("net: unexpected layer_7 size"); return HFNET_ERR_SHAPE; }
    }
    compute_offsets(*this, c.batch);
    HF_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (c.global && c.local) {
        HF_HIP(hipStreamCreateWithFlags(&stream_global, hipStreamNonBlocking));
        HF_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HF_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        two_streams = e->opt.two_streams;
    }
    const int first_layer = c.from_intermediate ? 7 : 1;
    const int last_layer = c.global ? 18 : 7;
    size_t exp_max = 0, dw_max = 0;
    for (int L = first_layer; L <= last_layer; ++L) {
        const bool stem_elided = L == 1 && fuse_stem && fuse_blocks && stem_block_fusable(w.stem_out, w.blocks[0]);
        if (L == 1) stem_elems_max = (size_t)pix[L][HFNET_MAX_LEVELS] * layer_channels(w, L);
        if (!stem_elided) HF_TRY(dalloc(allocs, &act[L], (size_t)pix[L][HFNET_MAX_LEVELS] * layer_channels(w, L)));   // (elided: allocated by the first tap request)
        if (L >= 2 && L > first_layer) {
            const BlockPack& b = w.blocks[L - 2];
            exp_max = std::max(exp_max, (size_t)pix[L - 1][HFNET_MAX_LEVELS] * b.expand);
            dw_max = std::max(dw_max, (size_t)pix[L][HFNET_MAX_LEVELS] * b.expand);
        }
    }
    // layers 8.. only cover level 0, but their input (layer 7) buffer spans all levels: size by level-0 pixels
    HF_TRY(dalloc(allocs, &exp_buf, exp_max));
    HF_TRY(dalloc(allocs, &dw_buf, std::max(dw_max, exp_max)));   // (the single-frame chain ping-pongs expanded tensors between the two)
    if (c.local) {
        const size_t pc = (size_t)pix_cell[HFNET_MAX_LEVELS], pi = (size_t)pix_img[HFNET_MAX_LEVELS];
        HF_TRY(dalloc(allocs, &desc_hidden, pc * HFNET_DESC_DIM));
        HF_TRY(dalloc(allocs, &desc_raw, pc * HFNET_DESC_DIM));
        HF_TRY(dalloc(allocs, &desc_norm, pc * HFNET_DESC_DIM));
        HF_TRY(dalloc(allocs, &det_hidden, pc * w.det_hidden));
        HF_TRY(dalloc(allocs, &logits, pc * 65));
        HF_TRY(dalloc(allocs, &dense, pi));
        HF_TRY(dalloc(allocs, &nms, pi));
        size_t mask_words = 0;                          // one word per (32-row block, column) of every image
        for (int l = 0; l < c.n_levels; ++l) mask_words += (size_t)c.batch * ((lp[l].Hc + 31) / 32) * lp[l].Wc;
        HF_TRY(dalloc(allocs, &nms_mask, mask_words));
        HF_TRY(dalloc(allocs, &nms_flags, mask_words));
        cand_stride = 0;
        for (int l = 0; l < c.n_levels; ++l) cand_stride = std::max(cand_stride, (long long)lp[l].Hc * lp[l].Wc);
        const size_t images = (size_t)c.n_levels * c.batch;
        HF_TRY(dalloc(allocs, &cand, images * (size_t)cand_stride));
        HF_TRY(dalloc(allocs, &counters, images * HFNET_COUNTER_STRIDE));
        HF_TRY(dalloc(allocs, &kps_level, images * (size_t)c.max_keypoints));
        const size_t rows = images * (size_t)c.max_keypoints * 4;
        HF_TRY(dalloc(allocs, &rows_hidden, rows * HFNET_DESC_DIM));
        HF_TRY(dalloc(allocs, &rows_raw, rows * HFNET_DESC_DIM));
        HF_TRY(dalloc(allocs, &n_level, images));
        // distinct tap cells of the sparse descriptor head (launch_tap_cells); the flags start (and are left) clean
        cell_stride = 0;
        for (int l = 0; l < c.n_levels; ++l) cell_stride = std::max(cell_stride, (long long)lp[l].h[7] * lp[l].w[7]);
        HF_TRY(dalloc(allocs, &tap_flags, images * (size_t)cell_stride));
        HF_HIP(hipMemset(tap_flags, 0, images * (size_t)cell_stride));
        HF_TRY(dalloc(allocs, &tap_cell_row, images * (size_t)cell_stride));
        HF_TRY(dalloc(allocs, &tap_cells, rows));
        HF_TRY(dalloc(allocs, &tap_nrows, images));
    }
    if (c.global) {
        const size_t pg = (size_t)c.batch * lp[0].h[18] * lp[0].w[18];
        const size_t N = (size_t)w.n_clusters * w.c_global;
        HF_TRY(dalloc(allocs, &memb, pg * w.n_clusters));
        HF_TRY(dalloc(allocs, &vlad_raw, (size_t)c.batch * N * vlad_scratch_parts()));
        HF_TRY(dalloc(allocs, &vlad_tap, (size_t)c.batch * N));
        HF_TRY(dalloc(allocs, &vlad_out, (size_t)c.batch * N));
        HF_TRY(dalloc(allocs, &fc_raw, (size_t)c.batch * w.global_dim));
        HF_TRY(dalloc(allocs, &fc_part, fc_scratch_floats(w.fc, c.batch)));
        HF_TRY(dalloc(allocs, &global_out, (size_t)c.batch * w.global_dim));
    }
    HF_TRY(dalloc(allocs, &inter_logical, (size_t)c.batch * lp[0].h[7] * lp[0].w[7] * w.c_local));
    return HFNET_OK;
}

void Net::release() {
    for (void* p : allocs) (void)hipFree(p);
    allocs.clear();
    if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    if (stream_global) { (void)hipStreamDestroy(stream_global); stream_global = nullptr; }
    if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
    if (ev_join) { (void)hipEventDestroy(ev_join); ev_join = nullptr; }
}

// layer_in == 0: the cropped input image
Geom Net::geom(int layer_in, int layer_out, int first_level, int n_used) const {
    Geom g;
    std::memset(&g, 0, sizeof g);
    g.n_levels = n_used;
    g.batch = cfg.batch;
    for (int i = 0; i < n_used; ++i) {
        const int l = first_level + i;
        const LevelPlan& p = lp[l];
        LevelGeom& v = g.lv[i];
        v.H = layer_in == 0 ? p.Hc : p.h[layer_in];
        v.W = layer_in == 0 ? p.Wc : p.w[layer_in];
        v.Ho = p.h[layer_out]; v.Wo = p.w[layer_out];
        v.pt = p.pt[layer_out]; v.pl = p.pl[layer_out];
        v.in_off = layer_in == 0 ? 0 : pix[layer_in][l];
        v.out_off = pix[layer_out][l];
    }
    return g;
}

// fused launch: always for the high-resolution layers; for the 30x47 layers only when the batch gives
// the launch enough workgroups to fill the chip (measured: 96 WGs lose to three launches, 384 win)
static bool block_runs_fused(const Net& n, int L) {
    const BlockPack& b = n.e->w.blocks[L - 2];
    bool fuse = n.fuse_blocks && L <= n.fuse_max_layer && block_fusable(b, n.fused_variant);
    if (fuse && L > 7) {
        const LevelPlan& p0 = n.lp[0];
        // (in 128-pixel tiles: 96 lose to three launches, 384 win)
        const long long wgs = (long long)((p0.w[L] + 15) / 16) * ((p0.h[L] + 7) / 8) * n.cfg.batch;
        fuse = wgs >= n.fuse_min_wgs;
    }
    return fuse;
}

static int run_block(Net& n, int L, int n_used, hipStream_t st) {   // layer L = block L-2, input act[L-1]
    Engine* e = n.e;
    const BlockPack& b = e->w.blocks[L - 2];
    const long long p_in = n.pix[L - 1][n_used == 1 ? 1 : HFNET_MAX_LEVELS];
    const long long p_out = n.pix[L][n_used == 1 ? 1 : HFNET_MAX_LEVELS];
    const bool fuse = block_runs_fused(n, L);
    if (fuse) {
        char fn[32];
        snprintf(fn, sizeof fn, "block_L%02d", L);
        const Geom gf = n.geom(L - 1, L, 0, n_used);
        HF_LAUNCH(e, st, fn, launch_block_fused(n.act[L - 1], b, n.act[L], gf, n.fused_variant, st));
        return HFNET_OK;
    }
    const float* src = n.act[L - 1];
    char nm[3][32];
    snprintf(nm[0], sizeof nm[0], "expand_L%02d", L);
    snprintf(nm[1], sizeof nm[1], "depthwise_L%02d", L);
    snprintf(nm[2], sizeof nm[2], "project_L%02d", L);
    // option global_bf16x3 (layers past the index-deciding part of the network only): the two 1x1 convolutions on split-bf16 operands
    const bool bf = n.global_bf16x3 && L > 7 && b.pr_bf && (!b.has_expand || b.ex_bf);
    if (bf) { std::strcat(nm[0], "_bf16x3"); std::strcat(nm[2], "_bf16x3"); }
    if (b.has_expand) {
        if (bf) HF_LAUNCH(e, st, nm[0], launch_pointwise_bf16x3(n.act[L - 1], b.ex, b.ex_bf, nullptr, n.exp_buf, p_in, 1, st));
        else HF_LAUNCH(e, st, nm[0], launch_pointwise(n.act[L - 1], b.ex, nullptr, n.exp_buf, p_in, 1, st));
        src = n.exp_buf;
    }
    const Geom g = n.geom(L - 1, L, 0, n_used);
    HF_LAUNCH(e, st, nm[1], launch_depthwise(src, b.dw, b.stride, n.dw_buf, g, st));
    if (bf) HF_LAUNCH(e, st, nm[2], launch_pointwise_bf16x3(n.dw_buf, b.pr, b.pr_bf, b.residual ? n.act[L - 1] : nullptr, n.act[L], p_out, 0, st));
    else HF_LAUNCH(e, st, nm[2], launch_pointwise(n.dw_buf, b.pr, b.residual ? n.act[L - 1] : nullptr, n.act[L], p_out, 0, st));
    return HFNET_OK;
}

int Net::forward(const ImageSet& imgs, float threshold, const TopkBudget& budget, bool defer_global, bool caller_joins) {
    const DeviceWeights& w = e->w;
    const int NL = cfg.n_levels;
    if (!cfg.from_intermediate) {
        const Geom gs = geom(0, 1, 0, NL);
        int first = 2;
        last_imgs = imgs; stem_valid = false;
        if (fuse_stem && fuse_blocks && stem_block_fusable(w.stem_out, w.blocks[0])) {
            // stem + layer_2 in one launch; the stem tensor (act[1]) is not materialised
            HF_LAUNCH(e, stream, "stem_block_L02", launch_stem_block(imgs, w.stem_w, w.stem_bias, w.blocks[0], act[2], gs,
                                                                   geom(1, 2, 0, NL), stream));
            first = 3;
        } else {
            HF_LAUNCH(e, stream, "stem", launch_stem(imgs, w.stem_w, w.stem_bias, w.stem_out, act[1], gs, stream));
            stem_valid = true;
        }
        for (int L = first; L <= 7; ++L) {
            // the previous step's global branch (deferred join) still reads layer 7 on its own stream
            if (L == 7 && join_pending) { HF_HIP(hipStreamWaitEvent(stream, ev_join, 0)); join_pending = false; }
            HF_TRY(run_block(*this, L, NL, stream));
        }
    }
    if (join_pending) { HF_HIP(hipStreamWaitEvent(stream, ev_join, 0)); join_pending = false; }   // (intermediate-input models)
    // the per-launch profile of EVERY kernel (no name filter) wants each kernel alone on the GPU: one stream for that pass
    const bool fork = cfg.global && cfg.local && two_streams && stream_global && !(e->prof.enabled && e->prof.filter.empty());
    // deferring needs every layer up to 7 fused (the unfused chain shares its scratch tensors with the global branch)
    bool front_fused = fuse_blocks && fuse_max_layer >= 7;
    for (int L = 3; L <= 7 && front_fused; ++L) front_fused = block_fusable(w.blocks[L - 2], fused_variant);
    front_fused = front_fused && (block_fusable(w.blocks[0], fused_variant) || (fuse_stem && stem_block_fusable(w.stem_out, w.blocks[0])));
    const bool defer = defer_global && fork && two_streams == 3 && front_fused;
    // few frames per call: the global branch (a chain of ~40 small launches) is the critical path, the detector conv does not
    // fill the chip -- fork right after layer 7 (0.852 -> 0.840 ms per 752x480 frame)
    const bool fork_early = fork && (two_streams == 1 || (!defer && cfg.batch <= 4));
    // (The fork point is recorded here, but the branch itself is enqueued AFTER the local heads: a captured graph hands its
    //  nodes to the queues in creation order at ~6 us per node, and with the ~35 launches of the global branch first the
    //  local heads of a single frame started 200 us after layer 7 had finished -- both branches ended together at 620 us.)
    // Few frames per call: the global branch is the critical path, and a captured graph hands its nodes to the queues in
    // creation order at ~6 us per node -- whichever branch is captured second starts that much later per node of the first
    // (global first: the local heads of a single frame started 200 us after layer 7; local first: the global branch started
    // 85 us after it).  So the steps of the global branch are enqueued BETWEEN the launches of the local heads.
    int g_next = 0, g_total = 1;
    auto pump_global = [&](int n) -> int {
        if (!fork_early || !interleave || g_next >= g_total) return HFNET_OK;
        HF_TRY(forward_global(stream_global, g_next, n, &g_total));
        g_next += n;
        return HFNET_OK;
    };
    if (fork_early) {
        HF_HIP(hipEventRecord(ev_fork, stream));
        HF_HIP(hipStreamWaitEvent(stream_global, ev_fork, 0));
    }
    if (cfg.local) {
        const long long pc = pix_cell[HFNET_MAX_LEVELS];
        Geom gh = geom(7, 7, 0, NL);
        for (int l = 0; l < NL; ++l) { gh.lv[l].pt = gh.lv[l].pl = 1; gh.lv[l].out_off = pix_cell[l]; }
        HF_LAUNCH(e, stream, "conv3x3_det", launch_conv3x3(act[7], w.det1, det_hidden, 1, gh, conv_wlds, stream));
        HF_TRY(pump_global(interleave));
        if (fork && !fork_early) {
            // the global branch starts after the (chip-filling, MFMA-bound) detector conv: it overlaps the long tail of
            // small kernels (softmax, NMS, top-K, sparse descriptor head) instead of time-sharing with that conv
            HF_HIP(hipEventRecord(ev_fork, stream));
            HF_HIP(hipStreamWaitEvent(stream_global, ev_fork, 0));
            HF_TRY(forward_global(stream_global));
            HF_HIP(hipEventRecord(ev_join, stream_global));
        }
        Geom gd = geom(7, 7, 0, NL);
        for (int l = 0; l < NL; ++l) { gd.lv[l].Ho = lp[l].Hc; gd.lv[l].Wo = lp[l].Wc; gd.lv[l].in_off = pix_cell[l]; gd.lv[l].out_off = pix_img[l]; }
        logits_valid = !(det_fuse && det_tail_supported(w.det2));
        if (!logits_valid) {
            // 1x1 conv + softmax + depth_to_space in one launch: the logits never reach HBM (their tap recomputes them on demand)
            HF_LAUNCH(e, stream, "det_tail", launch_det_tail(det_hidden, w.det2, dense, gd, stream));
            HF_TRY(pump_global(2));
        } else {
            HF_LAUNCH(e, stream, "pointwise_det", launch_pointwise(det_hidden, w.det2, nullptr, logits, pc, 0, stream));
            HF_TRY(pump_global(1));
            HF_LAUNCH(e, stream, "softmax_d2s", launch_softmax_d2s(logits, 65, dense, gd, stream));
            HF_TRY(pump_global(1));
        }
        Geom gn = gd;
        for (int l = 0; l < NL; ++l) { gn.lv[l].H = lp[l].Hc; gn.lv[l].W = lp[l].Wc; gn.lv[l].in_off = pix_img[l]; }
        HF_HIP(hipMemsetAsync(counters, 0, sizeof(unsigned int) * (size_t)NL * cfg.batch * HFNET_COUNTER_STRIDE, stream));
        HF_LAUNCH(e, stream, "nms", launch_nms(dense, nullptr, nms_mask, nms_flags, cand, counters, cand_stride, threshold, gn, stream));
        HF_TRY(pump_global(2));
        HF_LAUNCH(e, stream, "topk", launch_topk(cand, counters, cand_stride, budget, kps_level, cfg.max_keypoints, n_level, gn, stream));
        HF_TRY(pump_global(1));
        // Descriptor head.  Only the 4 bilinear taps of every selected keypoint are ever read
        // (HFNetTFModelV2.cc:153-167), so unless the budget covers most of the cell grid the head is
        // evaluated at those taps only (same arithmetic per cell -> bit-identical descriptors).
        long long tap_rows = 0;
        for (int l = 0; l < NL; ++l) tap_rows += 4ll * std::min(budget.k[l], cfg.max_keypoints) * cfg.batch;
        last_sparse = !force_dense && tap_rows * 5 < pc * 4;
        dense_valid = false;
        nms_valid = false; last_threshold = threshold;
        if (last_sparse) {
            Geom gt = gn;   // H, W: score map; Ho, Wo: cell grid; in_off: first cell of the level
            for (int l = 0; l < NL; ++l) { gt.lv[l].Ho = lp[l].h[7]; gt.lv[l].Wo = lp[l].w[7]; gt.lv[l].in_off = pix_cell[l]; }
            int kmaxb = 0;
            for (int l = 0; l < NL; ++l) kmaxb = std::max(kmaxb, std::min(budget.k[l], cfg.max_keypoints));
            const long long rows = ((long long)(NL * cfg.batch - 1) * cfg.max_keypoints + kmaxb) * 4;
            last_dedupe = dedupe_taps != 0;
            if (last_dedupe) {
                // taps shared by neighbouring keypoints are evaluated once: the rows of an image are its DISTINCT tap cells
                HF_LAUNCH(e, stream, "tap_cells", launch_tap_cells(kps_level, n_level, cfg.max_keypoints, tap_flags, tap_cell_row, tap_cells, tap_nrows, cell_stride, gt, stream));
                HF_TRY(pump_global(1));
                if (desc_bf16x3 && w.desc1_bf && w.desc2_bf) {
                    // option: the head on the bf16 matrix pipe (split operands, three products): tolerance instead of the oracle's bits
                    HF_LAUNCH(e, stream, "conv3x3_desc_taps_bf16x3", launch_conv3x3_cells_bf16x3(act[7], w.desc1, w.desc1_bf, rows_hidden, 1, cfg.max_keypoints, budget.k, gt, tap_cells, tap_nrows, stream));
                    HF_TRY(pump_global(2));
                    HF_LAUNCH(e, stream, "pointwise_desc_taps_bf16x3", launch_pointwise_bf16x3(rows_hidden, w.desc2, w.desc2_bf, nullptr, rows_raw, rows, 0, stream, tap_nrows, 4 * cfg.max_keypoints, 1));
                } else {
                HF_LAUNCH(e, stream, "conv3x3_desc_taps", launch_conv3x3_taps(act[7], w.desc1, rows_hidden, 1, kps_level, n_level, cfg.max_keypoints, budget.k, gt, conv_wlds, stream, tap_cells, tap_nrows));
                HF_TRY(pump_global(2));
                HF_LAUNCH(e, stream, "pointwise_desc_taps", launch_pointwise(rows_hidden, w.desc2, nullptr, rows_raw, rows, 0, stream, tap_nrows, 4 * cfg.max_keypoints, 1));
                }
            } else {
                HF_LAUNCH(e, stream, "conv3x3_desc_taps", launch_conv3x3_taps(act[7], w.desc1, rows_hidden, 1, kps_level, n_level, cfg.max_keypoints, budget.k, gt, conv_wlds, stream));
                HF_LAUNCH(e, stream, "pointwise_desc_taps", launch_pointwise(rows_hidden, w.desc2, nullptr, rows_raw, rows, 0, stream, n_level, 4 * cfg.max_keypoints, 4));
            }
        } else {
            HF_TRY(run_dense_desc());
        }
    }
    if (fork_early) {
        HF_TRY(forward_global(stream_global, g_next, 1 << 20, nullptr));   // whatever is left of the branch
        HF_HIP(hipEventRecord(ev_join, stream_global));
    }
    if (cfg.global && fork_early && caller_joins) join_pending = true;
    else if (cfg.global && fork && defer) join_pending = true;
    else if (cfg.global && fork) HF_HIP(hipStreamWaitEvent(stream, ev_join, 0));
    else if (cfg.global) HF_TRY(forward_global(stream));
    return HFNET_OK;
}

// layers 8-18, NetVLAD, dimensionality reduction on stream st
// few frames per call (what Tracking does): the chain of ~35 small launches of layers 8-18 is the critical path of the call.
// Single-frame kernels (kernels_tail.hip): one launch per block -- depthwise + projection (+ residual) + the NEXT block's
// expansion, the NetVLAD memberships conv after layer 18 -- with every accumulation chain on the short-latency MFMA.
bool Net::tail_chain() const {
    const DeviceWeights& w = e->w;
    if (!tail_fuse || cfg.batch > tail_fuse || !w.memb16.w || w.memb16.cin != w.blocks[16].cout || w.n_clusters > 64) return false;
    for (int L = 8; L <= 18; ++L) {
        const BlockPack& b = w.blocks[L - 2];
        if (!dwproject_supported(b) || block_runs_fused(*this, L) || (L > 8 && (!b.ex16.w || b.ex16.cin != w.blocks[L - 3].cout))) return false;
    }
    return true;
}

int Net::forward_global(hipStream_t st, int first, int count, int* total) {
    const DeviceWeights& w = e->w;
    const int P = lp[0].h[18] * lp[0].w[18];
    // a "step" is one launch group; [first, first + count) are enqueued by this call (forward() interleaves the steps of
    // this branch with the launches of the local heads when both go into one captured graph, see there)
    int step = 0;
    const long long last = (long long)first + count;
#define HF_GSTEP(...)                                          \
    do {                                                       \
        if (step >= first && step < last) { __VA_ARGS__; }     \
        ++step;                                                \
    } while (0)
    bool tail = false;
    if (tail_chain()) {
        // expanded tensors ping-pong between exp_buf and dw_buf (the depthwise tensor itself never exists on this path)
        float* ebuf[2] = {exp_buf, dw_buf};
        HF_GSTEP(HF_LAUNCH(e, st, "expand_L08", launch_pointwise(act[7], w.blocks[6].ex, nullptr, ebuf[0], pix[7][1], 1, st)));
        for (int L = 8; L <= 18; ++L) {
            const BlockPack& b = w.blocks[L - 2];
            const ConvPack16* next = L < 18 ? &w.blocks[L - 1].ex16 : &w.memb16;
            const float* next_bias = L < 18 ? w.blocks[L - 1].ex.bias : w.memb.bias;
            char fn[32];
            snprintf(fn, sizeof fn, "tail_block_L%02d", L);
            HF_GSTEP(HF_LAUNCH(e, st, fn, launch_dwproject(ebuf[L & 1], b, b.residual ? act[L - 1] : nullptr, act[L], next, next_bias,
                                                           L < 18 ? ebuf[(L + 1) & 1] : memb, L < 18 ? 1 : 0, L < 18 ? 0 : 1, geom(L - 1, L, 0, 1), st)));
        }
        tail = true;                                             // (layer 18's launch leaves the SOFTMAXED memberships)
    } else {
        for (int L = 8; L <= 18; ++L) HF_GSTEP(HF_TRY(run_block(*this, L, 1, st)));
        HF_GSTEP(HF_LAUNCH(e, st, "pointwise_memberships", launch_pointwise(act[18], w.memb, nullptr, memb, (long long)cfg.batch * P, 0, st)));
    }
    if (!tail) HF_GSTEP(HF_LAUNCH(e, st, "softmax_memberships", launch_softmax_rows(memb, (long long)cfg.batch * P, w.n_clusters, w.n_clusters, st)));
    HF_GSTEP(HF_LAUNCH(e, st, "vlad", launch_vlad_aggregate(act[18], memb, w.clusters, vlad_raw, cfg.batch, P, w.c_global, w.n_clusters, st));
             HF_LAUNCH(e, st, "vlad_norm", launch_vlad_norm(vlad_raw, vlad_tap, vlad_out, cfg.batch, w.c_global, w.n_clusters, st)));
    HF_GSTEP(HF_LAUNCH(e, st, "fc_l2", launch_fc_l2(vlad_out, w.fc, fc_part, fc_raw, global_dst ? global_dst : global_out, cfg.batch, st, global_host)));
#undef HF_GSTEP
    if (total) *total = step;
    return HFNET_OK;
}

int Net::run_dense_desc() {
    const DeviceWeights& w = e->w;
    const long long pc = pix_cell[HFNET_MAX_LEVELS];
    Geom gh = geom(7, 7, 0, cfg.n_levels);
    for (int l = 0; l < cfg.n_levels; ++l) { gh.lv[l].pt = gh.lv[l].pl = 1; gh.lv[l].out_off = pix_cell[l]; }
    HF_LAUNCH(e, stream, "conv3x3_desc", launch_conv3x3(act[7], w.desc1, desc_hidden, 1, gh, conv_wlds, stream));
    HF_LAUNCH(e, stream, "pointwise_desc", launch_pointwise(desc_hidden, w.desc2, nullptr, desc_raw, pc, 0, stream));
    HF_LAUNCH(e, stream, "l2norm_desc", launch_l2norm256(desc_raw, desc_norm, pc, stream));
    dense_valid = true;
    return HFNET_OK;
}

int Net::tap(int id, std::vector<float>& out) {
    if ((id == 18 || id == 19 || id == 26) && cfg.local && !dense_valid) HF_TRY(run_dense_desc());
    if (id == 25 && cfg.local && !nms_valid) {
        // the product path only needs the candidate list; the suppressed map is produced on demand
        const int NL = cfg.n_levels;
        Geom gn = geom(7, 7, 0, NL);
        for (int l = 0; l < NL; ++l) { gn.lv[l].H = gn.lv[l].Ho = lp[l].Hc; gn.lv[l].W = gn.lv[l].Wo = lp[l].Wc; gn.lv[l].in_off = gn.lv[l].out_off = pix_img[l]; }
        HF_LAUNCH(e, stream, "nms_map", launch_nms(dense, nms, nms_mask, nms_flags, nullptr, nullptr, cand_stride, last_threshold, gn, stream));
        nms_valid = true;
    }
    const DeviceWeights& w = e->w;
    if (id == 21 && cfg.local && !logits_valid) {
        HF_LAUNCH(e, stream, "pointwise_det_tap", launch_pointwise(det_hidden, w.det2, nullptr, logits, pix_cell[HFNET_MAX_LEVELS], 0, stream));
        logits_valid = true;
    }
    if (id == 0 && !cfg.from_intermediate && !stem_valid) {
        // the fused stem + layer_2 kernel never writes the stem tensor: produce it for the tap from the last input
        if (!act[1]) HF_TRY(dalloc(allocs, &act[1], stem_elems_max));
        HF_LAUNCH(e, stream, "stem_tap", launch_stem(last_imgs, w.stem_w, w.stem_bias, w.stem_out, act[1], geom(0, 1, 0, cfg.n_levels), stream));
        stem_valid = true;
    }
    const float* src = nullptr;
    size_t count = 0;
    int permute_c = 0;
    const long long pc = pix_cell[HFNET_MAX_LEVELS], pi = pix_img[HFNET_MAX_LEVELS];
    if (id >= 0 && id <= 17) {
        const int L = id + 1;
        if (!act[L]) { set_error("tap %d not computed by this model", id); return HFNET_ERR_INVALID_ARG; }
        src = act[L]; permute_c = layer_channels(w, L); count = (size_t)pix[L][HFNET_MAX_LEVELS] * permute_c;
    } else if (id == 18 && cfg.local) { src = desc_hidden; permute_c = HFNET_DESC_DIM; count = (size_t)pc * HFNET_DESC_DIM; }
    else if (id == 19 && cfg.local) { src = desc_raw; count = (size_t)pc * HFNET_DESC_DIM; }
    else if (id == 20 && cfg.local) { src = det_hidden; permute_c = w.det_hidden; count = (size_t)pc * w.det_hidden; }
    else if (id == 21 && cfg.local) { src = logits; count = (size_t)pc * 65; }
    else if (id == 22 && cfg.local) { src = dense; count = (size_t)pi; }
    else if (id == 23 && cfg.global) { src = memb; count = (size_t)cfg.batch * lp[0].h[18] * lp[0].w[18] * w.n_clusters; }
    else if (id == 24 && cfg.global) { src = vlad_tap; count = (size_t)cfg.batch * w.n_clusters * w.c_global; }
    else if (id == 25 && cfg.local) { src = nms; count = (size_t)pi; }
    else if (id == 26 && cfg.local) { src = desc_norm; count = (size_t)pc * HFNET_DESC_DIM; }
    else { set_error("unknown or unavailable tap %d", id); return HFNET_ERR_INVALID_ARG; }
    out.resize(count);
    if (permute_c) {
        float* tmp = nullptr;
        HF_HIP(hipMalloc((void**)&tmp, count * sizeof(float)));
        hipError_t er = launch_permute_channels(src, tmp, (long long)(count / permute_c), permute_c, 1, stream);
        if (er == hipSuccess) er = hipMemcpyAsync(out.data(), tmp, count * sizeof(float), hipMemcpyDeviceToHost, stream);
        if (er == hipSuccess) er = hipStreamSynchronize(stream);
        (void)hipFree(tmp);
        if (er != hipSuccess) { set_error("tap copy failed: %s", hipGetErrorString(er)); return HFNET_ERR_DEVICE; }
    } else {
        HF_HIP(hipMemcpyAsync(out.data(), src, count * sizeof(float), hipMemcpyDeviceToHost, stream));
        HF_HIP(hipStreamSynchronize(stream));
    }
    return HFNET_OK;
}

// ------------------------------------------------------------------------------------ tables
// HFextractor ctor (HFextractor.cc:82-119) and ComputePyramid sizes (:159-166)
static void extractor_tables(int nfeatures, int nlevels, float scale_factor, int width, int height, float* sf, int* fpl, int* lw, int* lh) {
    sf[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) sf[i] = sf[i - 1] * scale_factor;
    for (int i = 0; i < nlevels; ++i) {
        const float inv = 1.0f / sf[i];
        lw[i] = i == 0 ? width : cv_round((float)width * inv);
        lh[i] = i == 0 ? height : cv_round((float)height * inv);
    }
    if (nlevels == 1) { fpl[0] = nfeatures; return; }
    const float factor = 1.0f / scale_factor;
    float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        fpl[l] = cv_round(desired);
        sum += fpl[l];
        desired *= factor;
    }
    fpl[nlevels - 1] = std::max(nfeatures - sum, 0);
}

static short sat_short(float v) { const int i = cv_round(v); return (short)std::min(std::max(i, -32768), 32767); }

// coefficient tables of cv::resize(INTER_LINEAR) for CV_8U (OpenCV 4.2 imgproc/src/resize.cpp)
static void resize_tables(int sw, int sh, int dw, int dh, std::vector<int>& xofs, std::vector<short>& ialpha, std::vector<int>& yofs,
                          std::vector<short>& ibeta) {
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
    xofs.resize(dw); ialpha.resize(2 * dw); yofs.resize(dh); ibeta.resize(2 * dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short((1.f - fx) * 2048.f);
        ialpha[2 * dx + 1] = sat_short(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat_short((1.f - fy) * 2048.f);
        ibeta[2 * dy + 1] = sat_short(fy * 2048.f);
    }
}

}  // namespace hfnet

// ================================================================================================ C ABI
using namespace hfnet;

#define API_GUARD(ptr, what)                                               \
    do {                                                                   \
        if (!(ptr)) { set_error(what " is null"); return HFNET_ERR_INVALID_ARG; } \
    } while (0)

extern "C" {

const char* hfnet_last_error(void) { return get_error(); }
int hfnet_abi_version(void) { return HFNET_ABI_VERSION; }
#ifndef HFNET_BUILD_ID
#define HFNET_BUILD_ID "hfnet-build-id:unknown"
#endif
// (the marker prefix lets hfnet_slam_amd/build.py read the id from the file without loading it)
const char* hfnet_build_id(void) { static const char id[] = HFNET_BUILD_ID; return &id[sizeof("hfnet-build-id:") - 1]; }

int hfnet_device_count(void) {
    int n = 0;
    const hipError_t er = hipGetDeviceCount(&n);
    if (er != hipSuccess || n <= 0) { set_error("no HIP device visible (%s)", hipGetErrorString(er)); return 0; }
    return n;
}

int hfnet_engine_create(int device, const char* weights_path, hfnet_engine** out) {
    API_GUARD(out, "out");
    *out = nullptr;
    API_GUARD(weights_path, "weights_path");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible: libhfnet_hip needs a gfx950 GPU"); return HFNET_ERR_DEVICE; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (0..%d)", device, ndev - 1); return HFNET_ERR_INVALID_ARG; }
    HF_HIP(hipSetDevice(device));
    WeightFile wf;
    HF_TRY(wf.load(weights_path));
    std::unique_ptr<hfnet_engine> e(new hfnet_engine());
    e->impl.device = device;
    HF_HIP(hipStreamCreateWithFlags(&e->impl.stream, hipStreamNonBlocking));
    HF_TRY(e->impl.w.build(wf));
    HF_HIP(hipDeviceSynchronize());
    *out = e.release();
    return HFNET_OK;
}

void hfnet_engine_destroy(hfnet_engine* e) { delete e; }

int hfnet_engine_info(const hfnet_engine* e, int what) {
    if (!e) return -1;
    const DeviceWeights& w = e->impl.w;
    switch (what) {
        case 0: return w.stem_out;
        case 1: return w.c_local;
        case 2: return w.c_global;
        case 3: return w.n_clusters;
        case 4: return w.global_dim;
        case 5: return e->impl.device;
        default: return -1;
    }
}

int hfnet_engine_set_option(hfnet_engine* e, const char* name, int value) {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.mu);              // (the database / matcher entry points read options under this lock)
    int* p = e->impl.opt.find(name);
    if (!p) { set_error("unknown engine option '%s'", name ? name : "(null)"); return HFNET_ERR_INVALID_ARG; }
    if (value < 0) { set_error("engine option '%s': negative value %d", name, value); return HFNET_ERR_INVALID_ARG; }
    *p = value;
    if (std::strcmp(name, "tri_screen_bf16") == 0) {          // (writing the option also forgets what earlier calls found: Engine::tri_skip)
        e->impl.tri_skip = 0;
        if (e->impl.h_tri_stat) { (void)hipStreamSynchronize(e->impl.stream); e->impl.tri_stat_pending = false; }
    }
    return HFNET_OK;
}
int hfnet_engine_get_option(hfnet_engine* e, const char* name, int* value) {
    API_GUARD(e, "engine"); API_GUARD(value, "value");
    std::lock_guard<std::mutex> lk(e->impl.mu);
    const int* p = e->impl.opt.find(name);
    if (!p) { set_error("unknown engine option '%s'", name ? name : "(null)"); return HFNET_ERR_INVALID_ARG; }
    *value = *p;
    return HFNET_OK;
}

int hfnet_engine_synchronize(hfnet_engine* e) {
    API_GUARD(e, "engine");
    HF_HIP(hipSetDevice(e->impl.device));
    HF_HIP(hipDeviceSynchronize());
    return HFNET_OK;
}

int hfnet_engine_fence(hfnet_engine* eh) {
    API_GUARD(eh, "engine");
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    std::lock_guard<std::mutex> lk2(e.ev_mu);
    if (!e.ev_match) HF_HIP(hipEventCreateWithFlags(&e.ev_match, hipEventDisableTiming));
    HF_HIP(hipEventRecord(e.ev_match, e.stream));
    e.ev_match_set = true;
    return HFNET_OK;
}

// ---------------------------------------------------------------------------------------- BaseModel
int hfnet_model_create(hfnet_engine* e, hfnet_mode mode, int height, int width, int max_keypoints, hfnet_model** out) {
    API_GUARD(out, "out");
    *out = nullptr;
    API_GUARD(e, "engine");
    if (mode < HFNET_IMAGE_TO_LOCAL_AND_GLOBAL || mode > HFNET_INTERMEDIATE_TO_GLOBAL) { set_error("unknown mode %d", (int)mode); return HFNET_ERR_INVALID_ARG; }
    if (height <= 0 || width <= 0) { set_error("bad input shape %dx%d", width, height); return HFNET_ERR_SHAPE; }
    if (max_keypoints < 1) max_keypoints = 1;
    if (max_keypoints > HFNET_MAX_KEYPOINTS) { set_error("max_keypoints %d > %d", max_keypoints, HFNET_MAX_KEYPOINTS); return HFNET_ERR_CAPACITY; }
    HF_HIP(hipSetDevice(e->impl.device));
    std::unique_ptr<hfnet_model> m(new hfnet_model());
    m->eng = e; m->mode = mode; m->height = height; m->width = width; m->max_keypoints = max_keypoints;
    NetConfig c;
    c.n_levels = 1; c.width[0] = width; c.height[0] = height; c.batch = 1; c.max_keypoints = max_keypoints;
    c.local = mode != HFNET_INTERMEDIATE_TO_GLOBAL;
    c.global = mode == HFNET_IMAGE_TO_LOCAL_AND_GLOBAL || mode == HFNET_INTERMEDIATE_TO_GLOBAL;
    c.from_intermediate = mode == HFNET_INTERMEDIATE_TO_GLOBAL;
    HF_TRY(m->net.build(&e->impl, c));
    if (c.local) {
        HF_TRY(dalloc(m->net.allocs, &m->d_image, (size_t)height * width));
        HF_TRY(dalloc(m->net.allocs, &m->d_kps, (size_t)max_keypoints));
        HF_TRY(dalloc(m->net.allocs, &m->d_desc, (size_t)max_keypoints * HFNET_DESC_DIM));
        HF_TRY(dalloc(m->net.allocs, &m->d_n, 2));
    }
    m->valid = true;
    *out = m.release();
    return HFNET_OK;
}

void hfnet_model_destroy(hfnet_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->eng->impl.device);
    delete m;
}

int hfnet_model_is_valid(const hfnet_model* m) { return m && m->valid ? 1 : 0; }
int hfnet_model_mode(const hfnet_model* m) { return m ? (int)m->mode : -1; }

int hfnet_model_detect(hfnet_model* m, const uint8_t* image, int row_stride, int n_keypoints, float threshold, hfnet_keypoint* kps,
                       float* local_desc, float* aux, int* n_out) {
    API_GUARD(m, "model");
    if (n_out) *n_out = 0;
    if (!m->valid) { set_error("model is not valid"); return HFNET_ERR_INVALID_ARG; }
    if (m->mode == HFNET_INTERMEDIATE_TO_GLOBAL) { set_error("Detect(image, ...) called on an IntermediateToGlobal model"); return HFNET_ERR_WRONG_MODE; }
    if ((m->mode == HFNET_IMAGE_TO_LOCAL) != (aux == nullptr)) {
        // the 5-argument overload only exists for kImageToLocal, the 6-argument one for the other two (HFNetTFModelV2.cc:65,81)
        set_error("Detect overload does not match the model mode"); return HFNET_ERR_WRONG_MODE; }
    API_GUARD(image, "image"); API_GUARD(kps, "kps"); API_GUARD(local_desc, "local_desc"); API_GUARD(n_out, "n_out");
    if (row_stride < m->width) { set_error("row_stride %d < width %d", row_stride, m->width); return HFNET_ERR_SHAPE; }
    if (n_keypoints < 0 || n_keypoints > m->max_keypoints) { set_error("n_keypoints %d outside [0, %d]", n_keypoints, m->max_keypoints); return HFNET_ERR_CAPACITY; }
    std::lock_guard<std::mutex> lk(m->mu);
    Net& net = m->net;
    HF_HIP(hipSetDevice(m->eng->impl.device));
    HF_HIP(hipMemcpy2DAsync(m->d_image, m->width, image, row_stride, m->width, m->height, hipMemcpyHostToDevice, net.stream));
    ImageSet imgs;
    std::memset(&imgs, 0, sizeof imgs);
    imgs.ptr[0] = m->d_image; imgs.row_stride[0] = m->width; imgs.frame_stride[0] = (long long)m->width * m->height;
    TopkBudget budget;
    std::memset(&budget, 0, sizeof budget);
    budget.k[0] = n_keypoints;
    HF_TRY(net.forward(imgs, threshold, budget));
    SampleArgs sa;
    std::memset(&sa, 0, sizeof sa);
    sa.desc_map = net.sample_source(); sa.sparse = net.last_sparse ? 1 : 0; sa.cell_row = net.last_sparse && net.last_dedupe ? net.tap_cell_row : nullptr; sa.cell_stride = net.cell_stride; sa.kps_in = net.kps_level; sa.n_in = net.n_level; sa.kps_stride = net.cfg.max_keypoints;
    sa.kps_out = m->d_kps; sa.desc_out = m->d_desc; sa.n_out_frame = m->d_n; sa.n_out_level = nullptr;
    sa.out_frame_stride = m->max_keypoints; sa.scale_factor[0] = 1.0f; sa.set_octave = 0;
    Geom gs = net.geom(7, 7, 0, 1);
    gs.lv[0].H = net.lp[0].Hc; gs.lv[0].W = net.lp[0].Wc; gs.lv[0].Ho = net.lp[0].h[7]; gs.lv[0].Wo = net.lp[0].w[7];
    gs.lv[0].in_off = net.pix_cell[0];
    HF_LAUNCH(&m->eng->impl, net.stream, "sample", launch_sample(sa, gs, net.stream));
    int n = 0;
    HF_HIP(hipMemcpyAsync(&n, m->d_n, sizeof(int), hipMemcpyDeviceToHost, net.stream));
    if (m->mode == HFNET_IMAGE_TO_LOCAL_AND_GLOBAL) {
        HF_HIP(hipMemcpyAsync(aux, net.global_out, sizeof(float) * m->eng->impl.w.global_dim, hipMemcpyDeviceToHost, net.stream));
    } else if (m->mode == HFNET_IMAGE_TO_LOCAL_AND_INTERMEDIATE) {
        const long long P = (long long)net.lp[0].h[7] * net.lp[0].w[7];
        const int C = m->eng->impl.w.c_local;
        HF_LAUNCH(&m->eng->impl, net.stream, "permute", launch_permute_channels(net.act[7], net.inter_logical, P, C, 1, net.stream));
        HF_HIP(hipMemcpyAsync(aux, net.inter_logical, sizeof(float) * P * C, hipMemcpyDeviceToHost, net.stream));
    }
    HF_HIP(hipStreamSynchronize(net.stream));
    if (n > 0) {
        HF_HIP(hipMemcpyAsync(kps, m->d_kps, sizeof(hfnet_keypoint) * n, hipMemcpyDeviceToHost, net.stream));
        HF_HIP(hipMemcpyAsync(local_desc, m->d_desc, sizeof(float) * HFNET_DESC_DIM * n, hipMemcpyDeviceToHost, net.stream));
        HF_HIP(hipStreamSynchronize(net.stream));
    }
    *n_out = n;
    return HFNET_OK;
}

int hfnet_model_detect_global(hfnet_model* m, const float* intermediate, float* global_desc) {
    API_GUARD(m, "model");
    if (!m->valid) { set_error("model is not valid"); return HFNET_ERR_INVALID_ARG; }
    if (m->mode != HFNET_INTERMEDIATE_TO_GLOBAL) { set_error("Detect(intermediate, global) called on an image model"); return HFNET_ERR_WRONG_MODE; }
    API_GUARD(intermediate, "intermediate"); API_GUARD(global_desc, "global_desc");
    std::lock_guard<std::mutex> lk(m->mu);
    Net& net = m->net;
    Engine& eng = m->eng->impl;
    HF_HIP(hipSetDevice(eng.device));
    const long long P = (long long)m->height * m->width;
    const int C = eng.w.c_local;
    HF_HIP(hipMemcpyAsync(net.inter_logical, intermediate, sizeof(float) * P * C, hipMemcpyHostToDevice, net.stream));
    HF_LAUNCH(&eng, net.stream, "permute", launch_permute_channels(net.inter_logical, net.act[7], P, C, 0, net.stream));
    ImageSet imgs;
    std::memset(&imgs, 0, sizeof imgs);
    TopkBudget budget;
    std::memset(&budget, 0, sizeof budget);
    HF_TRY(net.forward(imgs, 0.f, budget));
    HF_HIP(hipMemcpyAsync(global_desc, net.global_out, sizeof(float) * eng.w.global_dim, hipMemcpyDeviceToHost, net.stream));
    HF_HIP(hipStreamSynchronize(net.stream));
    return HFNET_OK;
}

int hfnet_model_tap(hfnet_model* m, int tap, float* out, size_t capacity, size_t* count) {
    API_GUARD(m, "model"); API_GUARD(out, "out"); API_GUARD(count, "count");
    std::lock_guard<std::mutex> lk(m->mu);
    HF_HIP(hipSetDevice(m->eng->impl.device));
    std::vector<float> v;
    HF_TRY(m->net.tap(tap, v));
    *count = v.size();
    if (v.size() > capacity) { set_error("tap %d needs %zu floats, buffer holds %zu", tap, v.size(), capacity); return HFNET_ERR_CAPACITY; }
    std::memcpy(out, v.data(), v.size() * sizeof(float));
    return HFNET_OK;
}

// ---------------------------------------------------------------------------------------- HFextractor
int hfnet_extractor_create(hfnet_engine* e, int width, int height, int n_features, float threshold, float scale_factor, int n_levels,
                           int max_batch, hfnet_extractor** out) {
    API_GUARD(out, "out");
    *out = nullptr;
    API_GUARD(e, "engine");
    if (n_levels < 1 || n_levels > HFNET_MAX_LEVELS) { set_error("n_levels %d outside [1, %d]", n_levels, HFNET_MAX_LEVELS); return HFNET_ERR_INVALID_ARG; }
    if (width <= 0 || height <= 0 || n_features < 1 || max_batch < 1 || !(scale_factor >= 1.0f)) { set_error("bad extractor parameters"); return HFNET_ERR_INVALID_ARG; }
    if (n_features > HFNET_MAX_KEYPOINTS) { set_error("n_features %d > %d", n_features, HFNET_MAX_KEYPOINTS); return HFNET_ERR_CAPACITY; }
    HF_HIP(hipSetDevice(e->impl.device));
    std::unique_ptr<hfnet_extractor> x(new hfnet_extractor());
    x->eng = e; x->width = width; x->height = height; x->n_features = n_features; x->n_levels = n_levels; x->max_batch = max_batch;
    x->threshold = threshold; x->scale_factor = scale_factor;
    extractor_tables(n_features, n_levels, scale_factor, width, height, x->scale_factors, x->features_per_level, x->level_w, x->level_h);
    {   // the per-level model shapes of InitAllModels (BaseModel.cc:33-63) must agree with the pyramid sizes
        float scale = 1.0f;
        for (int l = 0; l < n_levels; ++l) {
            const int mh = cv_round(height * scale), mw = cv_round(width * scale);
            if (mh != x->level_h[l] || mw != x->level_w[l]) {
                set_error("level %d: pyramid size %dx%d differs from the model shape %dx%d the reference would build", l, x->level_w[l], x->level_h[l], mw, mh);
                return HFNET_ERR_SHAPE; }
            scale /= scale_factor;
        }
    }
    NetConfig c;
    c.n_levels = n_levels; c.batch = max_batch; c.local = true; c.global = true; c.from_intermediate = false;
    x->use_graph = e->impl.opt.graph;
    x->host_global = e->impl.opt.host_global;
    c.max_keypoints = 1;
    for (int l = 0; l < n_levels; ++l) { c.width[l] = x->level_w[l]; c.height[l] = x->level_h[l]; c.max_keypoints = std::max(c.max_keypoints, x->features_per_level[l]); }
    HF_TRY(x->net.build(&e->impl, c));
    for (int l = 0; l < n_levels; ++l) {
        HF_TRY(dalloc(x->allocs, &x->d_pyr[l], (size_t)max_batch * ((x->level_w[l] + 3) & ~3) * x->level_h[l]));   // levels >= 1: rows padded to 4 bytes
        if (l == 0) continue;
        std::vector<int> xofs, yofs;
        std::vector<short> ia, ib;
        resize_tables(x->level_w[l - 1], x->level_h[l - 1], x->level_w[l], x->level_h[l], xofs, ia, yofs, ib);
        HF_TRY(dalloc(x->allocs, &x->d_xofs[l], xofs.size()));
        HF_TRY(dalloc(x->allocs, &x->d_ialpha[l], ia.size()));
        HF_TRY(dalloc(x->allocs, &x->d_yofs[l], yofs.size()));
        HF_TRY(dalloc(x->allocs, &x->d_ibeta[l], ib.size()));
        HF_HIP(hipMemcpy(x->d_xofs[l], xofs.data(), xofs.size() * sizeof(int), hipMemcpyHostToDevice));
        HF_HIP(hipMemcpy(x->d_ialpha[l], ia.data(), ia.size() * sizeof(short), hipMemcpyHostToDevice));
        HF_HIP(hipMemcpy(x->d_yofs[l], yofs.data(), yofs.size() * sizeof(int), hipMemcpyHostToDevice));
        HF_HIP(hipMemcpy(x->d_ibeta[l], ib.data(), ib.size() * sizeof(short), hipMemcpyHostToDevice));
    }
    HF_TRY(dalloc(x->allocs, &x->d_kps, (size_t)max_batch * n_features));
    HF_TRY(dalloc(x->allocs, &x->d_desc, (size_t)max_batch * n_features * HFNET_DESC_DIM));
    HF_TRY(dalloc(x->allocs, &x->d_n, (size_t)max_batch));
    HF_TRY(dalloc(x->allocs, &x->d_n_level, (size_t)max_batch * n_levels));
    x->last_n.assign((size_t)max_batch, -1);
    {   // pinned block of the latency path (see hfnet_extractor::h_pin); without it the pageable path is used
        const int pf = std::min(max_batch, e->impl.opt.pinned_frames);
        if (pf > 0) {
            auto up = [](size_t b) { return (b + 255) / 256 * 256; };
            size_t off = up((size_t)pf * width * height);
            x->pin_res = off;
            const size_t res_bytes = x->result_offsets(pf, e->impl.w.global_dim).total;
            off += res_bytes;
            x->pin_flag = off;
            off += 256;
            HF_TRY(dalloc(x->allocs, &x->d_blk, res_bytes));
            HF_TRY(dalloc(x->allocs, &x->d_seq, 3));
            HF_HIP(hipMemset(x->d_seq, 0, 3 * sizeof(int)));
            void* hp = nullptr;
            // (coherent: kernels write results and the call's number into this block while the host spins on it mid-graph)
            if (hipHostMalloc(&hp, off, hipHostMallocCoherent) == hipSuccess) { x->h_pin = (unsigned char*)hp; x->pinned_frames = pf; *(volatile int*)(x->h_pin + x->pin_flag) = 0; *(volatile int*)(x->h_pin + x->pin_flag + 128) = 0; }
            else (void)hipGetLastError();
        }
    }
    *out = x.release();
    return HFNET_OK;
}

void hfnet_extractor_destroy(hfnet_extractor* x) {
    if (!x) return;
    (void)hipSetDevice(x->eng->impl.device);
    // (a single-frame call returns when its results are in the caller's buffers, which is before its graph has retired)
    if (x->net.stream) (void)hipStreamSynchronize(x->net.stream);
    if (x->net.stream_global) (void)hipStreamSynchronize(x->net.stream_global);
    for (auto& kv : x->graphs) (void)hipGraphExecDestroy(kv.second);
    for (void* p : x->allocs) (void)hipFree(p);
    if (x->h_pin) (void)hipHostFree(x->h_pin);
    for (int s = 0; s < 2; ++s) {
        if (x->pipe.h_in[s]) (void)hipHostFree(x->pipe.h_in[s]);
        if (x->pipe.h_out[s]) (void)hipHostFree(x->pipe.h_out[s]);
        for (hipEvent_t ev : {x->pipe.ev_up[s], x->pipe.ev_comp[s], x->pipe.ev_down[s]}) if (ev) (void)hipEventDestroy(ev);
    }
    if (x->pipe.s_up) (void)hipStreamDestroy(x->pipe.s_up);
    if (x->pipe.s_down) (void)hipStreamDestroy(x->pipe.s_down);
    delete x;
}

int hfnet_extractor_tables(const hfnet_extractor* x, float* scale_factors, int* features_per_level, int* level_width, int* level_height) {
    API_GUARD(x, "extractor");
    for (int l = 0; l < x->n_levels; ++l) {
        if (scale_factors) scale_factors[l] = x->scale_factors[l];
        if (features_per_level) features_per_level[l] = x->features_per_level[l];
        if (level_width) level_width[l] = x->level_w[l];
        if (level_height) level_height[l] = x->level_h[l];
    }
    return HFNET_OK;
}

// one chunk of nb <= max_batch frames; all pointers device pointers except when host_* is given
static int extract_chunk(hfnet_extractor* x, int nb, const uint8_t* d_images, int row_stride, long long frame_stride, hfnet_keypoint* d_kps,
                         float* d_desc, float* d_global, int* d_n, int* d_n_level, bool caller_joins = false) {
    Net& net = x->net;
    Engine& eng = x->eng->impl;
    if (net.cfg.batch != nb) { net.cfg.batch = nb; compute_offsets(net, nb); }
    ImageSet imgs;
    std::memset(&imgs, 0, sizeof imgs);
    imgs.ptr[0] = d_images; imgs.row_stride[0] = row_stride; imgs.frame_stride[0] = frame_stride;
    // calls of a few frames: the pyramid chain as ONE launch (three dependent 7 us launches otherwise)
    const bool chain = nb <= 4 && eng.opt.pyramid_fuse && x->n_levels >= 2 && pyramid_chain_supported(x->n_levels - 1, x->level_w, x->level_h);
    if (chain) {
        uint8_t* dst[HFNET_MAX_LEVELS] = {nullptr};
        int d_row[HFNET_MAX_LEVELS] = {0};
        long long d_frame[HFNET_MAX_LEVELS] = {0};
        for (int l = 1; l < x->n_levels; ++l) {
            const int dwp = (x->level_w[l] + 3) & ~3;
            dst[l] = x->d_pyr[l]; d_row[l] = dwp; d_frame[l] = (long long)dwp * x->level_h[l];
            imgs.ptr[l] = x->d_pyr[l]; imgs.row_stride[l] = dwp; imgs.frame_stride[l] = d_frame[l];
        }
        HF_LAUNCH(&eng, net.stream, "pyramid_resize",
                  launch_pyramid_chain(d_images, row_stride, frame_stride, x->n_levels - 1, x->level_w, x->level_h, dst, d_row, d_frame, x->d_xofs,
                                       x->d_ialpha, x->d_yofs, x->d_ibeta, nb, net.stream));
    }
    for (int l = 1; l < x->n_levels && !chain; ++l) {
        const int sw = x->level_w[l - 1], sh = x->level_h[l - 1], dw = x->level_w[l], dh = x->level_h[l];
        const int dwp = (dw + 3) & ~3;              // pyramid rows are padded to 4 bytes (packed stores)
        HF_LAUNCH(&eng, net.stream, "pyramid_resize",
                  launch_resize_u8(imgs.ptr[l - 1], sw, sh, imgs.row_stride[l - 1], imgs.frame_stride[l - 1], x->d_pyr[l], dw, dh, dwp,
                                   (long long)dwp * dh, x->d_xofs[l], x->d_ialpha[l], x->d_yofs[l], x->d_ibeta[l], nb, net.stream));
        imgs.ptr[l] = x->d_pyr[l]; imgs.row_stride[l] = dwp; imgs.frame_stride[l] = (long long)dwp * dh;
    }
    TopkBudget budget;
    std::memset(&budget, 0, sizeof budget);
    for (int l = 0; l < x->n_levels; ++l) budget.k[l] = x->features_per_level[l];
    const bool defer = d_n_level == nullptr;      // device-resident call: nothing of the global branch is needed on this stream
    HF_TRY(net.forward(imgs, x->threshold, budget, defer, caller_joins));
    SampleArgs sa;
    std::memset(&sa, 0, sizeof sa);
    sa.desc_map = net.sample_source(); sa.sparse = net.last_sparse ? 1 : 0; sa.cell_row = net.last_sparse && net.last_dedupe ? net.tap_cell_row : nullptr; sa.cell_stride = net.cell_stride; sa.kps_in = net.kps_level; sa.n_in = net.n_level; sa.kps_stride = net.cfg.max_keypoints;
    sa.kps_out = d_kps; sa.desc_out = d_desc; sa.n_out_frame = d_n; sa.n_out_level = d_n_level;
    sa.out_frame_stride = x->n_features; sa.set_octave = 1;
    for (int l = 0; l < x->n_levels; ++l) sa.scale_factor[l] = x->scale_factors[l];
    Geom gs = net.geom(7, 7, 0, x->n_levels);
    for (int l = 0; l < x->n_levels; ++l) {
        gs.lv[l].H = net.lp[l].Hc; gs.lv[l].W = net.lp[l].Wc; gs.lv[l].Ho = net.lp[l].h[7]; gs.lv[l].Wo = net.lp[l].w[7];
        gs.lv[l].in_off = net.pix_cell[l];
    }
    HF_LAUNCH(&eng, net.stream, "sample", launch_sample(sa, gs, net.stream));
    if (d_global)
        HF_HIP(hipMemcpyAsync(d_global, net.global_out, sizeof(float) * (size_t)nb * eng.w.global_dim, hipMemcpyDeviceToDevice,
                              net.join_pending ? net.stream_global : net.stream));
    return HFNET_OK;
}

// host-pointer latency path (chunks of up to pinned_frames frames through the pinned block): the chunk's copies and launches
// always use the extractor's own staging buffers, so they are captured once per chunk size into a graph (both streams: the global branch forks and joins inside it) and replayed afterwards
static int extract_chunk_graphed(hfnet_extractor* x, int nb) {
    Engine& eng = x->eng->impl;
    Net& net = x->net;
    hipStream_t st = net.stream;
    const int G = eng.w.global_dim;
    auto direct = [&]() -> int {
        const size_t img_bytes = (size_t)x->width * x->height;
        HF_HIP(hipMemcpyAsync(x->d_pyr[0], x->h_pin, img_bytes * nb, hipMemcpyHostToDevice, st));
        // results of the whole chunk at full capacity (sizes are static) into ONE device block laid out like the pinned one
        const hfnet_extractor::ResOff o = x->result_offsets(nb, G);
        net.global_dst = net.cfg.global ? (float*)(x->d_blk + o.g) : nullptr;
        const bool host_global = x->global_to_host(nb);
        if (host_global) net.global_host = FcHostOut{(float*)(x->h_pin + x->pin_res + o.g), (int*)(x->h_pin + x->pin_flag + 128), x->d_seq + 1};
        const int rc = extract_chunk(x, nb, x->d_pyr[0], x->width, (long long)img_bytes, (hfnet_keypoint*)(x->d_blk + o.k), (float*)(x->d_blk + o.d), nullptr,
                                     (int*)(x->d_blk + o.n), (int*)(x->d_blk + o.nl), /*caller_joins=*/true);
        net.global_dst = nullptr;
        net.global_host = FcHostOut();
        HF_TRY(rc);
        // the local results come down as soon as the sampler is done, followed by the "they are down" counter; the global
        // descriptors follow when the global branch -- the longer one for a single frame -- has joined
        HF_HIP(hipMemcpyAsync(x->h_pin + x->pin_res, x->d_blk, o.g, hipMemcpyDeviceToHost, st));
        HF_LAUNCH(&eng, st, "bump_seq", launch_bump_seq(x->d_seq, st));
        HF_HIP(hipMemcpyAsync(x->h_pin + x->pin_flag, x->d_seq, sizeof(int), hipMemcpyDeviceToHost, st));
        if (net.join_pending) { HF_HIP(hipStreamWaitEvent(st, net.ev_join, 0)); net.join_pending = false; }
        if (net.cfg.global && !host_global) HF_HIP(hipMemcpyAsync(x->h_pin + x->pin_res + o.g, x->d_blk + o.g, o.total - o.g, hipMemcpyDeviceToHost, st));
        return HFNET_OK;
    };
    if (!x->use_graph || eng.prof.enabled) return direct();
    if (net.join_pending) { HF_HIP(hipStreamWaitEvent(st, net.ev_join, 0)); net.join_pending = false; }   // (not capturable: recorded outside)
    const int key = nb;
    auto it = x->graphs.find(key);
    if (it == x->graphs.end()) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); x->use_graph = 0; return direct(); }
        const int rc = direct();
        const hipError_t er = hipStreamEndCapture(st, &graph);
        if (rc != HFNET_OK || er != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            x->use_graph = 0;                                     // capture is not available here: plain launches from now on
            return direct();
        }
        (void)hipGraphDestroy(graph);
        it = x->graphs.emplace(key, exec).first;
    }
    HF_HIP(hipGraphLaunch(it->second, st));
    return HFNET_OK;
}

// device copies of a host-pointer chunk for the attached store (hfnet_extractor_attach_store), on the extractor's stream
static int copy_chunk_to_store(hfnet_extractor* x, int first_frame, int nb, const float* d_desc, const int* d_n, hipStream_t st) {
    hfnet_store* s = x->att_store;
    if (!s) return HFNET_OK;
    const int rows = std::min(x->n_features, s->max_rows);
    // consecutive frames go to consecutive slots (modulo the store size): when a store row block is exactly a frame's
    // descriptor block, a run of frames is ONE copy (64 frames per chunk: 3 calls instead of 192 on the compute stream)
    const bool same_shape = s->max_rows == x->n_features && s->dim == HFNET_DESC_DIM;
    for (int f = 0; f < nb;) {
        const int slot = (x->att_first + first_frame + f) % s->n_sets;
        const int run = same_shape ? std::min(nb - f, s->n_sets - slot) : 1;
        HF_HIP(hipMemcpyAsync(s->d_desc + (size_t)slot * s->max_rows * s->dim, d_desc + (size_t)f * x->n_features * HFNET_DESC_DIM,
                              sizeof(float) * (size_t)(same_shape ? run * s->max_rows : rows) * s->dim, hipMemcpyDeviceToDevice, st));
        HF_HIP(hipMemcpyAsync(s->d_rows + slot, d_n + f, sizeof(int32_t) * run, hipMemcpyDeviceToDevice, st));
        HF_HIP(hipMemsetAsync(s->d_flags + (size_t)slot * s->max_rows, 0, (size_t)s->max_rows * run, st));
        f += run;
    }
    return HFNET_OK;
}

// ---- caller memory registered for DMA (hfnet_host_register): process-wide, like the page locks themselves
static std::mutex g_reg_mu;
static std::map<uintptr_t, size_t> g_registered;                 // start -> bytes
static bool host_range_registered(const void* p, size_t bytes) {
    if (!p) return false;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_registered.upper_bound((uintptr_t)p);
    if (it == g_registered.begin()) return false;
    --it;
    return (uintptr_t)p + bytes <= it->first + it->second;
}
int hfnet_host_register(void* ptr, size_t bytes) {
    if (!ptr || !bytes) { set_error("hfnet_host_register: null range"); return HFNET_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_registered.upper_bound((uintptr_t)ptr + bytes - 1);
    if (it != g_registered.begin()) {
        auto prev = std::prev(it);
        if (prev->first + prev->second > (uintptr_t)ptr) { set_error("hfnet_host_register: range overlaps a registered one"); return HFNET_ERR_INVALID_ARG; }
    }
    HF_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    g_registered[(uintptr_t)ptr] = bytes;
    return HFNET_OK;
}
int hfnet_host_unregister(void* ptr) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_registered.find((uintptr_t)ptr);
    if (it == g_registered.end()) { set_error("hfnet_host_unregister: not a registered range"); return HFNET_ERR_INVALID_ARG; }
    g_registered.erase(it);
    HF_HIP(hipHostUnregister(ptr));
    return HFNET_OK;
}

static int host_pipe_init(hfnet_extractor* x) {
    hfnet_extractor::HostPipe& p = x->pipe;
    if (p.ready) return HFNET_OK;
    Engine& eng = x->eng->impl;
    const size_t B = (size_t)x->max_batch, img = (size_t)x->width * x->height, G = (size_t)eng.w.global_dim;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t off = 0;
    p.o_n = off; off += up(sizeof(int) * B);
    p.o_nl = off; off += up(sizeof(int) * B * x->n_levels);
    p.o_g = off; off += up(sizeof(float) * B * G);
    p.o_k = off; off += up(sizeof(hfnet_keypoint) * B * x->n_features);
    p.o_d = off; off += up(sizeof(float) * HFNET_DESC_DIM * B * x->n_features);
    p.out_bytes = off;
    for (int s = 0; s < 2; ++s) {
        HF_HIP(hipHostMalloc((void**)&p.h_in[s], B * img, hipHostMallocDefault));
        HF_HIP(hipHostMalloc((void**)&p.h_out[s], p.out_bytes, hipHostMallocDefault));
        HF_HIP(hipEventCreateWithFlags(&p.ev_up[s], hipEventDisableTiming));
        HF_HIP(hipEventCreateWithFlags(&p.ev_comp[s], hipEventDisableTiming));
        HF_HIP(hipEventCreateWithFlags(&p.ev_down[s], hipEventDisableTiming));
        HF_TRY(dalloc(x->allocs, &p.d_glob[s], B * G));
    }
    p.d_in[0] = x->d_pyr[0]; p.d_kps[0] = x->d_kps; p.d_desc[0] = x->d_desc; p.d_n[0] = x->d_n; p.d_nl[0] = x->d_n_level;
    HF_TRY(dalloc(x->allocs, &p.d_in[1], B * img));
    HF_TRY(dalloc(x->allocs, &p.d_kps[1], B * x->n_features));
    HF_TRY(dalloc(x->allocs, &p.d_desc[1], B * x->n_features * HFNET_DESC_DIM));
    HF_TRY(dalloc(x->allocs, &p.d_n[1], B));
    HF_TRY(dalloc(x->allocs, &p.d_nl[1], B * x->n_levels));
    HF_HIP(hipStreamCreateWithFlags(&p.s_up, hipStreamNonBlocking));
    HF_HIP(hipStreamCreateWithFlags(&p.s_down, hipStreamNonBlocking));
    {
        // helper threads of the pageable <-> pinned staging copies: engine option "copy_threads" (several replicas on one host
        // share its cores: bench.py gives each rank cores / world), by default 3 on a host with >= 8 hardware threads
        const unsigned hc = std::thread::hardware_concurrency();
        const int want = x->eng->impl.opt.copy_threads;
        p.pool.reset(new hfnet::CopyPool(want < 64 ? want : hc >= 8 ? 3 : hc >= 4 ? 1 : 0));
    }
    p.ready = true;
    return HFNET_OK;
}

// host buffers in and out, frames [f0, n_frames): chunk c computes on the extractor's streams while chunk c + 1's images go
// up (pinned block -> device, copy stream 1) and chunk c - 1's results come down (device -> pinned block, copy stream 2) and
// are handed to the caller's buffers by this thread
static int extract_host_pipelined(hfnet_extractor* x, int f0, int n_frames, const uint8_t* images, int row_stride, size_t frame_stride,
                                  hfnet_keypoint* kps, float* local_desc, float* global_desc, int* n_out) {
    HF_TRY(host_pipe_init(x));
    hfnet_extractor::HostPipe& p = x->pipe;
    Engine& eng = x->eng->impl;
    hipStream_t st = x->net.stream;
    const size_t img = (size_t)x->width * x->height, G = (size_t)eng.w.global_dim, NF = (size_t)x->n_features;
    const int n_chunks = (n_frames - f0 + x->max_batch - 1) / x->max_batch;
    // registered caller memory (hfnet_host_register): the copy engines move every byte straight between the caller's buffers and
    // the device -- the pinned staging blocks and the host's staging copies are not used
    const size_t nf_all = (size_t)(n_frames - f0);
    const bool direct = row_stride == x->width && frame_stride == img &&
                        host_range_registered(images + (size_t)f0 * frame_stride, nf_all * img) &&
                        host_range_registered(kps + (size_t)f0 * NF, nf_all * NF * sizeof(hfnet_keypoint)) &&
                        host_range_registered(local_desc + (size_t)f0 * NF * HFNET_DESC_DIM, nf_all * NF * HFNET_DESC_DIM * sizeof(float)) &&
                        host_range_registered(n_out + f0, nf_all * sizeof(int)) &&
                        (!global_desc || host_range_registered(global_desc + (size_t)f0 * G, nf_all * G * sizeof(float)));
    if (direct) {
        auto finish = [&](int c) -> int {                    // chunk c's results are in the caller's buffers
            const int s = c & 1, c0 = f0 + c * x->max_batch, nb = std::min(x->max_batch, n_frames - c0);
            HF_HIP(hipEventSynchronize(p.ev_down[s]));
            if (x->att_store)
                for (int f = 0; f < nb; ++f) x->att_store->rows[(x->att_first + c0 + f) % x->att_store->n_sets] = std::min(n_out[c0 + f], x->att_store->max_rows);
            if (c == n_chunks - 1) {
                std::fill(x->last_n.begin(), x->last_n.end(), -1);
                for (int f = 0; f < nb; ++f) x->last_n[f] = n_out[c0 + f];
                x->last_desc = p.d_desc[s]; x->last_cnt = p.d_n[s];
                if (x->h_pin && x->pinned_frames >= 1) {
                    x->pin_nl_last = x->pin_res + x->result_offsets(1, (int)G).nl;
                    HF_HIP(hipMemcpy(x->h_pin + x->pin_nl_last, p.d_nl[s], sizeof(int) * x->n_levels, hipMemcpyDeviceToHost));
                }
                else HF_HIP(hipMemcpy(x->d_n_level, p.d_nl[s], sizeof(int) * x->n_levels, hipMemcpyDeviceToDevice));
            }
            return HFNET_OK;
        };
        for (int c = 0; c < n_chunks; ++c) {
            const int s = c & 1, c0 = f0 + c * x->max_batch, nb = std::min(x->max_batch, n_frames - c0);
            // slot s: chunk c - 2's download has been waited for (finish(c - 2)), hence its compute and its upload are complete
            HF_HIP(hipMemcpyAsync(p.d_in[s], images + (size_t)c0 * frame_stride, img * nb, hipMemcpyHostToDevice, p.s_up));
            HF_HIP(hipEventRecord(p.ev_up[s], p.s_up));
            HF_HIP(hipStreamWaitEvent(st, p.ev_up[s], 0));
            HF_TRY(extract_chunk(x, nb, p.d_in[s], x->width, (long long)img, p.d_kps[s], p.d_desc[s], p.d_glob[s], p.d_n[s], p.d_nl[s]));
            HF_TRY(copy_chunk_to_store(x, c0, nb, p.d_desc[s], p.d_n[s], st));
            HF_HIP(hipEventRecord(p.ev_comp[s], st));
            HF_HIP(hipStreamWaitEvent(p.s_down, p.ev_comp[s], 0));
            HF_HIP(hipMemcpyAsync(n_out + c0, p.d_n[s], sizeof(int) * nb, hipMemcpyDeviceToHost, p.s_down));
            if (global_desc) HF_HIP(hipMemcpyAsync(global_desc + (size_t)c0 * G, p.d_glob[s], sizeof(float) * (size_t)nb * G, hipMemcpyDeviceToHost, p.s_down));
            HF_HIP(hipMemcpyAsync(kps + (size_t)c0 * NF, p.d_kps[s], sizeof(hfnet_keypoint) * (size_t)nb * NF, hipMemcpyDeviceToHost, p.s_down));
            HF_HIP(hipMemcpyAsync(local_desc + (size_t)c0 * NF * HFNET_DESC_DIM, p.d_desc[s], sizeof(float) * HFNET_DESC_DIM * (size_t)nb * NF, hipMemcpyDeviceToHost, p.s_down));
            HF_HIP(hipEventRecord(p.ev_down[s], p.s_down));
            if (c >= 1) HF_TRY(finish(c - 1));
        }
        HF_TRY(finish(n_chunks - 1));
        return HFNET_OK;
    }
    auto drain = [&](int c) -> int {
        const int s = c & 1, c0 = f0 + c * x->max_batch, nb = std::min(x->max_batch, n_frames - c0);
        HF_HIP(hipEventSynchronize(p.ev_down[s]));
        const unsigned char* h = p.h_out[s];
        const int* hn = (const int*)(h + p.o_n);
        p.pool->run(nb, [&](int f) {
            const int n = hn[f];
            n_out[c0 + f] = n;
            if (global_desc) std::memcpy(global_desc + (size_t)(c0 + f) * G, h + p.o_g + sizeof(float) * (size_t)f * G, sizeof(float) * G);
            if (n <= 0) return;
            std::memcpy(kps + (size_t)(c0 + f) * NF, h + p.o_k + sizeof(hfnet_keypoint) * (size_t)f * NF, sizeof(hfnet_keypoint) * n);
            std::memcpy(local_desc + (size_t)(c0 + f) * NF * HFNET_DESC_DIM, h + p.o_d + sizeof(float) * HFNET_DESC_DIM * (size_t)f * NF,
                        sizeof(float) * HFNET_DESC_DIM * n);
        });
        if (x->att_store)
            for (int f = 0; f < nb; ++f) x->att_store->rows[(x->att_first + c0 + f) % x->att_store->n_sets] = std::min(hn[f], x->att_store->max_rows);
        if (c == n_chunks - 1) {                      // what hfnet_store_put_extracted / n_per_level see: the last chunk
            std::fill(x->last_n.begin(), x->last_n.end(), -1);
            for (int f = 0; f < nb; ++f) x->last_n[f] = hn[f];
            x->last_desc = p.d_desc[s]; x->last_cnt = p.d_n[s];
            if (x->h_pin && x->pinned_frames >= 1) {
                x->pin_nl_last = x->pin_res + x->result_offsets(1, (int)G).nl;
                std::memcpy(x->h_pin + x->pin_nl_last, h + p.o_nl, sizeof(int) * x->n_levels);
            }
            else HF_HIP(hipMemcpy(x->d_n_level, p.d_nl[s], sizeof(int) * x->n_levels, hipMemcpyDeviceToDevice));
        }
        return HFNET_OK;
    };
    for (int c = 0; c < n_chunks; ++c) {
        const int s = c & 1, c0 = f0 + c * x->max_batch, nb = std::min(x->max_batch, n_frames - c0);
        // (slot s is free: chunk c - 2 was drained -- its download, hence its compute and upload, are complete)
        p.pool->run(nb, [&](int f) {
            const uint8_t* src = images + (size_t)(c0 + f) * frame_stride;
            unsigned char* dst = p.h_in[s] + (size_t)f * img;
            if (row_stride == x->width) std::memcpy(dst, src, img);
            else for (int y = 0; y < x->height; ++y) std::memcpy(dst + (size_t)y * x->width, src + (size_t)y * row_stride, (size_t)x->width);
        });
        HF_HIP(hipMemcpyAsync(p.d_in[s], p.h_in[s], img * nb, hipMemcpyHostToDevice, p.s_up));
        HF_HIP(hipEventRecord(p.ev_up[s], p.s_up));
        HF_HIP(hipStreamWaitEvent(st, p.ev_up[s], 0));
        HF_TRY(extract_chunk(x, nb, p.d_in[s], x->width, (long long)img, p.d_kps[s], p.d_desc[s], p.d_glob[s], p.d_n[s], p.d_nl[s]));
        HF_TRY(copy_chunk_to_store(x, c0, nb, p.d_desc[s], p.d_n[s], st));
        HF_HIP(hipEventRecord(p.ev_comp[s], st));
        HF_HIP(hipStreamWaitEvent(p.s_down, p.ev_comp[s], 0));
        unsigned char* h = p.h_out[s];
        HF_HIP(hipMemcpyAsync(h + p.o_n, p.d_n[s], sizeof(int) * nb, hipMemcpyDeviceToHost, p.s_down));
        HF_HIP(hipMemcpyAsync(h + p.o_nl, p.d_nl[s], sizeof(int) * (size_t)x->n_levels * nb, hipMemcpyDeviceToHost, p.s_down));
        HF_HIP(hipMemcpyAsync(h + p.o_g, p.d_glob[s], sizeof(float) * (size_t)nb * G, hipMemcpyDeviceToHost, p.s_down));
        HF_HIP(hipMemcpyAsync(h + p.o_k, p.d_kps[s], sizeof(hfnet_keypoint) * (size_t)nb * NF, hipMemcpyDeviceToHost, p.s_down));
        HF_HIP(hipMemcpyAsync(h + p.o_d, p.d_desc[s], sizeof(float) * HFNET_DESC_DIM * (size_t)nb * NF, hipMemcpyDeviceToHost, p.s_down));
        HF_HIP(hipEventRecord(p.ev_down[s], p.s_down));
        if (c >= 1) HF_TRY(drain(c - 1));
    }
    HF_TRY(drain(n_chunks - 1));
    return HFNET_OK;
}

int hfnet_extractor_attach_store(hfnet_extractor* x, hfnet_store* s, int first_slot) {
    API_GUARD(x, "extractor");
    std::lock_guard<std::mutex> lk(x->mu);
    if (s) {
        if (s->eng != x->eng) { set_error("store and extractor belong to different engines"); return HFNET_ERR_INVALID_ARG; }
        if (s->dim != HFNET_DESC_DIM) { set_error("store: descriptor width %d, extractor produces %d", s->dim, HFNET_DESC_DIM); return HFNET_ERR_SHAPE; }
        if (s->max_rows < x->n_features) { set_error("store: %d rows per slot < the extractor's %d features", s->max_rows, x->n_features); return HFNET_ERR_CAPACITY; }
        if (first_slot < 0 || first_slot >= s->n_sets) { set_error("store: first_slot %d outside [0, %d)", first_slot, s->n_sets); return HFNET_ERR_INVALID_ARG; }
    }
    x->att_store = s; x->att_first = s ? first_slot : 0;
    return HFNET_OK;
}

int hfnet_extractor_extract_batch(hfnet_extractor* x, int n_frames, const uint8_t* images, int row_stride, size_t frame_stride,
                                  hfnet_keypoint* kps, float* local_desc, float* global_desc, int* n_out, int on_device) {
    API_GUARD(x, "extractor");
    if (n_frames < 0) { set_error("n_frames < 0"); return HFNET_ERR_INVALID_ARG; }
    if (n_frames == 0) return HFNET_OK;
    API_GUARD(images, "images"); API_GUARD(kps, "kps"); API_GUARD(local_desc, "local_desc"); API_GUARD(n_out, "n_out");
    if (row_stride < x->width || frame_stride < (size_t)row_stride * x->height) { set_error("bad image strides"); return HFNET_ERR_SHAPE; }
    std::lock_guard<std::mutex> lk(x->mu);
    Engine& eng = x->eng->impl;
    HF_HIP(hipSetDevice(eng.device));
    hipStream_t st = x->net.stream;
    const int G = eng.w.global_dim;
    HF_HIP(eng.wait_fence(st));          // (device-resident callers' hfnet_engine_fence; hfnet_store_put_extracted's copies out of the staging block)
    if (on_device) std::fill(x->last_n.begin(), x->last_n.end(), -1);
    for (int f0 = 0; f0 < n_frames; f0 += x->max_batch) {
        const int nb = std::min(x->max_batch, n_frames - f0);
        if (!on_device) std::fill(x->last_n.begin() + nb, x->last_n.end(), -1);   // staging frames this chunk does not write
        if (on_device) {
            HF_TRY(extract_chunk(x, nb, images + (size_t)f0 * frame_stride, row_stride, (long long)frame_stride, kps + (size_t)f0 * x->n_features,
                                 local_desc + (size_t)f0 * x->n_features * HFNET_DESC_DIM, global_desc ? global_desc + (size_t)f0 * G : nullptr,
                                 n_out + f0, nullptr));
        } else if (nb <= x->pinned_frames && x->h_pin) {
            // latency path: image -> pinned block (CPU), one graph (upload, ~75 kernels on two streams, downloads), one sync,
            // pinned block -> caller's buffers (CPU, only the rows that exist)
            const auto t_enter = std::chrono::steady_clock::now();
            auto stamp = [&](int i) { x->t_last[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count(); };
            const size_t img_bytes = (size_t)x->width * x->height;
            for (int f = 0; f < nb; ++f) {
                const uint8_t* src = images + (size_t)(f0 + f) * frame_stride;
                unsigned char* dst = x->h_pin + (size_t)f * img_bytes;
                if (row_stride == x->width) std::memcpy(dst, src, img_bytes);
                else for (int y = 0; y < x->height; ++y) std::memcpy(dst + (size_t)y * x->width, src + (size_t)y * row_stride, (size_t)x->width);
            }
            stamp(0);
            const int expected = ++x->seq_host;
            volatile int* flag = (volatile int*)(x->h_pin + x->pin_flag);
            volatile int* gflag = (volatile int*)(x->h_pin + x->pin_flag + 128);
            // an error between here and the waits below leaves the host's numbering ahead of the device's (the graph that bumps
            // it may never have been enqueued): drain the stream and take the numbers the device really wrote, or the next
            // call would spin its full 20 ms for a number that never comes
            auto resync = [&]() { (void)hipStreamSynchronize(st); (void)hipGetLastError(); x->seq_host = *flag; x->gseq_host = *gflag; };
            if (int rc = extract_chunk_graphed(x, nb)) { resync(); return rc; }
            const hfnet_extractor::ResOff o = x->result_offsets(nb, G);
            const float* blk_desc = (const float*)(x->d_blk + o.d);
            const int* blk_n = (const int*)(x->d_blk + o.n);
            if (int rc = copy_chunk_to_store(x, f0, nb, blk_desc, blk_n, st)) { resync(); return rc; }
            stamp(1);
            // the keypoints and descriptors (1 MB per frame) are unpacked while the GPU is still busy with the global branch:
            // spin until the counter that follows them into the pinned block shows this call's number (bounded; a call that
            // never sees it simply waits for the stream)
            {
                const auto t_spin = std::chrono::steady_clock::now();
                for (unsigned it = 0; *flag != expected; ++it) {
                    cpu_relax();
                    if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(20)) break;
                }
                if (*flag != expected && hipStreamSynchronize(st) != hipSuccess) { resync(); set_error("hipStreamSynchronize failed in the latency path"); return HFNET_ERR_DEVICE; }
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            stamp(2);
            x->last_desc = blk_desc; x->last_cnt = blk_n;
            const unsigned char* res = x->h_pin + x->pin_res;
            x->pin_nl_last = x->pin_res + o.nl;
            const int* hn = (const int*)(res + o.n);
            for (int f = 0; f < nb; ++f) {
                const int n = hn[f];
                n_out[f0 + f] = n;
                x->last_n[f] = n;
                if (x->att_store) x->att_store->rows[(x->att_first + f0 + f) % x->att_store->n_sets] = std::min(n, x->att_store->max_rows);
                if (n <= 0) continue;
                std::memcpy(kps + (size_t)(f0 + f) * x->n_features, res + o.k + sizeof(hfnet_keypoint) * (size_t)f * x->n_features, sizeof(hfnet_keypoint) * n);
                std::memcpy(local_desc + (size_t)(f0 + f) * x->n_features * HFNET_DESC_DIM,
                            res + o.d + sizeof(float) * HFNET_DESC_DIM * (size_t)f * x->n_features, sizeof(float) * HFNET_DESC_DIM * n);
            }
            stamp(3);
            if (x->global_to_host(nb)) {
                // the global descriptors arrive the same way: written into the pinned block by the last kernel of the branch,
                // followed by the call's number (no copy after the join, no stream synchronisation on the way out)
                const int gexpected = ++x->gseq_host;
                const auto t_spin = std::chrono::steady_clock::now();
                for (unsigned it = 0; *gflag != gexpected; ++it) {
                    cpu_relax();
                    if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(20)) break;
                }
                if (*gflag != gexpected) HF_HIP(hipStreamSynchronize(st));   // (the graph / the join event bring the branch's stream in)
                std::atomic_thread_fence(std::memory_order_acquire);
                x->gseq_host = *gflag;
            } else {
                HF_HIP(hipStreamSynchronize(st));
            }
            stamp(4);
            x->seq_host = *flag;                                  // (re-synchronise the numbering, whatever happened)
            if (global_desc)
                for (int f = 0; f < nb; ++f) std::memcpy(global_desc + (size_t)(f0 + f) * G, res + o.g + sizeof(float) * (size_t)f * G, sizeof(float) * G);
            stamp(5);
        } else {
            // everything that is left, as a double-buffered pipeline over its chunks
            HF_TRY(extract_host_pipelined(x, f0, n_frames, images, row_stride, frame_stride, kps, local_desc, global_desc, n_out));
            break;
        }
    }
    if (on_device) HF_HIP(eng.note_extract(st));
    return HFNET_OK;
}

int hfnet_extractor_last_timing(hfnet_extractor* x, double* us, int n) {
    API_GUARD(x, "extractor"); API_GUARD(us, "us");
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->t_last[5] < 0) { set_error("no latency-path call yet"); return HFNET_ERR_INVALID_ARG; }
    for (int i = 0; i < n && i < 6; ++i) us[i] = x->t_last[i];
    return HFNET_OK;
}

int hfnet_extractor_extract(hfnet_extractor* x, const uint8_t* image, int row_stride, hfnet_keypoint* kps, float* local_desc,
                            float* global_desc, int* n_out, int* n_per_level) {
    if (n_out) *n_out = -1;
    API_GUARD(x, "extractor"); API_GUARD(n_out, "n_out");
    if (!image) { set_error("empty image"); return HFNET_ERR_INVALID_ARG; }   // HFextractor.cc:145 returns -1
    int n = 0;
    HF_TRY(hfnet_extractor_extract_batch(x, 1, image, row_stride, (size_t)row_stride * x->height, kps, local_desc, global_desc, &n, 0));
    *n_out = n;
    if (n_per_level) {
        std::lock_guard<std::mutex> lk(x->mu);
        if (x->h_pin && x->pinned_frames >= 1) std::memcpy(n_per_level, x->h_pin + x->pin_nl_last, sizeof(int) * x->n_levels);   // came down with the frame
        else HF_HIP(hipMemcpy(n_per_level, x->d_n_level, sizeof(int) * x->n_levels, hipMemcpyDeviceToHost));
    }
    return HFNET_OK;
}

// ---------------------------------------------------------------------------------------- Matcher
static int stage_rows(Engine& e, DevMem& m, const float* src, size_t count, int on_device, const float** out) {
    if (on_device) { *out = src; return HFNET_OK; }
    HF_TRY(m.ensure(std::max<size_t>(count, 1) * sizeof(float)));
    if (count) HF_HIP(hipMemcpyAsync(m.p, src, count * sizeof(float), hipMemcpyHostToDevice, e.stream));
    *out = m.as<float>();
    return HFNET_OK;
}

int hfnet_descriptor_distance(hfnet_engine* eh, const float* a, const float* b, int dim, float* out) {
    API_GUARD(eh, "engine"); API_GUARD(a, "a"); API_GUARD(b, "b"); API_GUARD(out, "out");
    if (dim <= 0) { set_error("dim <= 0"); return HFNET_ERR_INVALID_ARG; }
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    const float *da, *db;
    HF_TRY(stage_rows(e, e.m_a, a, dim, 0, &da));
    HF_TRY(stage_rows(e, e.m_b, b, dim, 0, &db));
    HF_TRY(e.m_f0.ensure(sizeof(float)));
    HF_LAUNCH(&e, e.stream, "descriptor_distance", launch_descriptor_distance(da, db, dim, e.m_f0.as<float>(), e.stream));
    HF_HIP(hipMemcpyAsync(out, e.m_f0.p, sizeof(float), hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

// scratch for n_pairs x (max_rows x max_rows) similarity matrices, norms, keys and the pair descriptors
// neither matcher stores an n x m matrix: SearchByBoW keeps candidate slots per train row, SearchForTriangulation
// (maximum, index) partials per row / column and 64-wide tile
// the split-row scratch of the screened SearchForTriangulation, or null when this call takes the full path (see Engine::tri_skip);
// resets the device statistics the call will add to
static int tri_screen_begin(Engine& e, int n_pairs, int max_rows, void** split, int** stat) {
    *split = nullptr; *stat = nullptr;
    if (!e.opt.tri_screen_bf16 || n_pairs < 4) return HFNET_OK;
    if (!e.h_tri_stat) {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return HFNET_OK; }
        e.h_tri_stat = (int*)hp; e.h_tri_stat[0] = 0; e.h_tri_stat[1] = 0;
    }
    HF_TRY(e.m_tri_stat.ensure(2 * sizeof(int)));
    // the counts of the last screened call come down behind it without a synchronisation: they are looked at only once the event
    // behind that copy has completed (the host never writes the pinned words, so it does not race the DMA engine).  A device-
    // resident caller that runs ahead of the GPU simply decides one call later -- the matches are the same bits either way;
    // worst case of the adaptive state: one call in 17 pays the screened path's overflow (the full f32 path re-run for the
    // overflowed pairs, ~2x the call) on descriptor sets in which most products exceed the threshold.
    if (e.tri_stat_pending) {
        const hipError_t q = hipEventQuery(e.ev_tri_stat);
        if (q == hipSuccess) {
            e.tri_stat_pending = false;
            volatile int* h = e.h_tri_stat;
            if (h[1] > 0 && h[0] * 4 >= h[1]) e.tri_skip = 16;
        } else if (q != hipErrorNotReady) HF_HIP(q);
        else (void)hipGetLastError();
    }
    if (e.tri_skip > 0) { --e.tri_skip; return HFNET_OK; }
    HF_HIP(hipMemsetAsync(e.m_tri_stat.p, 0, 2 * sizeof(int), e.stream));
    *split = (unsigned char*)e.m_s.p + tri_split_offset_bytes(n_pairs, max_rows);
    *stat = e.m_tri_stat.as<int>();
    return HFNET_OK;
}
static int tri_screen_end(Engine& e, int* stat) {
    if (stat) {
        HF_HIP(hipMemcpyAsync(e.h_tri_stat, stat, 2 * sizeof(int), hipMemcpyDeviceToHost, e.stream));
        if (!e.ev_tri_stat) HF_HIP(hipEventCreateWithFlags(&e.ev_tri_stat, hipEventDisableTiming));
        HF_HIP(hipEventRecord(e.ev_tri_stat, e.stream));
        e.tri_stat_pending = true;
    }
    return HFNET_OK;
}

static int bow_scratch(Engine& e, int n_pairs, int max_rows, int dim, bool triangulation) {
    const size_t np = (size_t)std::max(n_pairs, 1), mr = (size_t)std::max(max_rows, 1);
    HF_TRY(e.m_s.ensure(triangulation ? tri_scratch_bytes((int)np, (int)mr, std::max(dim, 4)) : bow_scratch_bytes((int)np, (int)mr, std::max(dim, 4))));
    HF_TRY(e.m_qn.ensure(sizeof(float) * np * mr));
    HF_TRY(e.m_tn.ensure(sizeof(float) * np * mr));
    HF_TRY(e.m_key.ensure(sizeof(unsigned long long) * np * mr));
    HF_TRY(e.m_pairs.ensure(sizeof(BowPair) * np));
    return HFNET_OK;
}

int hfnet_match_search_by_bow(hfnet_engine* eh, const float* query, int n_query, const float* train, int n_train, int dim, float th_low,
                              int32_t* match_q2t, float* dist, int* n_matches, int on_device) {
    API_GUARD(eh, "engine"); API_GUARD(match_q2t, "match_q2t"); API_GUARD(dist, "dist"); API_GUARD(n_matches, "n_matches");
    if (n_query < 0 || n_train < 0 || dim <= 0 || dim % 64) { set_error("bad matcher sizes (dim must be a multiple of 64)"); return HFNET_ERR_INVALID_ARG; }
    if ((n_query && !query) || (n_train && !train)) { set_error("null descriptor matrix"); return HFNET_ERR_INVALID_ARG; }
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    if (on_device) HF_HIP(e.wait_extract());
    if (n_query == 0) { if (!on_device) *n_matches = 0; else HF_HIP(hipMemsetAsync(n_matches, 0, sizeof(int), e.stream)); return HFNET_OK; }
    const float *dq, *dt;
    HF_TRY(stage_rows(e, e.m_a, query, (size_t)n_query * dim, on_device, &dq));
    HF_TRY(stage_rows(e, e.m_b, train, (size_t)n_train * dim, on_device, &dt));
    const int max_rows = std::max(n_query, n_train);
    HF_TRY(bow_scratch(e, 1, max_rows, dim, false));
    int32_t* d_match = match_q2t; float* d_dist = dist; int* d_cnt = n_matches;
    if (!on_device) {
        HF_TRY(e.m_i0.ensure(sizeof(int32_t) * n_query)); HF_TRY(e.m_f0.ensure(sizeof(float) * n_query)); HF_TRY(e.m_cnt.ensure(sizeof(int)));
        d_match = e.m_i0.as<int32_t>(); d_dist = e.m_f0.as<float>(); d_cnt = e.m_cnt.as<int>();
    }
    BowPair P;
    P.q = dq; P.t = dt; P.St = e.m_s.as<float>(); P.qn = e.m_qn.as<float>(); P.tn = e.m_tn.as<float>(); P.qkey = e.m_key.as<unsigned long long>();
    P.match = d_match; P.dist = d_dist; P.cnt = d_cnt; P.nq = n_query; P.nt = n_train;
    HF_HIP(hipMemcpyAsync(e.m_pairs.p, &P, sizeof P, hipMemcpyHostToDevice, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));     // P lives on this stack frame
    HF_LAUNCH(&e, e.stream, "match_bow", launch_bow_pairs(e.m_pairs.as<BowPair>(), 1, max_rows, dim, th_low, e.m_s.p, e.stream, e.opt.match_screen_bf16));
    if (!on_device) {
        HF_HIP(hipMemcpyAsync(match_q2t, d_match, sizeof(int32_t) * n_query, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipMemcpyAsync(dist, d_dist, sizeof(float) * n_query, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipMemcpyAsync(n_matches, d_cnt, sizeof(int), hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipStreamSynchronize(e.stream));
    }
    return HFNET_OK;
}

static int match_pairs_batch(hfnet_engine* eh, int n_pairs, const float* desc_base, size_t set_stride, const int32_t* n_rows, int n_sets,
                             const int32_t* query_set, const int32_t* train_set, int max_rows, int dim, float th, int32_t* match_q2t,
                             float* dist, int32_t* n_matches, int on_device, bool triangulation) {
    API_GUARD(eh, "engine");
    if (n_pairs < 0 || n_sets < 0 || max_rows < 1 || dim <= 0 || dim % 64 || set_stride < (size_t)max_rows * dim) {
        set_error("bad batched matcher arguments (dim multiple of 64, set_stride >= max_rows * dim)"); return HFNET_ERR_INVALID_ARG; }
    if (n_pairs == 0) return HFNET_OK;
    API_GUARD(desc_base, "desc_base"); API_GUARD(n_rows, "n_rows"); API_GUARD(query_set, "query_set"); API_GUARD(train_set, "train_set");
    API_GUARD(match_q2t, "match"); API_GUARD(n_matches, "n_matches");
    if (!triangulation) API_GUARD(dist, "dist");
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    if (on_device) HF_HIP(e.wait_extract());
    HF_TRY(bow_scratch(e, n_pairs, max_rows, dim, triangulation));
    const float* d_base = desc_base; const int32_t *d_rows = n_rows, *d_qs = query_set, *d_ts = train_set;
    int32_t* d_match = match_q2t; float* d_dist = dist ? dist : (float*)match_q2t; int32_t* d_cnt = n_matches;
    if (!on_device) {
        for (int p = 0; p < n_pairs; ++p)
            if (query_set[p] < 0 || query_set[p] >= n_sets || train_set[p] < 0 || train_set[p] >= n_sets) { set_error("pair %d references a set outside [0, %d)", p, n_sets); return HFNET_ERR_INVALID_ARG; }
        for (int s = 0; s < n_sets; ++s)
            if (n_rows[s] < 0 || n_rows[s] > max_rows) { set_error("set %d has %d rows, outside [0, %d]", s, n_rows[s], max_rows); return HFNET_ERR_INVALID_ARG; }
        HF_TRY(e.m_a.ensure(sizeof(float) * (size_t)std::max(n_sets, 1) * set_stride));
        HF_TRY(e.m_b.ensure(sizeof(int32_t) * ((size_t)n_sets + 2 * (size_t)n_pairs)));
        HF_TRY(e.m_i0.ensure(sizeof(int32_t) * (size_t)n_pairs * max_rows)); HF_TRY(e.m_f0.ensure(sizeof(float) * (size_t)n_pairs * max_rows));
        HF_TRY(e.m_cnt.ensure(sizeof(int32_t) * n_pairs));
        HF_HIP(hipMemcpyAsync(e.m_a.p, desc_base, sizeof(float) * (size_t)n_sets * set_stride, hipMemcpyHostToDevice, e.stream));
        int32_t* ib = e.m_b.as<int32_t>();
        HF_HIP(hipMemcpyAsync(ib, n_rows, sizeof(int32_t) * n_sets, hipMemcpyHostToDevice, e.stream));
        HF_HIP(hipMemcpyAsync(ib + n_sets, query_set, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, e.stream));
        HF_HIP(hipMemcpyAsync(ib + n_sets + n_pairs, train_set, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, e.stream));
        d_base = e.m_a.as<float>(); d_rows = ib; d_qs = ib + n_sets; d_ts = ib + n_sets + n_pairs;
        d_match = e.m_i0.as<int32_t>(); d_dist = e.m_f0.as<float>(); d_cnt = e.m_cnt.as<int32_t>();
        // rows at or beyond a pair's query count are not written by the kernels: the caller gets -1 there (and 0xFF.. = NaN
        // in the distances of such rows)
        HF_HIP(hipMemsetAsync(d_match, 0xFF, sizeof(int32_t) * (size_t)n_pairs * max_rows, e.stream));
        if (!triangulation) HF_HIP(hipMemsetAsync(d_dist, 0xFF, sizeof(float) * (size_t)n_pairs * max_rows, e.stream));
    }
    HF_LAUNCH(&e, e.stream, "match_bow_setup",
              launch_bow_setup(e.m_pairs.as<BowPair>(), n_pairs, d_base, (long long)set_stride, d_rows, d_qs, d_ts, max_rows, e.m_s.as<float>(),
                               triangulation ? (long long)tri_scratch_floats(max_rows) : 0, e.m_qn.as<float>(), e.m_tn.as<float>(), e.m_key.as<unsigned long long>(), d_match, d_dist, d_cnt, max_rows, e.stream));
    if (triangulation) {
        const float threshold = (float)(-0.5 * th * th + 1);   // Matcher.cc:851
        void* split = nullptr; int* stat = nullptr;
        HF_TRY(tri_screen_begin(e, n_pairs, max_rows, &split, &stat));
        HF_LAUNCH(&e, e.stream, "match_tri", launch_tri_pairs(e.m_pairs.as<BowPair>(), n_pairs, max_rows, dim, threshold, e.stream, split, stat));
        HF_TRY(tri_screen_end(e, stat));
    } else {
        HF_LAUNCH(&e, e.stream, "match_bow", launch_bow_pairs(e.m_pairs.as<BowPair>(), n_pairs, max_rows, dim, th, e.m_s.p, e.stream, e.opt.match_screen_bf16));
    }
    if (!on_device) {
        HF_HIP(hipMemcpyAsync(match_q2t, d_match, sizeof(int32_t) * (size_t)n_pairs * max_rows, hipMemcpyDeviceToHost, e.stream));
        if (!triangulation) HF_HIP(hipMemcpyAsync(dist, d_dist, sizeof(float) * (size_t)n_pairs * max_rows, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipMemcpyAsync(n_matches, d_cnt, sizeof(int32_t) * n_pairs, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipStreamSynchronize(e.stream));
    }
    return HFNET_OK;
}

int hfnet_match_search_by_bow_batch(hfnet_engine* eh, int n_pairs, const float* desc_base, size_t set_stride, const int32_t* n_rows, int n_sets,
                                    const int32_t* query_set, const int32_t* train_set, int max_rows, int dim, float th_low, int32_t* match_q2t,
                                    float* dist, int32_t* n_matches, int on_device) {
    return match_pairs_batch(eh, n_pairs, desc_base, set_stride, n_rows, n_sets, query_set, train_set, max_rows, dim, th_low, match_q2t, dist,
                             n_matches, on_device, false);
}

int hfnet_match_search_for_triangulation_batch(hfnet_engine* eh, int n_pairs, const float* desc_base, size_t set_stride, const int32_t* n_rows,
                                               int n_sets, const int32_t* set1, const int32_t* set2, int max_rows, int dim, float th_high,
                                               int32_t* match12, int32_t* n_matches, int on_device) {
    return match_pairs_batch(eh, n_pairs, desc_base, set_stride, n_rows, n_sets, set1, set2, max_rows, dim, th_high, match12, nullptr, n_matches,
                             on_device, true);
}

// ---------------------------------------------------------------------------------------- descriptor store
int hfnet_store_create(hfnet_engine* eh, int n_sets, int max_rows, int dim, hfnet_store** out) {
    API_GUARD(out, "out");
    *out = nullptr;
    API_GUARD(eh, "engine");
    if (n_sets < 1 || max_rows < 1 || dim <= 0 || dim % 64) { set_error("store: n_sets, max_rows >= 1 and dim a multiple of 64 required"); return HFNET_ERR_INVALID_ARG; }
    HF_HIP(hipSetDevice(eh->impl.device));
    std::unique_ptr<hfnet_store> st(new hfnet_store);
    st->eng = eh; st->n_sets = n_sets; st->max_rows = max_rows; st->dim = dim;
    st->rows.assign(n_sets, 0);
    HF_HIP(hipMalloc((void**)&st->d_desc, sizeof(float) * (size_t)n_sets * max_rows * dim));
    if (hipMalloc((void**)&st->d_rows, sizeof(int32_t) * n_sets) != hipSuccess) { (void)hipFree(st->d_desc); set_error("store: out of device memory"); return HFNET_ERR_DEVICE; }
    if (hipMalloc((void**)&st->d_flags, (size_t)n_sets * max_rows) != hipSuccess) { (void)hipFree(st->d_desc); (void)hipFree(st->d_rows); set_error("store: out of device memory"); return HFNET_ERR_DEVICE; }
    {   // on the engine's (non-blocking) stream, which every later put / match uses: see hfnet_db_create
        Engine& e = eh->impl;
        std::lock_guard<std::mutex> lk(e.mu);
        HF_HIP(hipMemsetAsync(st->d_rows, 0, sizeof(int32_t) * n_sets, e.stream));
        HF_HIP(hipMemsetAsync(st->d_flags, 0, (size_t)n_sets * max_rows, e.stream));
        HF_HIP(hipStreamSynchronize(e.stream));
    }
    *out = st.release();
    return HFNET_OK;
}

void hfnet_store_destroy(hfnet_store* st) {
    if (!st) return;
    (void)hipSetDevice(st->eng->impl.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(st->d_desc);
    (void)hipFree(st->d_rows);
    (void)hipFree(st->d_flags);
    delete st;
}

int hfnet_store_put(hfnet_store* st, int slot, const float* rows, int n_rows) {
    API_GUARD(st, "store");
    if (slot < 0 || slot >= st->n_sets || n_rows < 0 || n_rows > st->max_rows) { set_error("store: slot %d / %d rows outside [0, %d) / [0, %d]", slot, n_rows, st->n_sets, st->max_rows); return HFNET_ERR_INVALID_ARG; }
    if (n_rows && !rows) { set_error("null descriptor matrix"); return HFNET_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(st->mu);
    Engine& e = st->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    HF_HIP(hipSetDevice(e.device));
    const int32_t n = n_rows;
    if (n_rows) HF_HIP(hipMemcpyAsync(st->d_desc + (size_t)slot * st->max_rows * st->dim, rows, sizeof(float) * (size_t)n_rows * st->dim, hipMemcpyHostToDevice, e.stream));
    HF_HIP(hipMemcpyAsync(st->d_rows + slot, &n, sizeof n, hipMemcpyHostToDevice, e.stream));
    HF_HIP(hipMemsetAsync(st->d_flags + (size_t)slot * st->max_rows, 0, (size_t)st->max_rows, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));                          // the host buffers may go away
    st->rows[slot] = n;
    return HFNET_OK;
}

int hfnet_store_rows(const hfnet_store* st, int slot) {
    if (!st || slot < 0 || slot >= st->n_sets) return -1;
    return st->rows[slot];
}

int hfnet_store_set_flags(hfnet_store* st, int slot, const uint8_t* flags, int n_rows) {
    API_GUARD(st, "store");
    if (slot < 0 || slot >= st->n_sets || n_rows < 0 || n_rows > st->max_rows) { set_error("store: slot %d / %d rows outside [0, %d) / [0, %d]", slot, n_rows, st->n_sets, st->max_rows); return HFNET_ERR_INVALID_ARG; }
    if (n_rows == 0) return HFNET_OK;
    API_GUARD(flags, "flags");
    std::lock_guard<std::mutex> lk(st->mu);
    Engine& e = st->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    HF_HIP(hipSetDevice(e.device));
    HF_HIP(hipMemcpyAsync(st->d_flags + (size_t)slot * st->max_rows, flags, (size_t)n_rows, hipMemcpyHostToDevice, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

int hfnet_store_put_extracted(hfnet_store* st, int slot, hfnet_extractor* x, int frame) {
    API_GUARD(st, "store"); API_GUARD(x, "extractor");
    if (st->eng != x->eng) { set_error("store and extractor belong to different engines"); return HFNET_ERR_INVALID_ARG; }
    if (slot < 0 || slot >= st->n_sets || frame < 0 || frame >= x->max_batch) { set_error("store: slot %d / frame %d out of range", slot, frame); return HFNET_ERR_INVALID_ARG; }
    if (st->dim != HFNET_DESC_DIM) { set_error("store: descriptor width %d, extractor produces %d", st->dim, HFNET_DESC_DIM); return HFNET_ERR_SHAPE; }
    std::lock_guard<std::mutex> lkx(x->mu);
    const int n = x->last_n[frame];
    if (n < 0) { set_error("store: no host-pointer extraction result in staging frame %d", frame); return HFNET_ERR_INVALID_ARG; }
    if (n > st->max_rows) { set_error("store: %d rows > capacity %d", n, st->max_rows); return HFNET_ERR_CAPACITY; }
    std::lock_guard<std::mutex> lk(st->mu);
    Engine& e = st->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    HF_HIP(hipSetDevice(e.device));
    // Invariant relied on (no stream drain any more: with host_global the host-pointer call returns while the global branch may
    // still run): the LOCAL section of the extractor's device block (descriptors, counts) is complete once the host has seen the
    // local-results flag -- the call does not return before that, and the flag follows the download of that section on the
    // stream -- and nothing writes it again before the NEXT extraction, which waits for the event recorded below
    // (Engine::wait_fence, unconditional at the top of every extraction).  Later matches are ordered behind these copies by the
    // engine stream.
    const float* src_desc = x->last_desc ? x->last_desc : x->d_desc;
    const int* src_n = x->last_cnt ? x->last_cnt : x->d_n;
    if (n) HF_HIP(hipMemcpyAsync(st->d_desc + (size_t)slot * st->max_rows * st->dim, src_desc + (size_t)frame * x->n_features * HFNET_DESC_DIM,
                                 sizeof(float) * (size_t)n * st->dim, hipMemcpyDeviceToDevice, e.stream));
    HF_HIP(hipMemcpyAsync(st->d_rows + slot, src_n + frame, sizeof(int32_t), hipMemcpyDeviceToDevice, e.stream));
    HF_HIP(hipMemsetAsync(st->d_flags + (size_t)slot * st->max_rows, 0, (size_t)st->max_rows, e.stream));
    // no host synchronisation: later matches follow on the same stream, and the next extraction (which overwrites the staging
    // block these copies read) waits for this point by event, like hfnet_engine_fence
    {
        std::lock_guard<std::mutex> lk3(e.ev_mu);
        if (!e.ev_match) HF_HIP(hipEventCreateWithFlags(&e.ev_match, hipEventDisableTiming));
        HF_HIP(hipEventRecord(e.ev_match, e.stream));
        e.ev_match_set = true;
    }
    st->rows[slot] = n;
    return HFNET_OK;
}

// pairs of resident sets -> host results.  Only the pair lists go up and the matches come down.
static int match_store(hfnet_store* st, int n_pairs, const int32_t* set1, const int32_t* set2, int rows1, int rows2, float th, int32_t* match,
                       float* dist, int32_t* n_matches, bool triangulation) {
    API_GUARD(st, "store");
    if (n_pairs < 0) { set_error("n_pairs < 0"); return HFNET_ERR_INVALID_ARG; }
    if (rows1 < HFNET_ROWS_ALL || rows1 > HFNET_ROWS_UNFLAGGED || rows2 < HFNET_ROWS_ALL || rows2 > HFNET_ROWS_UNFLAGGED) { set_error("row filter must be HFNET_ROWS_ALL / _FLAGGED / _UNFLAGGED"); return HFNET_ERR_INVALID_ARG; }
    if (n_pairs == 0) return HFNET_OK;
    API_GUARD(set1, "set1"); API_GUARD(set2, "set2"); API_GUARD(match, "match"); API_GUARD(n_matches, "n_matches");
    if (!triangulation) API_GUARD(dist, "dist");
    for (int p = 0; p < n_pairs; ++p)
        if (set1[p] < 0 || set1[p] >= st->n_sets || set2[p] < 0 || set2[p] >= st->n_sets) { set_error("pair %d references a set outside [0, %d)", p, st->n_sets); return HFNET_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lks(st->mu);
    Engine& e = st->eng->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    const int mr = st->max_rows;
    const long long stride = (long long)mr * st->dim;
    // filtered sides: one compacted copy per distinct (slot, filter)
    std::vector<int32_t> host;                                     // [qsel | tsel | c_slot | c_filter]
    std::vector<int32_t> qsel(set1, set1 + n_pairs), tsel(set2, set2 + n_pairs), c_slot, c_filter;
    if (rows1 != HFNET_ROWS_ALL || rows2 != HFNET_ROWS_ALL) {
        std::map<std::pair<int, int>, int> seen;
        auto compacted = [&](int slot, int filter) {
            auto it = seen.find({slot, filter});
            if (it == seen.end()) { it = seen.emplace(std::make_pair(slot, filter), (int)c_slot.size()).first; c_slot.push_back(slot); c_filter.push_back(filter); }
            return ~it->second;
        };
        for (int p = 0; p < n_pairs; ++p) {
            if (rows1 != HFNET_ROWS_ALL) qsel[p] = compacted(set1[p], rows1);
            if (rows2 != HFNET_ROWS_ALL) tsel[p] = compacted(set2[p], rows2);
        }
    }
    const int nc = (int)c_slot.size();
    host.insert(host.end(), qsel.begin(), qsel.end()); host.insert(host.end(), tsel.begin(), tsel.end());
    host.insert(host.end(), c_slot.begin(), c_slot.end()); host.insert(host.end(), c_filter.begin(), c_filter.end());
    HF_TRY(bow_scratch(e, n_pairs, mr, st->dim, triangulation));
    // m_b: [qsel | tsel | c_slot | c_filter | c_rows | map nc*mr | inv nc*mr]
    HF_TRY(e.m_b.ensure(sizeof(int32_t) * (2 * (size_t)n_pairs + 3 * (size_t)nc + 2 * (size_t)nc * mr)));
    HF_TRY(e.m_i0.ensure(sizeof(int32_t) * (size_t)n_pairs * mr)); HF_TRY(e.m_f0.ensure(sizeof(float) * (size_t)n_pairs * mr));
    HF_TRY(e.m_cnt.ensure(sizeof(int32_t) * n_pairs));
    if (nc) { HF_TRY(e.m_a.ensure(sizeof(float) * (size_t)nc * stride)); HF_TRY(e.m_i1.ensure(sizeof(int32_t) * (size_t)n_pairs * mr)); HF_TRY(e.m_f1.ensure(sizeof(float) * (size_t)n_pairs * mr)); }
    int32_t* ib = e.m_b.as<int32_t>();
    int32_t *d_qsel = ib, *d_tsel = ib + n_pairs, *d_cslot = ib + 2 * n_pairs, *d_cfilter = d_cslot + nc, *d_crows = d_cfilter + nc, *d_map = d_crows + nc,
            *d_inv = d_map + (size_t)nc * mr;
    HF_HIP(hipMemcpyAsync(ib, host.data(), sizeof(int32_t) * host.size(), hipMemcpyHostToDevice, e.stream));
    int32_t* d_match = e.m_i0.as<int32_t>(); float* d_dist = e.m_f0.as<float>(); int32_t* d_cnt = e.m_cnt.as<int32_t>();
    int32_t* w_match = nc ? e.m_i1.as<int32_t>() : d_match; float* w_dist = nc ? e.m_f1.as<float>() : d_dist;   // results in compacted numbering
    if (nc)
        HF_LAUNCH(&e, e.stream, "store_compact",
                  launch_store_compact(st->d_desc, st->d_flags, stride, st->d_rows, nc, d_cslot, d_cfilter, mr, st->dim, d_map, d_inv, d_crows,
                                       e.m_a.as<float>(), e.stream));
    HF_LAUNCH(&e, e.stream, "store_setup",
              launch_store_setup(e.m_pairs.as<BowPair>(), n_pairs, st->d_desc, e.m_a.as<float>(), stride, st->d_rows, d_crows, d_qsel, d_tsel, mr,
                                 e.m_s.as<float>(), triangulation ? (long long)tri_scratch_floats(mr) : 0, e.m_qn.as<float>(), e.m_tn.as<float>(),
                                 e.m_key.as<unsigned long long>(), w_match, w_dist, d_cnt, e.stream));
    if (triangulation) {
        const float threshold = (float)(-0.5 * th * th + 1);       // Matcher.cc:851
        void* split = nullptr; int* stat = nullptr;
        HF_TRY(tri_screen_begin(e, n_pairs, mr, &split, &stat));
        HF_LAUNCH(&e, e.stream, "match_tri", launch_tri_pairs(e.m_pairs.as<BowPair>(), n_pairs, mr, st->dim, threshold, e.stream, split, stat));
        HF_TRY(tri_screen_end(e, stat));
    } else {
        HF_LAUNCH(&e, e.stream, "match_bow", launch_bow_pairs(e.m_pairs.as<BowPair>(), n_pairs, mr, st->dim, th, e.m_s.p, e.stream, e.opt.match_screen_bf16));
    }
    if (nc)
        HF_LAUNCH(&e, e.stream, "store_remap",
                  launch_store_remap(n_pairs, d_qsel, d_tsel, d_cslot, st->d_rows, d_map, d_inv, mr, w_match, triangulation ? nullptr : w_dist, d_match,
                                     triangulation ? nullptr : d_dist, e.stream));
    // results: through the engine's pinned block when they fit (copies into pageable memory are staged and synchronous one by one)
    const size_t b_match = sizeof(int32_t) * (size_t)n_pairs * mr, b_dist = triangulation ? 0 : sizeof(float) * (size_t)n_pairs * mr,
                 b_cnt = sizeof(int32_t) * (size_t)n_pairs;
    if (e.pinned_results(b_match + b_dist + b_cnt)) {
        unsigned char* hp = e.h_res;
        HF_HIP(hipMemcpyAsync(hp, d_match, b_match, hipMemcpyDeviceToHost, e.stream));
        if (b_dist) HF_HIP(hipMemcpyAsync(hp + b_match, d_dist, b_dist, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipMemcpyAsync(hp + b_match + b_dist, d_cnt, b_cnt, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipStreamSynchronize(e.stream));
        std::memcpy(match, hp, b_match);
        if (b_dist) std::memcpy(dist, hp + b_match, b_dist);
        std::memcpy(n_matches, hp + b_match + b_dist, b_cnt);
        return HFNET_OK;
    }
    HF_HIP(hipMemcpyAsync(match, d_match, b_match, hipMemcpyDeviceToHost, e.stream));
    if (!triangulation) HF_HIP(hipMemcpyAsync(dist, d_dist, b_dist, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipMemcpyAsync(n_matches, d_cnt, b_cnt, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

int hfnet_store_search_by_bow(hfnet_store* st, int n_pairs, const int32_t* query_set, const int32_t* train_set, int query_rows, int train_rows,
                              float th_low, int32_t* match_q2t, float* dist, int32_t* n_matches) {
    return match_store(st, n_pairs, query_set, train_set, query_rows, train_rows, th_low, match_q2t, dist, n_matches, false);
}

int hfnet_store_search_for_triangulation(hfnet_store* st, int n_pairs, const int32_t* set1, const int32_t* set2, int rows1, int rows2, float th_high,
                                         int32_t* match12, int32_t* n_matches) {
    return match_store(st, n_pairs, set1, set2, rows1, rows2, th_high, match12, nullptr, n_matches, true);
}

int hfnet_match_search_for_triangulation(hfnet_engine* eh, const float* d1, int n1, const float* d2, int n2, int dim, float th_high,
                                         int32_t* match12, int* n_matches, int on_device) {
    API_GUARD(eh, "engine"); API_GUARD(match12, "match12"); API_GUARD(n_matches, "n_matches");
    if (n1 < 0 || n2 < 0 || dim <= 0 || dim % 64) { set_error("bad matcher sizes (dim must be a multiple of 64)"); return HFNET_ERR_INVALID_ARG; }
    if ((n1 && !d1) || (n2 && !d2)) { set_error("null descriptor matrix"); return HFNET_ERR_INVALID_ARG; }
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    if (on_device) HF_HIP(e.wait_extract());
    if (n1 == 0) { if (!on_device) *n_matches = 0; else HF_HIP(hipMemsetAsync(n_matches, 0, sizeof(int), e.stream)); return HFNET_OK; }
    const float *da, *db;
    HF_TRY(stage_rows(e, e.m_a, d1, (size_t)n1 * dim, on_device, &da));
    HF_TRY(stage_rows(e, e.m_b, d2, (size_t)n2 * dim, on_device, &db));
    const int max_rows = std::max(n1, n2);
    HF_TRY(bow_scratch(e, 1, max_rows, dim, true));
    int32_t* d_match = match12; int* d_cnt = n_matches;
    if (!on_device) {
        HF_TRY(e.m_i0.ensure(sizeof(int32_t) * n1)); HF_TRY(e.m_cnt.ensure(sizeof(int)));
        d_match = e.m_i0.as<int32_t>(); d_cnt = e.m_cnt.as<int>();
    }
    BowPair P;
    P.q = da; P.t = db; P.St = e.m_s.as<float>(); P.qn = e.m_qn.as<float>(); P.tn = e.m_tn.as<float>(); P.qkey = e.m_key.as<unsigned long long>();
    P.match = d_match; P.dist = nullptr; P.cnt = d_cnt; P.nq = n1; P.nt = n2;
    HF_HIP(hipMemcpyAsync(e.m_pairs.p, &P, sizeof P, hipMemcpyHostToDevice, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));     // P lives on this stack frame
    const float threshold = (float)(-0.5 * th_high * th_high + 1);   // Matcher.cc:851
    HF_LAUNCH(&e, e.stream, "match_tri", launch_tri_pairs(e.m_pairs.as<BowPair>(), 1, max_rows, dim, threshold, e.stream, nullptr, nullptr));
    if (!on_device) {
        HF_HIP(hipMemcpyAsync(match12, d_match, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipMemcpyAsync(n_matches, d_cnt, sizeof(int), hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipStreamSynchronize(e.stream));
    }
    return HFNET_OK;
}

int hfnet_match_candidates(hfnet_engine* eh, const float* query, int n_query, const float* train, int n_train, const int32_t* train_level, int dim,
                           const int32_t* cand_offsets, const int32_t* cand_index, int32_t* best_idx, float* best_dist, int32_t* best_level,
                           float* second_dist, int32_t* second_level, int on_device) {
    API_GUARD(eh, "engine");
    if (n_query < 0 || n_train < 0 || dim <= 0 || dim % 4) { set_error("match_candidates: bad sizes (dim must be a multiple of 4)"); return HFNET_ERR_INVALID_ARG; }
    if (n_query == 0) return HFNET_OK;
    API_GUARD(query, "query"); API_GUARD(cand_offsets, "cand_offsets");
    API_GUARD(best_idx, "best_idx"); API_GUARD(best_dist, "best_dist"); API_GUARD(best_level, "best_level"); API_GUARD(second_dist, "second_dist"); API_GUARD(second_level, "second_level");
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    if (on_device) {
        HF_HIP(e.wait_extract());
        HF_LAUNCH(&e, e.stream, "match_candidates", launch_match_candidates(query, n_query, train, train_level, dim, cand_offsets, cand_index, best_idx,
                                                                       best_dist, best_level, second_dist, second_level, e.stream));
        return HFNET_OK;
    }
    const int total = cand_offsets[n_query];
    if (cand_offsets[0] != 0 || total < 0) { set_error("match_candidates: cand_offsets must start at 0 and be non-decreasing"); return HFNET_ERR_INVALID_ARG; }
    for (int i = 0; i < n_query; ++i) if (cand_offsets[i + 1] < cand_offsets[i]) { set_error("match_candidates: cand_offsets decrease at %d", i); return HFNET_ERR_INVALID_ARG; }
    if (total && (!cand_index || !train)) { set_error("match_candidates: null candidate list / train matrix"); return HFNET_ERR_INVALID_ARG; }
    for (int c = 0; c < total; ++c) if (cand_index[c] < 0 || cand_index[c] >= n_train) { set_error("match_candidates: candidate %d names row %d outside [0, %d)", c, cand_index[c], n_train); return HFNET_ERR_INVALID_ARG; }
    const float *dq, *dt;
    HF_TRY(stage_rows(e, e.m_a, query, (size_t)n_query * dim, 0, &dq));
    HF_TRY(stage_rows(e, e.m_b, train, (size_t)n_train * dim, 0, &dt));
    // [offsets n_query+1 | index total | level n_train] and the five outputs
    HF_TRY(e.m_i0.ensure(sizeof(int32_t) * ((size_t)n_query + 1 + (size_t)total + (size_t)n_train)));
    HF_TRY(e.m_i1.ensure(sizeof(int32_t) * 3 * (size_t)n_query)); HF_TRY(e.m_f0.ensure(sizeof(float) * 2 * (size_t)n_query));
    int32_t* ib = e.m_i0.as<int32_t>();
    HF_HIP(hipMemcpyAsync(ib, cand_offsets, sizeof(int32_t) * ((size_t)n_query + 1), hipMemcpyHostToDevice, e.stream));
    if (total) HF_HIP(hipMemcpyAsync(ib + n_query + 1, cand_index, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, e.stream));
    int32_t* d_level = nullptr;
    if (train_level && n_train) { d_level = ib + n_query + 1 + total; HF_HIP(hipMemcpyAsync(d_level, train_level, sizeof(int32_t) * (size_t)n_train, hipMemcpyHostToDevice, e.stream)); }
    int32_t* oi = e.m_i1.as<int32_t>(); float* of = e.m_f0.as<float>();
    HF_LAUNCH(&e, e.stream, "match_candidates", launch_match_candidates(dq, n_query, dt, d_level, dim, ib, ib + n_query + 1, oi, of, oi + n_query, of + n_query,
                                                                   oi + 2 * (size_t)n_query, e.stream));
    HF_HIP(hipMemcpyAsync(best_idx, oi, sizeof(int32_t) * n_query, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipMemcpyAsync(best_level, oi + n_query, sizeof(int32_t) * n_query, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipMemcpyAsync(second_level, oi + 2 * (size_t)n_query, sizeof(int32_t) * n_query, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipMemcpyAsync(best_dist, of, sizeof(float) * n_query, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipMemcpyAsync(second_dist, of + n_query, sizeof(float) * n_query, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

int hfnet_distinctive_descriptors(hfnet_engine* eh, const float* desc, const int32_t* set_offsets, int n_sets, int dim, int32_t* best) {
    API_GUARD(eh, "engine");
    if (n_sets < 0 || dim <= 0 || dim % 4) { set_error("distinctive_descriptors: bad sizes (dim must be a multiple of 4)"); return HFNET_ERR_INVALID_ARG; }
    if (n_sets == 0) return HFNET_OK;
    API_GUARD(set_offsets, "set_offsets"); API_GUARD(best, "best");
    if (set_offsets[0] != 0) { set_error("distinctive_descriptors: set_offsets must start at 0"); return HFNET_ERR_INVALID_ARG; }
    for (int s = 0; s < n_sets; ++s) {
        const int n = set_offsets[s + 1] - set_offsets[s];
        if (n < 0) { set_error("distinctive_descriptors: set_offsets decrease at %d", s); return HFNET_ERR_INVALID_ARG; }
        if (n > distinctive_max_rows()) { set_error("distinctive_descriptors: set %d has %d rows (> %d)", s, n, distinctive_max_rows()); return HFNET_ERR_CAPACITY; }
    }
    const int total = set_offsets[n_sets];
    if (total) API_GUARD(desc, "desc");
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    const float* dd;
    HF_TRY(stage_rows(e, e.m_a, desc, (size_t)total * dim, 0, &dd));
    HF_TRY(e.m_i0.ensure(sizeof(int32_t) * ((size_t)n_sets + 1))); HF_TRY(e.m_i1.ensure(sizeof(int32_t) * (size_t)n_sets));
    HF_HIP(hipMemcpyAsync(e.m_i0.p, set_offsets, sizeof(int32_t) * ((size_t)n_sets + 1), hipMemcpyHostToDevice, e.stream));
    HF_LAUNCH(&e, e.stream, "distinctive", launch_distinctive(dd, e.m_i0.as<int>(), n_sets, dim, e.m_i1.as<int>(), e.stream));
    HF_HIP(hipMemcpyAsync(best, e.m_i1.p, sizeof(int32_t) * (size_t)n_sets, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

int hfnet_resampler(hfnet_engine* eh, const float* data, const float* warp, float* output, int batch_size, int data_height, int data_width,
                    int data_channels, int num_sampling_points) {
    API_GUARD(eh, "engine"); API_GUARD(data, "data"); API_GUARD(output, "output");
    if (batch_size < 0 || data_height <= 0 || data_width <= 0 || data_channels <= 0 || num_sampling_points < 0) { set_error("resampler: bad sizes"); return HFNET_ERR_INVALID_ARG; }
    if (batch_size == 0 || num_sampling_points == 0) return HFNET_OK;
    API_GUARD(warp, "warp");
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    const size_t nd = (size_t)batch_size * data_height * data_width * data_channels, nw = (size_t)batch_size * num_sampling_points * 2;
    const size_t no = (size_t)batch_size * num_sampling_points * data_channels;
    HF_TRY(e.m_s.ensure(nd * sizeof(float))); HF_TRY(e.m_a.ensure(nw * sizeof(float))); HF_TRY(e.m_b.ensure(no * sizeof(float)));
    HF_HIP(hipMemcpyAsync(e.m_s.p, data, nd * sizeof(float), hipMemcpyHostToDevice, e.stream));
    HF_HIP(hipMemcpyAsync(e.m_a.p, warp, nw * sizeof(float), hipMemcpyHostToDevice, e.stream));
    HF_LAUNCH(&e, e.stream, "resampler", launch_resampler(e.m_s.as<float>(), e.m_a.as<float>(), e.m_b.as<float>(), batch_size, data_height, data_width,
                                                         data_channels, num_sampling_points, e.stream));
    HF_HIP(hipMemcpyAsync(output, e.m_b.p, no * sizeof(float), hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

// ---------------------------------------------------------------------------------------- KeyFrameDatabase
int hfnet_db_create(hfnet_engine* eh, int capacity, int dim, hfnet_db** out) {
    API_GUARD(out, "out");
    *out = nullptr;
    API_GUARD(eh, "engine");
    if (capacity < 1 || dim < 256 || dim % 256) { set_error("db: capacity >= 1 and dim a multiple of 256 required"); return HFNET_ERR_INVALID_ARG; }
    HF_HIP(hipSetDevice(eh->impl.device));
    std::unique_ptr<hfnet_db> db(new hfnet_db());
    db->eng = eh; db->capacity = capacity; db->dim = dim;
    HF_HIP(hipMalloc((void**)&db->d_db, sizeof(float) * (size_t)capacity * dim));
    HF_HIP(hipMalloc((void**)&db->d_occ, (size_t)capacity));
    HF_HIP(hipMalloc((void**)&db->d_q, sizeof(float) * dim));
    HF_HIP(hipMalloc((void**)&db->d_norm, sizeof(float) * capacity));
    HF_HIP(hipMalloc(&db->d_hi, (size_t)2 * capacity * dim));
    HF_HIP(hipMalloc((void**)&db->d_scores, sizeof(float) * capacity));
    HF_HIP(hipMalloc((void**)&db->d_cand_score, sizeof(float) * capacity));
    HF_HIP(hipMalloc((void**)&db->d_cand_slot, sizeof(int32_t) * capacity));
    HF_HIP(hipMalloc((void**)&db->d_best, sizeof(float)));
    HF_HIP(hipMalloc((void**)&db->d_n, sizeof(int)));
    HF_HIP(hipMalloc((void**)&db->d_best_bits, sizeof(unsigned int) * 4 * (size_t)db_scan_workgroups(capacity)));   // per-wave partial maxima
    {
        // on the stream the adds and scans use: it is non-blocking, i.e. NOT ordered with the null stream, and a hipMemset there
        // is not host-synchronous -- it could land after the first hfnet_db_add had set its occupancy byte
        Engine& e = eh->impl;
        std::lock_guard<std::mutex> lk(e.mu);
        HF_HIP(hipMemsetAsync(db->d_occ, 0, (size_t)capacity, e.stream));
        HF_HIP(hipMemsetAsync(db->d_norm, 0, sizeof(float) * capacity, e.stream));
        HF_HIP(hipStreamSynchronize(e.stream));
    }
    *out = db.release();
    return HFNET_OK;
}

void hfnet_db_destroy(hfnet_db* db) {
    if (!db) return;
    (void)hipSetDevice(db->eng->impl.device);
    for (void* p : {(void*)db->d_db, (void*)db->d_occ, (void*)db->d_q, (void*)db->d_scores, (void*)db->d_cand_score, (void*)db->d_cand_slot,
                    (void*)db->d_best, (void*)db->d_n, (void*)db->d_best_bits, (void*)db->d_norm, db->d_hi})
        if (p) (void)hipFree(p);
    delete db;
}

int hfnet_db_add(hfnet_db* db, int slot, const float* descriptor) {
    API_GUARD(db, "db"); API_GUARD(descriptor, "descriptor");
    if (slot < 0 || slot >= db->capacity) { set_error("db: slot %d outside [0, %d)", slot, db->capacity); return HFNET_ERR_CAPACITY; }
    std::lock_guard<std::mutex> lk(db->mu);
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    HF_HIP(hipSetDevice(e.device));
    // on the stream the scans run on (created non-blocking: the null stream would not order with it)
    HF_HIP(hipMemcpyAsync(db->d_db + (size_t)slot * db->dim, descriptor, sizeof(float) * db->dim, hipMemcpyHostToDevice, e.stream));
    HF_HIP(hipMemsetAsync(db->d_occ + slot, 1, 1, e.stream));
    db->norm_dirty = true;
    HF_HIP(hipStreamSynchronize(e.stream));                          // the host buffer may go away
    return HFNET_OK;
}

int hfnet_db_erase(hfnet_db* db, int slot) {
    API_GUARD(db, "db");
    if (slot < 0 || slot >= db->capacity) { set_error("db: slot %d outside [0, %d)", slot, db->capacity); return HFNET_ERR_CAPACITY; }
    std::lock_guard<std::mutex> lk(db->mu);
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    HF_HIP(hipSetDevice(e.device));
    HF_HIP(hipMemsetAsync(db->d_occ + slot, 0, 1, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

int hfnet_db_clear(hfnet_db* db) {
    API_GUARD(db, "db");
    std::lock_guard<std::mutex> lk(db->mu);
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    HF_HIP(hipSetDevice(e.device));
    HF_HIP(hipMemsetAsync(db->d_occ, 0, (size_t)db->capacity, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

int hfnet_db_query(hfnet_db* db, const float* query, int mode, int32_t* cand_slot, float* cand_score, int* n_cand, float* best_score,
                   float* scores_all) {
    API_GUARD(db, "db"); API_GUARD(query, "query"); API_GUARD(cand_slot, "cand_slot"); API_GUARD(cand_score, "cand_score"); API_GUARD(n_cand, "n_cand");
    if (mode != 0 && mode != 1) { set_error("db: mode must be 0 or 1"); return HFNET_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(db->mu);   // KeyFrameDatabase.cc:82 holds mMutex over the scan
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    HF_HIP(hipSetDevice(e.device));
    HF_HIP(hipMemcpyAsync(db->d_q, query, sizeof(float) * db->dim, hipMemcpyHostToDevice, e.stream));
    HF_LAUNCH(&e, e.stream, "db_scores", launch_db_scores(db->d_q, db->d_db, db->d_occ, db->capacity, db->dim, db->d_scores, db->d_best_bits, e.stream));
    HF_LAUNCH(&e, e.stream, "db_filter", launch_db_filter(db->d_scores, db->capacity, mode, db->d_best_bits, 4 * db_scan_workgroups(db->capacity), db->d_cand_slot, db->d_cand_score, db->d_n, db->d_best, 1, e.stream));
    int n = 0;
    float best = 0.f;
    HF_HIP(hipMemcpyAsync(&n, db->d_n, sizeof(int), hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipMemcpyAsync(&best, db->d_best, sizeof(float), hipMemcpyDeviceToHost, e.stream));
    if (scores_all) HF_HIP(hipMemcpyAsync(scores_all, db->d_scores, sizeof(float) * db->capacity, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    if (n > 0) {
        HF_HIP(hipMemcpyAsync(cand_slot, db->d_cand_slot, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipMemcpyAsync(cand_score, db->d_cand_score, sizeof(float) * n, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipStreamSynchronize(e.stream));
    }
    *n_cand = n;
    if (best_score) *best_score = best;
    return HFNET_OK;
}

int hfnet_db_query_batch(hfnet_db* db, int n_queries, const float* queries, int mode, int32_t* cand_slot, float* cand_score, int32_t* n_cand,
                         float* best_score, float* scores_all) {
    API_GUARD(db, "db");
    if (n_queries < 0) { set_error("db: n_queries < 0"); return HFNET_ERR_INVALID_ARG; }
    if (n_queries == 0) return HFNET_OK;
    API_GUARD(queries, "queries"); API_GUARD(cand_slot, "cand_slot"); API_GUARD(cand_score, "cand_score"); API_GUARD(n_cand, "n_cand");
    if (mode != 0 && mode != 1) { set_error("db: mode must be 0 or 1"); return HFNET_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(db->mu);
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    const bool gemm = n_queries >= e.opt.db_gemm_min_queries && db->dim % 512 == 0;
    if (!gemm && db->dim > 4096) { set_error("db: the exact batched scan supports dim <= 4096"); return HFNET_ERR_INVALID_ARG; }
    HF_HIP(hipSetDevice(e.device));
    const size_t Q = (size_t)n_queries, cap = (size_t)db->capacity;
    // per-call scratch: [Q][dim] queries, [Q][cap] scores / candidates, [Q] best / counts
    HF_TRY(e.m_a.ensure(sizeof(float) * Q * db->dim));
    HF_TRY(e.m_s.ensure(sizeof(float) * Q * cap));
    HF_TRY(e.m_f0.ensure(sizeof(float) * Q * cap));
    HF_TRY(e.m_i0.ensure(sizeof(int32_t) * Q * cap));
    HF_TRY(e.m_cnt.ensure(sizeof(int32_t) * Q));
    HF_TRY(e.m_qn.ensure(sizeof(float) * Q));
    const int parts = gemm ? db_gemm_partials(db->capacity) : 4 * db_batch_workgroups(db->capacity);
    HF_TRY(e.m_key.ensure(sizeof(unsigned int) * Q * parts));
    if (gemm) {
        HF_TRY(e.m_tn.ensure(sizeof(float) * Q)); HF_TRY(e.m_b.ensure(sizeof(float) * db_gemm_scratch_floats(db->capacity, n_queries)));
        HF_TRY(e.m_f1.ensure((size_t)2 * Q * db->dim));             // bf16 copies of the queries
    }
    float* d_q = e.m_a.as<float>(); float* d_scores = e.m_s.as<float>(); float* d_cs = e.m_f0.as<float>();
    int32_t* d_slot = e.m_i0.as<int32_t>(); int* d_n = e.m_cnt.as<int>(); float* d_best = e.m_qn.as<float>();
    unsigned int* d_bits = e.m_key.as<unsigned int>();
    HF_HIP(hipMemcpyAsync(d_q, queries, sizeof(float) * Q * db->dim, hipMemcpyHostToDevice, e.stream));
    if (gemm) {
        if (db->norm_dirty) {
            HF_LAUNCH(&e, e.stream, "db_norm", launch_db_prep_hi(db->d_db, db->capacity, db->dim, db->d_norm, db->d_hi, e.stream));
            db->norm_dirty = false;
        }
        HF_LAUNCH(&e, e.stream, "db_qnorm", launch_db_prep_hi(d_q, n_queries, db->dim, e.m_tn.as<float>(), e.m_f1.p, e.stream));
        HF_LAUNCH(&e, e.stream, "db_screen", launch_db_screen(d_q, e.m_f1.p, n_queries, e.m_tn.as<float>(), db->d_db, db->d_hi, db->d_norm, db->d_occ, db->capacity,
                                                         db->dim, d_scores, d_bits, e.m_b.as<float>(), e.stream));
    } else {
        HF_LAUNCH(&e, e.stream, "db_scores_batch", launch_db_scores_batch(d_q, n_queries, db->d_db, db->d_occ, db->capacity, db->dim, d_scores, d_bits, e.stream));
    }
    HF_LAUNCH(&e, e.stream, "db_filter", launch_db_filter(d_scores, db->capacity, mode, d_bits, parts, d_slot, d_cs, d_n, d_best, n_queries, e.stream));
    HF_HIP(hipMemcpyAsync(n_cand, d_n, sizeof(int32_t) * Q, hipMemcpyDeviceToHost, e.stream));
    if (best_score) HF_HIP(hipMemcpyAsync(best_score, d_best, sizeof(float) * Q, hipMemcpyDeviceToHost, e.stream));
    if (scores_all) HF_HIP(hipMemcpyAsync(scores_all, d_scores, sizeof(float) * Q * cap, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipStreamSynchronize(e.stream));
    for (size_t qi = 0; qi < Q; ++qi) {
        const int n = n_cand[qi];
        if (n <= 0) continue;
        HF_HIP(hipMemcpyAsync(cand_slot + qi * cap, d_slot + qi * cap, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e.stream));
        HF_HIP(hipMemcpyAsync(cand_score + qi * cap, d_cs + qi * cap, sizeof(float) * n, hipMemcpyDeviceToHost, e.stream));
    }
    HF_HIP(hipStreamSynchronize(e.stream));
    return HFNET_OK;
}

// ---------------------------------------------------------------------------------------- profiling
int hfnet_profile_enable(hfnet_engine* e, int on) {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    if (!on) e->impl.prof.flush();
    e->impl.prof.enabled = on != 0;
    return HFNET_OK;
}
int hfnet_profile_reset(hfnet_engine* e) {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    e->impl.prof.reset();
    return HFNET_OK;
}
int hfnet_profile_filter(hfnet_engine* e, const char* name) {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    e->impl.prof.filter = name ? name : "";
    return HFNET_OK;
}
int hfnet_profile_count(hfnet_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    e->impl.prof.flush();
    return (int)e->impl.prof.names.size();
}
int hfnet_profile_get(hfnet_engine* e, int i, char* name, int name_cap, int* launches, double* total_ms) {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    Profiler& p = e->impl.prof;
    p.flush();
    if (i < 0 || i >= (int)p.names.size()) { set_error("profile index %d out of range", i); return HFNET_ERR_INVALID_ARG; }
    if (name && name_cap > 0) { std::strncpy(name, p.names[i].c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = p.launches[i];
    if (total_ms) *total_ms = p.total_ms[i];
    return HFNET_OK;
}

}  // extern "C"
