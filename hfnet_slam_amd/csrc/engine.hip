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
#include <stdexcept>
#include <thread>

namespace hfnet {

int* Options::find(const char* name) {
    if (!name) return nullptr;
    const struct { const char* n; int* p; } tab[] = {{"fuse_blocks", &fuse_blocks}, {"fuse_max_layer", &fuse_max_layer}, {"fused_variant", &fused_variant},
                                                       {"fuse_stem", &fuse_stem}, {"dense_desc", &dense_desc}, {"two_streams", &two_streams},
                                                       {"graph", &graph}, {"pinned_frames", &pinned_frames}, {"db_gemm_min_queries", &db_gemm_min_queries}, {"db_screen_min_rows", &db_screen_min_rows},
                                                       {"conv_wlds", &conv_wlds}, {"fuse_min_wgs", &fuse_min_wgs}, {"copy_threads", &copy_threads}, {"tail_fuse", &tail_fuse}, {"dedupe_taps", &dedupe_taps}, {"pyramid_fuse", &pyramid_fuse}, {"resize_band", &resize_band}, {"fc_tile", &fc_tile}, {"interleave", &interleave}, {"host_global", &host_global}, {"det_fuse", &det_fuse}, {"match_screen_bf16", &match_screen_bf16}, {"tri_screen_bf16", &tri_screen_bf16}, {"desc_bf16x3", &desc_bf16x3}, {"global_bf16x3", &global_bf16x3}, {"scores_bf16x3", &scores_bf16x3}, {"join_fused_branch", &join_fused_branch}, {"match_stats", &match_stats}};
    for (const auto& t : tab) if (std::strcmp(t.n, name) == 0) return t.p;
    return nullptr;
}

// ------------------------------------------------------------------------------------ memory
int DevMem::ensure(size_t n) {
    if (n <= bytes) return HFNET_OK;
    if (p) { (void)dev_free(p); p = nullptr; bytes = 0; }
    HF_HIP(dev_malloc(&p, n));
    bytes = n;
    return HFNET_OK;
}
void DevMem::release() {
    if (p) (void)dev_free(p);
    p = nullptr;
    bytes = 0;
}

Engine::~Engine() {
    (void)hipSetDevice(device);
    prof.flush();
    for (auto ev : prof.pool) (void)hipEventDestroy(ev);
    if (stream) (void)hipStreamSynchronize(stream);          // (the statistics copy of a screened SearchForTriangulation may still be in flight)
    for (DevMem* m : {&m_a, &m_b, &m_s, &m_qn, &m_tn, &m_key, &m_i0, &m_i1, &m_f0, &m_f1, &m_cnt, &m_pairs, &m_tri_stat, &m_bow_stat}) m->release();
    w.release();
    if (h_res) (void)hipHostFree(h_res);
    if (bounce.base) (void)hipHostFree(bounce.base);
    if (h_tri_stat) (void)hipHostFree(h_tri_stat);
    if (ev_extract) (void)hipEventDestroy(ev_extract);
    if (ev_match) (void)hipEventDestroy(ev_match);
    if (ev_tri_stat) (void)hipEventDestroy(ev_tri_stat);
    if (stream) (void)hipStreamDestroy(stream);
}

int Engine::sync_host() {
    HF_HIP(hipStreamSynchronize(stream));
    for (const HostBounce::Pending& p : bounce.pending) std::memcpy(p.dst, p.src, p.bytes);
    bounce.pending.clear();
    bounce.used = 0;
    return HFNET_OK;
}
int Engine::bounce_take(size_t bytes, unsigned char** out) {
    const size_t need = (bytes + 255) / 256 * 256;
    if (need > HostBounce::kMaxBytes) { set_error("bounce_take: %zu bytes in one piece", bytes); return HFNET_ERR_INTERNAL; }
    if (bounce.used + need > bounce.cap) {
        HF_TRY(sync_host());                                  // nothing is in flight through the block any more
        if (need > bounce.cap) {
            const size_t cap = std::min(std::max<size_t>(std::max(need, bounce.cap * 2), (size_t)1 << 20), HostBounce::kMaxBytes);
            if (bounce.base) { (void)hipHostFree(bounce.base); bounce.base = nullptr; bounce.cap = 0; }
            void* hp = nullptr;
            HF_HIP(hipHostMalloc(&hp, cap, hipHostMallocDefault));
            bounce.base = (unsigned char*)hp; bounce.cap = cap;
        }
    }
    *out = bounce.base + bounce.used;
    bounce.used += need;
    return HFNET_OK;
}
// (transfers above HostBounce::kPiece cross in pieces: the pinned block stays at <= kMaxBytes whatever a caller uploads -- a batch of
//  descriptor sets can be hundreds of MB -- and a full block is drained, not grown)
int Engine::h2d(void* dst_dev, const void* src_host, size_t bytes) {
    for (size_t off = 0; off < bytes; off += HostBounce::kPiece) {
        const size_t n = std::min(HostBounce::kPiece, bytes - off);
        unsigned char* b = nullptr;
        HF_TRY(bounce_take(n, &b));
        std::memcpy(b, (const unsigned char*)src_host + off, n);
        HF_HIP(hipMemcpyAsync((unsigned char*)dst_dev + off, b, n, hipMemcpyHostToDevice, stream));
    }
    return HFNET_OK;
}
int* Engine::bow_stat() {
    if (!m_bow_stat.p) {
        if (m_bow_stat.ensure(2 * sizeof(int)) != HFNET_OK) return nullptr;
        if (hipMemsetAsync(m_bow_stat.p, 0, 2 * sizeof(int), stream) != hipSuccess) return nullptr;
    }
    return m_bow_stat.as<int>();
}
int Engine::d2h(void* dst_host, const void* src_dev, size_t bytes) {
    for (size_t off = 0; off < bytes; off += HostBounce::kPiece) {
        const size_t n = std::min(HostBounce::kPiece, bytes - off);
        unsigned char* b = nullptr;
        HF_TRY(bounce_take(n, &b));
        HF_HIP(hipMemcpyAsync(b, (const unsigned char*)src_dev + off, n, hipMemcpyDeviceToHost, stream));
        bounce.pending.push_back({(unsigned char*)dst_host + off, b, n});
    }
    return HFNET_OK;
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

// ------------------------------------------------------------------------------------ Net
static int layer_channels(const DeviceWeights& w, int layer) { return layer == 1 ? w.stem_out : w.blocks[layer - 2].cout; }

void compute_offsets(Net& n, int batch) {
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
    fuse_blocks = e->opt.fuse_blocks; fuse_max_layer = e->opt.fuse_max_layer; fused_variant = e->opt.fused_variant; fuse_min_wgs = e->opt.fuse_min_wgs; tail_fuse = e->opt.tail_fuse; dedupe_taps = e->opt.dedupe_taps; interleave = e->opt.interleave; det_fuse = e->opt.det_fuse; desc_bf16x3 = e->opt.desc_bf16x3; global_bf16x3 = e->opt.global_bf16x3; scores_bf16x3 = e->opt.scores_bf16x3;
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
        HF_TRY(dalloc(allocs, &tap_cell_row, images * (size_t)cell_stride));
        HF_TRY(dalloc(allocs, &tap_cells, rows));
        HF_TRY(dalloc(allocs, &tap_nrows, images));
        HF_TRY(dalloc(allocs, &dev_fault, 1));
        // On the network's OWN stream, and waited for: every forward() runs on this (non-blocking) stream, which is not ordered with the
        // null stream, and hipMemset is not host-synchronous on this runtime.  Until round 5 this was a null-stream hipMemset:
        // hfnet_model_create returned with the clear possibly still queued, hfnet_model_detect's k_tap_compact could then see the
        // allocation's previous contents, number up to H/8 * W/8 "marked" cells into a row list sized 4 * max_keypoints and hand that
        // count to the gathered descriptor head -- writes past the end of three buffers, a GPU memory fault, the host process aborted
        // (GPUTEST_r04; reproduced by tools/dev/null_stream_race.py, NOTEBOOK.md R5.1).  k_tap_compact also bounds its row numbers now.
        HF_HIP(hipMemsetAsync(tap_flags, 0, images * (size_t)cell_stride, stream));
        HF_HIP(hipMemsetAsync(tap_nrows, 0, sizeof(int) * images, stream));
        HF_HIP(hipMemsetAsync(n_level, 0, sizeof(int) * images, stream));
        HF_HIP(hipMemsetAsync(dev_fault, 0, sizeof(unsigned int), stream));
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
    HF_HIP(hipStreamSynchronize(stream));                    // the clears above have landed when build() returns
    return HFNET_OK;
}

void Net::release() {
    for (void* p : allocs) (void)dev_free(p);
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
    bool fuse = block_runs_fused(n, L);
    // blocks that only have a split-bf16 fused form (layers 16, 17: 120 -> 720 -> 120): with the option on and a launch that fills the chip they take
    // it instead of three launches (the 720-channel expanded tensor then never reaches HBM)
    bool bf_only = false;
    if (!fuse && n.global_bf16x3 && n.fuse_blocks && L > 14 && block_fused_bf16x3_supported(b)) {
        const LevelPlan& p0 = n.lp[0];
        const long long wgs = (long long)((p0.w[L] + 15) / 16) * ((p0.h[L] + 7) / 8) * n.cfg.batch;
        bf_only = wgs >= std::max(n.fuse_min_wgs, 1);            // (calls of <= tail_fuse frames never get here: Net::tail_chain)
        fuse = bf_only;
    }
    if (fuse) {
        // option global_bf16x3 (layers past the index-deciding part of the network only): the block's 1x1 convolutions on split-bf16 operands
        // option scores_bf16x3: the same for layers 3-7 -- the score map then moves within its stated tolerance, and NMS / top-K are exact ON IT
        const bool bfb = bf_only || (((n.global_bf16x3 && L > 7) || (n.scores_bf16x3 && L <= 7)) && block_fused_bf16x3_supported(b));
        if (L > 7) n.branch_fused_used = true;
        char fn[32];
        snprintf(fn, sizeof fn, bfb ? "block_L%02d_bf16x3" : "block_L%02d", L);
        const Geom gf = n.geom(L - 1, L, 0, n_used);
        HF_LAUNCH(e, st, fn, launch_block_fused(n.act[L - 1], b, n.act[L], gf, n.fused_variant, st, bfb ? 1 : 0));
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
        if (scores_bf16x3 && w.det1_bf && conv3x3_dense_bf16x3_supported(w.det1, gh))
            HF_LAUNCH(e, stream, "conv3x3_det_bf16x3", launch_conv3x3_dense_bf16x3(act[7], w.det1, w.det1_bf, det_hidden, 1, gh, stream));
        else
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
            if (scores_bf16x3 && w.det2_bf && det_tail_bf16x3_supported(w.det2))
                HF_LAUNCH(e, stream, "det_tail_bf16x3", launch_det_tail_bf16x3(det_hidden, w.det2, w.det2_bf, dense, gd, stream));
            else
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
        // Descriptor head: sparse or dense is a function of the budget alone (known here), see below
        long long tap_rows = 0;
        for (int l = 0; l < NL; ++l) tap_rows += 4ll * std::min(budget.k[l], cfg.max_keypoints) * cfg.batch;
        last_sparse = !force_dense && tap_rows * 5 < pc * 4;
        last_dedupe = last_sparse && dedupe_taps != 0;
        Geom gt = gn;   // H, W: score map; Ho, Wo: cell grid; in_off: first cell of the level
        for (int l = 0; l < NL; ++l) { gt.lv[l].Ho = lp[l].h[7]; gt.lv[l].Wo = lp[l].w[7]; gt.lv[l].in_off = pix_cell[l]; }
        if (last_dedupe && dedupe_taps == 2) {
            // the two-launch form of the same row numbering (mark, then compact): kept reachable so that it cannot rot (tests' variant "dedupe_two_launch")
            HF_LAUNCH(e, stream, "topk", launch_topk(cand, counters, cand_stride, budget, kps_level, cfg.max_keypoints, n_level, gn, stream));
            HF_LAUNCH(e, stream, "tap_cells", launch_tap_cells(kps_level, n_level, cfg.max_keypoints, tap_flags, tap_cell_row, tap_cells, tap_nrows, cell_stride, gt, stream, dev_fault));
        } else if (last_dedupe) {
            // top-K and the distinct tap cells of its keypoints in one launch (both are one workgroup per image)
            HF_LAUNCH(e, stream, "topk", launch_topk_taps(cand, counters, cand_stride, budget, kps_level, cfg.max_keypoints, n_level, tap_flags, tap_cell_row, tap_cells,
                                                          tap_nrows, cell_stride, gt, stream, dev_fault));
        } else {
            HF_LAUNCH(e, stream, "topk", launch_topk(cand, counters, cand_stride, budget, kps_level, cfg.max_keypoints, n_level, gn, stream));
        }
        HF_TRY(pump_global(1));
        // Descriptor head.  Only the 4 bilinear taps of every selected keypoint are ever read
        // (HFNetTFModelV2.cc:153-167), so unless the budget covers most of the cell grid the head is
        // evaluated at those taps only (same arithmetic per cell -> bit-identical descriptors).
        dense_valid = false;
        nms_valid = false; last_threshold = threshold;
        if (last_sparse) {
            int kmaxb = 0;
            for (int l = 0; l < NL; ++l) kmaxb = std::max(kmaxb, std::min(budget.k[l], cfg.max_keypoints));
            const long long rows = ((long long)(NL * cfg.batch - 1) * cfg.max_keypoints + kmaxb) * 4;
            if (last_dedupe) {
                // taps shared by neighbouring keypoints are evaluated once: the rows of an image are its DISTINCT tap cells (numbered by the
                // top-K launch above)
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
    // NOTEBOOK.md R4.8: with fused-block kernels in the global branch of a call of <= 4 frames (fuse_min_wgs lowered) the sampler, which runs while
    // that branch is still going, returned wrong rows -- its compiler-packed v_pk_mul_f32 / v_pk_add_f32 gave wrong values in lanes 48-63
    // beside those kernels.  The library is built without packed f32 instructions now (build.py); join_fused_branch = 1 brings back the stop-gap
    // of joining the branch first.
    const bool early_join = caller_joins && branch_fused_used && e->opt.join_fused_branch;
    branch_fused_used = false;
    if (cfg.global && fork_early && caller_joins && !early_join) join_pending = true;
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
        if (global_bf16x3 && w.memb_bf)
            HF_GSTEP(HF_LAUNCH(e, st, "pointwise_memberships_bf16x3", launch_pointwise_bf16x3(act[18], w.memb, w.memb_bf, nullptr, memb, (long long)cfg.batch * P, 0, st)));
        else
            HF_GSTEP(HF_LAUNCH(e, st, "pointwise_memberships", launch_pointwise(act[18], w.memb, nullptr, memb, (long long)cfg.batch * P, 0, st)));
    }
    if (!tail) HF_GSTEP(HF_LAUNCH(e, st, "softmax_memberships", launch_softmax_rows(memb, (long long)cfg.batch * P, w.n_clusters, w.n_clusters, st)));
    HF_GSTEP(HF_LAUNCH(e, st, "vlad", launch_vlad_aggregate(act[18], memb, w.clusters, vlad_raw, cfg.batch, P, w.c_global, w.n_clusters, st));
             HF_LAUNCH(e, st, "vlad_norm", launch_vlad_norm(vlad_raw, vlad_tap, vlad_out, cfg.batch, w.c_global, w.n_clusters, st)));
    if (global_bf16x3 && w.fc_bf && cfg.batch >= 64 && !global_host.out)
        HF_GSTEP(HF_LAUNCH(e, st, "fc_l2_bf16x3", launch_fc_l2_bf16x3(vlad_out, w.fc, w.fc_bf, fc_part, fc_raw, global_dst ? global_dst : global_out, cfg.batch, st)));
    else
        HF_GSTEP(HF_LAUNCH(e, st, "fc_l2", launch_fc_l2(vlad_out, w.fc, fc_part, fc_raw, global_dst ? global_dst : global_out, cfg.batch, st, global_host, e->opt.fc_tile)));
#undef HF_GSTEP
    if (total) *total = step;
    return HFNET_OK;
}

int Net::clear_faults(unsigned int seen) {
    sticky_faults |= seen;
    if (dev_fault) HF_HIP(hipMemsetAsync(dev_fault, 0, sizeof(unsigned int), stream));
    return HFNET_OK;
}

int Net::read_faults(unsigned int* out) {
    *out = sticky_faults;
    if (!dev_fault) return HFNET_OK;
    void* hp = nullptr;                                       // (a pinned word: no pageable memory is handed to the runtime, see HostBounce)
    HF_HIP(hipHostMalloc(&hp, 64, hipHostMallocDefault));
    hipError_t er = hipMemcpyAsync(hp, dev_fault, sizeof(unsigned int), hipMemcpyDeviceToHost, stream);
    if (er == hipSuccess) er = hipStreamSynchronize(stream);
    if (er == hipSuccess) *out = *(volatile unsigned int*)hp | sticky_faults;
    (void)hipHostFree(hp);
    HF_HIP(er);
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
        HF_HIP(dev_malloc((void**)&tmp, count * sizeof(float)));
        hipError_t er = launch_permute_channels(src, tmp, (long long)(count / permute_c), permute_c, 1, stream);
        if (er != hipSuccess) { (void)dev_free(tmp); set_error("tap permute failed: %s", hipGetErrorString(er)); return HFNET_ERR_DEVICE; }
        src = tmp;
    }
    // (diagnostics: through a pinned block of its own -- no pageable memory is handed to the runtime anywhere, see HostBounce)
    void* hp = nullptr;
    hipError_t er = hipHostMalloc(&hp, std::max<size_t>(count, 1) * sizeof(float), hipHostMallocDefault);
    if (er == hipSuccess) er = hipMemcpyAsync(hp, src, count * sizeof(float), hipMemcpyDeviceToHost, stream);
    if (er == hipSuccess) er = hipStreamSynchronize(stream);
    if (er == hipSuccess) std::memcpy(out.data(), hp, count * sizeof(float));
    if (hp) (void)hipHostFree(hp);
    if (permute_c) (void)dev_free((void*)src);
    if (er != hipSuccess) { set_error("tap copy failed: %s", hipGetErrorString(er)); return HFNET_ERR_DEVICE; }
    return HFNET_OK;
}

// ------------------------------------------------------------------------------------ tables
// HFextractor ctor (HFextractor.cc:82-119) and ComputePyramid sizes (:159-166)
void extractor_tables(int nfeatures, int nlevels, float scale_factor, int width, int height, float* sf, int* fpl, int* lw, int* lh) {
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
void resize_tables(int sw, int sh, int dw, int dh, std::vector<int>& xofs, std::vector<short>& ialpha, std::vector<int>& yofs,
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

extern "C" {


const char* hfnet_last_error(void) { return get_error(); }
int hfnet_abi_version(void) { return HFNET_ABI_VERSION; }
#ifndef HFNET_BUILD_ID
#define HFNET_BUILD_ID "hfnet-build-id:unknown"
#endif
// (the marker prefix lets hfnet_slam_amd/build.py read the id from the file without loading it)
const char* hfnet_build_id(void) { static const char id[] = HFNET_BUILD_ID; return &id[sizeof("hfnet-build-id:") - 1]; }

int hfnet_device_count(void) try {
    int n = 0;
    const hipError_t er = hipGetDeviceCount(&n);
    if (er != hipSuccess || n <= 0) { set_error("no HIP device visible (%s)", hipGetErrorString(er)); return 0; }
    return n;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_engine_create(int device, const char* weights_path, hfnet_engine** out) try {
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
} catch (...) { return ::hfnet::api_exception(); }

void hfnet_engine_destroy(hfnet_engine* e) { delete e; }

int hfnet_engine_info(const hfnet_engine* e, int what) try {
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
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_engine_set_option(hfnet_engine* e, const char* name, int value) try {
    // test hook of the exception barrier every entry point ends in (tests/test_abi.py; needs no engine and no GPU): name "debug_throw"
    // -- only with HFNET_TEST_HOOKS=1 in the environment: a production caller that passes this name gets "unknown option" like any other
    if (name && std::strcmp(name, "debug_throw") == 0 && std::getenv("HFNET_TEST_HOOKS") && std::strcmp(std::getenv("HFNET_TEST_HOOKS"), "1") == 0) {
        if (value == 1) throw std::bad_alloc();
        throw std::runtime_error("debug_throw");
    }
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
} catch (...) { return ::hfnet::api_exception(); }
int hfnet_engine_get_option(hfnet_engine* e, const char* name, int* value) try {
    API_GUARD(e, "engine"); API_GUARD(value, "value");
    std::lock_guard<std::mutex> lk(e->impl.mu);
    const int stat_ix = !name ? -1 : std::strcmp(name, "stat_bow_exact") == 0 ? 0 : std::strcmp(name, "stat_db_exact") == 0 ? 1 : -1;
    if (stat_ix >= 0) {
        // read-only statistics (engine option match_stats): exact distance evaluations of the SearchByBoW calls / exact scores of the screened
        // batched database queries since the last read -- what the screens let through: a broken screen still returns the right results
        // (everything is then evaluated exactly), and this is where it shows; waits for the stream
        Engine& en = e->impl;
        HF_HIP(hipSetDevice(en.device));
        int v = 0;
        if (en.m_bow_stat.p) {
            HF_TRY(en.d2h(&v, en.m_bow_stat.as<int>() + stat_ix, sizeof(int)));
            HF_HIP(hipMemsetAsync(en.m_bow_stat.as<int>() + stat_ix, 0, sizeof(int), en.stream));
            HF_TRY(en.sync_host());
        }
        *value = v;
        return HFNET_OK;
    }
    const int* p = e->impl.opt.find(name);
    if (!p) { set_error("unknown engine option '%s'", name ? name : "(null)"); return HFNET_ERR_INVALID_ARG; }
    *value = *p;
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_engine_synchronize(hfnet_engine* e) try {
    API_GUARD(e, "engine");
    HF_HIP(hipSetDevice(e->impl.device));
    HF_HIP(hipDeviceSynchronize());
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_engine_fence(hfnet_engine* eh) try {
    API_GUARD(eh, "engine");
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    HF_HIP(hipSetDevice(e.device));
    std::lock_guard<std::mutex> lk2(e.ev_mu);
    if (!e.ev_match) HF_HIP(hipEventCreateWithFlags(&e.ev_match, hipEventDisableTiming));
    HF_HIP(hipEventRecord(e.ev_match, e.stream));
    e.ev_match_set = true;
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

// ---------------------------------------------------------------------------------------- profiling
int hfnet_profile_enable(hfnet_engine* e, int on) try {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    if (!on) e->impl.prof.flush();
    e->impl.prof.enabled = on != 0;
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }
int hfnet_profile_reset(hfnet_engine* e) try {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    e->impl.prof.reset();
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }
int hfnet_profile_filter(hfnet_engine* e, const char* name) try {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    e->impl.prof.filter = name ? name : "";
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }
int hfnet_profile_count(hfnet_engine* e) try {
    if (!e) return 0;
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    e->impl.prof.flush();
    return (int)e->impl.prof.names.size();
} catch (...) { return ::hfnet::api_exception(); }
int hfnet_profile_get(hfnet_engine* e, int i, char* name, int name_cap, int* launches, double* total_ms) try {
    API_GUARD(e, "engine");
    std::lock_guard<std::mutex> lk(e->impl.prof_mu);
    Profiler& p = e->impl.prof;
    p.flush();
    if (i < 0 || i >= (int)p.names.size()) { set_error("profile index %d out of range", i); return HFNET_ERR_INVALID_ARG; }
    if (name && name_cap > 0) { std::strncpy(name, p.names[i].c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = p.launches[i];
    if (total_ms) *total_ms = p.total_ms[i];
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

}  // extern "C"
