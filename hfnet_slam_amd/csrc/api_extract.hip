// api_extract.hip -- C ABI: BaseModel (include/Extractors/BaseModel.h:38-54) and HFextractor (src/Extractors/HFextractor.cc:82-284) over the
// engine's per-level networks: single frames (captured graph + pinned block), batches (device-resident / staged / registered host buffers).
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace hfnet;

extern "C" {

// ---------------------------------------------------------------------------------------- BaseModel
int hfnet_model_create(hfnet_engine* e, hfnet_mode mode, int height, int width, int max_keypoints, hfnet_model** out) try {
    API_GUARD(out, "out");
    *out = nullptr;
    API_GUARD(e, "engine");
    if (mode < HFNET_IMAGE_TO_LOCAL_AND_GLOBAL || mode > HFNET_INTERMEDIATE_TO_GLOBAL) { set_error("unknown mode %d", (int)mode); return HFNET_ERR_INVALID_ARG; }
    if (height <= 0 || width <= 0) { set_error("bad input shape %dx%d", width, height); return HFNET_ERR_SHAPE; }
    if (max_keypoints < 1) max_keypoints = 1;
    if (max_keypoints > HFNET_MAX_KEYPOINTS) { set_error("max_keypoints %d > %d", max_keypoints, HFNET_MAX_KEYPOINTS); return HFNET_ERR_CAPACITY; }
    HF_HIP(hipSetDevice(e->impl.device));
    struct Cleanup { void operator()(hfnet_model* d) const { hfnet_model_destroy(d); } };      // (frees whatever had been allocated when a later step fails)
    std::unique_ptr<hfnet_model, Cleanup> m(new hfnet_model());
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
    {
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const Engine& eng = e->impl;
        const size_t img = c.local ? up((size_t)height * width) : 0;
        const size_t inter = sizeof(float) * (size_t)(c.from_intermediate ? height * width : (height / 8) * (width / 8)) * eng.w.c_local;
        const size_t aux = up(std::max(inter, sizeof(float) * (size_t)eng.w.global_dim));
        m->o_n = img; m->o_aux = m->o_n + 256; m->o_kps = m->o_aux + aux;
        m->o_desc = m->o_kps + up(sizeof(hfnet_keypoint) * (size_t)max_keypoints);
        m->stage_bytes = m->o_desc + up(sizeof(float) * HFNET_DESC_DIM * (size_t)max_keypoints);
        void* hp = nullptr;
        HF_HIP(hipHostMalloc(&hp, m->stage_bytes, hipHostMallocDefault));
        m->h_stage = (unsigned char*)hp;
    }
    m->valid = true;
    *out = m.release();
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

void hfnet_model_destroy(hfnet_model* m) try {
    if (!m) return;
    (void)hipSetDevice(m->eng->impl.device);
    if (m->h_stage) { if (m->net.stream) (void)hipStreamSynchronize(m->net.stream); (void)hipHostFree(m->h_stage); }
    delete m;
} catch (...) { (void)::hfnet::api_exception(); }

int hfnet_model_is_valid(const hfnet_model* m) { return m && m->valid ? 1 : 0; }
int hfnet_model_mode(const hfnet_model* m) { return m ? (int)m->mode : -1; }

int hfnet_model_detect(hfnet_model* m, const uint8_t* image, int row_stride, int n_keypoints, float threshold, hfnet_keypoint* kps,
                       float* local_desc, float* aux, int* n_out) try {
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
    // the image goes up through the model's pinned block (the caller's rows are read by the host only: see hfnet_model::h_stage)
    if (row_stride == m->width) std::memcpy(m->h_stage, image, (size_t)m->width * m->height);
    else for (int y = 0; y < m->height; ++y) std::memcpy(m->h_stage + (size_t)y * m->width, image + (size_t)y * row_stride, (size_t)m->width);
    HF_HIP(hipMemcpyAsync(m->d_image, m->h_stage, (size_t)m->width * m->height, hipMemcpyHostToDevice, net.stream));
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
    sa.fault = net.dev_fault;
    Geom gs = net.geom(7, 7, 0, 1);
    gs.lv[0].H = net.lp[0].Hc; gs.lv[0].W = net.lp[0].Wc; gs.lv[0].Ho = net.lp[0].h[7]; gs.lv[0].Wo = net.lp[0].w[7];
    gs.lv[0].in_off = net.pix_cell[0];
    HF_LAUNCH(&m->eng->impl, net.stream, "sample", launch_sample(sa, gs, net.stream));
    int* h_n = (int*)(m->h_stage + m->o_n);                      // [0] keypoints, [1] fault word
    h_n[0] = 0; h_n[1] = 0;
    float* h_aux = (float*)(m->h_stage + m->o_aux);
    size_t aux_bytes = 0;
    HF_HIP(hipMemcpyAsync(h_n, m->d_n, sizeof(int), hipMemcpyDeviceToHost, net.stream));
    if (net.dev_fault) HF_HIP(hipMemcpyAsync(h_n + 1, net.dev_fault, sizeof(unsigned int), hipMemcpyDeviceToHost, net.stream));
    if (m->mode == HFNET_IMAGE_TO_LOCAL_AND_GLOBAL) {
        aux_bytes = sizeof(float) * m->eng->impl.w.global_dim;
        HF_HIP(hipMemcpyAsync(h_aux, net.global_out, aux_bytes, hipMemcpyDeviceToHost, net.stream));
    } else if (m->mode == HFNET_IMAGE_TO_LOCAL_AND_INTERMEDIATE) {
        const long long P = (long long)net.lp[0].h[7] * net.lp[0].w[7];
        const int C = m->eng->impl.w.c_local;
        HF_LAUNCH(&m->eng->impl, net.stream, "permute", launch_permute_channels(net.act[7], net.inter_logical, P, C, 1, net.stream));
        aux_bytes = sizeof(float) * P * C;
        HF_HIP(hipMemcpyAsync(h_aux, net.inter_logical, aux_bytes, hipMemcpyDeviceToHost, net.stream));
    }
    HF_HIP(hipStreamSynchronize(net.stream));
    const int n = std::min(std::max(h_n[0], 0), n_keypoints);
    const unsigned int faults = (unsigned int)h_n[1];
    if (faults) {
        // the reference's contract is a per-call `return false` (HFNetTFModelV2.cc:62-98): THIS call fails, the word is cleared on the stream so that
        // the next call (whose forward pass rebuilds every flag it reads) starts clean; hfnet_model_device_faults keeps the bits for the object's lifetime
        HF_TRY(net.clear_faults(faults));
        set_error("device-side bound hit (fault bits 0x%x): results of this call discarded", faults);
        return HFNET_ERR_DEVICE;
    }
    if (aux_bytes) std::memcpy(aux, h_aux, aux_bytes);
    if (n > 0) {
        HF_HIP(hipMemcpyAsync(m->h_stage + m->o_kps, m->d_kps, sizeof(hfnet_keypoint) * n, hipMemcpyDeviceToHost, net.stream));
        HF_HIP(hipMemcpyAsync(m->h_stage + m->o_desc, m->d_desc, sizeof(float) * HFNET_DESC_DIM * n, hipMemcpyDeviceToHost, net.stream));
        HF_HIP(hipStreamSynchronize(net.stream));
        std::memcpy(kps, m->h_stage + m->o_kps, sizeof(hfnet_keypoint) * n);
        std::memcpy(local_desc, m->h_stage + m->o_desc, sizeof(float) * HFNET_DESC_DIM * n);
    }
    *n_out = n;
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_model_detect_global(hfnet_model* m, const float* intermediate, float* global_desc) try {
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
    float* h_aux = (float*)(m->h_stage + m->o_aux);               // (pinned block: see hfnet_model::h_stage)
    std::memcpy(h_aux, intermediate, sizeof(float) * P * C);
    HF_HIP(hipMemcpyAsync(net.inter_logical, h_aux, sizeof(float) * P * C, hipMemcpyHostToDevice, net.stream));
    HF_LAUNCH(&eng, net.stream, "permute", launch_permute_channels(net.inter_logical, net.act[7], P, C, 0, net.stream));
    ImageSet imgs;
    std::memset(&imgs, 0, sizeof imgs);
    TopkBudget budget;
    std::memset(&budget, 0, sizeof budget);
    HF_TRY(net.forward(imgs, 0.f, budget));
    HF_HIP(hipStreamSynchronize(net.stream));                    // (the upload has left the block before the result overwrites it)
    HF_HIP(hipMemcpyAsync(h_aux, net.global_out, sizeof(float) * eng.w.global_dim, hipMemcpyDeviceToHost, net.stream));
    HF_HIP(hipStreamSynchronize(net.stream));
    std::memcpy(global_desc, h_aux, sizeof(float) * eng.w.global_dim);
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_model_device_faults(hfnet_model* m, unsigned int* bits) try {
    API_GUARD(m, "model"); API_GUARD(bits, "bits");
    std::lock_guard<std::mutex> lk(m->mu);
    HF_HIP(hipSetDevice(m->eng->impl.device));
    return m->net.read_faults(bits);
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_model_tap(hfnet_model* m, int tap, float* out, size_t capacity, size_t* count) try {
    API_GUARD(m, "model"); API_GUARD(out, "out"); API_GUARD(count, "count");
    std::lock_guard<std::mutex> lk(m->mu);
    HF_HIP(hipSetDevice(m->eng->impl.device));
    std::vector<float> v;
    HF_TRY(m->net.tap(tap, v));
    *count = v.size();
    if (v.size() > capacity) { set_error("tap %d needs %zu floats, buffer holds %zu", tap, v.size(), capacity); return HFNET_ERR_CAPACITY; }
    std::memcpy(out, v.data(), v.size() * sizeof(float));
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

// ---------------------------------------------------------------------------------------- HFextractor
int hfnet_extractor_create(hfnet_engine* e, int width, int height, int n_features, float threshold, float scale_factor, int n_levels,
                           int max_batch, hfnet_extractor** out) try {
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
    int opt_resize_band;
    {   // (the engine's option block is written under this lock: hfnet_engine_set_option)
        std::lock_guard<std::mutex> lk(e->impl.mu);
        x->use_graph = e->impl.opt.graph;
        x->host_global = e->impl.opt.host_global;
        opt_resize_band = e->impl.opt.resize_band;
    }
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
        HF_HIP(copy_h2d_blocking(x->d_xofs[l], xofs.data(), xofs.size() * sizeof(int)));
        HF_HIP(copy_h2d_blocking(x->d_ialpha[l], ia.data(), ia.size() * sizeof(short)));
        HF_HIP(copy_h2d_blocking(x->d_yofs[l], yofs.data(), yofs.size() * sizeof(int)));
        HF_HIP(copy_h2d_blocking(x->d_ibeta[l], ib.data(), ib.size() * sizeof(short)));
        x->pyr_band_rows[l] = opt_resize_band ? resize_band_rows(yofs.data(), x->level_h[l], x->level_h[l - 1]) : 0;
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
            HF_HIP(hipMemsetAsync(x->d_seq, 0, 3 * sizeof(int), x->net.stream));   // (the stream its users run on; a null-stream clear is not ordered with it)
            HF_HIP(hipStreamSynchronize(x->net.stream));
            void* hp = nullptr;
            // (coherent: kernels write results and the call's number into this block while the host spins on it mid-graph)
            if (hipHostMalloc(&hp, off, hipHostMallocCoherent) == hipSuccess) { x->h_pin = (unsigned char*)hp; x->pinned_frames = pf; *(volatile int*)(x->h_pin + x->pin_flag) = 0; *(volatile int*)(x->h_pin + x->pin_flag + 128) = 0; }
            else (void)hipGetLastError();
        }
    }
    *out = x.release();
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

void hfnet_extractor_destroy(hfnet_extractor* x) try {
    if (!x) return;
    (void)hipSetDevice(x->eng->impl.device);
    // (a single-frame call returns when its results are in the caller's buffers, which is before its graph has retired)
    if (x->net.stream) (void)hipStreamSynchronize(x->net.stream);
    if (x->net.stream_global) (void)hipStreamSynchronize(x->net.stream_global);
    for (auto& kv : x->graphs) (void)hipGraphExecDestroy(kv.second);
    for (void* p : x->allocs) (void)dev_free(p);
    if (x->h_pin) (void)hipHostFree(x->h_pin);
    for (int s = 0; s < 2; ++s) {
        if (x->pipe.h_in[s]) (void)hipHostFree(x->pipe.h_in[s]);
        if (x->pipe.h_out[s]) (void)hipHostFree(x->pipe.h_out[s]);
        for (hipEvent_t ev : {x->pipe.ev_up[s], x->pipe.ev_comp[s], x->pipe.ev_down[s]}) if (ev) (void)hipEventDestroy(ev);
    }
    if (x->pipe.s_up) (void)hipStreamDestroy(x->pipe.s_up);
    if (x->pipe.s_down) (void)hipStreamDestroy(x->pipe.s_down);
    delete x;
} catch (...) { (void)::hfnet::api_exception(); }

int hfnet_extractor_tables(const hfnet_extractor* x, float* scale_factors, int* features_per_level, int* level_width, int* level_height) try {
    API_GUARD(x, "extractor");
    for (int l = 0; l < x->n_levels; ++l) {
        if (scale_factors) scale_factors[l] = x->scale_factors[l];
        if (features_per_level) features_per_level[l] = x->features_per_level[l];
        if (level_width) level_width[l] = x->level_w[l];
        if (level_height) level_height[l] = x->level_h[l];
    }
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

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
    const bool chain = nb <= eng.opt.pyramid_fuse && x->n_levels >= 2 && pyramid_chain_supported(x->n_levels - 1, x->level_w, x->level_h);
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
                                   (long long)dwp * dh, x->d_xofs[l], x->d_ialpha[l], x->d_yofs[l], x->d_ibeta[l], nb, net.stream, x->pyr_band_rows[l]));
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
    sa.fault = net.dev_fault;
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
int hfnet_host_register(void* ptr, size_t bytes) try {
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
} catch (...) { return ::hfnet::api_exception(); }
int hfnet_host_unregister(void* ptr) try {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_registered.find((uintptr_t)ptr);
    if (it == g_registered.end()) { set_error("hfnet_host_unregister: not a registered range"); return HFNET_ERR_INVALID_ARG; }
    HF_HIP(hipHostUnregister(ptr));        // (first: on failure the range is still page-locked AND still known here, a retry can succeed)
    g_registered.erase(it);
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

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
        // The copy engines read the caller's images and write the caller's result buffers: no error return may leave such a copy in
        // flight (the caller is free to unregister / free the buffers once the call has returned).  Whatever way the loop is left,
        // the three streams are drained first.
        struct Quiesce {
            hipStream_t a, b, c; bool armed = true;
            ~Quiesce() { if (armed) { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(c); } }
        } quiesce{p.s_up, st, p.s_down};
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
        quiesce.armed = false;                                // (every download has been waited for)
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

int hfnet_extractor_attach_store(hfnet_extractor* x, hfnet_store* s, int first_slot) try {
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
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_extractor_extract_batch(hfnet_extractor* x, int n_frames, const uint8_t* images, int row_stride, size_t frame_stride,
                                  hfnet_keypoint* kps, float* local_desc, float* global_desc, int* n_out, int on_device) try {
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
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_extractor_device_faults(hfnet_extractor* x, unsigned int* bits) try {
    API_GUARD(x, "extractor"); API_GUARD(bits, "bits");
    std::lock_guard<std::mutex> lk(x->mu);
    HF_HIP(hipSetDevice(x->eng->impl.device));
    return x->net.read_faults(bits);
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_extractor_tap(hfnet_extractor* x, int tap, float* out, size_t capacity, size_t* count) try {
    API_GUARD(x, "extractor"); API_GUARD(out, "out"); API_GUARD(count, "count");
    std::lock_guard<std::mutex> lk(x->mu);
    HF_HIP(hipSetDevice(x->eng->impl.device));
    std::vector<float> v;
    HF_TRY(x->net.tap(tap, v));
    *count = v.size();
    if (v.size() > capacity) { set_error("tap %d needs %zu floats, buffer holds %zu", tap, v.size(), capacity); return HFNET_ERR_CAPACITY; }
    std::memcpy(out, v.data(), v.size() * sizeof(float));
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_extractor_last_timing(hfnet_extractor* x, double* us, int n) try {
    API_GUARD(x, "extractor"); API_GUARD(us, "us");
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->t_last[5] < 0) { set_error("no latency-path call yet"); return HFNET_ERR_INVALID_ARG; }
    for (int i = 0; i < n && i < 6; ++i) us[i] = x->t_last[i];
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_extractor_extract(hfnet_extractor* x, const uint8_t* image, int row_stride, hfnet_keypoint* kps, float* local_desc,
                            float* global_desc, int* n_out, int* n_per_level) try {
    if (n_out) *n_out = -1;
    API_GUARD(x, "extractor"); API_GUARD(n_out, "n_out");
    if (!image) { set_error("empty image"); return HFNET_ERR_INVALID_ARG; }   // HFextractor.cc:145 returns -1
    int n = 0;
    HF_TRY(hfnet_extractor_extract_batch(x, 1, image, row_stride, (size_t)row_stride * x->height, kps, local_desc, global_desc, &n, 0));
    *n_out = n;
    if (n_per_level) {
        std::lock_guard<std::mutex> lk(x->mu);
        if (x->h_pin && x->pinned_frames >= 1) std::memcpy(n_per_level, x->h_pin + x->pin_nl_last, sizeof(int) * x->n_levels);   // came down with the frame
        else HF_HIP(copy_d2h_blocking(n_per_level, x->d_n_level, sizeof(int) * x->n_levels));
    }
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

}  // extern "C"
