// api_match.hip -- C ABI: the brute-force bodies of Matcher (src/Matcher.cc), the device-resident keyframe descriptor store, the windowed
// matchers' candidate loop, ComputeDistinctiveDescriptors and the free-standing Resampler.  Host code; the kernels are in kernels_match.hip.
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace hfnet;

namespace hfnet {
int stage_rows(Engine& e, DevMem& m, const float* src, size_t count, int on_device, const float** out) {
    if (on_device) { *out = src; return HFNET_OK; }
    HF_TRY(m.ensure(std::max<size_t>(count, 1) * sizeof(float)));
    if (count) HF_TRY(e.h2d(m.p, src, count * sizeof(float)));
    *out = m.as<float>();
    return HFNET_OK;
}
}  // namespace hfnet

extern "C" {

// ---------------------------------------------------------------------------------------- Matcher

int hfnet_descriptor_distance(hfnet_engine* eh, const float* a, const float* b, int dim, float* out) try {
    API_GUARD(eh, "engine"); API_GUARD(a, "a"); API_GUARD(b, "b"); API_GUARD(out, "out");
    if (dim <= 0) { set_error("dim <= 0"); return HFNET_ERR_INVALID_ARG; }
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    const float *da, *db;
    HF_TRY(stage_rows(e, e.m_a, a, dim, 0, &da));
    HF_TRY(stage_rows(e, e.m_b, b, dim, 0, &db));
    HF_TRY(e.m_f0.ensure(sizeof(float)));
    HF_LAUNCH(&e, e.stream, "descriptor_distance", launch_descriptor_distance(da, db, dim, e.m_f0.as<float>(), e.stream));
    HF_TRY(e.d2h(out, e.m_f0.p, sizeof(float)));
    HF_TRY(e.sync_host());
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

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
                              int32_t* match_q2t, float* dist, int* n_matches, int on_device) try {
    API_GUARD(eh, "engine"); API_GUARD(match_q2t, "match_q2t"); API_GUARD(dist, "dist"); API_GUARD(n_matches, "n_matches");
    if (n_query < 0 || n_train < 0 || dim <= 0 || dim % 64) { set_error("bad matcher sizes (dim must be a multiple of 64)"); return HFNET_ERR_INVALID_ARG; }
    if ((n_query && !query) || (n_train && !train)) { set_error("null descriptor matrix"); return HFNET_ERR_INVALID_ARG; }
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    e.bounce_discard();
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
    HF_TRY(e.h2d(e.m_pairs.p, &P, sizeof P));
    HF_TRY(e.sync_host());     // P lives on this stack frame
    HF_LAUNCH(&e, e.stream, "match_bow", launch_bow_pairs(e.m_pairs.as<BowPair>(), 1, max_rows, dim, th_low, e.m_s.p, e.stream, e.opt.match_screen_bf16, e.opt.match_stats ? e.bow_stat() : nullptr));
    if (!on_device) {
        HF_TRY(e.d2h(match_q2t, d_match, sizeof(int32_t) * n_query));
        HF_TRY(e.d2h(dist, d_dist, sizeof(float) * n_query));
        HF_TRY(e.d2h(n_matches, d_cnt, sizeof(int)));
        HF_TRY(e.sync_host());
    }
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

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
    e.bounce_discard();
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
        HF_TRY(e.h2d(e.m_a.p, desc_base, sizeof(float) * (size_t)n_sets * set_stride));
        int32_t* ib = e.m_b.as<int32_t>();
        HF_TRY(e.h2d(ib, n_rows, sizeof(int32_t) * n_sets));
        HF_TRY(e.h2d(ib + n_sets, query_set, sizeof(int32_t) * n_pairs));
        HF_TRY(e.h2d(ib + n_sets + n_pairs, train_set, sizeof(int32_t) * n_pairs));
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
        HF_LAUNCH(&e, e.stream, "match_bow", launch_bow_pairs(e.m_pairs.as<BowPair>(), n_pairs, max_rows, dim, th, e.m_s.p, e.stream, e.opt.match_screen_bf16, e.opt.match_stats ? e.bow_stat() : nullptr));
    }
    if (!on_device) {
        HF_TRY(e.d2h(match_q2t, d_match, sizeof(int32_t) * (size_t)n_pairs * max_rows));
        if (!triangulation) HF_TRY(e.d2h(dist, d_dist, sizeof(float) * (size_t)n_pairs * max_rows));
        HF_TRY(e.d2h(n_matches, d_cnt, sizeof(int32_t) * n_pairs));
        HF_TRY(e.sync_host());
    }
    return HFNET_OK;
}

int hfnet_match_search_by_bow_batch(hfnet_engine* eh, int n_pairs, const float* desc_base, size_t set_stride, const int32_t* n_rows, int n_sets,
                                    const int32_t* query_set, const int32_t* train_set, int max_rows, int dim, float th_low, int32_t* match_q2t,
                                    float* dist, int32_t* n_matches, int on_device) try {
    return match_pairs_batch(eh, n_pairs, desc_base, set_stride, n_rows, n_sets, query_set, train_set, max_rows, dim, th_low, match_q2t, dist,
                             n_matches, on_device, false);
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_match_search_for_triangulation_batch(hfnet_engine* eh, int n_pairs, const float* desc_base, size_t set_stride, const int32_t* n_rows,
                                               int n_sets, const int32_t* set1, const int32_t* set2, int max_rows, int dim, float th_high,
                                               int32_t* match12, int32_t* n_matches, int on_device) try {
    return match_pairs_batch(eh, n_pairs, desc_base, set_stride, n_rows, n_sets, set1, set2, max_rows, dim, th_high, match12, nullptr, n_matches,
                             on_device, true);
} catch (...) { return ::hfnet::api_exception(); }

// ---------------------------------------------------------------------------------------- descriptor store
int hfnet_store_create(hfnet_engine* eh, int n_sets, int max_rows, int dim, hfnet_store** out) try {
    API_GUARD(out, "out");
    *out = nullptr;
    API_GUARD(eh, "engine");
    if (n_sets < 1 || max_rows < 1 || dim <= 0 || dim % 64) { set_error("store: n_sets, max_rows >= 1 and dim a multiple of 64 required"); return HFNET_ERR_INVALID_ARG; }
    HF_HIP(hipSetDevice(eh->impl.device));
    struct Cleanup { void operator()(hfnet_store* d) const { hfnet_store_destroy(d); } };     // (frees whatever had been allocated when a later step fails)
    std::unique_ptr<hfnet_store, Cleanup> st(new hfnet_store);
    st->eng = eh; st->n_sets = n_sets; st->max_rows = max_rows; st->dim = dim;
    st->rows.assign(n_sets, 0);
    HF_HIP(dev_malloc((void**)&st->d_desc, sizeof(float) * (size_t)n_sets * max_rows * dim));
    HF_HIP(dev_malloc((void**)&st->d_rows, sizeof(int32_t) * n_sets));
    HF_HIP(dev_malloc((void**)&st->d_flags, (size_t)n_sets * max_rows));
    {   // on the engine's (non-blocking) stream, which every later put / match uses: see hfnet_db_create
        Engine& e = eh->impl;
        std::lock_guard<std::mutex> lk(e.mu);
    e.bounce_discard();
        HF_HIP(hipMemsetAsync(st->d_rows, 0, sizeof(int32_t) * n_sets, e.stream));
        HF_HIP(hipMemsetAsync(st->d_flags, 0, (size_t)n_sets * max_rows, e.stream));
        HF_TRY(e.sync_host());
    }
    *out = st.release();
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

void hfnet_store_destroy(hfnet_store* st) try {
    if (!st) return;
    (void)hipSetDevice(st->eng->impl.device);
    (void)hipDeviceSynchronize();
    (void)dev_free(st->d_desc);
    (void)dev_free(st->d_rows);
    (void)dev_free(st->d_flags);
    delete st;
} catch (...) { (void)::hfnet::api_exception(); }

int hfnet_store_put(hfnet_store* st, int slot, const float* rows, int n_rows) try {
    API_GUARD(st, "store");
    if (slot < 0 || slot >= st->n_sets || n_rows < 0 || n_rows > st->max_rows) { set_error("store: slot %d / %d rows outside [0, %d) / [0, %d]", slot, n_rows, st->n_sets, st->max_rows); return HFNET_ERR_INVALID_ARG; }
    if (n_rows && !rows) { set_error("null descriptor matrix"); return HFNET_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(st->mu);
    Engine& e = st->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    const int32_t n = n_rows;
    if (n_rows) HF_TRY(e.h2d(st->d_desc + (size_t)slot * st->max_rows * st->dim, rows, sizeof(float) * (size_t)n_rows * st->dim));
    HF_TRY(e.h2d(st->d_rows + slot, &n, sizeof n));
    HF_HIP(hipMemsetAsync(st->d_flags + (size_t)slot * st->max_rows, 0, (size_t)st->max_rows, e.stream));
    HF_TRY(e.sync_host());                          // the host buffers may go away
    st->rows[slot] = n;
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_store_rows(const hfnet_store* st, int slot) try {
    if (!st || slot < 0 || slot >= st->n_sets) return -1;
    return st->rows[slot];
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_store_set_flags(hfnet_store* st, int slot, const uint8_t* flags, int n_rows) try {
    API_GUARD(st, "store");
    if (slot < 0 || slot >= st->n_sets || n_rows < 0 || n_rows > st->max_rows) { set_error("store: slot %d / %d rows outside [0, %d) / [0, %d]", slot, n_rows, st->n_sets, st->max_rows); return HFNET_ERR_INVALID_ARG; }
    if (n_rows == 0) return HFNET_OK;
    API_GUARD(flags, "flags");
    std::lock_guard<std::mutex> lk(st->mu);
    Engine& e = st->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    HF_TRY(e.h2d(st->d_flags + (size_t)slot * st->max_rows, flags, (size_t)n_rows));
    HF_TRY(e.sync_host());
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_store_put_extracted(hfnet_store* st, int slot, hfnet_extractor* x, int frame) try {
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
    e.bounce_discard();
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
} catch (...) { return ::hfnet::api_exception(); }

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
    e.bounce_discard();
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
    HF_TRY(e.h2d(ib, host.data(), sizeof(int32_t) * host.size()));
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
        HF_LAUNCH(&e, e.stream, "match_bow", launch_bow_pairs(e.m_pairs.as<BowPair>(), n_pairs, mr, st->dim, th, e.m_s.p, e.stream, e.opt.match_screen_bf16, e.opt.match_stats ? e.bow_stat() : nullptr));
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
        HF_TRY(e.sync_host());
        std::memcpy(match, hp, b_match);
        if (b_dist) std::memcpy(dist, hp + b_match, b_dist);
        std::memcpy(n_matches, hp + b_match + b_dist, b_cnt);
        return HFNET_OK;
    }
    HF_TRY(e.d2h(match, d_match, b_match));
    if (!triangulation) HF_TRY(e.d2h(dist, d_dist, b_dist));
    HF_TRY(e.d2h(n_matches, d_cnt, b_cnt));
    HF_TRY(e.sync_host());
    return HFNET_OK;
}

int hfnet_store_search_by_bow(hfnet_store* st, int n_pairs, const int32_t* query_set, const int32_t* train_set, int query_rows, int train_rows,
                              float th_low, int32_t* match_q2t, float* dist, int32_t* n_matches) try {
    return match_store(st, n_pairs, query_set, train_set, query_rows, train_rows, th_low, match_q2t, dist, n_matches, false);
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_store_search_for_triangulation(hfnet_store* st, int n_pairs, const int32_t* set1, const int32_t* set2, int rows1, int rows2, float th_high,
                                         int32_t* match12, int32_t* n_matches) try {
    return match_store(st, n_pairs, set1, set2, rows1, rows2, th_high, match12, nullptr, n_matches, true);
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_match_search_for_triangulation(hfnet_engine* eh, const float* d1, int n1, const float* d2, int n2, int dim, float th_high,
                                         int32_t* match12, int* n_matches, int on_device) try {
    API_GUARD(eh, "engine"); API_GUARD(match12, "match12"); API_GUARD(n_matches, "n_matches");
    if (n1 < 0 || n2 < 0 || dim <= 0 || dim % 64) { set_error("bad matcher sizes (dim must be a multiple of 64)"); return HFNET_ERR_INVALID_ARG; }
    if ((n1 && !d1) || (n2 && !d2)) { set_error("null descriptor matrix"); return HFNET_ERR_INVALID_ARG; }
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    e.bounce_discard();
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
    HF_TRY(e.h2d(e.m_pairs.p, &P, sizeof P));
    HF_TRY(e.sync_host());     // P lives on this stack frame
    const float threshold = (float)(-0.5 * th_high * th_high + 1);   // Matcher.cc:851
    HF_LAUNCH(&e, e.stream, "match_tri", launch_tri_pairs(e.m_pairs.as<BowPair>(), 1, max_rows, dim, threshold, e.stream, nullptr, nullptr));
    if (!on_device) {
        HF_TRY(e.d2h(match12, d_match, sizeof(int32_t) * n1));
        HF_TRY(e.d2h(n_matches, d_cnt, sizeof(int)));
        HF_TRY(e.sync_host());
    }
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_match_candidates(hfnet_engine* eh, const float* query, int n_query, const float* train, int n_train, const int32_t* train_level, int dim,
                           const int32_t* cand_offsets, const int32_t* cand_index, int32_t* best_idx, float* best_dist, int32_t* best_level,
                           float* second_dist, int32_t* second_level, int on_device) try {
    API_GUARD(eh, "engine");
    if (n_query < 0 || n_train < 0 || dim <= 0 || dim % 4) { set_error("match_candidates: bad sizes (dim must be a multiple of 4)"); return HFNET_ERR_INVALID_ARG; }
    if (n_query == 0) return HFNET_OK;
    API_GUARD(query, "query"); API_GUARD(cand_offsets, "cand_offsets");
    API_GUARD(best_idx, "best_idx"); API_GUARD(best_dist, "best_dist"); API_GUARD(best_level, "best_level"); API_GUARD(second_dist, "second_dist"); API_GUARD(second_level, "second_level");
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    e.bounce_discard();
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
    HF_TRY(e.h2d(ib, cand_offsets, sizeof(int32_t) * ((size_t)n_query + 1)));
    if (total) HF_TRY(e.h2d(ib + n_query + 1, cand_index, sizeof(int32_t) * (size_t)total));
    int32_t* d_level = nullptr;
    if (train_level && n_train) { d_level = ib + n_query + 1 + total; HF_TRY(e.h2d(d_level, train_level, sizeof(int32_t) * (size_t)n_train)); }
    int32_t* oi = e.m_i1.as<int32_t>(); float* of = e.m_f0.as<float>();
    HF_LAUNCH(&e, e.stream, "match_candidates", launch_match_candidates(dq, n_query, dt, d_level, dim, ib, ib + n_query + 1, oi, of, oi + n_query, of + n_query,
                                                                   oi + 2 * (size_t)n_query, e.stream));
    HF_TRY(e.d2h(best_idx, oi, sizeof(int32_t) * n_query));
    HF_TRY(e.d2h(best_level, oi + n_query, sizeof(int32_t) * n_query));
    HF_TRY(e.d2h(second_level, oi + 2 * (size_t)n_query, sizeof(int32_t) * n_query));
    HF_TRY(e.d2h(best_dist, of, sizeof(float) * n_query));
    HF_TRY(e.d2h(second_dist, of + n_query, sizeof(float) * n_query));
    HF_TRY(e.sync_host());
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_distinctive_descriptors(hfnet_engine* eh, const float* desc, const int32_t* set_offsets, int n_sets, int dim, int32_t* best) try {
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
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    const float* dd;
    HF_TRY(stage_rows(e, e.m_a, desc, (size_t)total * dim, 0, &dd));
    HF_TRY(e.m_i0.ensure(sizeof(int32_t) * ((size_t)n_sets + 1))); HF_TRY(e.m_i1.ensure(sizeof(int32_t) * (size_t)n_sets));
    HF_TRY(e.h2d(e.m_i0.p, set_offsets, sizeof(int32_t) * ((size_t)n_sets + 1)));
    HF_LAUNCH(&e, e.stream, "distinctive", launch_distinctive(dd, e.m_i0.as<int>(), n_sets, dim, e.m_i1.as<int>(), e.stream));
    HF_TRY(e.d2h(best, e.m_i1.p, sizeof(int32_t) * (size_t)n_sets));
    HF_TRY(e.sync_host());
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_resampler(hfnet_engine* eh, const float* data, const float* warp, float* output, int batch_size, int data_height, int data_width,
                    int data_channels, int num_sampling_points) try {
    API_GUARD(eh, "engine"); API_GUARD(data, "data"); API_GUARD(output, "output");
    if (batch_size < 0 || data_height <= 0 || data_width <= 0 || data_channels <= 0 || num_sampling_points < 0) { set_error("resampler: bad sizes"); return HFNET_ERR_INVALID_ARG; }
    if (batch_size == 0 || num_sampling_points == 0) return HFNET_OK;
    API_GUARD(warp, "warp");
    Engine& e = eh->impl;
    std::lock_guard<std::mutex> lk(e.mu);
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    const size_t nd = (size_t)batch_size * data_height * data_width * data_channels, nw = (size_t)batch_size * num_sampling_points * 2;
    const size_t no = (size_t)batch_size * num_sampling_points * data_channels;
    HF_TRY(e.m_s.ensure(nd * sizeof(float))); HF_TRY(e.m_a.ensure(nw * sizeof(float))); HF_TRY(e.m_b.ensure(no * sizeof(float)));
    HF_TRY(e.h2d(e.m_s.p, data, nd * sizeof(float)));
    HF_TRY(e.h2d(e.m_a.p, warp, nw * sizeof(float)));
    HF_LAUNCH(&e, e.stream, "resampler", launch_resampler(e.m_s.as<float>(), e.m_a.as<float>(), e.m_b.as<float>(), batch_size, data_height, data_width,
                                                         data_channels, num_sampling_points, e.stream));
    HF_TRY(e.d2h(output, e.m_b.p, no * sizeof(float)));
    HF_TRY(e.sync_host());
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

}  // extern "C"
