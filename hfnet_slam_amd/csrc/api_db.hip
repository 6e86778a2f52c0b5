// api_db.hip -- C ABI: the global-descriptor store and scans of KeyFrameDatabase (src/KeyFrameDatabase.cc:36-104, 178-197).
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace hfnet;

// The screened scan of n_queries device rows d_q (kernels_match.hip: the database's 8-bit copy streamed once per 64 queries, the bound test and
// the exact chain for what is left in the same kernel): scores [n_queries][capacity], per-tile maxima [n_queries][db_gemm_partials(capacity)].
// Brings the copy up to date for the slots added since the last screened scan.  Engine lock held by the caller.
static int db_screened_scan(Engine& e, hfnet_db* db, const float* d_q, int n_queries, float* d_scores, unsigned int* d_bits) {
    HF_TRY(e.m_tn.ensure(sizeof(float) * db_stat_floats(n_queries)));
    HF_TRY(e.m_f1.ensure(db_hi_bytes((n_queries + 63) & ~63, db->dim)));      // 8-bit copies of the queries (a launch reads whole groups of 32 / 64)
    int* db_stat = e.opt.match_stats && e.bow_stat() ? e.bow_stat() + 1 : nullptr;
    if (db->dirty_lo < db->dirty_hi) {
        // only the 32-row tiles the added slots fall into (keyframes arrive in slot order: usually one tile): a tile's 8-bit steps are
        // contiguous in the copy, its rows' statistics in the array
        const int r0 = db->dirty_lo & ~31, r1 = std::min(db->capacity, (db->dirty_hi + 31) & ~31);
        HF_LAUNCH(&e, e.stream, "db_norm", launch_db_prep_hi(db->d_db + (size_t)r0 * db->dim, r1 - r0, db->dim, db->d_norm + db_stat_floats(r0),
                                                        (unsigned char*)db->d_hi + db_hi_bytes(r0, db->dim), e.stream));
        db->dirty_lo = db->dirty_hi = 0;
    }
    HF_LAUNCH(&e, e.stream, "db_qnorm", launch_db_prep_hi(d_q, n_queries, db->dim, e.m_tn.as<float>(), e.m_f1.p, e.stream));
    for (int q0 = 0; q0 < n_queries; q0 += 64)
        HF_LAUNCH(&e, e.stream, "db_screen", launch_db_sweep(d_q, e.m_f1.p, n_queries, q0, e.m_tn.as<float>(), db->d_db, db->d_hi, db->d_norm, db->d_occ, db->capacity,
                                                            db->dim, d_scores, d_bits, e.stream, db_stat));
    return HFNET_OK;
}

extern "C" {

// ---------------------------------------------------------------------------------------- KeyFrameDatabase
int hfnet_db_create(hfnet_engine* eh, int capacity, int dim, hfnet_db** out) try {
    API_GUARD(out, "out");
    *out = nullptr;
    API_GUARD(eh, "engine");
    if (capacity < 1 || dim < 256 || dim % 256) { set_error("db: capacity >= 1 and dim a multiple of 256 required"); return HFNET_ERR_INVALID_ARG; }
    HF_HIP(hipSetDevice(eh->impl.device));
    struct Cleanup { void operator()(hfnet_db* d) const { hfnet_db_destroy(d); } };      // (frees whatever had been allocated when a later step fails)
    std::unique_ptr<hfnet_db, Cleanup> db(new hfnet_db());
    db->eng = eh; db->capacity = capacity; db->dim = dim;
    HF_HIP(dev_malloc((void**)&db->d_db, sizeof(float) * (size_t)capacity * dim));
    HF_HIP(dev_malloc((void**)&db->d_occ, (size_t)capacity));
    HF_HIP(dev_malloc((void**)&db->d_q, sizeof(float) * dim));
    HF_HIP(dev_malloc((void**)&db->d_norm, sizeof(float) * db_stat_floats(capacity)));
    HF_HIP(dev_malloc(&db->d_hi, db_hi_bytes(capacity, dim)));
    HF_HIP(dev_malloc((void**)&db->d_scores, sizeof(float) * capacity));
    HF_HIP(dev_malloc((void**)&db->d_cand_score, sizeof(float) * capacity));
    HF_HIP(dev_malloc((void**)&db->d_cand_slot, sizeof(int32_t) * capacity));
    HF_HIP(dev_malloc((void**)&db->d_best, sizeof(float)));
    HF_HIP(dev_malloc((void**)&db->d_n, sizeof(int)));
    HF_HIP(dev_malloc((void**)&db->d_best_bits, sizeof(unsigned int) * std::max((size_t)4 * db_scan_workgroups(capacity), (size_t)db_gemm_partials(capacity))));   // per-wave / per-tile partial maxima
    {
        // on the stream the adds and scans use: it is non-blocking, i.e. NOT ordered with the null stream, and a hipMemset there
        // is not host-synchronous -- it could land after the first hfnet_db_add had set its occupancy byte
        Engine& e = eh->impl;
        std::lock_guard<std::mutex> lk(e.mu);
        e.bounce_discard();
        HF_HIP(hipMemsetAsync(db->d_occ, 0, (size_t)capacity, e.stream));
        HF_HIP(hipMemsetAsync(db->d_norm, 0, sizeof(float) * db_stat_floats(capacity), e.stream));
        HF_TRY(e.sync_host());
    }
    *out = db.release();
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

void hfnet_db_destroy(hfnet_db* db) try {
    if (!db) return;
    (void)hipSetDevice(db->eng->impl.device);
    for (void* p : {(void*)db->d_db, (void*)db->d_occ, (void*)db->d_q, (void*)db->d_scores, (void*)db->d_cand_score, (void*)db->d_cand_slot,
                    (void*)db->d_best, (void*)db->d_n, (void*)db->d_best_bits, (void*)db->d_norm, db->d_hi})
        if (p) (void)dev_free(p);
    delete db;
} catch (...) { (void)::hfnet::api_exception(); }

int hfnet_db_add(hfnet_db* db, int slot, const float* descriptor) try {
    API_GUARD(db, "db"); API_GUARD(descriptor, "descriptor");
    if (slot < 0 || slot >= db->capacity) { set_error("db: slot %d outside [0, %d)", slot, db->capacity); return HFNET_ERR_CAPACITY; }
    std::lock_guard<std::mutex> lk(db->mu);
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    // on the stream the scans run on (created non-blocking: the null stream would not order with it)
    HF_TRY(e.h2d(db->d_db + (size_t)slot * db->dim, descriptor, sizeof(float) * db->dim));
    HF_HIP(hipMemsetAsync(db->d_occ + slot, 1, 1, e.stream));
    if (db->dirty_lo >= db->dirty_hi) { db->dirty_lo = slot; db->dirty_hi = slot + 1; }
    else { db->dirty_lo = std::min(db->dirty_lo, slot); db->dirty_hi = std::max(db->dirty_hi, slot + 1); }
    HF_TRY(e.sync_host());                          // the host buffer may go away
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_db_erase(hfnet_db* db, int slot) try {
    API_GUARD(db, "db");
    if (slot < 0 || slot >= db->capacity) { set_error("db: slot %d outside [0, %d)", slot, db->capacity); return HFNET_ERR_CAPACITY; }
    std::lock_guard<std::mutex> lk(db->mu);
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    HF_HIP(hipMemsetAsync(db->d_occ + slot, 0, 1, e.stream));
    HF_TRY(e.sync_host());
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_db_clear(hfnet_db* db) try {
    API_GUARD(db, "db");
    std::lock_guard<std::mutex> lk(db->mu);
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    HF_HIP(hipMemsetAsync(db->d_occ, 0, (size_t)db->capacity, e.stream));
    HF_TRY(e.sync_host());
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_db_query(hfnet_db* db, const float* query, int mode, int32_t* cand_slot, float* cand_score, int* n_cand, float* best_score,
                   float* scores_all) try {
    API_GUARD(db, "db"); API_GUARD(query, "query"); API_GUARD(cand_slot, "cand_slot"); API_GUARD(cand_score, "cand_score"); API_GUARD(n_cand, "n_cand");
    if (mode != 0 && mode != 1) { set_error("db: mode must be 0 or 1"); return HFNET_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(db->mu);   // KeyFrameDatabase.cc:82 holds mMutex over the scan
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    e.bounce_discard();
    HF_HIP(hipSetDevice(e.device));
    HF_TRY(e.h2d(db->d_q, query, sizeof(float) * db->dim));
    // a large database: the screened form reads a quarter of the bytes (the 8-bit copy + 32 KB per keyframe that can be closer than 1) for the same
    // bits; below "db_screen_min_rows" its two extra launches (the query's fragments) cost what it saves
    const bool screened = e.opt.db_screen_min_rows > 0 && db->capacity >= e.opt.db_screen_min_rows && db_screen_supported(db->dim);
    if (screened) HF_TRY(db_screened_scan(e, db, db->d_q, 1, db->d_scores, db->d_best_bits));
    else HF_LAUNCH(&e, e.stream, "db_scores", launch_db_scores(db->d_q, db->d_db, db->d_occ, db->capacity, db->dim, db->d_scores, db->d_best_bits, e.stream));
    HF_LAUNCH(&e, e.stream, "db_filter", launch_db_filter(db->d_scores, db->capacity, mode, db->d_best_bits, screened ? db_gemm_partials(db->capacity) : 4 * db_scan_workgroups(db->capacity), db->d_cand_slot, db->d_cand_score, db->d_n, db->d_best, 1, e.stream));
    // count, best score and the first candidates in ONE round trip (a place-recognition query returns a handful): every synchronisation of the
    // host with the stream is ~10 us of a ~50 us call
    constexpr int kFirst = 64;
    const int first = std::min(db->capacity, kFirst);
    int n = 0;
    float best = 0.f;
    int32_t slot0[kFirst]; float score0[kFirst];
    HF_TRY(e.d2h(&n, db->d_n, sizeof(int)));
    HF_TRY(e.d2h(&best, db->d_best, sizeof(float)));
    HF_TRY(e.d2h(slot0, db->d_cand_slot, sizeof(int32_t) * first));
    HF_TRY(e.d2h(score0, db->d_cand_score, sizeof(float) * first));
    if (scores_all) HF_TRY(e.d2h(scores_all, db->d_scores, sizeof(float) * db->capacity));
    HF_TRY(e.sync_host());
    if (n < 0 || n > db->capacity) { set_error("db: candidate count %d outside [0, %d]", n, db->capacity); return HFNET_ERR_DEVICE; }
    std::memcpy(cand_slot, slot0, sizeof(int32_t) * std::min(n, first));          // (beyond n: whatever an earlier query left there -- not handed out)
    std::memcpy(cand_score, score0, sizeof(float) * std::min(n, first));
    if (n > first) {
        HF_TRY(e.d2h(cand_slot + first, db->d_cand_slot + first, sizeof(int32_t) * (n - first)));
        HF_TRY(e.d2h(cand_score + first, db->d_cand_score + first, sizeof(float) * (n - first)));
        HF_TRY(e.sync_host());
    }
    *n_cand = n;
    if (best_score) *best_score = best;
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

int hfnet_db_query_batch(hfnet_db* db, int n_queries, const float* queries, int mode, int32_t* cand_slot, float* cand_score, int32_t* n_cand,
                         float* best_score, float* scores_all) try {
    API_GUARD(db, "db");
    if (n_queries < 0) { set_error("db: n_queries < 0"); return HFNET_ERR_INVALID_ARG; }
    if (n_queries == 0) return HFNET_OK;
    API_GUARD(queries, "queries"); API_GUARD(cand_slot, "cand_slot"); API_GUARD(cand_score, "cand_score"); API_GUARD(n_cand, "n_cand");
    if (mode != 0 && mode != 1) { set_error("db: mode must be 0 or 1"); return HFNET_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(db->mu);
    Engine& e = db->eng->impl;
    std::lock_guard<std::mutex> lk2(e.mu);
    e.bounce_discard();
    // the screened form: from "db_gemm_min_queries" queries on, and -- any number of queries -- against a database large enough that a quarter of the
    // bytes pays for the queries' fragments ("db_screen_min_rows", as hfnet_db_query)
    const bool gemm = db_screen_supported(db->dim) &&
                      (n_queries >= e.opt.db_gemm_min_queries || (e.opt.db_screen_min_rows > 0 && db->capacity >= e.opt.db_screen_min_rows));
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
    float* d_q = e.m_a.as<float>(); float* d_scores = e.m_s.as<float>(); float* d_cs = e.m_f0.as<float>();
    int32_t* d_slot = e.m_i0.as<int32_t>(); int* d_n = e.m_cnt.as<int>(); float* d_best = e.m_qn.as<float>();
    unsigned int* d_bits = e.m_key.as<unsigned int>();
    HF_TRY(e.h2d(d_q, queries, sizeof(float) * Q * db->dim));
    if (gemm) {
        HF_TRY(db_screened_scan(e, db, d_q, n_queries, d_scores, d_bits));
    } else {
        HF_LAUNCH(&e, e.stream, "db_scores_batch", launch_db_scores_batch(d_q, n_queries, db->d_db, db->d_occ, db->capacity, db->dim, d_scores, d_bits, e.stream));
    }
    HF_LAUNCH(&e, e.stream, "db_filter", launch_db_filter(d_scores, db->capacity, mode, d_bits, parts, d_slot, d_cs, d_n, d_best, n_queries, e.stream));
    // results: counts, best scores and the first candidates of EVERY query in one round trip -- two strided copies (the first kFirst entries of each
    // query's row) instead of two small copies per query behind a first synchronisation (128 copies of a few bytes for 64 queries: a multiple of
    // the scan's own time); a query with more candidates fetches its rest in a second one
    constexpr size_t kFirst = 32;
    const size_t first = std::min(cap, kFirst);
    HF_TRY(e.d2h(n_cand, d_n, sizeof(int32_t) * Q));
    if (best_score) HF_TRY(e.d2h(best_score, d_best, sizeof(float) * Q));
    if (scores_all) HF_TRY(e.d2h(scores_all, d_scores, sizeof(float) * Q * cap));
    // (ONE piece of the pinned block, taken last: a later bounce_take may drain and reuse -- or replace -- the block)
    unsigned char* b_slot = nullptr;
    HF_TRY(e.bounce_take((sizeof(int32_t) + sizeof(float)) * Q * first, &b_slot));
    unsigned char* b_score = b_slot + sizeof(int32_t) * Q * first;
    HF_HIP(hipMemcpy2DAsync(b_slot, sizeof(int32_t) * first, d_slot, sizeof(int32_t) * cap, sizeof(int32_t) * first, Q, hipMemcpyDeviceToHost, e.stream));
    HF_HIP(hipMemcpy2DAsync(b_score, sizeof(float) * first, d_cs, sizeof(float) * cap, sizeof(float) * first, Q, hipMemcpyDeviceToHost, e.stream));
    HF_TRY(e.sync_host());
    bool more = false;
    for (size_t qi = 0; qi < Q; ++qi) {
        const int n = n_cand[qi];
        if (n < 0 || (size_t)n > cap) { set_error("db: candidate count %d of query %zu outside [0, %zu]", n, qi, cap); return HFNET_ERR_DEVICE; }
        const size_t n0 = std::min((size_t)n, first);
        std::memcpy(cand_slot + qi * cap, b_slot + sizeof(int32_t) * qi * first, sizeof(int32_t) * n0);
        std::memcpy(cand_score + qi * cap, b_score + sizeof(float) * qi * first, sizeof(float) * n0);
        more = more || (size_t)n > first;
    }
    if (more) {
        for (size_t qi = 0; qi < Q; ++qi) {
            const size_t n = (size_t)n_cand[qi];
            if (n <= first) continue;
            HF_TRY(e.d2h(cand_slot + qi * cap + first, d_slot + qi * cap + first, sizeof(int32_t) * (n - first)));
            HF_TRY(e.d2h(cand_score + qi * cap + first, d_cs + qi * cap + first, sizeof(float) * (n - first)));
        }
        HF_TRY(e.sync_host());
    }
    return HFNET_OK;
} catch (...) { return ::hfnet::api_exception(); }

}  // extern "C"
