// engine.hpp -- host-side objects behind the C ABI (include/hfnet_hip.h).
#pragma once
#include "kernels.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <thread>

namespace hfnet {

// A few helper threads for the pageable <-> pinned copies of the host-pointer batch pipeline (one memcpy thread moves
// ~7 GB/s; a 64-frame call stages 23 MB in and 67 MB out, as long as the GPU needs to compute it).  run(n, fn) calls
// fn(0..n-1) on the helpers and the caller and returns when all are done.
class CopyPool {
public:
    explicit CopyPool(int helpers) {
        for (int i = 0; i < helpers; ++i) th_.emplace_back([this] { worker(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        // A helper registers in active_ (under m_) before it enters drain() and leaves it the same way.  The run state (fn_, n_,
        // pending_, next_) is only ever written while no helper is inside drain(), and run() only returns when none is:
        // a helper that claimed an index past the end of one run can therefore never compare it with the next run's larger
        // n_ (it would call that run's function with a stale index -- the index processed twice, pending_ decremented once
        // too often, run() back in the caller while a staging copy is still in flight), and fn is never used after scope.
        {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [this] { return active_ == 0; });
            fn_ = &fn; n_ = n; pending_.store(n); next_.store(0); ++gen_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_.load() == 0 && active_ == 0; });
        fn_ = nullptr;
    }
private:
    void drain() {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) return;
            (*fn_)(i);
            if (pending_.fetch_sub(1) == 1) { std::lock_guard<std::mutex> lk(m_); done_.notify_all(); }
        }
    }
    void worker() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                ++active_;
            }
            drain();
            { std::lock_guard<std::mutex> lk(m_); if (--active_ == 0) done_.notify_all(); }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0;
    std::atomic<int> next_{0}, pending_{0};
    unsigned long long gen_ = 0;
    int active_ = 0;               // helpers inside drain() (under m_)
    bool stop_ = false;
};

struct DevMem {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n);   // grow-only
    void release();
    template <class T> T* as() const { return (T*)p; }
};

// diagnostics / A-B switches (hfnet_engine_set_option): read when a model / extractor is created from the engine
struct Options {
    int fuse_blocks = 1;      // 0: expand / depthwise / project as three launches per block (the reference variant of the tests)
    int fuse_max_layer = 14;  // last layer that may use a fused block kernel
    int fused_variant = 4;    // 4: wave-autonomous tiles where they exist (5: the 4 x 8 form only, at any size; 3: its three-waves-per-SIMD forms at any size; 6 / 7: the 6 x 8 form at any size / everywhere); 2: the barrier-phased kernel everywhere
    int fuse_stem = 1;        // stem + layer_2 in one launch
    int dense_desc = 0;       // 1: always evaluate the dense descriptor head (default: only the taps of the selected keypoints)
    int two_streams = 3;      // 0 one stream; 1 fork after layer 7; 2 fork after the detector conv; 3 = 2 + deferred join
    int graph = 1;            // host-pointer extractor calls replay a captured graph
    int pinned_frames = 4;    // chunks up to this many frames move through one pinned block inside that graph
    int conv_wlds = 1;        // 3x3 heads: weights staged through LDS once per workgroup (0: every wave reads them from L1 / L2)
    int db_gemm_min_queries = 8;   // hfnet_db_query_batch: from this many queries on, the scores come from the screened form (integer matrix pipe + exact chain)
    int db_screen_min_rows = 6144;    // hfnet_db_query (ONE query): databases of this capacity or more take the screened form too (a quarter of the bytes; 0: never)
    int pyramid_fuse = 4;     // calls of up to this many frames: the pyramid chain as one launch (0: never)
    int fc_tile = 1;          // FC 7680 -> 4096 of calls above 16 frames: 1 blocked kernel (weights shared through LDS, range partials in registers) when its
                              // workgroups fill the chip, 2 / 4: that kernel with 32 / 64 columns per workgroup at any size (tests), 0: one column tile per wave
    int resize_band = 1;      // level-to-level resize of larger calls: source band of a workgroup staged through LDS (0: thread-per-column gathers from L2 / HBM)
    int det_fuse = 1;         // detector tail (1x1 conv, softmax, depth_to_space) as one launch
    int host_global = 1;      // host-pointer calls of up to four frames: the global descriptors are written into the pinned block by the branch's last kernel
    int interleave = 3;       // calls of up to four frames: launch groups of the global branch enqueued between the local heads' launches,
                              // this many right after the detector conv (0: the whole branch after the local heads)
    int dedupe_taps = 1;      // sparse descriptor head: taps shared by neighbouring keypoints are evaluated once
    int tri_screen_bf16 = 1;    // SearchForTriangulation (calls of >= 4 pairs): threshold screen on the bf16 matrix pipe + exact chains for the listed products; 0: full f32 GEMM
    int match_screen_bf16 = 1;  // SearchByBoW pre-selection on the bf16 matrix pipe (split operands, wider band); 0: f32 MFMA.  The matches are the exact ones either way
    int tail_fuse = 4;        // calls of up to this many frames run layers 8-18 with the single-frame kernels (0: never)
    int global_bf16x3 = 0;    // 1: the 1x1 convolutions of layers 9-18 (fused blocks 9-14 of calls of more than four frames; the three-launch blocks
                              //    15-18) on split-bf16 operands: the global descriptor within the stated tolerance of the exact path (include/hfnet_hip.h)
    int scores_bf16x3 = 0;    // 1: the 1x1 convolutions of layers 3-7 and the detector head on split-bf16 operands: the SCORE MAP within the stated tolerance of
                              //    the exact path; NMS, threshold scan and top-K run exactly on that map (include/hfnet_hip.h)
    int match_stats = 0;      // 1: the SearchByBoW calls and the screened database queries count their exact evaluations (read-only options stat_bow_exact, stat_db_exact; an atomic per wave / workgroup: tests / diagnosis)
    int join_fused_branch = 0;   // 1: a global branch that contains fused-block kernels is joined before the sampler of a few-frame call (the stop-gap of
                                 // NOTEBOOK.md R4.8 before its cause -- packed f32 instructions, now compiled out -- was found; kept as a diagnostic)
    int desc_bf16x3 = 0;      // 1: the sparse descriptor head (3x3 + 1x1 at the distinct tap cells) on split-bf16 operands: descriptors within the stated
                              //    tolerance of the exact path instead of its bits; keypoints, scores and every index stay exact (include/hfnet_hip.h)
    int copy_threads = 64;    // helper threads of the host-pointer batch pipeline's staging copies (>= 64: chosen from the core count)
    int fuse_min_wgs = 256;   // layers 8-14 take their fused kernel from this many 128-pixel tiles per launch on (0: always; tests)
    int* find(const char* name);
};

// Every byte a host-pointer entry point of the matcher / store / database moves across the PCIe boundary goes through this pinned block: the
// caller's (pageable) arrays are only ever touched by host memcpy.  Handing pageable caller memory to hipMemcpy*Async makes the runtime pin
// it in place for the GPU; with the BaseModel path that ended in "Memory access fault by GPU ... on address <a page of the host heap>" once
// in ~15 runs of the GPU suite (NOTEBOOK.md R5.4).  Uploads are copied into the block at once; downloads land in the block and are handed
// to the caller's buffers by Engine::sync_host() (which every such entry point ends with).  Grow-only; all under Engine::mu.
struct HostBounce {
    unsigned char* base = nullptr;
    size_t cap = 0, used = 0;
    struct Pending { void* dst; const unsigned char* src; size_t bytes; };
    std::vector<Pending> pending;
    static constexpr size_t kMaxBytes = (size_t)64 << 20;    // the block never grows beyond this: larger transfers go through it in pieces
    static constexpr size_t kPiece = (size_t)16 << 20;
};

struct Engine {
    int device = 0;
    HostBounce bounce;
    int bounce_take(size_t bytes, unsigned char** out);     // room in the block (drains the stream and grows the block, up to HostBounce::kMaxBytes, when it is full)
    // at the top of every host-pointer entry point: an earlier call that returned an error half-way must not deliver into buffers that may be gone --
    // its downloads are dropped, and (its copies may still be in flight through the block) the stream is drained before the block is reused
    void bounce_discard() {
        if (!bounce.pending.empty()) { (void)hipStreamSynchronize(stream); bounce.pending.clear(); bounce.used = 0; }
    }
    int h2d(void* dst_dev, const void* src_host, size_t bytes);    // on `stream`
    int d2h(void* dst_host, const void* src_dev, size_t bytes);    // on `stream`; dst_host is written by sync_host()
    int sync_host();                                        // hipStreamSynchronize(stream) + the pending downloads' memcpys
    Options opt;
    hipStream_t stream = nullptr;   // matcher / database work
    DeviceWeights w;
    Profiler prof;
    std::mutex mu;                  // engine-level scratch + stream
    std::mutex prof_mu;
    DevMem m_a, m_b, m_s, m_qn, m_tn, m_key, m_i0, m_i1, m_f0, m_f1, m_cnt, m_pairs;   // matcher scratch
    // device-side ordering between extractor streams and the matcher stream for on_device callers
    std::mutex ev_mu;
    hipEvent_t ev_extract = nullptr, ev_match = nullptr;   // last on_device extraction / last hfnet_engine_fence
    bool ev_extract_set = false, ev_match_set = false;
    unsigned char* h_res = nullptr;                         // 1 MB pinned block for the small results of the store matchers (under mu)
    // screened SearchForTriangulation (tri_screen_bf16) pays when few products exceed the threshold; on sets where most do, every pair
    // overflows its list and runs the full path as well.  The screened path counts {overflowed pairs, pairs} on the device, the counts
    // come down behind the call (no synchronisation: whatever has arrived by the next call is used), and after a call in which a
    // quarter of the pairs overflowed the next tri_skip calls go straight to the full path.  The matches are the same either way.
    DevMem m_bow_stat;                                      // two ints: exact evaluations of the SearchByBoW calls / of the screened database queries since they were last read (bow_stat(), + 1)
    int* bow_stat();                                        // its device address (allocated and zeroed on first use; null if that fails)
    DevMem m_tri_stat;
    int* h_tri_stat = nullptr;                              // pinned {overflowed, pairs}
    int tri_skip = 0;
    hipEvent_t ev_tri_stat = nullptr;                       // recorded behind the statistics copy of the last screened call
    bool tri_stat_pending = false;                          // a copy into h_tri_stat has been enqueued and not been consumed yet
    bool pinned_results(size_t bytes);                      // the block exists and holds `bytes`
    hipError_t note_extract(hipStream_t net_stream);        // record: extraction enqueued up to here
    hipError_t wait_extract();                              // matcher stream waits for it
    hipError_t wait_fence(hipStream_t net_stream);          // extractor stream waits for the last fence
    ~Engine();
};

// per-level spatial plan of the network for one input size
struct LevelPlan {
    int H = 0, W = 0;        // raw image
    int Hc = 0, Wc = 0;      // cropped to multiples of 8 (hf_net.py:188-190)
    int h[19] = {0}, w[19] = {0};   // output rows / cols of layer_1 .. layer_18 (index = layer)
    int pt[19] = {0}, pl[19] = {0}; // 'SAME' padding before, of the 3x3 conv in that layer
};

struct NetConfig {
    int n_levels = 1;
    int width[HFNET_MAX_LEVELS] = {0}, height[HFNET_MAX_LEVELS] = {0};
    int batch = 1;
    bool local = true;          // run the local heads + keypoint selection
    bool global = false;        // run layers 8..18 + NetVLAD on level 0
    bool from_intermediate = false;   // input is the layer_7 map (mode IntermediateToGlobal)
    int max_keypoints = 1000;   // per image
};

// The network over a ragged batch: [level][frame] images, all levels in every launch.
struct Net {
    Engine* e = nullptr;
    NetConfig cfg;
    hipStream_t stream = nullptr;
    // the global branch (layers 8-18, NetVLAD, FC) only depends on layer 7: it runs on a second stream next to the
    // local heads (small launches that do not fill the chip next to MFMA-bound ones)
    hipStream_t stream_global = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int two_streams = 1;
    LevelPlan lp[HFNET_MAX_LEVELS];
    long long pix[19][HFNET_MAX_LEVELS + 1];   // pixel offset of level l (frame 0) in layer L's tensor; [n_levels] = total
    long long pix_img[HFNET_MAX_LEVELS + 1];   // same for the cropped full-resolution maps
    long long pix_cell[HFNET_MAX_LEVELS + 1];  // same for the H/8 x W/8 maps
    std::vector<void*> allocs;
    float* act[19] = {nullptr};                // layer outputs (device layout)
    float *exp_buf = nullptr, *dw_buf = nullptr;
    float *desc_hidden = nullptr, *desc_raw = nullptr, *desc_norm = nullptr, *det_hidden = nullptr, *logits = nullptr;
    // sparse descriptor head: rows (image*max_keypoints + i)*4 + tap, see launch_conv3x3_taps
    float *rows_hidden = nullptr, *rows_raw = nullptr;
    bool last_sparse = false;      // which descriptor path the last forward() took
    bool last_dedupe = false;      // ... with de-duplicated tap rows (tap_cell_row is then the sampler's row lookup)
    int dedupe_taps = 1;
    unsigned char* tap_flags = nullptr;        // [image][cell_stride] scratch of launch_tap_cells
    int *tap_cell_row = nullptr, *tap_cells = nullptr, *tap_nrows = nullptr;
    long long cell_stride = 0;
    unsigned int* dev_fault = nullptr;         // HFNET_FAULT_* bits set by kernels that had to bound an index read from device memory (0 in a healthy run)
    unsigned int sticky_faults = 0;            // every bit a call has reported so far (the device word itself is cleared once reported: per-call errors)
    int read_faults(unsigned int* out);        // sticky bits | the device word; waits for the stream; hfnet_model_device_faults / hfnet_extractor_device_faults
    int clear_faults(unsigned int seen);       // a call saw `seen` in the device word: remember it, clear the word on the stream (the next call starts clean)
    bool dense_valid = false;      // dense descriptor tensors match the last forward()
    bool nms_valid = false;        // the suppressed score map (tap 25) matches the last forward()
    float last_threshold = 0.f;
    int force_dense = 0;           // diagnostics / A-B: always run the dense descriptor head
    int fuse_blocks = 1;           // fused inverted-residual kernel for layers <= fuse_max_layer
    int fuse_max_layer = 14;
    int fused_variant = 4;
    int fuse_min_wgs = 256;
    int tail_fuse = 4;
    int interleave = 3;
    int det_fuse = 1;
    int desc_bf16x3 = 0, global_bf16x3 = 0, scores_bf16x3 = 0;
    bool logits_valid = false;     // the logits tensor holds the last forward's values (the fused detector tail does not write it)
    int fuse_stem = 1;             // stem + layer_2 in one launch: the stem tensor is not materialised (its tap recomputes it on demand)
    int conv_wlds = 1;             // 3x3 heads with LDS-staged weights
    ImageSet last_imgs;            // input of the last forward (for that tap)
    bool stem_valid = false;
    size_t stem_elems_max = 0;     // size of the stem tensor at the configured (largest) batch
    float *dense = nullptr, *nms = nullptr;
    float* global_dst = nullptr;   // when set: forward_global() writes the global descriptors here instead of global_out
    FcHostOut global_host;         // when set: ... and into the caller's pinned block, followed by the call's number
    unsigned *nms_mask = nullptr, *nms_flags = nullptr;   // bit-column masks of the NMS passes (max_mask, supp)
    unsigned long long* cand = nullptr;
    unsigned int* counters = nullptr;
    long long cand_stride = 0;
    hfnet_keypoint* kps_level = nullptr;       // [image][max_keypoints]
    int* n_level = nullptr;                    // [image]
    float *memb = nullptr, *vlad_raw = nullptr, *vlad_tap = nullptr, *vlad_out = nullptr, *fc_raw = nullptr, *fc_part = nullptr, *global_out = nullptr;
    float* inter_logical = nullptr;            // level-0 layer_7 map in logical order [batch x hd x wd x C]
    int build(Engine* eng, const NetConfig& c);
    void release();
    Geom geom(int layer_in, int layer_out, int first_level, int n_levels_used) const;
    // enqueue the whole forward pass; imgs: per-level u8 sources (ignored when from_intermediate)
    // defer_global: do not wait for the global branch at the end; the caller consumes global_out on stream_global and the
    // next forward() waits for it before layer 7 is overwritten
    // caller_joins: (few frames per call) the global branch forks right after layer 7 and forward() does NOT wait for it:
    // join_pending is set, the caller finishes its work on the local results first and then waits for ev_join itself
    int forward(const ImageSet& imgs, float threshold, const TopkBudget& budget, bool defer_global = false, bool caller_joins = false);
    bool join_pending = false;
    bool branch_fused_used = false;   // a fused-block kernel was enqueued for a layer of the global branch by the forward() in progress (see its end)
    int tap(int id, std::vector<float>& out);
    int run_dense_desc();
    int forward_global(hipStream_t st, int first = 0, int count = 1 << 20, int* total = nullptr);   // launch groups [first, first + count)
    bool tail_chain() const;       // layers 8-18 run as one launch per block (single-frame kernels)
    const float* sample_source() const { return last_sparse ? rows_raw : desc_norm; }   // sparse rows are normalised by k_sample
    ~Net() { release(); }
};

// ---- internals shared by the translation units of the host side (engine.hip, api_extract.hip, api_match.hip, api_db.hip)
// one polite spin iteration of the host waits on the pinned flags (the pause intrinsic is x86-only)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

template <class T>
inline int dalloc(std::vector<void*>& allocs, T** out, size_t count) {
    void* p = nullptr;
    HF_HIP(dev_malloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    allocs.push_back(p);
    *out = (T*)p;
    return HFNET_OK;
}

// HFNET_TRACE_LAUNCHES=1 in the environment (diagnostics, with the guarded allocator of devmem.cpp): every launch group is named on
// stderr before it is enqueued and waited for afterwards (unless the stream is being captured into a graph), so that the last name
// printed before a "Memory access fault" is the kernel that faulted.
bool trace_launches();
void trace_launch(const char* name, hipStream_t s, bool after);

#define HF_LAUNCH(eng, strm, name, call)                                                   \
    do {                                                                                   \
        hipError_t er__;                                                                   \
        if (::hfnet::trace_launches()) ::hfnet::trace_launch(name, strm, false);           \
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
        if (::hfnet::trace_launches()) ::hfnet::trace_launch(name, strm, true);            \
    } while (0)

#define API_GUARD(ptr, what)                                               \
    do {                                                                   \
        if (!(ptr)) { ::hfnet::set_error(what " is null"); return HFNET_ERR_INVALID_ARG; } \
    } while (0)

// HFextractor's constructor tables (HFextractor.cc:82-139) and cv::resize's fixed-point tables (engine.hip)
void extractor_tables(int nfeatures, int nlevels, float scale_factor, int width, int height, float* sf, int* fpl, int* lw, int* lh);
void resize_tables(int sw, int sh, int dw, int dh, std::vector<int>& xofs, std::vector<short>& ialpha, std::vector<int>& yofs, std::vector<short>& ibeta);
// tensor offsets of a Net for `batch` frames per level (engine.hip)
void compute_offsets(Net& n, int batch);
// host rows -> device scratch (or the caller's device pointer as it is)
int stage_rows(Engine& e, DevMem& m, const float* src, size_t count, int on_device, const float** out);

}  // namespace hfnet

// ---- C ABI objects ------------------------------------------------------------------------------
struct hfnet_engine { hfnet::Engine impl; };

struct hfnet_model {
    hfnet_engine* eng = nullptr;
    hfnet_mode mode = HFNET_IMAGE_TO_LOCAL;
    int height = 0, width = 0, max_keypoints = 0;
    bool valid = false;
    hfnet::Net net;
    uint8_t* d_image = nullptr;          // [H x W]
    hfnet_keypoint* d_kps = nullptr;     // [max_keypoints]
    float* d_desc = nullptr;             // [max_keypoints x 256]
    int* d_n = nullptr;
    // Pinned block every byte of a call crosses the PCIe boundary through: [image | n, fault word | aux | keypoints | descriptors].
    // The caller's (pageable) buffers are only touched by host memcpy.  Until round 5 they were handed to hipMemcpy2DAsync / hipMemcpyAsync
    // directly; the runtime pins such memory in place for the GPU, and once in ~15 runs of the GPU suite a call died with "Memory access
    // fault by GPU ... on address <a page of the host heap>" (GPUTEST_r04, NOTEBOOK.md R5.4).
    unsigned char* h_stage = nullptr;
    size_t o_n = 0, o_aux = 0, o_kps = 0, o_desc = 0, stage_bytes = 0;
    std::mutex mu;
};

struct hfnet_extractor {
    hfnet_engine* eng = nullptr;
    int width = 0, height = 0, n_features = 0, n_levels = 0, max_batch = 0;
    float threshold = 0.f, scale_factor = 1.f;
    float scale_factors[HFNET_MAX_LEVELS];
    int features_per_level[HFNET_MAX_LEVELS], level_w[HFNET_MAX_LEVELS], level_h[HFNET_MAX_LEVELS];
    hfnet::Net net;
    std::vector<void*> allocs;
    uint8_t* d_pyr[HFNET_MAX_LEVELS] = {nullptr};   // level l >= 1: [max_batch][h][w]; level 0: staging for host input
    int* d_xofs[HFNET_MAX_LEVELS] = {nullptr};
    short* d_ialpha[HFNET_MAX_LEVELS] = {nullptr};
    int* d_yofs[HFNET_MAX_LEVELS] = {nullptr};
    int pyr_band_rows[HFNET_MAX_LEVELS] = {0};      // level l >= 1: resize_band_rows of its table (0: engine option resize_band off)
    short* d_ibeta[HFNET_MAX_LEVELS] = {nullptr};
    hfnet_keypoint* d_kps = nullptr;     // [max_batch][n_features]
    float* d_desc = nullptr;             // [max_batch][n_features][256]
    int* d_n = nullptr;                  // [max_batch]
    int* d_n_level = nullptr;            // [max_batch][n_levels]
    // host-pointer calls replay a captured graph of the ~75 launches of a chunk (same staging buffers every call): a
    // frame at a time the launches themselves are a good part of the latency.  One executable graph per chunk size.
    std::map<int, hipGraphExec_t> graphs;
    int use_graph = 1;
    // Small chunks (<= pinned_frames) also move their input and results through one pinned block inside the same graph:
    // one launch and one host synchronisation per call instead of three blocking pageable copies.
    // [images | n | n_level | keypoints | descriptors | global | flag] for pinned_frames frames
    unsigned char* h_pin = nullptr;
    std::vector<int> last_n;             // keypoint counts of the last host-pointer call per staging frame (-1: unknown)
    int pinned_frames = 0;
    // Result sections [n | n_level | keypoints | descriptors | global] packed for the frames of the call (offsets from pin_res;
    // result_offsets()); the device side keeps the SAME layout in one block (d_blk), so a call's results come down with ONE copy
    size_t pin_res = 0, pin_nl_last = 0, pin_flag = 0;
    // single-frame calls: the last kernel of the global branch writes the descriptors into the pinned block itself
    bool global_to_host(int nb) const { return net.cfg.global && h_pin && host_global && hfnet::fc_host_out_supported(nb); }
    int host_global = 1;           // (engine option of the same name, read at creation)
    double t_last[6] = {-1, -1, -1, -1, -1, -1};   // hfnet_extractor_last_timing
    // "the local results are down": a device counter the graph bumps and copies into the pinned block right after them; the
    // host spins on it, unpacks keypoints and descriptors while the global branch is still running, then waits for the rest
    int* d_seq = nullptr;          // [0] calls whose local results are down, [1..2] the same for the global descriptors (FcHostOut::seq)
    int seq_host = 0, gseq_host = 0;
    unsigned char* d_blk = nullptr;
    struct ResOff { size_t n, nl, g, k, d, total; };
    ResOff result_offsets(int nb, int global_dim) const {
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        // the local results first and contiguous ([0, g): they come down as soon as the local heads are done), the global
        // descriptors last (they follow when the global branch has joined)
        ResOff o;
        o.n = 0; o.nl = up(sizeof(int) * nb); o.k = o.nl + up(sizeof(int) * (size_t)nb * n_levels);
        o.d = o.k + up(sizeof(hfnet_keypoint) * (size_t)nb * n_features);
        o.g = o.d + up(sizeof(float) * HFNET_DESC_DIM * (size_t)nb * n_features);
        o.total = o.g + up(sizeof(float) * (size_t)nb * global_dim);
        return o;
    }
    // Larger host-pointer calls run as a double-buffered pipeline over their chunks: while chunk c computes, chunk c + 1's
    // images go up and chunk c - 1's results come down through pinned blocks on two copy streams (built on first use).
    // Slot 0 of the device side is the staging above (d_pyr[0], d_kps, d_desc, d_n, d_n_level).
    struct HostPipe {
        bool ready = false;
        unsigned char* h_in[2] = {nullptr, nullptr};
        unsigned char* h_out[2] = {nullptr, nullptr};         // [n | n_level | global | keypoints | descriptors] at full capacity
        size_t o_n = 0, o_nl = 0, o_g = 0, o_k = 0, o_d = 0, out_bytes = 0;
        uint8_t* d_in[2] = {nullptr, nullptr};
        hfnet_keypoint* d_kps[2] = {nullptr, nullptr};
        float* d_desc[2] = {nullptr, nullptr};
        float* d_glob[2] = {nullptr, nullptr};
        int* d_n[2] = {nullptr, nullptr};
        int* d_nl[2] = {nullptr, nullptr};
        hipStream_t s_up = nullptr, s_down = nullptr;
        hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
        std::unique_ptr<hfnet::CopyPool> pool;                       // helpers for the staging copies
    } pipe;
    const float* last_desc = nullptr;    // device descriptors / counts of the last host-pointer chunk (hfnet_store_put_extracted)
    const int* last_cnt = nullptr;
    hfnet_store* att_store = nullptr;    // hfnet_extractor_attach_store: device copies of every host-pointer frame
    int att_first = 0;
    std::mutex mu;
};

// device-resident descriptor sets of keyframes: uploaded once, matched many times
struct hfnet_store {
    hfnet_engine* eng = nullptr;
    int n_sets = 0, max_rows = 0, dim = 0;
    float* d_desc = nullptr;       // [n_sets][max_rows][dim]
    int32_t* d_rows = nullptr;     // [n_sets]
    unsigned char* d_flags = nullptr;   // [n_sets][max_rows] "has a MapPoint" flag per row
    std::vector<int32_t> rows;     // host mirror of d_rows
    std::mutex mu;
};

struct hfnet_db {
    hfnet_engine* eng = nullptr;
    int capacity = 0, dim = 0;
    float* d_db = nullptr;
    unsigned char* d_occ = nullptr;
    float *d_q = nullptr, *d_scores = nullptr, *d_cand_score = nullptr, *d_best = nullptr;
    float* d_norm = nullptr;       // per slot: |d|^2 (tree256 order), scale and scaled 1-norm of its 8-bit steps (db_stat_floats), and
    void* d_hi = nullptr;          // the 8-bit copy of every row (fragment order), for the screened batched query: both refreshed in one launch
    int dirty_lo = 0, dirty_hi = 0;   // by the first batched query after rows were added: the slots [dirty_lo, dirty_hi) (whole 32-row tiles of them)
    int32_t* d_cand_slot = nullptr;
    int* d_n = nullptr;
    unsigned int* d_best_bits = nullptr;
    std::mutex mu;
};
