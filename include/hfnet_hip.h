/*
 * hfnet_hip.h -- C ABI of the MI355X-native HF-Net feature front end (libhfnet_hip.so).
 *
 * Drop-in boundary for the feature front end of LiuLimingCode/HFNet_SLAM: every entry point
 * below replaces one reference interface (cited as path:line under /root/reference).  Plain
 * pointers and sizes only; caller-owned buffers; no memory owned by the library crosses the
 * boundary; nothing here throws or exits (the reference's factory exit(-1)s on a bad model,
 * src/Extractors/BaseModel.cc:119-125 -- the caller of this ABI decides instead).
 * Host buffers handed to an entry point are only read / written by host code of the library (the bytes move through pinned blocks the
 * library owns); the GPU never accesses caller memory, except arrays the caller registered with hfnet_host_register.  A C++ exception
 * inside the library (out of host memory, ...) is caught at the boundary and returned as HFNET_ERR_INTERNAL.
 *
 * All entry points return HFNET_OK (0) or an error code; hfnet_last_error() gives the text for
 * the calling thread.  Objects are internally serialised per object (the reference enters its
 * shared global model from two SLAM threads without a lock, Tracking.cc:2026 /
 * LocalMapping.cc:367); different objects may be used concurrently from different threads.
 *
 * Numeric contract: fp32 end to end (f32 MFMA, fused multiply-add chains in the order fixed by
 * the CPU oracle, oracle/hfnet_oracle.h) -- keypoint coordinates / match indices / candidate
 * indices are bit-exact against the oracle, float outputs are bit-exact (the tolerance mode -- engine options "scores_bf16x3",
 * "desc_bf16x3", "global_bf16x3", all off by default -- states its own contract below).
 */
#ifndef HFNET_HIP_H
#define HFNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HFNET_ABI_VERSION 2          /* 2 (round 6): HFNET_ERR_INTERNAL, hfnet_*_device_faults, hfnet_extractor_tap, option "scores_bf16x3",
                                       "pyramid_fuse" is a frame limit (was on / off), options "match_stats" / "stat_bow_exact" / "stat_db_exact" /
                                       "db_screen_min_rows", the screened database queries take dim <= 4096 (was: any multiple of 512); 1: rounds 1-5 */
#define HFNET_DESC_DIM 256          /* local descriptor length  (HFNetTFModelV2.cc:153)            */
#define HFNET_MAX_LEVELS 8
#define HFNET_MAX_KEYPOINTS 8192    /* per image and call; mono init asks for 5*nFeatures (Tracking.cc:693) */

typedef enum {
    HFNET_OK = 0,
    HFNET_ERR_INVALID_ARG = 1,
    HFNET_ERR_WRONG_MODE = 2,       /* == the reference's `return false` on a mode mismatch (HFNetTFModelV2.cc:65,81,92) */
    HFNET_ERR_SHAPE = 3,            /* input size differs from the size given at construction (HFNetTFModelV2.cc:103)  */
    HFNET_ERR_DEVICE = 4,           /* HIP runtime error / no gfx950 device                                           */
    HFNET_ERR_IO = 5,               /* weight container missing or malformed                                           */
    HFNET_ERR_CAPACITY = 6,
    HFNET_ERR_INTERNAL = 7          /* out of host memory / a C++ exception inside the library, caught at this boundary        */
} hfnet_status;

/* include/Extractors/BaseModel.h:16-21 */
typedef enum {
    HFNET_IMAGE_TO_LOCAL_AND_GLOBAL = 0,
    HFNET_IMAGE_TO_LOCAL = 1,
    HFNET_IMAGE_TO_LOCAL_AND_INTERMEDIATE = 2,
    HFNET_INTERMEDIATE_TO_GLOBAL = 3
} hfnet_mode;

/* The cv::KeyPoint fields the path writes (HFNetTFModelV2.cc:122-138, HFextractor.cc:272-279):
 * pt.x, pt.y, response, octave; angle is always 0. */
typedef struct { float x, y, response; int32_t octave; } hfnet_keypoint;

typedef struct hfnet_engine hfnet_engine;        /* one GPU: weights resident in HBM, streams, scratch */
typedef struct hfnet_model hfnet_model;          /* == one BaseModel instance (fixed input shape + mode) */
typedef struct hfnet_extractor hfnet_extractor;  /* == one HFextractor (pyramid + per-level budget)      */
typedef struct hfnet_store hfnet_store;          /* device-resident descriptor sets of keyframes          */
typedef struct hfnet_db hfnet_db;                /* == KeyFrameDatabase's descriptor store + scan        */

const char* hfnet_last_error(void);
int hfnet_abi_version(void);
/* hash of the library's sources, compiled in by hfnet_slam_amd/build.py: says which tree a measured binary was built from */
const char* hfnet_build_id(void);
/* number of visible HIP devices; HFNET_ERR_DEVICE text in hfnet_last_error() when none */
int hfnet_device_count(void);

/* ---- engine: model files + device (replaces LoadHFNetTRModel / LoadSavedModel,
 *      src/Extractors/HFNetRTModel.cc:208-254, HFNetTFModelV2.cc:180-202) ------------------------ */
int hfnet_engine_create(int device, const char* weights_path, hfnet_engine** out);
void hfnet_engine_destroy(hfnet_engine* e);
/* what: 0 stem channels, 1 local (intermediate) channels, 2 global-branch channels,
 *       3 NetVLAD clusters, 4 global descriptor length (4096), 5 device ordinal */
int hfnet_engine_info(const hfnet_engine* e, int what);
/* Diagnostics / A-B switches, read when a model or extractor is created from the engine (no environment variables):
 *   "fuse_blocks" (1)   0: every inverted-residual block as three launches (the tests' reference variant)
 *   "fuse_max_layer" (14), "fused_variant" (4: wave-autonomous tiles -- 6 x 8 on the 16x16x4 MFMA for the stride-1 blocks from
 *                       layer 6 on, 4 x 8 on 32x32x2 otherwise; 5: the 4 x 8 form only, at any launch size; 3: their three-waves-per-SIMD forms at any
 *                       launch size; 6 / 7: the 6 x 8 tiles at any launch size / for every stride-1 block; 2: barrier-phased
 *                       kernel), "fuse_stem" (1)
 *   "fuse_min_wgs" (256) layers 8-14 take their fused kernel from this many 128-pixel tiles per launch on (0: always)
 *   "dense_desc" (0)    1: dense descriptor head instead of the taps of the selected keypoints
 *   "dedupe_taps" (1)   sparse descriptor head: taps shared by neighbouring keypoints are evaluated once (2: the row numbering as two launches,
 *                       mark + compact, instead of inside the top-K launch; same rows)
 *   "two_streams" (3)   0 one stream; 1 fork after layer 7; 2 fork after the detector conv; 3 = 2 + deferred join
 *   "conv_wlds" (1)     3x3 head convolutions: weights staged through LDS once per workgroup (0: every wave reads them)
 *   "graph" (1), "pinned_frames" (4): host-pointer extractor calls
 *   "db_gemm_min_queries" (8): hfnet_db_query_batch screens on the integer matrix pipe from this many queries on (same bits either way)
 *   "db_screen_min_rows" (6144): hfnet_db_query (one query) and hfnet_db_query_batch of fewer than "db_gemm_min_queries" queries screen too when the
 *                       database has this many slots or more (0: never; same bits)
 *   "tail_fuse" (4)     calls of up to this many frames run layers 8-18 with the single-frame kernels (depthwise + projection
 *                       in one launch, short-latency MFMA chains); 0: never
 *   "pyramid_fuse" (4)  calls of up to this many frames: the pyramid resize chain as one launch
 *   "fc_tile" (1)       dimensionality reduction of calls above 16 frames: blocked kernel (a workgroup's weight pieces shared through LDS, the 16
 *                       range partials merged in registers in the balanced tree's order) when its workgroups fill the chip; 2 / 4: that kernel
 *                       with 32 / 64 columns per workgroup at any size; 0: one column tile per wave.  Same bits either way
 *   "resize_band" (1)   larger calls: each level-to-level resize stages a workgroup's source rows through LDS once (0: per-thread
 *                       byte gathers from L2 / HBM); same bytes either way
 *   "interleave" (3)    calls of up to four frames: the launch groups of the global branch are enqueued between the launches
 *                       of the local heads, this many right after the detector conv (0: the whole branch after the local heads)
 *   "det_fuse" (1)      detector tail (1x1 conv 128 -> 65, softmax, depth_to_space) as one launch: the logits stay in LDS
 *   "host_global" (1)   host-pointer calls of up to four frames: the last kernel of the global branch writes the descriptors
 *                       into the pinned result block itself (no copy after the join, the call returns without draining the stream)
 *   "match_screen_bf16" (1)  SearchByBoW pre-selects on the bf16 matrix pipe (operands split into two bf16 pieces, three products,
 *                       a wider rounding band); 0: on the f32 MFMA.  Every candidate inside the band is re-evaluated exactly
 *                       either way: matches and distances are the same bits
 *   "tri_screen_bf16" (1)  SearchForTriangulation, calls of >= 4 pairs: the products that can exceed the similarity threshold are
 *                       found on the bf16 matrix pipe (split operands, rigorous bound) and evaluated as the exact fma chains; a pair
 *                       with too many of them goes through the full f32 GEMM instead, and after a call in which a quarter of the
 *                       pairs did, the next 16 calls skip the screen (writing the option resets that).  Same matches, bit for bit
 *   "copy_threads" (64 = by core count): helper threads for the staging copies of host-pointer batch calls (replicas sharing a host)
 *   "match_stats" (0)   1: the SearchByBoW calls and the screened batched database queries count what their screens let through -- exact
 *                       distance evaluations / exactly scored (query, keyframe) pairs -- into the read-only options "stat_bow_exact" /
 *                       "stat_db_exact" (hfnet_engine_get_option reads AND clears; waits for the stream).  A screen that lets everything
 *                       through still returns the right results; this is where it shows (tests, diagnosis)
 * Values are >= 0.
 * Every setting of the extractor and matcher switches ABOVE produces the same bits (tests/test_gpu_parity.py).
 *
 * Three options trade the oracle's bits of FLOAT outputs for speed, within a stated tolerance; all default to 0.  The first two touch
 * nothing that decides an index -- backbone layers 1-7, detector head, NMS, threshold scan and top-K run the exact f32 chains whatever
 * they say, so keypoint counts, positions, responses and octaves stay bit-identical; the third ("scores_bf16x3") is the full tolerance mode:
 *   "desc_bf16x3" (0)   the sparse descriptor head (3x3 96 -> 256 + ReLU6, 1x1 256 -> 256 at the distinct tap cells) on the bf16 matrix
 *                       pipe: every f32 operand is split into two bf16 pieces x = hi + lo + e, |e| <= 2^-16 |x|, and a product
 *                       a.w is taken as ah.wh + ah.wl + al.wh (v_mfma_f32_32x32x16_bf16; exact products, fp32 accumulation in the
 *                       unit's order).  Per output of a K-term convolution |error| <= (3 * 2^-16 + K * 2^-23) * sum_k |a_k||w_k|
 *                       in the worst case; the errors are rounding residues of random sign, and on the unit-norm 256-D rows the
 *                       extractor returns the deviation from the exact path is <= 2e-6 as measured (752x480 and 512x512, 4 levels;
 *                       tests/test_gpu_fullsize.py).  STATED TOLERANCE: 1e-5 absolute per component of a (unit-norm) descriptor row.
 *   "global_bf16x3" (0) the same for the 1x1 convolutions of the global branch's blocks: layers 9-14 inside their fused kernel (calls that take
 *                       the fused kernels, i.e. more than four frames; the depthwise stage between the two stays exact f32) and layers 15-18
 *                       (three launches per block).  STATED TOLERANCE: 2e-5 absolute per component of the (unit-norm) 4096-D global
 *                       descriptor (measured <= 6e-6 through the ten blocks; a model with a D-dimensional global descriptor: 2e-5 *
 *                       sqrt(4096 / D) -- the components of a unit vector scale that way).  Also layer 8 (fused; calls that take the fused kernels)
 *                       and the NetVLAD memberships conv, and -- calls of >= 64 frames -- the dimensionality reduction (7680 x 4096 on the same
 *                       split products, K in eight parts; its own deviation ~1e-6); the NetVLAD aggregation stays exact f32.
 *   "scores_bf16x3" (0) the rest of the network on the same split-bf16 products: the 1x1 convolutions of layers 3-7 inside their fused kernels (the
 *                       depthwise stage between them stays exact f32), the detector head's 3x3 conv 96 -> 128 (halo staged through LDS already split,
 *                       k_conv3x3_dense_bf16x3) and its 1x1 128 -> 65 (k_det_tail_bf16x3).  Stem + layer 2, every depthwise stage, softmax, NMS,
 *                       threshold scan, top-K, sampler, NetVLAD aggregation and the FC stay f32.  With this option THE SCORE MAP IS A TOLERANCE TENSOR,
 *                       and "bit-exact keypoints" changes meaning the way SURVEY.md section 7 spells it out: NMS / threshold / top-K are exact ON THE SCORE
 *                       MAP THE DEVICE PRODUCED -- the oracle's hfo_simple_nms + hfo_select_keypoints run on the dense scores read back through
 *                       hfnet_extractor_tap(22) give the device's keypoints, array_equal (tests/test_gpu_scores_bf16x3.py; bench.py checks it on the
 *                       last timed chunk) -- while the map itself is within a stated tolerance of the oracle's.  STATED TOLERANCES of this full tolerance
 *                       mode (all three options on; every one of the 18 layers then feeds the deviation, which is why they are wider than those of the two
 *                       index-exact options above): dense scores 5e-4 absolute and 2e-3 relative to the score (measured 1.5e-4 / 5e-4: a softmax output
 *                       moves by s |d logit|, the logits by <= 5e-4 after six layers); unit-norm descriptors of the keypoints both modes select 2e-5
 *                       (measured <= 1.5e-5); global descriptor 1e-4 per component at the reference's image sizes (measured <= 5.9e-5 over 2 200 random
 *                       cases and six weight sets; components are ~1/64, the typical deviation is ten times smaller), 2.5e-4 for images of a few hundred
 *                       cells, where NetVLAD averages far fewer pixels (measured <= 8e-5).  Keypoint-set overlap with the exact
 *                       mode: >= 99 % (measured 99.996 % over those cases, tools/dev/soak_tolerance.py; synthetic weights are the hard case: scores
 *                       near 1/65 everywhere).  With "global_bf16x3" the option set also moves layer 8 (fused, one wave per SIMD), layers 15-18 (fused,
 *                       split-bf16 forms only) and the NetVLAD memberships conv onto split-bf16 operands.
 *   "join_fused_branch" (0) diagnostic: calls of <= 4 frames whose global branch contains fused-block kernels (only with "fuse_min_wgs"
 *                       lowered) join the branch before the sampler instead of after it (NOTEBOOK.md R4.8)
 * The matcher and the database are exact FOR THE DESCRIPTORS THEY ARE GIVEN in either mode. */
int hfnet_engine_set_option(hfnet_engine* e, const char* name, int value);
int hfnet_engine_get_option(hfnet_engine* e, const char* name, int* value);
int hfnet_engine_synchronize(hfnet_engine* e);
/* Stream ordering for on_device callers (no host wait anywhere):
 *  - matcher calls with on_device != 0 run after every on_device extraction enqueued before them
 *    (their inputs are the extractor's outputs);
 *  - hfnet_engine_fence: on_device extraction enqueued AFTER this call runs after all matcher work
 *    enqueued BEFORE it -- call it before re-using descriptor buffers a pending match still reads;
 *  - the library's streams are NOT ordered with the caller's: whatever the caller writes on the device (images, pair
 *    lists, row counts it produces itself) must be complete before the call, and results are final only after
 *    hfnet_engine_synchronize. */
int hfnet_engine_fence(hfnet_engine* e);

/* ---- BaseModel (include/Extractors/BaseModel.h:38-54; ctor HFNetTFModelV2.cc:12-60) ----------- */
/* height/width: the input image for image modes; for HFNET_INTERMEDIATE_TO_GLOBAL the
 * intermediate map's H/8 x W/8 (BaseModel.cc:70).  max_keypoints bounds nKeypointsNum. */
int hfnet_model_create(hfnet_engine* e, hfnet_mode mode, int height, int width, int max_keypoints,
                       hfnet_model** out);
void hfnet_model_destroy(hfnet_model* m);
int hfnet_model_is_valid(const hfnet_model* m);                 /* BaseModel::IsValid            */
int hfnet_model_mode(const hfnet_model* m);
/* BaseModel::Detect(image, kps, local, global, N, thr)  and  Detect(image, kps, local, N, thr)
 * (HFNetTFModelV2.cc:62-87).  image: 8-bit gray, `row_stride` bytes between rows.
 * kps / local_desc: caller buffers for n_keypoints rows (local_desc is n x 256 floats).
 * aux: mode LOCAL_AND_GLOBAL -> global descriptor (global_dim floats);
 *      mode LOCAL_AND_INTERMEDIATE -> intermediate map (H/8 x W/8 x local_channels floats, NHWC);
 *      mode LOCAL -> must be NULL.   Keypoint order is the oracle's canonical order. */
int hfnet_model_detect(hfnet_model* m, const uint8_t* image, int row_stride, int n_keypoints,
                       float threshold, hfnet_keypoint* kps, float* local_desc, float* aux, int* n_out);
/* BaseModel::Detect(intermediate, global) (HFNetTFModelV2.cc:89-98) */
int hfnet_model_detect_global(hfnet_model* m, const float* intermediate, float* global_desc);
/* Diagnostics: copy an intermediate tensor of the last detect call to the host (logical channel
 * order, NHWC).  tap ids as in oracle/hfnet_oracle.h (HFO_TAP_*), plus 25 = scores after NMS,
 * 26 = normalised dense descriptor map.  *count receives the number of floats written. */
int hfnet_model_tap(hfnet_model* m, int tap, float* out, size_t capacity, size_t* count);
/* Diagnostics: a word of HFNET_DEVICE_FAULT_* bits, 0 in a healthy run, sticky for the lifetime of the object.  The kernels
 * bound every index they read from device memory before it becomes an address (a valid call can then never take the host
 * process down through a GPU memory fault -- the reference's contract is `return false`, HFNetTFModelV2.cc:62-98, never an
 * abort); a bit here says that such a bound was hit, i.e. that device state was inconsistent and results may be wrong.
 * hfnet_model_detect returns HFNET_ERR_DEVICE for THE CALL whose launches set a bit (the reference's per-call `return false`) and clears
 * the device word behind it, so the object stays usable; this function keeps reporting every bit ever seen.  Waits for the object's stream. */
#define HFNET_DEVICE_FAULT_TAP_ROWS 1u    /* more marked tap cells than rows in an image's slot of the sparse descriptor head */
#define HFNET_DEVICE_FAULT_SAMPLE_ROW 2u  /* a bilinear tap of a selected keypoint had no descriptor row                      */
int hfnet_model_device_faults(hfnet_model* m, unsigned int* bits);

/* ---- HFextractor (include/Extractors/HFextractor.h:26-27; src/Extractors/HFextractor.cc:82-284;
 *      model set-up mirrors InitAllModels, BaseModel.cc:24-93) ----------------------------------- */
int hfnet_extractor_create(hfnet_engine* e, int width, int height, int n_features, float threshold,
                           float scale_factor, int n_levels, int max_batch, hfnet_extractor** out);
void hfnet_extractor_destroy(hfnet_extractor* x);
/* tables HFextractor computes in its ctor: scale factors, per-level budget, per-level size */
int hfnet_extractor_tables(const hfnet_extractor* x, float* scale_factors, int* features_per_level,
                           int* level_width, int* level_height);
/* HFextractor::operator()(image, keypoints, localDescriptors, globalDescriptors).  Returns the
 * number of keypoints through *n_out (-1 with HFNET_ERR_INVALID_ARG on a bad image, as the
 * reference returns -1, HFextractor.cc:145).  kps / local_desc hold n_features rows. */
int hfnet_extractor_extract(hfnet_extractor* x, const uint8_t* image, int row_stride,
                            hfnet_keypoint* kps, float* local_desc, float* global_desc,
                            int* n_out, int* n_per_level /* n_levels or NULL */);
/* Host-side time stamps of the last host-pointer call that went through the latency path (chunks of up to "pinned_frames"
 * frames), in microseconds since the call was entered: [0] image in the pinned block, [1] everything enqueued, [2] local
 * results seen on the host, [3] local results unpacked, [4] stream drained (global descriptor down), [5] return.
 * Writes min(n, 6) values; returns HFNET_ERR_INVALID_ARG before the first such call. */
int hfnet_extractor_last_timing(hfnet_extractor* x, double* us, int n);
/* see hfnet_model_device_faults */
int hfnet_extractor_device_faults(hfnet_extractor* x, unsigned int* bits);
/* Diagnostics: hfnet_model_tap for the extractor's network -- the tensor of the LAST call (its last chunk of frames), all pyramid
 * levels and frames concatenated in [level][frame][y][x][channel] order (levels have their own sizes).  Tap 22 (dense scores) is
 * what the tolerance-mode tests run the oracle's NMS / top-K on (tests/test_gpu_scores_bf16x3.py). */
int hfnet_extractor_tap(hfnet_extractor* x, int tap, float* out, size_t capacity, size_t* count);
/* Batched form (independent frames, BASELINE config 4): images are n_frames buffers of
 * height x row_stride bytes, `frame_stride` bytes apart; outputs are n_frames slots of n_features
 * rows each.  `on_device` != 0: every pointer is a device pointer on the engine's GPU and the call
 * only enqueues work (use hfnet_engine_synchronize; the matcher entry points order themselves behind it, see
 * hfnet_engine_fence).  `on_device` == 0 with more frames than the extractor's max_batch: the chunks run as a
 * double-buffered pipeline (pinned staging, two copy streams): chunk c + 1 goes up and chunk c - 1 comes down while chunk c
 * computes.  The global descriptors of such a call are produced on a second stream that may still run
 * while the next call's backbone executes: they are complete after hfnet_engine_synchronize. */
int hfnet_extractor_extract_batch(hfnet_extractor* x, int n_frames, const uint8_t* images,
                                  int row_stride, size_t frame_stride, hfnet_keypoint* kps,
                                  float* local_desc, float* global_desc, int* n_out, int on_device);
/* Caller memory registered for DMA (the H2D / D2H boundary of the reference's GPU back end: cudaMemcpy from / into the
 * cv::Mat buffers, HFNetRTModel.cc:128,134).  A host-pointer hfnet_extractor_extract_batch call whose image block
 * (frames contiguous: row_stride == width, frame_stride == width * height) and whose kps / local_desc / n_out
 * (and global_desc, if given) buffers ALL lie inside registered ranges moves its data straight between the caller's memory
 * and the GPU -- no pinned staging block, no staging memcpy on the host: the images of chunk c + 1 are read and the results of
 * chunk c - 1 written by the copy engines while chunk c computes.  Such a call writes all n_features rows of every frame slot
 * (rows from n_out[f] on are unspecified); results are the same bits.  Anything else takes the staged pipeline above.
 * hfnet_host_register page-locks [ptr, ptr + bytes) (hipHostRegister); the range must stay allocated until
 * hfnet_host_unregister(ptr).  Registering a range twice or unregistering an unknown pointer is HFNET_ERR_INVALID_ARG. */
int hfnet_host_register(void* ptr, size_t bytes);
int hfnet_host_unregister(void* ptr);

/* ---- Matcher brute-force bodies (src/Matcher.cc) ------------------------------------------------ */
/* Matcher::DescriptorDistance (Matcher.cc:1893-1900) */
int hfnet_descriptor_distance(hfnet_engine* e, const float* a, const float* b, int dim, float* out);
/* SearchByBoW body (Matcher.cc:229-260, 574-618): cv::BFMatcher(NORM_L2, crossCheck=true).match
 * followed by distance < th_low.  match_q2t[i] = train row matched to query row i or -1.
 * Rows are `dim` floats, contiguous.  on_device as above. */
int hfnet_match_search_by_bow(hfnet_engine* e, const float* query, int n_query, const float* train,
                              int n_train, int dim, float th_low, int32_t* match_q2t, float* dist,
                              int* n_matches, int on_device);
/* The same over many descriptor-set pairs in four launches (LoopClosing matches a keyframe against all its
 * candidates' covisibles, LoopClosing.cc:606-712; offline pipelines match every frame against its predecessor).
 * Set s holds n_rows[s] rows of `dim` floats at desc_base + s * set_stride (floats); pair p matches query set
 * query_set[p] against train set train_set[p].  match_q2t / dist: [n_pairs][max_rows], n_matches: [n_pairs].
 * on_device != 0: EVERY pointer (n_rows, query_set, train_set included) is a device pointer, the call only
 * enqueues -- the keypoint counts written by hfnet_extractor_extract_batch never have to visit the host. */
int hfnet_match_search_by_bow_batch(hfnet_engine* e, int n_pairs, const float* desc_base, size_t set_stride,
                                    const int32_t* n_rows, int n_sets, const int32_t* query_set,
                                    const int32_t* train_set, int max_rows, int dim, float th_low,
                                    int32_t* match_q2t, float* dist, int32_t* n_matches, int on_device);
/* SearchForTriangulation body (Matcher.cc:845-889): S = D1 * D2^T, threshold 1 - th_high^2 / 2,
 * row arg-max (strict >) + column cross-check.  match12[i] = row of d2 or -1. */
int hfnet_match_search_for_triangulation(hfnet_engine* e, const float* d1, int n1, const float* d2,
                                         int n2, int dim, float th_high, int32_t* match12,
                                         int* n_matches, int on_device);

/* SearchForTriangulation over many pairs (a new keyframe against its ~30 covisible neighbours,
 * LocalMapping.cc:516-520): same set store / pair description as hfnet_match_search_by_bow_batch; pair p matches the
 * rows of set1[p] against set2[p]; match12: [n_pairs][max_rows], n_matches: [n_pairs]. */
int hfnet_match_search_for_triangulation_batch(hfnet_engine* e, int n_pairs, const float* desc_base,
                                               size_t set_stride, const int32_t* n_rows, int n_sets,
                                               const int32_t* set1, const int32_t* set2, int max_rows, int dim,
                                               float th_high, int32_t* match12, int32_t* n_matches, int on_device);

/* The candidate loop the windowed matchers share (SearchByProjection x5, SearchForInitialization, Fuse x2, SearchBySim3:
 * Matcher.cc:74-110, 126-160, 313-341, 1652-1690, ...), for all queries of a call at once: query i (a MapPoint's descriptor)
 * is compared with the train rows cand_index[cand_offsets[i] .. cand_offsets[i+1]) (the keypoints the caller's grid lookup
 * returned, already filtered by its ownership / stereo tests) in list order with Matcher::DescriptorDistance; outputs are the
 * reference's bestIdx / bestDist / bestLevel / bestDist2 / bestLevel2 (strict-< updates; empty list: -1, FLT_MAX, -1).
 * train_level: octave of every train row (may be NULL: 0).  The thresholds and the ratio test that follow stay with the
 * caller.  cand_index entries must lie in [0, n_train).  on_device as above (then nothing is validated). */
int hfnet_match_candidates(hfnet_engine* e, const float* query, int n_query, const float* train, int n_train,
                           const int32_t* train_level, int dim, const int32_t* cand_offsets, const int32_t* cand_index,
                           int32_t* best_idx, float* best_dist, int32_t* best_level, float* second_dist,
                           int32_t* second_level, int on_device);
/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:366-400) for n_sets map points at once: set s = the descriptors of its
 * observations, rows [set_offsets[s], set_offsets[s+1]) of desc; best[s] = the row (index inside the set) with the least
 * median DescriptorDistance to the others, first one on ties; -1 for an empty set.  At most 96 rows per set
 * (HFNET_ERR_CAPACITY otherwise).  Host pointers. */
int hfnet_distinctive_descriptors(hfnet_engine* e, const float* desc, const int32_t* set_offsets, int n_sets, int dim,
                                  int32_t* best);

/* ---- device-resident descriptor store --------------------------------------------
 * Matcher.cc re-gathers and would re-upload the N x 256 blocks of both keyframes on every call (Matcher.cc:231-246,
 * 808-834).  A store keeps each keyframe's block on the GPU: hfnet_store_put uploads a set once (slot ids are managed by
 * the caller, like the database's), the two searches then take slot pairs and only move the pair lists and the matches.
 * Each row carries a flag byte ("has a MapPoint", hfnet_store_set_flags, 0 after put); a search can restrict either
 * side to the flagged or unflagged rows, which is the gather of Matcher.cc:231-246 (rows WITH a MapPoint) and :808-834
 * (rows WITHOUT one) done on the device; results keep the ORIGINAL row numbers of both sides, excluded rows get -1
 * (distance FLT_MAX).
 * match arrays: [n_pairs][max_rows] (row r of pair p: set1[p]'s descriptor r), n_matches: [n_pairs]. */
enum { HFNET_ROWS_ALL = 0, HFNET_ROWS_FLAGGED = 1, HFNET_ROWS_UNFLAGGED = 2 };
int hfnet_store_create(hfnet_engine* e, int n_sets, int max_rows, int dim, hfnet_store** out);
void hfnet_store_destroy(hfnet_store* s);
int hfnet_store_put(hfnet_store* s, int slot, const float* rows, int n_rows);
/* the descriptors of staging frame `frame` of the extractor's last host-pointer call, device to device: the block of a
 * frame that was just extracted never travels back up (slot flags are cleared, like hfnet_store_put) */
int hfnet_store_put_extracted(hfnet_store* s, int slot, hfnet_extractor* x, int frame);
/* Host pipelines that also match on the device: after this call, frame f of every host-pointer extraction
 * (hfnet_extractor_extract / _extract_batch with on_device == 0) additionally lands, device to device, in slot
 * (first_slot + f) % n_sets of `store` (flags cleared), so a whole batch can be matched by slot without its descriptors
 * going back up.  store == NULL detaches. */
int hfnet_extractor_attach_store(hfnet_extractor* x, hfnet_store* s, int first_slot);
int hfnet_store_rows(const hfnet_store* s, int slot);            /* rows of a slot, -1 for a bad slot */
int hfnet_store_set_flags(hfnet_store* s, int slot, const uint8_t* flags, int n_rows);
int hfnet_store_search_by_bow(hfnet_store* s, int n_pairs, const int32_t* query_set, const int32_t* train_set,
                              int query_rows, int train_rows, float th_low, int32_t* match_q2t, float* dist,
                              int32_t* n_matches);
int hfnet_store_search_for_triangulation(hfnet_store* s, int n_pairs, const int32_t* set1, const int32_t* set2,
                                         int rows1, int rows2, float th_high, int32_t* match12, int32_t* n_matches);

/* ---- Resampler (include/Extractors/BaseModel.h:78-80, src/Extractors/BaseModel.cc:491-562) ---------
 * tensorflow.contrib.resampler: bilinear sampling of an NHWC fp32 map at (x, y) warp points with zero
 * padding; output[b][p][c].  Host pointers (the fused path inside hfnet_*_detect never calls this). */
int hfnet_resampler(hfnet_engine* e, const float* data, const float* warp, float* output, int batch_size,
                    int data_height, int data_width, int data_channels, int num_sampling_points);

/* ---- KeyFrameDatabase scan (src/KeyFrameDatabase.cc:75-104, 170-197) --------------------------- */
int hfnet_db_create(hfnet_engine* e, int capacity, int dim, hfnet_db** out);
void hfnet_db_destroy(hfnet_db* db);
/* KeyFrameDatabase::add / erase: slot ids are managed by the caller (mirrored on the KeyFrame) */
int hfnet_db_add(hfnet_db* db, int slot, const float* descriptor);
int hfnet_db_erase(hfnet_db* db, int slot);
int hfnet_db_clear(hfnet_db* db);
/* mode 0: DetectNBestCandidates filter (score > 0.8 * best);
 * mode 1: DetectRelocalizationCandidates filter (score > max(0.5, 0.8 * best)).
 * score = max(0, 1 - ||q - d||).  cand_slot / cand_score: caller buffers of `capacity` entries,
 * filled in ascending slot order; scores_all (may be NULL): one score per slot, -1 for empty.
 * A database of "db_screen_min_rows" (6144) slots or more takes the screened form of hfnet_db_query_batch below (dim <= 4096): a quarter of
 * the bytes per scan, the same bits (measured, host pointers in and out: 10 000 slots 63 -> 52 us per call, 40 000 slots 164 -> 77 us; below ~5 000
 * slots the exact scan is the faster one). */
int hfnet_db_query(hfnet_db* db, const float* query, int mode, int32_t* cand_slot, float* cand_score,
                   int* n_cand, float* best_score, float* scores_all);
/* The same scan for n_queries descriptors at once (a burst of keyframes at loop closing / relocalisation,
 * BASELINE config 5).  queries: [n_queries][dim]; cand_slot / cand_score: [n_queries][capacity] (row q holds n_cand[q]
 * entries); best_score: [n_queries] or NULL; scores_all: [n_queries][capacity] or NULL.
 *  - fewer than "db_gemm_min_queries" (8) queries against a database of fewer than "db_screen_min_rows" slots: the exact scan, the database
 *    crosses HBM once per 8 queries; per query
 *    EVERY result equals hfnet_db_query's bit for bit (dim <= 4096);
 *  - otherwise (dim <= 4096): the score is EXACTLY 0 for every keyframe at distance >= 1 from the query, so a crude product on
 *    the integer matrix pipe only has to find the slots that can be closer.  The database keeps an 8-bit copy of its rows (+ 1 byte per
 *    element in the matrix unit's fragment order, + 16 bytes per row; refreshed with the first batched query after an add): every vector
 *    as steps of its own scale s = max|x| / 127.  The int32 product of two step vectors is exact, so
 *    d2~ = |q|^2 + |d|^2 - 2 s_q s_d sum a_i b_i deviates from the true squared distance by at most
 *    2 err = s_d sum|s_q a_i| + s_q sum|s_d b_i| + dim s_q s_d / 2 (the quantisation error; both sums are stored per row); a slot with
 *    d2~ >= 1 + 2.002 err + 1e-4 (|q|^2 + |d|^2) is written as 0, every other occupied slot is scored with hfnet_db_query's exact chain
 *    (for unit vectors of 4096 roughly Gaussian components: everything beyond d^2 ~ 1.055).
 *    EVERY result -- scores_all of every slot, best_score, the candidate set, cand_score -- equals hfnet_db_query's bit for bit,
 *    whatever the burst size.  Cost: one launch and one pass over the 8-bit copy per 64 queries + 32 KB per (query, keyframe the bound
 *    cannot rule out).  Engine option "match_stats" = 1 counts those pairs (read-only option "stat_db_exact": read and cleared). */
int hfnet_db_query_batch(hfnet_db* db, int n_queries, const float* queries, int mode, int32_t* cand_slot,
                         float* cand_score, int32_t* n_cand, float* best_score, float* scores_all);

/* ---- measurement hooks (bench.py: HIP events on the engine stream around every launch) -------- */
int hfnet_profile_enable(hfnet_engine* e, int on);
int hfnet_profile_reset(hfnet_engine* e);
/* record only launches whose name equals `name` (NULL or "" = every launch) */
int hfnet_profile_filter(hfnet_engine* e, const char* name);
/* number of distinct kernels seen since reset */
int hfnet_profile_count(hfnet_engine* e);
/* i-th kernel: name (<= 63 chars), launches, total milliseconds */
int hfnet_profile_get(hfnet_engine* e, int i, char* name, int name_cap, int* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* HFNET_HIP_H */
