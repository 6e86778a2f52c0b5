/*
 * hfnet_oracle.h -- CPU restatement of the HF-Net front end of LiuLimingCode/HFNet_SLAM.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / reported baseline.  The product path (hfnet_slam_amd/csrc) never links it.
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path,
 * its model runtimes (TensorFlow C++, TensorRT), OpenCV, Eigen and the weights are not in the
 * image, so none of its own code for the path can be built or run here.  This file restates the
 * algorithm from the cited reference sources and from the published semantics of the
 * un-vendored third-party pieces (TensorFlow 1.15/2.9 ops, OpenCV 4.2 resize / normalize /
 * BFMatcher, Eigen 3 norm / GEMM).  It is cross-checked against an independent PyTorch-CPU
 * restatement (oracle/torch_ref.py) and closed-form known-answer tests (tests/).
 *
 * Floating-point contract ("canonical order").  The reference leaves summation order to Eigen /
 * OpenCV SIMD code, i.e. unspecified.  The oracle fixes one order so results are reproducible
 * bit-for-bit by any implementation that follows it:
 *   - convolutions / matmuls (the 7680 -> 4096 dimensionality reduction included): inference BatchNorm is folded
 *     into the layer as inference engines do (w[..., c] *= gamma[c] / sqrt(var[c] + eps), one f32 rounding per weight;
 *     bias[c] = beta[c] - mean[c] * that scale); one accumulator per output, STARTED AT bias[c], updated with a fused
 *     multiply-add per term, terms in (ky, kx, cin) order; then ReLU6 where the layer has one.
 *   - short sums (softmax over 65 / 32 channels, intra-norm over K): left to right.
 *   - the NetVLAD sum over pixels: the pixels in 8 contiguous ranges of ceil(P / 8), one left-to-right chain per range from 0
 *     (r = c - f; t = r * m; acc = acc + t), the eight partial sums added as a balanced binary tree (global_head).
 *   - the dimensionality-reduction FC: sixteen partial fma chains over contiguous input ranges, balanced binary tree, + bias.
 *   - long vector reductions (L2 norms over 256 / 4096 / 7680 elements, descriptor distances): "tree256" --
 *     256 interleaved partial sums (element i goes to partial i % 256, in increasing i) followed
 *     by a binary tree (stride 128, 64, ..., 1).
 *   - exp() in the softmaxes is hfo_expf below (Cephes-style polynomial, the same family Eigen's
 *     pexp -- what TensorFlow's CPU softmax runs -- uses), so that it is reproducible.
 * Build with -ffp-contract=off: every fused operation is written as an explicit fmaf().
 *
 * Known deviations of this restatement from the third-party code it stands for (inside any fp32 tolerance, and exactly what
 * "unpinned" means): (1) Eigen's GEMM / norm kernels sum in SIMD-lane order, not in the canonical orders above.  (2) cv_l2 (the
 * distance behind cv::BFMatcher NORM_L2, hfnet_oracle.c) is OpenCV 4.2's GENERIC normL2Sqr template -- four-way unrolled, one
 * accumulator: s += v0^2 + v1^2 + v2^2 + v3^2 -- while an x86 / NEON build of OpenCV dispatches to the SIMD normL2Sqr_ (four
 * VECTOR accumulators updated with v_muladd, reduced at the end): another summation order, so a BFMatcher distance can differ in
 * its last ulp and a first-minimum tie can fall on the other row.
 */
#ifndef HFNET_ORACLE_H
#define HFNET_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hfo_model hfo_model;

/* BaseModel.h:16-21 */
enum { HFO_IMAGE_TO_LOCAL_AND_GLOBAL = 0, HFO_IMAGE_TO_LOCAL = 1,
       HFO_IMAGE_TO_LOCAL_AND_INTERMEDIATE = 2, HFO_INTERMEDIATE_TO_GLOBAL = 3 };

/* cv::KeyPoint fields the path writes (HFNetTFModelV2.cc:122-138, HFextractor.cc:272-279) */
typedef struct { float x, y, response; int32_t octave; } hfo_keypoint;

/* taps: optional dumps of intermediate tensors (NULL entries are skipped) */
enum { HFO_TAP_STEM = 0,           /* layer_1 output                         */
       HFO_TAP_BLOCK0 = 1,         /* layer_2 .. layer_18 outputs: 1..17     */
       HFO_TAP_DESC_HIDDEN = 18,   /* descriptor 3x3 conv + BN + ReLU6       */
       HFO_TAP_DESC_RAW = 19,      /* descriptor 1x1 conv + bias (pre-norm)  */
       HFO_TAP_DET_HIDDEN = 20,
       HFO_TAP_LOGITS = 21,        /* Hd x Wd x 65                           */
       HFO_TAP_SCORES_DENSE = 22,  /* H' x W' (before NMS)                  */
       HFO_TAP_MEMBERSHIPS = 23,   /* Hg x Wg x K after softmax              */
       HFO_TAP_VLAD = 24,          /* K*D after both VLAD normalisations     */
       HFO_N_TAPS = 32 };

hfo_model* hfo_model_load(const char* path);
void hfo_model_free(hfo_model* m);
/* queries: 0 stem_out, 1 local_channels, 2 global_channels, 3 n_clusters, 4 global_dim */
int hfo_model_info(const hfo_model* m, int what);
void hfo_set_threads(int n);

/* --- pieces --- */
float hfo_expf(float x);
float hfo_sumsq_tree256(const float* x, int n);
double hfo_sumsq_tree256_d(const float* x, int n);
void hfo_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride);
void hfo_simple_nms(const float* scores, int h, int w, int radius, int iterations, float* out);
void hfo_resampler(const float* data, const float* warp, float* output, int batch, int dh, int dw,
                   int channels, int npoints);
int hfo_nms_points(const hfo_keypoint* in, int n, int width, int height, int radius, hfo_keypoint* out);
int hfo_select_keypoints(const float* scores_nms, int h, int w, float threshold, int kmax,
                         hfo_keypoint* kps);
void hfo_sample_descriptors(const float* desc_map, int hd, int wd, int channels,
                            const hfo_keypoint* kps, int n, int h, int w, float* out);

/* --- network --- */
/* image: u8 H x W with row stride; outputs sized for the cropped H' = H/8*8, W' = W/8*8,
 * Hd = H'/8, Wd = W'/8.  Any output pointer may be NULL. */
int hfo_run_local(const hfo_model* m, const uint8_t* img, int h, int w, int stride,
                  float* scores_nms /*H'xW'*/, float* desc_map /*HdxWdx256*/,
                  float* intermediate /*HdxWdxC7*/, float* global_desc /*global_dim*/,
                  float** taps /*HFO_N_TAPS or NULL*/);
int hfo_run_global(const hfo_model* m, const float* intermediate, int hd, int wd,
                   float* global_desc, float** taps);

/* BaseModel::Detect x3 (HFNetTFModelV2.cc:62-98).  Return 1 ok, 0 = "false" (wrong mode). */
int hfo_detect(const hfo_model* m, int mode, const uint8_t* img, int h, int w, int stride,
               int nkeypoints, float threshold, hfo_keypoint* kps, float* local_desc,
               float* global_or_intermediate, int* n_out);
int hfo_detect_global(const hfo_model* m, int mode, const float* intermediate, int hd, int wd,
                      float* global_desc);

/* HFextractor (HFextractor.cc:82-284) */
void hfo_extractor_tables(int nfeatures, int nlevels, float scale_factor, int width, int height,
                          float* scale_factors, int* features_per_level, int* level_w, int* level_h);
int hfo_extract(const hfo_model* m, const uint8_t* img, int h, int w, int stride,
                int nfeatures, float threshold, int nlevels, float scale_factor,
                hfo_keypoint* kps, float* local_desc, float* global_desc,
                int* n_per_level /*nlevels or NULL*/);

/* --- matching (Matcher.cc) --- */
float hfo_descriptor_distance(const float* a, const float* b, int dim);
/* cv::BFMatcher(NORM_L2, crossCheck=true).match(query, train): train_idx[q] = -1 if unmatched */
void hfo_bfmatch_l2_crosscheck(const float* q, int nq, const float* t, int nt, int dim,
                               int32_t* train_idx, float* dist);
/* SearchByBoW brute-force body: BFMatcher + (distance < th_low).  Returns #matches. */
int hfo_search_by_bow(const float* q, int nq, const float* t, int nt, int dim, float th_low,
                      int32_t* match_q2t, float* dist);
/* SearchForTriangulation brute-force body (Matcher.cc:845-889): dot-product mutual NN. */
int hfo_search_for_triangulation(const float* d1, int n1, const float* d2, int n2, int dim,
                                 float th_high, int32_t* match12, float* sim /*n1xn2 or NULL*/);

/* the candidate loop of the windowed matchers (Matcher.cc:74-110 and its siblings): best / second best with levels */
void hfo_match_candidates(const float* query, int nq, const float* train, const int32_t* train_level, int dim,
                          const int32_t* cand_offsets, const int32_t* cand_index,
                          int32_t* best_idx, float* best_dist, int32_t* best_level, float* second_dist, int32_t* second_level);
/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:366-400) over many observation sets */
void hfo_distinctive_descriptors(const float* desc, const int32_t* set_offsets, int n_sets, int dim, int32_t* best);

/* --- place recognition (KeyFrameDatabase.cc:86-104,178-197) --- */
void hfo_db_scores(const float* query, const float* db, int n, int dim, float* scores);
/* mode 0: DetectNBestCandidates filter (> 0.8 best); mode 1: relocalisation (> max(0.5, 0.8 best)) */
int hfo_db_candidates(const float* scores, int n, int mode, int32_t* idx, float* best);

#ifdef __cplusplus
}
#endif
#endif
