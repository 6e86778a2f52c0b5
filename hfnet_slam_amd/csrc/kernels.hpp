// kernels.hpp -- launchers of every gfx950 kernel (definitions in kernels_*.hip).
// All launchers enqueue on `s` and return hipGetLastError().
#pragma once
#include "common.hpp"

namespace hfnet {

// ---- kernels_conv.hip ---------------------------------------------------------------------------
// u8 pyramid level -> level (cv::resize INTER_LINEAR, HFextractor.cc:159-173); tables built on host
hipError_t launch_resize_u8(const uint8_t* src, int sw, int sh, int s_row, long long s_frame,
                            uint8_t* dst, int dw, int dh, int d_row, long long d_frame,
                            const int* xofs, const short* ialpha, const int* yofs, const short* ibeta,
                            int batch, hipStream_t s, int band_rows = 0);
// band_rows > 0: the form that stages a workgroup's source band through LDS (each source row read once, as 16-byte pieces);
// band_rows = resize_band_rows(host yofs table, dh, sh), the largest band of the level.  0: the thread-per-column form.
int resize_band_rows(const int* yofs, int dh, int sh);
// the whole chain level 0 -> 1 -> .. -> n (n <= 3) in one launch, same bytes (arrays indexed by level, [0] of dst / tables unused);
// d_row: padded row length of a produced level (the padding repeats the last column, as launch_resize_u8 leaves it)
bool pyramid_chain_supported(int n, const int* w, const int* h);
hipError_t launch_pyramid_chain(const uint8_t* src, int s_row, long long s_frame, int n, const int* w, const int* h, uint8_t* const* dst,
                                const int* d_row, const long long* d_frame, const int* const* xofs, const short* const* ialpha,
                                const int* const* yofs, const short* const* ibeta, int batch, hipStream_t s);
// image prep + stem conv 3x3/2 + BN + ReLU6 (HFNetTFModelV2.cc:204-208, layers.py:6-7, hf_net.py:30,188-190)
hipError_t launch_stem(const ImageSet& imgs, const float* w, const float* bias, int cout, float* out, const Geom& g, hipStream_t s);
// 1x1 convolution on the matrix cores: out[P x n] = epilogue(A[P x cin] * W).  slot_units (optional, device): the rows are
// slots of slot_rows rows per image of which only the first slot_units[image] * rows_per_unit are in use (tap rows of the
// sparse descriptor head); 32-row tiles in the unused part are skipped, their output rows are left untouched.
hipError_t launch_pointwise(const float* A, const ConvPack& cp, const float* residual, float* out, long long P,
                            int relu6, hipStream_t s, const int* slot_units = nullptr, int slot_rows = 0, int rows_per_unit = 0);
// dense 3x3 stride-1 convolution (implicit GEMM on the matrix cores), per-image tiles; wlds != 0: weights staged through LDS
// once per workgroup (k_conv3x3_wlds) where the shape allows it
hipError_t launch_conv3x3(const float* A, const ConvPack& cp, float* out, int relu6, const Geom& g, int wlds, hipStream_t s);
// the same convolution evaluated only at the 4 bilinear taps of every selected keypoint ("sparse
// descriptor head"): row (image*kps_stride + i)*4 + t of `out` is tap t of keypoint i.  Geom: H, W =
// score-map size, Ho, Wo = cell grid, in_off = first cell of the level.
// cells / n_rows (optional, launch_tap_cells): keypoints at least 5 pixels apart on an 8-pixel cell grid share taps -- the rows of an
// image are then its n_rows[image] DISTINCT tap cells (row r of the slot is cell cells[image * kps_stride * 4 + r]).
// split-bf16 forms (engine option desc_bf16x3; kernels_conv.hip): NOT the oracle's bits -- within the tolerance of include/hfnet_hip.h
size_t bf16x3_pack_bytes(const ConvPack& cp);
// with_bias: the folded bias goes into the spare k slot of an odd channel-group count (1x1 only): see k_repack_bf16x3
hipError_t launch_repack_bf16x3(const ConvPack& cp, void* out, hipStream_t s, int with_bias = 0);
inline bool bf16x3_supported(const ConvPack& cp) { return (cp.taps == 1 && cp.cin % 8 == 0) || (cp.taps == 9 && cp.cin % 16 == 0); }
hipError_t launch_conv3x3_cells_bf16x3(const float* A, const ConvPack& cp, const void* Wb, float* out, int relu6, long long kps_stride,
                                       const int* level_keypoints, const Geom& g, const int* cells, const int* n_rows, hipStream_t s);
hipError_t launch_pointwise_bf16x3(const float* A, const ConvPack& cp, const void* Wb, const float* residual, float* out, long long P, int relu6,
                                   hipStream_t s, const int* slot_units = nullptr, int slot_rows = 0, int rows_per_unit = 0);
// dense 3 x 3 convolution on split-bf16 operands with the halo staged through LDS (engine option scores_bf16x3: the detector head)
bool conv3x3_dense_bf16x3_supported(const ConvPack& cp, const Geom& g);
hipError_t launch_conv3x3_dense_bf16x3(const float* A, const ConvPack& cp, const void* Wb, float* out, int relu6, const Geom& g, hipStream_t s);
hipError_t launch_conv3x3_taps(const float* A, const ConvPack& cp, float* out, int relu6, const hfnet_keypoint* kps, const int* n_in,
                               long long kps_stride, const int* level_keypoints /* upper bound of n_in per level */, const Geom& g, int wlds,
                               hipStream_t s, const int* cells = nullptr, const int* n_rows = nullptr);
// depthwise 3x3 (stride 1 / 2) + BN + ReLU6
hipError_t launch_depthwise(const float* in, const DwPack& dp, int stride, float* out, const Geom& g, hipStream_t s);
// whole inverted-residual block (expand -> depthwise -> project [+ residual]) in one launch; the expanded
// tensor lives in LDS only.  block_fusable(): the block's shape has a fused kernel (kernels_block.hip).
// variant: 4 = wave-autonomous tiles where available (default), 3 = the same with the three-waves-per-SIMD instantiations at any
// launch size, 2 = the barrier-phased kernel everywhere (A/B of the tests)
bool block_fusable(const BlockPack& b, int variant);
hipError_t launch_block_fused(const float* X, const BlockPack& b, float* out, const Geom& g, int variant, hipStream_t s, int bf16x3 = 0);
// bf16x3 != 0 takes the split-bf16 form of the block (engine option global_bf16x3) where this says so
bool block_fused_bf16x3_supported(const BlockPack& b);
// stem conv + layer_2 (no-expansion block) in one launch: the stem tensor stays in LDS
bool stem_block_fusable(int stem_out, const BlockPack& b);
hipError_t launch_stem_block(const ImageSet& imgs, const float* stem_w, const float* stem_bias, const BlockPack& b, float* out,
                             const Geom& g_stem, const Geom& g_block, hipStream_t s);
// channel-order conversion between the device layout and NHWC logical order (boundary tensors)
hipError_t launch_permute_channels(const float* in, float* out, long long P, int C, int to_logical, hipStream_t s);

// ---- kernels_tail.hip ---------------------------------------------------------------------------
// single-frame path of layers 8-18: depthwise 3x3 + BN + ReLU6 and the 1x1 projection + BN [+ residual] in one launch, the
// projection as v_mfma_f32_16x16x4_f32 chains (a quarter of the 32x32x2 chain's latency); `expanded`: the block's expanded
// tensor (output of its expansion conv).  Same bits as launch_depthwise + launch_pointwise.
bool dwproject_supported(const BlockPack& b);
// next (optional): the 1x1 convolution that consumes the block's output -- the next block's expansion (next_relu 1) or the
// NetVLAD memberships conv (0) -- evaluated in the same launch: next_out[pixel][next->n] = act(out[pixel] * W + next_bias)
// next_softmax: the rows of next_out are softmaxed as launch_softmax_rows would (memberships; next->n <= 64)
hipError_t launch_dwproject(const float* expanded, const BlockPack& b, const float* residual, float* out, const ConvPack16* next,
                            const float* next_bias, float* next_out, int next_relu, int next_softmax, const Geom& g, hipStream_t s);

// ---- kernels_detect.hip -------------------------------------------------------------------------
// softmax(65) -> drop dustbin -> depth_to_space(8) (hf_net.py:88-93); logits row stride ld
hipError_t launch_softmax_d2s(const float* logits, int ld, float* dense, const Geom& g, hipStream_t s);
// detector tail in one launch: 1x1 conv (65 outputs) + softmax + depth_to_space; same bits as launch_pointwise + launch_softmax_d2s
bool det_tail_supported(const ConvPack& cp);
hipError_t launch_det_tail(const float* hidden, const ConvPack& cp, float* dense, const Geom& g, hipStream_t s);
// the same on split-bf16 operands (engine option scores_bf16x3; Wb: launch_repack_bf16x3 of cp): the scores within the stated tolerance
bool det_tail_bf16x3_supported(const ConvPack& cp);
hipError_t launch_det_tail_bf16x3(const float* hidden, const ConvPack& cp, const void* Wb, float* dense, const Geom& g, hipStream_t s);
// simple_nms(radius 4, 2 iterations) (layers.py:10-32) + candidate emission (score >= threshold,
// HFNetTFModelV2.cc:127-140).  counters: one uint per image, zeroed by the caller.
// counters: one uint per image, HFNET_COUNTER_STRIDE words apart (a cache line each: atomics on neighbouring words
// serialise in one L2 channel), zeroed by the caller.
#define HFNET_COUNTER_STRIDE 32
// mask0 / supp: bit-column masks, one word per (32-row block, column) of every image (scratch of the three passes).
// nms == nullptr: the suppressed map is not written; cand == nullptr: no candidates are emitted.
hipError_t launch_nms(const float* dense, float* nms, unsigned* mask0, unsigned* supp, unsigned long long* cand, unsigned int* counters,
                      long long cand_stride, float threshold, const Geom& g, hipStream_t s);
// top-K of the candidates in canonical order (HFNetTFModelV2.cc:144-151); writes level-resolution
// keypoints {x, y, response, octave=0} and the per-image count
struct TopkBudget { int k[HFNET_MAX_LEVELS]; };
hipError_t launch_topk(const unsigned long long* cand, const unsigned int* counters, long long cand_stride,
                       const TopkBudget& kmax_per_level, hfnet_keypoint* kps, long long kps_stride, int* n_out,
                       const Geom& g, hipStream_t s);
// *seq += 1 (one thread): the "results are down" counter of the single-frame host path (engine.hip)
hipError_t launch_bump_seq(int* seq, hipStream_t s);
// per-pixel L2 normalisation of the dense descriptor map (hf_net.py:80)
hipError_t launch_l2norm256(const float* in, float* out, long long P, hipStream_t s);
// bilinear Resampler + cv::normalize + keypoint rescale / concat (HFNetTFModelV2.cc:153-167,
// BaseModel.cc:491-562, HFextractor.cc:267-281)
// the distinct tap cells of every image's selected keypoints, in ascending cell order (Geom as launch_conv3x3_taps):
//   flags   [images][cell_stride] bytes, scratch     cell_row [images][cell_stride]: row of a cell in the image's slot (or -1)
//   cells   [images][kps_stride * 4]: cell of a row   n_rows   [images]
// fault (optional): device word that collects HFNET_FAULT_* bits -- an index from device memory that had to be bounded
#define HFNET_FAULT_TAP_ROWS 1u     // k_tap_compact: more marked tap cells than rows in the image's slot (flags not clean)
#define HFNET_FAULT_SAMPLE_ROW 2u   // k_sample: a tap cell of a selected keypoint has no row
hipError_t launch_tap_cells(const hfnet_keypoint* kps, const int* n_in, long long kps_stride, unsigned char* flags, int* cell_row, int* cells,
                            int* n_rows, long long cell_stride, const Geom& g, hipStream_t s, unsigned int* fault = nullptr);
// launch_topk + launch_tap_cells in one launch (g: launch_tap_cells' geometry)
hipError_t launch_topk_taps(const unsigned long long* cand, const unsigned int* counters, long long cand_stride, const TopkBudget& kmax_per_level,
                            hfnet_keypoint* kps, long long kps_stride, int* n_out, unsigned char* flags, int* cell_row, int* cells, int* n_rows,
                            long long cell_stride, const Geom& g, hipStream_t s, unsigned int* fault = nullptr);
struct SampleArgs {
    const float* desc_map;        // dense: normalised [pixels x 256]; sparse: RAW tap rows [image][kps_stride*4][256] (normalised on the fly)
    const int* cell_row;          // sparse, de-duplicated taps: row of cell (y * Wo + x) in the image's slot; null: rows 4 i .. 4 i + 3
    long long cell_stride;
    int sparse;
    const hfnet_keypoint* kps_in; // per image slot of kps_stride entries (level coordinates)
    const int* n_in;              // per image count
    long long kps_stride;
    hfnet_keypoint* kps_out;      // per frame: out_frame_stride entries, levels concatenated
    float* desc_out;              // per frame: out_frame_stride x 256
    int* n_out_frame;             // per frame total (may be null)
    int* n_out_level;             // per frame x level counts (may be null)
    long long out_frame_stride;
    float scale_factor[HFNET_MAX_LEVELS];  // pt *= scale_factor[level]; octave = level
    int set_octave;               // 0: single model (octave stays 0, no rescale)
    unsigned int* fault;          // optional: HFNET_FAULT_* bits (see launch_tap_cells)
};
hipError_t launch_sample(const SampleArgs& a, const Geom& g, hipStream_t s);
// free-standing Resampler (BaseModel.cc:491-562): out[b][p][c] for NHWC data and (x, y) warp points
hipError_t launch_resampler(const float* data, const float* warp, float* out, int batch, int dh, int dw, int channels, int npoints,
                            hipStream_t s);

// ---- kernels_global.hip -------------------------------------------------------------------------
hipError_t launch_softmax_rows(float* x, long long rows, int n, int ld, hipStream_t s);
// NetVLAD aggregation + intra-normalisation + both L2 normalisations (layers.py:77-97)
// vlad_tap: logical order; out: the FC kernel's slot order within every group of 16 (fc_slot_of_logical)
int vlad_scratch_parts();      // partial sums per output the scratch tensor of launch_vlad holds ([frames][parts][K * D])
hipError_t launch_vlad_aggregate(const float* feat /*phys layout [frames x P x D]*/, const float* memb /*[frames x P x K]*/, const float* clusters,
                                 float* scratch /*[frames][parts][K * D]*/, int frames, int P, int D, int K, hipStream_t s);
hipError_t launch_vlad_norm(const float* scratch, float* vlad_tap /*optional*/, float* out /*FC slot order*/, int frames, int D, int K, hipStream_t s);
// dimensionality reduction: FC + bias as one MFMA GEMM over the frames of the batch, split 16 ways along the inputs (weights
// cross HBM once, 4096 workgroups stream them) + L2 normalise (layers.py:98-108).  x: [frames][n_in] in the FC slot order
// (fc_slot_of_logical), pack: FcPack of weights.cpp; partial: fc_scratch_floats() floats
size_t fc_scratch_floats(const FcPack& fc, int frames);
// the FC on split-bf16 operands (engine option global_bf16x3, calls of many frames): FC_BF_PARTS partial GEMMs over contiguous input ranges
// (k_conv_rows_bf16x3 with K split over the grid's z) + a sum with the bias + the L2 normalisation; Wb: launch_repack_fc_bf16x3 of the FcPack
#define FC_BF_PARTS 8
size_t fc_bf16x3_pack_bytes(const FcPack& fc);
bool fc_bf16x3_supported(const FcPack& fc);
hipError_t launch_repack_fc_bf16x3(const FcPack& fc, void* out, hipStream_t s);
hipError_t launch_fc_partials_bf16x3(const float* x, const FcPack& fc, const void* Wb, float* partial, int frames, hipStream_t s);
hipError_t launch_fc_l2_bf16x3(const float* x, const FcPack& fc, const void* Wb, float* partial, float* y_raw, float* out, int frames, hipStream_t s);
// optional second destination of the global descriptors in host-visible (pinned) memory: `out` rows, then the call number
// (seq[0], kept on the device; seq[1] counts the workgroups that are done) into `flag`; calls of up to four frames only
struct FcHostOut { float* out = nullptr; int* flag = nullptr; int* seq = nullptr; };
bool fc_host_out_supported(int frames);
// fc_tile (engine option): 1 = the blocked kernel (k_fc_mfma_tile) for calls whose 64-frame workgroups fill the chip, 2 / 4 = that kernel with
// 32 / 64 columns per workgroup at any size above 16 frames (tests), 0 = never
hipError_t launch_fc_l2(const float* x, const FcPack& fc, float* partial, float* y_raw, float* out, int frames, hipStream_t s, FcHostOut host = FcHostOut(), int fc_tile = 1);

// ---- kernels_match.hip --------------------------------------------------------------------------
// BFMatcher(NORM_L2, crossCheck) + distance < th_low (Matcher.cc:229-260), batched over descriptor-set pairs.
// One BowPair per (query set, train set); launch_bow_setup fills them on the device (row counts may be
// device-resident), launch_bow_pairs runs prep / GEMM / train pass / finalize for all pairs in four launches.
struct BowPair {
    const float* q; const float* t;
    float* St; float* qn; float* tn; unsigned long long* qkey;
    int32_t* match; float* dist; int* cnt;
    int nq, nt;
};
hipError_t launch_bow_setup(BowPair* pairs, int n_pairs, const float* base, long long set_stride, const int* n_rows, const int* qset,
                            const int* tset, int max_rows, float* St, long long st_stride, float* qn, float* tn, unsigned long long* qkey, int32_t* match,
                            float* dist, int* cnt, long long out_stride, hipStream_t s);
// row-filtered matching over a descriptor store (flag byte per row; sel >= 0 store slot, sel < 0 compacted set ~sel)
hipError_t launch_store_compact(const float* base, const unsigned char* flags, long long set_stride, const int* store_rows, int n_compact,
                                const int* c_slot, const int* c_filter, int max_rows, int dim, int* map, int* inv, int* c_rows, float* comp,
                                hipStream_t s);
hipError_t launch_store_setup(BowPair* pairs, int n_pairs, const float* base, const float* comp, long long set_stride, const int* store_rows,
                              const int* c_rows, const int* qsel, const int* tsel, int max_rows, float* St, long long st_stride, float* qn, float* tn,
                              unsigned long long* qkey, int32_t* match, float* dist, int* cnt, hipStream_t s);
hipError_t launch_store_remap(int n_pairs, const int* qsel, const int* tsel, const int* c_slot, const int* store_rows, const int* map,
                              const int* inv, int max_rows, const int32_t* c_match, const float* c_dist, int32_t* match, float* dist,
                              hipStream_t s);
// SearchForTriangulation over the same pair descriptors (q = set 1, t = set 2; dist / qn / qkey unused): S = D1 * D2^T as
// fused multiply-add chains over k = 0..dim-1 (Matcher.cc:845-849) with the mutual arg-max (Matcher.cc:851-889) in the GEMM
// epilogue.  Scratch per pair (BowPair::St): tri_scratch_floats(max_rows) floats of (maximum, index) partials, no n x m matrix.
size_t tri_scratch_floats(int max_rows);
size_t tri_split_offset_bytes(int n_pairs, int max_rows);
size_t tri_scratch_bytes(int n_pairs, int max_rows, int dim);
// split_scratch: tri_scratch_bytes' second part (null: the full f32 path for every pair); stat: two device ints {pairs whose list
// overflowed, pairs}, incremented by the screened path (may be null)
hipError_t launch_tri_pairs(const BowPair* pairs, int n_pairs, int max_rows, int dim, float threshold, hipStream_t s, void* split_scratch, int* stat);
// scratch: bow_scratch_bytes(n_pairs, max_rows, dim) bytes (candidate slots per train row and 64-query tile + the rows of every pair split
// into bf16 pieces for the screening GEMM; no n x m matrix)
size_t bow_scratch_bytes(int n_pairs, int max_rows, int dim);
// stat (may be null): += the number of exact distance evaluations (what the screen let through; engine read-only option stat_bow_exact)
hipError_t launch_bow_pairs(const BowPair* pairs, int n_pairs, int max_rows, int dim, float th_low, void* scratch, hipStream_t s, int screen_bf16, int* stat = nullptr);
hipError_t launch_descriptor_distance(const float* a, const float* b, int dim, float* out, hipStream_t s);
// the candidate loop of the windowed matchers (Matcher.cc:74-110 and siblings): best / second best with levels per query
hipError_t launch_match_candidates(const float* query, int nq, const float* train, const int* train_level, int dim, const int* cand_offsets,
                                   const int* cand_index, int* best_idx, float* best_dist, int* best_level, float* second_dist, int* second_level,
                                   hipStream_t s);
// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:366-400) over many observation sets (<= distinctive_max_rows() rows each)
int distinctive_max_rows();
hipError_t launch_distinctive(const float* desc, const int* set_offsets, int n_sets, int dim, int* best, hipStream_t s);
// KeyFrameDatabase scan (KeyFrameDatabase.cc:86-104, 178-197)
// best_partial: one word per wave of the scan (db_scan_workgroups(n) * 4, resp. db_batch_workgroups(n) * 4 per query);
// launch_db_filter reduces them (no atomics on the data path)
int db_scan_workgroups(int n);
int db_batch_workgroups(int n);
hipError_t launch_db_scores(const float* q, const float* db, const unsigned char* occupied, int n, int dim, float* scores,
                            unsigned int* best_partial, hipStream_t s);
// n_queries rows of n scores / candidates, one best / count per query
hipError_t launch_db_filter(const float* scores, int n, int mode, const unsigned int* best_partial, int n_partials, int32_t* cand_slot,
                            float* cand_score, int* n_cand, float* best, int n_queries, hipStream_t s);
// scores[q][slot] for n_queries queries (dim <= 4096): the database is read once per 8 queries
hipError_t launch_db_scores_batch(const float* q, int n_queries, const float* db, const unsigned char* occupied, int n, int dim,
                                  float* scores, unsigned int* best_partial, hipStream_t s);
// the same scan for many queries (n_queries >= 8), screened on the integer matrix pipe: a slot whose crude squared distance (8-bit copies of
// both vectors at their own scales, one exact int32 product, a rigorous bound of the quantisation error from the rows' scales and 1-norms) is
// >= 1 + the bound is at distance >= 1 whatever the rounding -- score exactly 0 --, every other occupied slot is scored with launch_db_scores'
// exact chain: ALL outputs equal the exact scan's bit for bit (kernels_match.hip).
// stat / hi: per row |x|^2 in tree256 order, scale, scaled 1-norm (db_stat_floats) and the 8-bit copy of every row in the matrix unit's
// fragment order (db_hi_bytes) -- launch_db_prep_hi: the database's when rows were added, the queries' per call; best_partial:
// [n_queries][db_gemm_partials(n)]
int db_gemm_partials(int n);
bool db_screen_supported(int dim);     // descriptor lengths the screened batched query takes (others: the exact batched scan)
size_t db_hi_bytes(int n_rows, int dim);   // the 8-bit copy of n_rows vectors: whole 32-row tiles in fragment order
size_t db_stat_floats(int n_rows);         // |x|^2, scale, 1-norm of the steps per row
hipError_t launch_db_prep_hi(const float* x, int n_rows, int dim, float* stat, void* hi, hipStream_t s);
// up to 64 queries from q0 on per launch; stat (may be null): += the pairs scored exactly (engine read-only option stat_db_exact)
hipError_t launch_db_sweep(const float* q, const void* qh, int n_queries, int q0, const float* qstat, const float* db, const void* dbh, const float* dstat,
                           const unsigned char* occupied, int n, int dim, float* scores, unsigned int* best_partial, hipStream_t s, int* stat);

}  // namespace hfnet
