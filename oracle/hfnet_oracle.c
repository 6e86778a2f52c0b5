/*
 * hfnet_oracle.c -- CPU restatement of the HF-Net front end (see hfnet_oracle.h for the
 * status of this file: test infrastructure, parity unpinned, canonical summation order).
 *
 * Reference sources restated here (paths relative to /root/reference):
 *   graph        hfnet/models/hf_net.py:13-52,55-96,184-237
 *                hfnet/models/utils/layers.py:6-7,10-32,57-109
 *                hfnet/models/backbones/utils/conv_blocks.py:163-312
 *                hfnet/models/backbones/utils/mobilenet.py:148-294
 *                hfnet/export_model.py:35-37
 *   post-proc    src/Extractors/HFNetTFModelV2.cc:62-178,204-237
 *                src/Extractors/BaseModel.cc:491-603
 *   extractor    src/Extractors/HFextractor.cc:82-284
 *   matching     src/Matcher.cc:33-34,220-263,561-621,845-889,1893-1900
 *   place recog. src/KeyFrameDatabase.cc:86-104,178-197
 * Third-party semantics restated from their published behaviour (not in the tree):
 *   TensorFlow 'SAME' padding, slim batch_norm (eps 1e-3), depth_to_space (NHWC), max_pool_v2,
 *   tf.nn.l2_normalize (x * rsqrt(max(sum x^2, 1e-12))), OpenCV 4.2 cv::resize(INTER_LINEAR, 8U),
 *   cv::normalize(NORM_L2), cv::BFMatcher(NORM_L2, crossCheck), Eigen norm().
 */
#include "hfnet_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HFO_FC_PARTS 16   /* partial chains of the dimensionality-reduction FC (see global_head) */
#define HFO_VLAD_PARTS 8  /* partial chains of the NetVLAD pixel sum (see global_head) */

#define HFO_BN_EPS 1e-3f
#define HFO_DESC_DIM 256
#define HFO_DET_CH 65
#define HFO_GRID 8
#define HFO_NBLOCKS 17

/* ------------------------------------------------------------------ weights container */

typedef struct { char name[96]; uint32_t ndim; uint32_t dims[4]; uint64_t offset; uint64_t nbytes; } hfo_entry;

_Static_assert(sizeof(hfo_entry) == 136, "container entry layout");

/* convolution + inference BatchNorm folded the way inference engines fold it (the reference's default backend,
 * TensorRT, does the same to the exported graph): w[..., c] *= scale[c] (one f32 rounding per weight) and
 * bias[c] = shift[c]; the accumulator then STARTS at bias[c].  w is owned. */
typedef struct { float* w; float* bias; } hfo_convbn;

typedef struct {
    int cin, expand, stride, cout, residual, has_expand;
    hfo_convbn ex, dw, pr;
} hfo_block;

struct hfo_model {
    unsigned char* blob; size_t blob_size; int n; hfo_entry* entries;
    int stem_out, c_local, c_global, n_clusters, global_dim, det_hidden;
    hfo_convbn stem;
    hfo_block blocks[HFO_NBLOCKS];
    hfo_convbn desc1; const float* desc2_w; const float* desc2_b;
    hfo_convbn det1;  const float* det2_w;  const float* det2_b;
    hfo_convbn memb;  const float* clusters; const float* fc_w; const float* fc_b;
    float* fc_wt;   /* [global_dim][K*D] transposed copy, built on first use */
};

static const int k_strides[HFO_NBLOCKS] = {1, 2, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1}; /* hf_net.py:31-50 */

static const hfo_entry* find_entry(const hfo_model* m, const char* name) {
    for (int i = 0; i < m->n; ++i) if (strcmp(m->entries[i].name, name) == 0) return &m->entries[i];
    return NULL;
}
static const float* tensor(const hfo_model* m, const char* name, const hfo_entry** e_out) {
    const hfo_entry* e = find_entry(m, name);
    if (e_out) *e_out = e;
    return e ? (const float*)(m->blob + e->offset) : NULL;
}

/* slim.batch_norm inference: y = x * scale + shift with scale = gamma / sqrt(var + eps), shift = beta - mean * scale,
 * folded into the convolution: wf[k][c] = w[k][c] * scale[c], bias[c] = shift[c].  gamma may be absent for one scope only:
 * slim.batch_norm defaults to scale=False, and the NetVLAD memberships conv is built outside the mobilenet arg_scope
 * (hfnet/models/utils/layers.py:71-76), so a real checkpoint has no gamma there -> 1.  A missing gamma anywhere else is a
 * truncated container: refuse it. */
static int fold_bn(const hfo_model* m, const char* scope, const float* w, size_t rows, int c, hfo_convbn* out) {
    char nm[160];
    const float *g, *b, *mu, *var;
    snprintf(nm, sizeof nm, "%s/BatchNorm/gamma", scope);           g = tensor(m, nm, NULL);
    snprintf(nm, sizeof nm, "%s/BatchNorm/beta", scope);            b = tensor(m, nm, NULL);
    snprintf(nm, sizeof nm, "%s/BatchNorm/moving_mean", scope);     mu = tensor(m, nm, NULL);
    snprintf(nm, sizeof nm, "%s/BatchNorm/moving_variance", scope); var = tensor(m, nm, NULL);
    if (!b || !mu || !var) return 0;
    if (!g && strcmp(scope, "global_head/vlad/memberships") != 0) return 0;
    out->w = (float*)malloc(sizeof(float) * rows * (size_t)c);
    out->bias = (float*)malloc(sizeof(float) * c);
    for (int i = 0; i < c; ++i) {
        float s = (g ? g[i] : 1.0f) / sqrtf(var[i] + HFO_BN_EPS);
        float ms = mu[i] * s;
        out->bias[i] = b[i] - ms;
        for (size_t r = 0; r < rows; ++r) out->w[r * (size_t)c + i] = w[r * (size_t)c + i] * s;
    }
    return 1;
}

/* weights: [..., cout] with cout the last (fastest) dimension in every layout used here (HWIO convs; depthwise
 * [3,3,C,1] read as [9][C]) */
static int load_convbn(const hfo_model* m, const char* scope, const char* wname, int cout_dim, hfo_convbn* out, int* cout) {
    char nm[160];
    const hfo_entry* e;
    snprintf(nm, sizeof nm, "%s/%s", scope, wname);
    const float* w = tensor(m, nm, &e);
    if (!w) return 0;
    *cout = (int)e->dims[cout_dim];
    size_t total = 1;
    for (uint32_t d = 0; d < e->ndim; ++d) total *= e->dims[d];
    return fold_bn(m, scope, w, total / (size_t)*cout, *cout, out);
}

hfo_model* hfo_model_load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    hfo_model* m = (hfo_model*)calloc(1, sizeof(hfo_model));
    m->blob = (unsigned char*)malloc((size_t)sz); m->blob_size = (size_t)sz;
    if (fread(m->blob, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); hfo_model_free(m); return NULL; }
    fclose(f);
    if (sz < 16 || memcmp(m->blob, "HFNETW1\0", 8) != 0) { hfo_model_free(m); return NULL; }
    uint32_t n; memcpy(&n, m->blob + 8, 4);
    m->n = (int)n; m->entries = (hfo_entry*)(m->blob + 16);

    int ok = 1, c;
    ok &= load_convbn(m, "MobilenetV2/Conv", "weights", 3, &m->stem, &m->stem_out);
    int cin = m->stem_out;
    for (int i = 0; i < HFO_NBLOCKS && ok; ++i) {
        hfo_block* b = &m->blocks[i];
        char scope[64], sub[96];
        if (i == 0) snprintf(scope, sizeof scope, "MobilenetV2/expanded_conv");
        else snprintf(scope, sizeof scope, "MobilenetV2/expanded_conv_%d", i);
        b->cin = cin; b->stride = k_strides[i];
        snprintf(sub, sizeof sub, "%s/expand", scope);
        b->has_expand = 0; b->expand = cin;
        {   char nm[160]; snprintf(nm, sizeof nm, "%s/weights", sub);
            if (find_entry(m, nm)) { ok &= load_convbn(m, sub, "weights", 3, &b->ex, &b->expand); b->has_expand = 1; } }
        snprintf(sub, sizeof sub, "%s/depthwise", scope);
        ok &= load_convbn(m, sub, "depthwise_weights", 2, &b->dw, &c);
        ok &= (c == b->expand);
        snprintf(sub, sizeof sub, "%s/project", scope);
        ok &= load_convbn(m, sub, "weights", 3, &b->pr, &b->cout);
        b->residual = (b->stride == 1 && b->cin == b->cout);   /* conv_blocks.py:304-311 */
        cin = b->cout;
    }
    m->c_local = m->blocks[5].cout;    /* layer_7  */
    m->c_global = m->blocks[16].cout;  /* layer_18 */
    ok &= load_convbn(m, "local_head/descriptor/Conv", "weights", 3, &m->desc1, &c); ok &= (c == HFO_DESC_DIM);
    m->desc2_w = tensor(m, "local_head/descriptor/Conv_1/weights", NULL);
    m->desc2_b = tensor(m, "local_head/descriptor/Conv_1/biases", NULL);
    ok &= load_convbn(m, "local_head/detector/Conv", "weights", 3, &m->det1, &m->det_hidden);
    m->det2_w = tensor(m, "local_head/detector/Conv_1/weights", NULL);
    m->det2_b = tensor(m, "local_head/detector/Conv_1/biases", NULL);
    ok &= load_convbn(m, "global_head/vlad/memberships", "weights", 3, &m->memb, &m->n_clusters);
    m->clusters = tensor(m, "global_head/vlad/clusters", NULL);
    const hfo_entry* e;
    m->fc_w = tensor(m, "global_head/dimensionality_reduction/weights", &e);
    m->fc_b = tensor(m, "global_head/dimensionality_reduction/biases", NULL);
    ok &= (m->desc2_w && m->desc2_b && m->det2_w && m->det2_b && m->clusters && m->fc_w && m->fc_b);
    if (ok) { m->global_dim = (int)e->dims[1]; ok &= ((int)e->dims[0] == m->n_clusters * m->c_global); }
    if (!ok) { hfo_model_free(m); return NULL; }
    return m;
}

static const float* fc_weights_t(const hfo_model* cm) {
    hfo_model* m = (hfo_model*)cm;
#pragma omp critical(hfo_fc_wt)
    if (!m->fc_wt) {
        const int N = m->n_clusters * m->c_global, G = m->global_dim;
        float* t = (float*)malloc(sizeof(float) * (size_t)N * G);
        for (int i = 0; i < N; ++i) for (int j = 0; j < G; ++j) t[(size_t)j * N + i] = m->fc_w[(size_t)i * G + j];
        m->fc_wt = t;
    }
    return m->fc_wt;
}

static void free_convbn(hfo_convbn* c) { free(c->w); free(c->bias); }

void hfo_model_free(hfo_model* m) {
    if (!m) return;
    free_convbn(&m->stem);
    for (int i = 0; i < HFO_NBLOCKS; ++i) { free_convbn(&m->blocks[i].ex); free_convbn(&m->blocks[i].dw); free_convbn(&m->blocks[i].pr); }
    free_convbn(&m->desc1); free_convbn(&m->det1); free_convbn(&m->memb);
    free(m->fc_wt); free(m->blob); free(m);
}

int hfo_model_info(const hfo_model* m, int what) {
    switch (what) { case 0: return m->stem_out; case 1: return m->c_local; case 2: return m->c_global;
                    case 3: return m->n_clusters; case 4: return m->global_dim; default: return -1; }
}

void hfo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ scalar helpers */

/* exp() used by both softmaxes: Cephes expf polynomial with explicit fused operations.
 * |rel err| < 2 ulp on [-87, 0]. */
float hfo_expf(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    return ldexpf(y, (int)n);
}

/* tree256: 256 interleaved partials (fma accumulate), then binary tree 128..1 */
float hfo_sumsq_tree256(const float* x, int n) {
    float p[256];
    for (int i = 0; i < 256; ++i) p[i] = 0.0f;
    for (int i = 0; i < n; ++i) p[i & 255] = fmaf(x[i], x[i], p[i & 255]);
    for (int off = 128; off >= 1; off >>= 1) for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
    return p[0];
}
double hfo_sumsq_tree256_d(const float* x, int n) {
    double p[256];
    for (int i = 0; i < 256; ++i) p[i] = 0.0;
    for (int i = 0; i < n; ++i) p[i & 255] = fma((double)x[i], (double)x[i], p[i & 255]);
    for (int off = 128; off >= 1; off >>= 1) for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
    return p[0];
}
static float sumsq_diff_tree256(const float* a, const float* b, int n) {
    float p[256];
    for (int i = 0; i < 256; ++i) p[i] = 0.0f;
    for (int i = 0; i < n; ++i) { float d = a[i] - b[i]; p[i & 255] = fmaf(d, d, p[i & 255]); }
    for (int off = 128; off >= 1; off >>= 1) for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
    return p[0];
}

/* tf.nn.l2_normalize over a contiguous vector: x * rsqrt(max(sum x^2, 1e-12)) */
static void l2_normalize_vec(float* x, int n) {
    float ss = hfo_sumsq_tree256(x, n);
    float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int i = 0; i < n; ++i) x[i] = x[i] * inv;
}

static inline float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

static void same_pad(int in, int k, int stride, int* out, int* before) {
    int o = (in + stride - 1) / stride;
    int total = (o - 1) * stride + k - in; if (total < 0) total = 0;
    *out = o; *before = total / 2;
}

/* ------------------------------------------------------------------ layers (NHWC, fp32) */

/* act: 0 none, 1 relu6.  Every accumulator starts at bias[c] (folded BN shift, or the layer's bias). */
static void conv2d(const float* x, int h, int w, int cin, const float* wt, int k, int stride, int cout,
                   const float* bias, int act, float* y, int* ho, int* wo) {
    int oh, ow, pt, pl;
    same_pad(h, k, stride, &oh, &pt); same_pad(w, k, stride, &ow, &pl);
    *ho = oh; *wo = ow;
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * cout);
#pragma omp for schedule(static)
        for (int oy = 0; oy < oh; ++oy) {
            for (int ox = 0; ox < ow; ++ox) {
                for (int c = 0; c < cout; ++c) acc[c] = bias[c];
                for (int ky = 0; ky < k; ++ky) {
                    int iy = oy * stride - pt + ky;
                    if (iy < 0 || iy >= h) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        int ix = ox * stride - pl + kx;
                        if (ix < 0 || ix >= w) continue;
                        const float* xp = x + ((size_t)iy * w + ix) * cin;
                        const float* wp = wt + (size_t)(ky * k + kx) * cin * cout;
                        for (int ci = 0; ci < cin; ++ci) {
                            float a = xp[ci];
                            const float* wr = wp + (size_t)ci * cout;
                            for (int c = 0; c < cout; ++c) acc[c] = fmaf(a, wr[c], acc[c]);
                        }
                    }
                }
                float* yp = y + ((size_t)oy * ow + ox) * cout;
                for (int c = 0; c < cout; ++c) {
                    float v = acc[c];
                    if (act) v = relu6f(v);
                    yp[c] = v;
                }
            }
        }
        free(acc);
    }
}

static void depthwise3x3(const float* x, int h, int w, int c, const float* wt, int stride,
                         const float* bias, float* y, int* ho, int* wo) {
    int oh, ow, pt, pl;
    same_pad(h, 3, stride, &oh, &pt); same_pad(w, 3, stride, &ow, &pl);
    *ho = oh; *wo = ow;
#pragma omp parallel for schedule(static)
    for (int oy = 0; oy < oh; ++oy) {
        for (int ox = 0; ox < ow; ++ox) {
            float* yp = y + ((size_t)oy * ow + ox) * c;
            for (int ch = 0; ch < c; ++ch) yp[ch] = bias[ch];
            for (int ky = 0; ky < 3; ++ky) {
                int iy = oy * stride - pt + ky;
                if (iy < 0 || iy >= h) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    int ix = ox * stride - pl + kx;
                    if (ix < 0 || ix >= w) continue;
                    const float* xp = x + ((size_t)iy * w + ix) * c;
                    const float* wp = wt + (size_t)(ky * 3 + kx) * c;
                    for (int ch = 0; ch < c; ++ch) yp[ch] = fmaf(xp[ch], wp[ch], yp[ch]);
                }
            }
            for (int ch = 0; ch < c; ++ch) yp[ch] = relu6f(yp[ch]);
        }
    }
}

static void tap_copy(float** taps, int id, const float* src, size_t n) {
    if (taps && taps[id]) memcpy(taps[id], src, n * sizeof(float));
}

/* conv_blocks.py:163-312: [expand 1x1 + BN + ReLU6] -> dw 3x3 + BN + ReLU6 -> project 1x1 + BN [+ input] */
static float* run_block(const hfo_block* b, float* x, int* h, int* w) {
    int hh = *h, ww = *w, oh, ow, th, tw;
    float* e = x;
    if (b->has_expand) {
        e = (float*)malloc(sizeof(float) * (size_t)hh * ww * b->expand);
        conv2d(x, hh, ww, b->cin, b->ex.w, 1, 1, b->expand, b->ex.bias, 1, e, &th, &tw);
    }
    same_pad(hh, 3, b->stride, &oh, &th); same_pad(ww, 3, b->stride, &ow, &tw);
    float* d = (float*)malloc(sizeof(float) * (size_t)oh * ow * b->expand);
    depthwise3x3(e, hh, ww, b->expand, b->dw.w, b->stride, b->dw.bias, d, &oh, &ow);
    if (e != x) free(e);
    float* y = (float*)malloc(sizeof(float) * (size_t)oh * ow * b->cout);
    conv2d(d, oh, ow, b->expand, b->pr.w, 1, 1, b->cout, b->pr.bias, 0, y, &th, &tw);
    free(d);
    if (b->residual) { size_t n = (size_t)oh * ow * b->cout; for (size_t i = 0; i < n; ++i) y[i] = y[i] + x[i]; }
    free(x);
    *h = oh; *w = ow;
    return y;
}

/* layers.py:10-32: max-pool based NMS; max_pool 'SAME' ignores out-of-image cells */
static void maxpool_same(const float* in, int h, int w, int r, float* out, float* tmp) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= w ? w - 1 : x + r;
            float m = in[(size_t)y * w + x0];
            for (int xx = x0 + 1; xx <= x1; ++xx) { float v = in[(size_t)y * w + xx]; if (v > m) m = v; }
            tmp[(size_t)y * w + x] = m;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= h ? h - 1 : y + r;
        for (int x = 0; x < w; ++x) {
            float m = tmp[(size_t)y0 * w + x];
            for (int yy = y0 + 1; yy <= y1; ++yy) { float v = tmp[(size_t)yy * w + x]; if (v > m) m = v; }
            out[(size_t)y * w + x] = m;
        }
    }
}

void hfo_simple_nms(const float* scores, int h, int w, int radius, int iterations, float* out) {
    size_t n = (size_t)h * w;
    float* pool = (float*)malloc(sizeof(float) * n);
    float* tmp = (float*)malloc(sizeof(float) * n);
    float* mask = (float*)malloc(sizeof(float) * n);    /* max_mask as 0/1 */
    float* supp = (float*)malloc(sizeof(float) * n);
    float* ss = (float*)malloc(sizeof(float) * n);
    maxpool_same(scores, h, w, radius, pool, tmp);
    for (size_t i = 0; i < n; ++i) mask[i] = (scores[i] == pool[i]) ? 1.0f : 0.0f;
    for (int it = 0; it < iterations - 1; ++it) {
        maxpool_same(mask, h, w, radius, supp, tmp);                 /* supp_mask = cast(max_pool(to_float(max_mask)), bool) */
        for (size_t i = 0; i < n; ++i) ss[i] = (supp[i] != 0.0f) ? 0.0f : scores[i];
        maxpool_same(ss, h, w, radius, pool, tmp);
        for (size_t i = 0; i < n; ++i) {
            int new_max = (ss[i] == pool[i]);
            if (new_max && supp[i] == 0.0f) mask[i] = 1.0f;
        }
    }
    for (size_t i = 0; i < n; ++i) out[i] = (mask[i] != 0.0f) ? scores[i] : 0.0f;
    free(pool); free(tmp); free(mask); free(supp); free(ss);
}

/* layers.py:57-109 NetVLAD + dimensionality reduction on the layer_18 feature map */
static void global_head(const hfo_model* m, const float* feat, int h, int w, float* out, float** taps) {
    const int K = m->n_clusters, D = m->c_global, P = h * w;
    int th, tw;
    float* mem = (float*)malloc(sizeof(float) * (size_t)P * K);
    conv2d(feat, h, w, D, m->memb.w, 1, 1, K, m->memb.bias, 0, mem, &th, &tw);
    for (int p = 0; p < P; ++p) {                                   /* softmax over K */
        float* r = mem + (size_t)p * K;
        float mx = r[0]; for (int k = 1; k < K; ++k) if (r[k] > mx) mx = r[k];
        float s = 0.0f;
        for (int k = 0; k < K; ++k) { r[k] = hfo_expf(r[k] - mx); s = s + r[k]; }
        for (int k = 0; k < K; ++k) r[k] = r[k] / s;
    }
    tap_copy(taps, HFO_TAP_MEMBERSHIPS, mem, (size_t)P * K);
    float* v = (float*)malloc(sizeof(float) * (size_t)K * D);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k)
        for (int d = 0; d < D; ++d) {                               /* sum_hw (c - f) * m   (layers.py:82-87).  The reference leaves the
                                                                       order of this sum to TensorFlow's reduce_sum; canonical here: the
                                                                       pixels in HFO_VLAD_PARTS contiguous ranges of ceil(P / parts), one
                                                                       chain in pixel order from 0 per range, the partial sums added as a
                                                                       balanced binary tree */
            const float c = m->clusters[(size_t)k * D + d];
            const int pp = (P + HFO_VLAD_PARTS - 1) / HFO_VLAD_PARTS;
            float part[HFO_VLAD_PARTS];
            for (int s = 0; s < HFO_VLAD_PARTS; ++s) {
                const int p0 = s * pp < P ? s * pp : P, p1 = p0 + pp < P ? p0 + pp : P;
                float acc = 0.0f;
                for (int p = p0; p < p1; ++p) { float r = c - feat[(size_t)p * D + d]; float t = r * mem[(size_t)p * K + k]; acc = acc + t; }
                part[s] = acc;
            }
            for (int n = HFO_VLAD_PARTS; n > 1; n >>= 1)
                for (int s = 0; s < n / 2; ++s) part[s] = part[2 * s] + part[2 * s + 1];
            v[(size_t)k * D + d] = part[0];
        }
    for (int d = 0; d < D; ++d) {                                   /* l2_normalize(axis=1): over clusters (layers.py:89) */
        float ss = 0.0f;
        for (int k = 0; k < K; ++k) ss = fmaf(v[(size_t)k * D + d], v[(size_t)k * D + d], ss);
        float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        for (int k = 0; k < K; ++k) v[(size_t)k * D + d] = v[(size_t)k * D + d] * inv;
    }
    l2_normalize_vec(v, K * D);                                     /* layers.py:92 (flatten is K-major) */
    tap_copy(taps, HFO_TAP_VLAD, v, (size_t)K * D);
    l2_normalize_vec(v, K * D);                                     /* layers.py:97 */
    const int G = m->global_dim, N = K * D;                         /* layers.py:99-107: x @ W + b.  The reference leaves the order of
                                                                       this 7680-term sum to Eigen; the canonical order here is the one a
                                                                       GEMM split 16 ways along the inputs produces: HFO_FC_PARTS partial
                                                                       fma chains from 0 (part p: the inputs of groups-of-16
                                                                       [p * gp, (p + 1) * gp), gp = ceil(N / 16 / 16), ascending), added
                                                                       as a balanced binary tree, then + b[j] */
    const float* wt = fc_weights_t(m);
    const int gp = ((N + 15) / 16 + HFO_FC_PARTS - 1) / HFO_FC_PARTS;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < G; ++j) {
        const float* wr = wt + (size_t)j * N;
        float part[HFO_FC_PARTS];
        for (int p = 0; p < HFO_FC_PARTS; ++p) {
            const int i0 = p * gp * 16 < N ? p * gp * 16 : N, i1 = (p + 1) * gp * 16 < N ? (p + 1) * gp * 16 : N;
            float acc = 0.0f;
            for (int i = i0; i < i1; ++i) acc = fmaf(v[i], wr[i], acc);
            part[p] = acc;
        }
        for (int n = HFO_FC_PARTS; n > 1; n >>= 1)
            for (int p = 0; p < n / 2; ++p) part[p] = part[2 * p] + part[2 * p + 1];
        out[j] = part[0] + m->fc_b[j];
    }
    l2_normalize_vec(out, G);                                       /* layers.py:108 */
    free(mem); free(v);
}

static float* run_backbone_tail(const hfo_model* m, float* x, int* h, int* w, int first_block, float** taps) {
    for (int i = first_block; i < HFO_NBLOCKS; ++i) {
        x = run_block(&m->blocks[i], x, h, w);
        tap_copy(taps, HFO_TAP_BLOCK0 + i, x, (size_t)(*h) * (*w) * m->blocks[i].cout);
    }
    return x;
}

int hfo_run_global(const hfo_model* m, const float* intermediate, int hd, int wd, float* global_desc, float** taps) {
    size_t n = (size_t)hd * wd * m->c_local;
    float* x = (float*)malloc(sizeof(float) * n);
    memcpy(x, intermediate, sizeof(float) * n);
    int h = hd, w = wd;
    x = run_backbone_tail(m, x, &h, &w, 6, taps);                  /* layer_8 .. layer_18 */
    global_head(m, x, h, w, global_desc, taps);
    free(x);
    return 1;
}

int hfo_run_local(const hfo_model* m, const uint8_t* img, int h, int w, int stride,
                  float* scores_nms, float* desc_map, float* intermediate, float* global_desc, float** taps) {
    const int hc = h / 8 * 8, wc = w / 8 * 8;                       /* hf_net.py:188-190 */
    if (hc < 8 || wc < 8) return 0;
    float* x = (float*)malloc(sizeof(float) * (size_t)hc * wc);
    for (int y = 0; y < hc; ++y)                                     /* Mat2Tensor + image_normalization */
        for (int xx = 0; xx < wc; ++xx) x[(size_t)y * wc + xx] = ((float)img[(size_t)y * stride + xx] - 128.0f) / 128.0f;
    int ch, cw;
    float* s = (float*)malloc(sizeof(float) * (size_t)((hc + 1) / 2) * ((wc + 1) / 2) * m->stem_out);
    conv2d(x, hc, wc, 1, m->stem.w, 3, 2, m->stem_out, m->stem.bias, 1, s, &ch, &cw);
    free(x);
    tap_copy(taps, HFO_TAP_STEM, s, (size_t)ch * cw * m->stem_out);
    x = s;
    for (int i = 0; i < 6; ++i) {                                    /* layer_2 .. layer_7 */
        x = run_block(&m->blocks[i], x, &ch, &cw);
        tap_copy(taps, HFO_TAP_BLOCK0 + i, x, (size_t)ch * cw * m->blocks[i].cout);
    }
    const int hd = ch, wd = cw, cl = m->c_local;
    const size_t P = (size_t)hd * wd;
    if (intermediate) memcpy(intermediate, x, sizeof(float) * P * cl);
    int th, tw;
    if (desc_map || (taps && (taps[HFO_TAP_DESC_HIDDEN] || taps[HFO_TAP_DESC_RAW]))) {      /* hf_net.py:74-80 */
        float* t1 = (float*)malloc(sizeof(float) * P * HFO_DESC_DIM);
        float* t2 = (float*)malloc(sizeof(float) * P * HFO_DESC_DIM);
        conv2d(x, hd, wd, cl, m->desc1.w, 3, 1, HFO_DESC_DIM, m->desc1.bias, 1, t1, &th, &tw);
        tap_copy(taps, HFO_TAP_DESC_HIDDEN, t1, P * HFO_DESC_DIM);
        conv2d(t1, hd, wd, HFO_DESC_DIM, m->desc2_w, 1, 1, HFO_DESC_DIM, m->desc2_b, 0, t2, &th, &tw);
        tap_copy(taps, HFO_TAP_DESC_RAW, t2, P * HFO_DESC_DIM);
        for (size_t p = 0; p < P; ++p) l2_normalize_vec(t2 + p * HFO_DESC_DIM, HFO_DESC_DIM);
        if (desc_map) memcpy(desc_map, t2, sizeof(float) * P * HFO_DESC_DIM);
        free(t1); free(t2);
    }
    if (scores_nms || (taps && (taps[HFO_TAP_DET_HIDDEN] || taps[HFO_TAP_LOGITS] || taps[HFO_TAP_SCORES_DENSE]))) {  /* hf_net.py:82-93 */
        float* t1 = (float*)malloc(sizeof(float) * P * m->det_hidden);
        float* lg = (float*)malloc(sizeof(float) * P * HFO_DET_CH);
        conv2d(x, hd, wd, cl, m->det1.w, 3, 1, m->det_hidden, m->det1.bias, 1, t1, &th, &tw);
        tap_copy(taps, HFO_TAP_DET_HIDDEN, t1, P * m->det_hidden);
        conv2d(t1, hd, wd, m->det_hidden, m->det2_w, 1, 1, HFO_DET_CH, m->det2_b, 0, lg, &th, &tw);
        tap_copy(taps, HFO_TAP_LOGITS, lg, P * HFO_DET_CH);
        float* dense = (float*)malloc(sizeof(float) * (size_t)hc * wc);
        for (int cy = 0; cy < hd; ++cy)
            for (int cx = 0; cx < wd; ++cx) {
                const float* r = lg + ((size_t)cy * wd + cx) * HFO_DET_CH;
                float mx = r[0]; for (int k = 1; k < HFO_DET_CH; ++k) if (r[k] > mx) mx = r[k];
                float e[HFO_DET_CH], sum = 0.0f;
                for (int k = 0; k < HFO_DET_CH; ++k) { e[k] = hfo_expf(r[k] - mx); sum = sum + e[k]; }
                for (int k = 0; k < HFO_GRID * HFO_GRID; ++k)        /* drop dustbin, depth_to_space(8) */
                    dense[(size_t)(cy * HFO_GRID + k / HFO_GRID) * wc + cx * HFO_GRID + k % HFO_GRID] = e[k] / sum;
            }
        tap_copy(taps, HFO_TAP_SCORES_DENSE, dense, (size_t)hc * wc);
        if (scores_nms) hfo_simple_nms(dense, hc, wc, 4, 2, scores_nms);  /* export_model.py:35,37 */
        free(t1); free(lg); free(dense);
    }
    if (global_desc) {
        x = run_backbone_tail(m, x, &ch, &cw, 6, taps);
        global_head(m, x, ch, cw, global_desc, taps);
    }
    free(x);
    return 1;
}

/* ------------------------------------------------------------------ post-processing */

/* BaseModel.cc:491-562 (tensorflow.contrib.resampler), same expression order */
void hfo_resampler(const float* data, const float* warp, float* output, int batch, int dh, int dw, int channels, int npoints) {
    for (int b = 0; b < batch; ++b)
        for (int s = 0; s < npoints; ++s) {
            const float x = warp[((size_t)b * npoints + s) * 2], y = warp[((size_t)b * npoints + s) * 2 + 1];
            float* o = output + ((size_t)b * npoints + s) * channels;
            const float* d = data + (size_t)b * dh * dw * channels;
            if (x > -1.0f && y > -1.0f && x < (float)dw && y < (float)dh) {
                const int fx = (int)floorf(x), fy = (int)floorf(y), cx = fx + 1, cy = fy + 1;
                const float dx = (float)cx - x, dy = (float)cy - y;
#define HFO_PT(xx, yy, c) (((xx) >= 0 && (yy) >= 0 && (xx) <= dw - 1 && (yy) <= dh - 1) ? d[(size_t)channels * ((size_t)(yy) * dw + (xx)) + (c)] : 0.0f)
                for (int c = 0; c < channels; ++c) {
                    const float a = dx * dy * HFO_PT(fx, fy, c);
                    const float bq = (1.0f - dx) * (1.0f - dy) * HFO_PT(cx, cy, c);
                    const float cq = dx * (1.0f - dy) * HFO_PT(fx, cy, c);
                    const float dq = (1.0f - dx) * dy * HFO_PT(cx, fy, c);
                    o[c] = a + bq + cq + dq;
                }
#undef HFO_PT
            } else {
                for (int c = 0; c < channels; ++c) o[c] = 0.0f;
            }
        }
}

/* BaseModel.cc:564-603.  Output order: the reference iterates an unordered_set (unspecified);
 * canonical order here = input order of the survivors. */
int hfo_nms_points(const hfo_keypoint* in, int n, int width, int height, int radius, hfo_keypoint* out) {
    int* grid = (int*)malloc(sizeof(int) * (size_t)width * height);
    char* alive = (char*)malloc((size_t)n);
    for (size_t i = 0; i < (size_t)width * height; ++i) grid[i] = -1;
    for (int i = 0; i < n; ++i) { grid[(size_t)((int)in[i].y) * width + (int)in[i].x] = i; alive[i] = 1; }
    for (int i = 0; i < n; ++i) {
        const int px = (int)in[i].x, py = (int)in[i].y;
        int done = 0;
        for (int dx = -radius; dx <= radius && !done; ++dx)
            for (int dy = -radius; dy <= radius && !done; ++dy) {
                const int x = px + dx, y = py + dy;
                if (x < 0 || y < 0 || x >= width || y >= height) continue;
                const int j = grid[(size_t)y * width + x];
                if (j < 0) continue;
                const int self = grid[(size_t)py * width + px];
                if (self < 0) { done = 1; break; }
                if (in[self].response < in[j].response) { alive[self] = 0; grid[(size_t)py * width + px] = -1; done = 1; }
            }
    }
    int m = 0;
    for (int i = 0; i < n; ++i) if (alive[i]) out[m++] = in[i];
    free(grid); free(alive);
    return m;
}

typedef struct { float score; int32_t idx; } hfo_cand;
static int cand_cmp(const void* a, const void* b) {
    const hfo_cand* p = (const hfo_cand*)a; const hfo_cand* q = (const hfo_cand*)b;
    if (p->score > q->score) return -1;
    if (p->score < q->score) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx);
}

/* HFNetTFModelV2.cc:122-151.  Scan is column-major, test is score >= threshold.  When more than
 * kmax candidates pass, the reference keeps the kmax largest responses via std::nth_element
 * (order and tie choice unspecified); canonical rule: sort by (response desc, col*H+row asc),
 * keep the first kmax, emit in that order.  With <= kmax candidates the scan order is kept. */
int hfo_select_keypoints(const float* scores_nms, int h, int w, float threshold, int kmax, hfo_keypoint* kps) {
    size_t cap = 1024, n = 0;
    hfo_cand* c = (hfo_cand*)malloc(sizeof(hfo_cand) * cap);
    for (int col = 0; col < w; ++col)
        for (int row = 0; row < h; ++row) {
            float s = scores_nms[(size_t)row * w + col];
            if (s >= threshold) {
                if (n == cap) { cap *= 2; c = (hfo_cand*)realloc(c, sizeof(hfo_cand) * cap); }
                c[n].score = s; c[n].idx = col * h + row; ++n;
            }
        }
    if (n > (size_t)kmax) { qsort(c, n, sizeof(hfo_cand), cand_cmp); n = (size_t)kmax; }
    for (size_t i = 0; i < n; ++i) {
        kps[i].x = (float)(c[i].idx / h); kps[i].y = (float)(c[i].idx % h);
        kps[i].response = c[i].score; kps[i].octave = 0;
    }
    free(c);
    return (int)n;
}

/* HFNetTFModelV2.cc:119-120,153-167: warp, Resampler, cv::normalize(row, row) (NORM_L2: norm
 * accumulated in double, row *= (float)(1/norm)) */
void hfo_sample_descriptors(const float* desc_map, int hd, int wd, int channels, const hfo_keypoint* kps, int n,
                            int h, int w, float* out) {
    if (n <= 0) return;
    const float sw = ((float)wd - 1.f) / (float)((float)w - 1.f);
    const float sh = ((float)hd - 1.f) / (float)((float)h - 1.f);
    float* warp = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    for (int i = 0; i < n; ++i) { warp[2 * i] = sw * kps[i].x; warp[2 * i + 1] = sh * kps[i].y; }
    hfo_resampler(desc_map, warp, out, 1, hd, wd, channels, n);
    for (int i = 0; i < n; ++i) {
        float* r = out + (size_t)i * channels;
        double nrm = sqrt(hfo_sumsq_tree256_d(r, channels));
        float sc = (float)(nrm > DBL_EPSILON ? 1.0 / nrm : 0.0);
        for (int c = 0; c < channels; ++c) r[c] = r[c] * sc;
    }
    free(warp);
}

int hfo_detect(const hfo_model* m, int mode, const uint8_t* img, int h, int w, int stride, int nkeypoints, float threshold,
               hfo_keypoint* kps, float* local_desc, float* global_or_intermediate, int* n_out) {
    if (mode != HFO_IMAGE_TO_LOCAL_AND_GLOBAL && mode != HFO_IMAGE_TO_LOCAL && mode != HFO_IMAGE_TO_LOCAL_AND_INTERMEDIATE) return 0;
    const int hc = h / 8 * 8, wc = w / 8 * 8, hd = hc / 8, wd = wc / 8;
    float* scores = (float*)malloc(sizeof(float) * (size_t)hc * wc);
    float* dmap = (float*)malloc(sizeof(float) * (size_t)hd * wd * HFO_DESC_DIM);
    int ok = hfo_run_local(m, img, h, w, stride, scores, dmap,
                           mode == HFO_IMAGE_TO_LOCAL_AND_INTERMEDIATE ? global_or_intermediate : NULL,
                           mode == HFO_IMAGE_TO_LOCAL_AND_GLOBAL ? global_or_intermediate : NULL, NULL);
    if (ok) {
        int n = hfo_select_keypoints(scores, hc, wc, threshold, nkeypoints, kps);
        hfo_sample_descriptors(dmap, hd, wd, HFO_DESC_DIM, kps, n, hc, wc, local_desc);
        *n_out = n;
    }
    free(scores); free(dmap);
    return ok;
}

int hfo_detect_global(const hfo_model* m, int mode, const float* intermediate, int hd, int wd, float* global_desc) {
    if (mode != HFO_INTERMEDIATE_TO_GLOBAL) return 0;
    return hfo_run_global(m, intermediate, hd, wd, global_desc, NULL);
}

/* ------------------------------------------------------------------ pyramid + extractor */

static int cv_round_f(float v) { return (int)lrintf(v); }
static short sat_short(float v) { int i = cv_round_f(v); return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i); }

/* OpenCV 4.2 cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 (imgproc/src/resize.cpp:
 * resizeGeneric_ + HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>) */
void hfo_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
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
#pragma omp parallel
    {
        int* r0 = (int*)malloc(sizeof(int) * dw);
        int* r1 = (int*)malloc(sizeof(int) * dw);
#pragma omp for schedule(static)
        for (int dy = 0; dy < dh; ++dy) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = (int)floorf(fy);
            fy -= sy;
            const short b0 = sat_short((1.f - fy) * 2048.f), b1 = sat_short(fy * 2048.f);
            int y0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
            int y1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
            const uint8_t* s0 = src + (size_t)y0 * sstride;
            const uint8_t* s1 = src + (size_t)y1 * sstride;
            for (int dx = 0; dx < dw; ++dx) {
                int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
                int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
                r0[dx] = s0[sx] * a0 + s0[sx1] * a1;
                r1[dx] = s1[sx] * a0 + s1[sx1] * a1;
            }
            uint8_t* d = dst + (size_t)dy * dstride;
            for (int dx = 0; dx < dw; ++dx) {
                int v = (((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2;
                d[dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
        free(r0); free(r1);
    }
    free(xofs); free(ialpha);
}

/* HFextractor.cc:82-119,159-166 */
void hfo_extractor_tables(int nfeatures, int nlevels, float scale_factor, int width, int height,
                          float* scale_factors, int* features_per_level, int* level_w, int* level_h) {
    scale_factors[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) scale_factors[i] = scale_factors[i - 1] * scale_factor;
    for (int i = 0; i < nlevels; ++i) {
        float inv = 1.0f / scale_factors[i];
        level_w[i] = i == 0 ? width : cv_round_f((float)width * inv);
        level_h[i] = i == 0 ? height : cv_round_f((float)height * inv);
    }
    if (nlevels == 1) { features_per_level[0] = nfeatures; return; }
    float factor = 1.0f / scale_factor;
    float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        features_per_level[l] = cv_round_f(desired);
        sum += features_per_level[l];
        desired *= factor;
    }
    features_per_level[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

/* HFextractor::operator() (HFextractor.cc:142-284): level 0 runs mode LocalAndGlobal, the others
 * Local; per-level keypoints get octave = level and pt *= scaleFactor^level; descriptors are
 * concatenated level by level.  Returns the number of keypoints, -1 on bad input. */
int hfo_extract(const hfo_model* m, const uint8_t* img, int h, int w, int stride, int nfeatures, float threshold,
                int nlevels, float scale_factor, hfo_keypoint* kps, float* local_desc, float* global_desc, int* n_per_level) {
    if (!img || h <= 0 || w <= 0 || nlevels < 1 || nlevels > 16) return -1;
    float sf[16]; int fpl[16], lw[16], lh[16];
    hfo_extractor_tables(nfeatures, nlevels, scale_factor, w, h, sf, fpl, lw, lh);
    uint8_t* pyr[16]; int pstride[16];
    pyr[0] = (uint8_t*)img; pstride[0] = stride;
    for (int l = 1; l < nlevels; ++l) {
        pyr[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]); pstride[l] = lw[l];
        hfo_resize_linear_u8(pyr[l - 1], lw[l - 1], lh[l - 1], pstride[l - 1], pyr[l], lw[l], lh[l], lw[l]);
    }
    int total = 0;
    for (int l = 0; l < nlevels; ++l) {
        int n = 0;
        int ok = hfo_detect(m, l == 0 ? HFO_IMAGE_TO_LOCAL_AND_GLOBAL : HFO_IMAGE_TO_LOCAL, pyr[l], lh[l], lw[l], pstride[l],
                            fpl[l], threshold, kps + total, local_desc + (size_t)total * HFO_DESC_DIM,
                            l == 0 ? global_desc : NULL, &n);
        if (!ok) n = 0;
        for (int i = 0; i < n; ++i) { kps[total + i].octave = l; kps[total + i].x *= sf[l]; kps[total + i].y *= sf[l]; }
        if (n_per_level) n_per_level[l] = n;
        total += n;
    }
    for (int l = 1; l < nlevels; ++l) free(pyr[l]);
    return total;
}

/* ------------------------------------------------------------------ matching */

/* Matcher.cc:1893-1900: (des1 - des2).norm() */
float hfo_descriptor_distance(const float* a, const float* b, int dim) { return sqrtf(sumsq_diff_tree256(a, b, dim)); }

/* OpenCV batchDistance L2 for CV_32F: sqrt(normL2Sqr(a, b)); normL2Sqr generic template,
 * unrolled by 4:  s += v0*v0 + v1*v1 + v2*v2 + v3*v3  (core/include/opencv2/core/base.hpp) */
static float cv_l2(const float* a, const float* b, int n) {
    float s = 0; int i = 0;
    for (; i <= n - 4; i += 4) {
        float v0 = a[i] - b[i], v1 = a[i + 1] - b[i + 1], v2 = a[i + 2] - b[i + 2], v3 = a[i + 3] - b[i + 3];
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; ++i) { float v = a[i] - b[i]; s += v * v; }
    return sqrtf(s);
}

/* cv::BFMatcher(NORM_L2, crossCheck=true)::match == batchDistance(..., K=1, crosscheck=true)
 * (core/src/batch_distance.cpp): for every train row t take its nearest query q*(t) (first
 * minimum); query q is matched to the nearest of the train rows that chose it (first minimum in
 * train order); queries chosen by no train row stay unmatched. */
void hfo_bfmatch_l2_crosscheck(const float* q, int nq, const float* t, int nt, int dim, int32_t* train_idx, float* dist) {
    for (int i = 0; i < nq; ++i) { train_idx[i] = -1; dist[i] = FLT_MAX; }
    int* best_q = (int*)malloc(sizeof(int) * (nt > 0 ? nt : 1));
    float* best_d = (float*)malloc(sizeof(float) * (nt > 0 ? nt : 1));
#pragma omp parallel for schedule(static)
    for (int j = 0; j < nt; ++j) {
        float bd = FLT_MAX; int bi = -1;
        for (int i = 0; i < nq; ++i) { float d = cv_l2(t + (size_t)j * dim, q + (size_t)i * dim, dim); if (d < bd) { bd = d; bi = i; } }
        best_q[j] = bi; best_d[j] = bd;
    }
    for (int j = 0; j < nt; ++j) {
        int i = best_q[j];
        if (i >= 0 && best_d[j] < dist[i]) { dist[i] = best_d[j]; train_idx[i] = j; }
    }
    free(best_q); free(best_d);
}

/* Matcher.cc:229-260 / 574-618 after the MapPoint gather: keep matches with distance < TH_LOW */
int hfo_search_by_bow(const float* q, int nq, const float* t, int nt, int dim, float th_low, int32_t* match_q2t, float* dist) {
    hfo_bfmatch_l2_crosscheck(q, nq, t, nt, dim, match_q2t, dist);
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (match_q2t[i] >= 0 && dist[i] < th_low) ++n; else match_q2t[i] = -1;
    }
    return n;
}

/* Matcher.cc:845-889: S = D1 * D2^T, threshold = 1 - 0.5*TH_HIGH^2, row arg-max with strict >,
 * column cross-check with the same rule.  (The epipolar tests that follow are CPU geometry.) */
int hfo_search_for_triangulation(const float* d1, int n1, const float* d2, int n2, int dim, float th_high,
                                 int32_t* match12, float* sim_out) {
    float* sim = sim_out ? sim_out : (float*)malloc(sizeof(float) * (size_t)(n1 > 0 ? n1 : 1) * (n2 > 0 ? n2 : 1));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < dim; ++k) acc = fmaf(d1[(size_t)i * dim + k], d2[(size_t)j * dim + k], acc);
            sim[(size_t)i * n2 + j] = acc;
        }
    const float threshold = (float)(-0.5 * th_high * th_high + 1);
    int n = 0;
    for (int i = 0; i < n1; ++i) {
        float best = threshold; int bj = -1;
        for (int j = 0; j < n2; ++j) { float d = sim[(size_t)i * n2 + j]; if (d > best) { best = d; bj = j; } }
        match12[i] = -1;
        if (bj != -1) {
            float cb = threshold; int ci = -1;
            for (int r = 0; r < n1; ++r) { float d = sim[(size_t)r * n2 + bj]; if (d > cb) { cb = d; ci = r; } }
            if (ci == i) { match12[i] = bj; ++n; }
        }
    }
    if (!sim_out) free(sim);
    return n;
}

/* The inner loop the windowed matchers share (SearchByProjection x5, SearchForInitialization, Fuse x2, SearchBySim3:
 * Matcher.cc:74-110, 126-160, 313-341, 1652-1690, ...): for one query descriptor, walk its candidate list (built on the CPU
 * from the frame grid / map geometry) in order, DescriptorDistance to every candidate, keep best and second best with their
 * pyramid levels:
 *     if (dist < bestDist) { bestDist2 = bestDist; bestLevel2 = bestLevel; bestDist = dist; bestLevel = level; bestIdx = idx; }
 *     else if (dist < bestDist2) { bestLevel2 = level; bestDist2 = dist; }
 * The thresholds / ratio test / ownership rules that follow differ per caller and stay on the CPU side.
 * cand_index[cand_offsets[i] .. cand_offsets[i+1]) are the train rows of query i; empty list -> idx -1, distances FLT_MAX. */
void hfo_match_candidates(const float* query, int nq, const float* train, const int32_t* train_level, int dim,
                          const int32_t* cand_offsets, const int32_t* cand_index,
                          int32_t* best_idx, float* best_dist, int32_t* best_level, float* second_dist, int32_t* second_level) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < nq; ++i) {
        float bd = FLT_MAX, bd2 = FLT_MAX;
        int bl = -1, bl2 = -1, bi = -1;
        for (int c = cand_offsets[i]; c < cand_offsets[i + 1]; ++c) {
            const int idx = cand_index[c];
            const float dist = hfo_descriptor_distance(query + (size_t)i * dim, train + (size_t)idx * dim, dim);
            const int level = train_level ? train_level[idx] : 0;
            if (dist < bd) { bd2 = bd; bl2 = bl; bd = dist; bl = level; bi = idx; }
            else if (dist < bd2) { bl2 = level; bd2 = dist; }
        }
        best_idx[i] = bi; best_dist[i] = bd; best_level[i] = bl; second_dist[i] = bd2; second_level[i] = bl2;
    }
}

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:366-400) for many map points at once: set s holds the descriptors of
 * its observations, desc rows [set_offsets[s], set_offsets[s+1]); all pairwise DescriptorDistance (diagonal 0), per row the
 * median = element (int)(0.5 * (N - 1)) of the sorted row, the row with the smallest median wins (first one on ties:
 * strict <).  best[s] = row index inside the set, -1 for an empty set. */
static int cmp_float(const void* a, const void* b) { const float x = *(const float*)a, y = *(const float*)b; return (x > y) - (x < y); }
void hfo_distinctive_descriptors(const float* desc, const int32_t* set_offsets, int n_sets, int dim, int32_t* best) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int s = 0; s < n_sets; ++s) {
        const int n = set_offsets[s + 1] - set_offsets[s];
        if (n <= 0) { best[s] = -1; continue; }
        const float* d = desc + (size_t)set_offsets[s] * dim;
        float* dist = (float*)malloc(sizeof(float) * (size_t)n * n);
        float* row = (float*)malloc(sizeof(float) * n);
        for (int i = 0; i < n; ++i) {
            dist[(size_t)i * n + i] = 0.0f;
            for (int j = i + 1; j < n; ++j) {
                const float v = hfo_descriptor_distance(d + (size_t)i * dim, d + (size_t)j * dim, dim);
                dist[(size_t)i * n + j] = v; dist[(size_t)j * n + i] = v;
            }
        }
        float best_median = FLT_MAX;
        int bi = 0;
        for (int i = 0; i < n; ++i) {
            memcpy(row, dist + (size_t)i * n, sizeof(float) * n);
            qsort(row, n, sizeof(float), cmp_float);
            const float median = row[(int)(0.5 * (n - 1))];
            if (median < best_median) { best_median = median; bi = i; }
        }
        best[s] = bi;
        free(dist); free(row);
    }
}

/* ------------------------------------------------------------------ place recognition */

/* KeyFrameDatabase.cc:86-96 / 178-188: score = max(0, 1 - ||q - d||) for every keyframe */
void hfo_db_scores(const float* query, const float* db, int n, int dim, float* scores) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float d = sqrtf(sumsq_diff_tree256(query, db + (size_t)i * dim, dim));
        float s = 1 - d;
        scores[i] = s > 0.f ? s : 0.f;
    }
}

/* KeyFrameDatabase.cc:94-104 (mode 0) / 188-197 (mode 1) */
int hfo_db_candidates(const float* scores, int n, int mode, int32_t* idx, float* best_out) {
    float best = 0;
    for (int i = 0; i < n; ++i) best = scores[i] > best ? scores[i] : best;
    float min_score = best * 0.8f;
    if (mode == 1) min_score = 0.5f > min_score ? 0.5f : min_score;
    int m = 0;
    for (int i = 0; i < n; ++i) if (scores[i] > min_score) idx[m++] = i;
    if (best_out) *best_out = best;
    return m;
}
