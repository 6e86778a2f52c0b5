#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small input/output vectors of the hot path.

The reference holds no golden vectors and none of its code for this path can run here (no
TensorFlow / OpenCV / Eigen / weights -- SURVEY.md 8c), so these fixtures are produced by the
build's own CPU oracle (oracle/hfnet_oracle.c) in this container, after it was cross-checked
against the independent PyTorch restatement (tests/test_oracle_vs_torch.py).  They pin the oracle
against regressions and give the GPU tests a data-only target that needs no oracle at run time.

    python tests/golden/make_golden.py          (rewrites the fixtures in place)
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def weights_digest(w) -> str:
    h = hashlib.sha256()
    for k, v in w.items():
        h.update(k.encode()); h.update(np.ascontiguousarray(v).tobytes()[:4096])
    return h.hexdigest()


def main():
    from conftest import synth_image
    from hfnet_slam_amd import weights
    from oracle import oracle as O
    O.build()
    w = weights.synthetic_weights(7)
    path = os.path.join(tempfile.gettempdir(), "golden_seed7.hfw")
    weights.save(path, w)
    m = O.Model(path)
    digest = weights_digest(w)

    # --- extractor: 2 levels on a 96x72 frame, 60 keypoints, incl. global descriptor
    img = synth_image(72, 96, 4242)
    n, kps, desc, g, npl = m.extract(img, 60, 0.01, 2, 1.2)
    r = m.run_local(img, want_global=True)
    np.savez_compressed(os.path.join(HERE, "extract_96x72.npz"), weights_digest=digest, image=img, n=n, kps=kps, desc=desc,
                        glob=g, n_per_level=npl, scores_nms_nonzero=np.argwhere(r["scores_nms"] > 0).astype(np.int16),
                        scores_nms_values=r["scores_nms"][r["scores_nms"] > 0])

    # --- matchers + database on seeded unit rows
    rng = np.random.default_rng(11)
    a = rng.standard_normal((96, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
    perm = np.random.default_rng(12).permutation(96)[:80]
    b = a[perm] + 0.03 * rng.standard_normal((80, 256)).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
    a = a.astype(np.float32); b = b.astype(np.float32)
    nb, mb, db_ = O.search_by_bow(a, b, 0.6)
    nt, mt = O.search_for_triangulation(a, b, 0.75)
    dbm = rng.standard_normal((40, 4096)).astype(np.float32); dbm /= np.linalg.norm(dbm, axis=1, keepdims=True)
    q = dbm[17] + 0.004 * rng.standard_normal(4096).astype(np.float32); q = (q / np.linalg.norm(q)).astype(np.float32)
    dbm = dbm.astype(np.float32)
    sc = O.db_scores(q, dbm)
    c0, best = O.db_candidates(sc, 0)
    np.savez_compressed(os.path.join(HERE, "match_db.npz"), seed_a=11, perm=perm.astype(np.int32), bow_n=nb, bow_match=mb, bow_dist=db_,
                        tri_n=nt, tri_match=mt, db_scores=sc, db_cand=c0, db_best=np.float32(best), a_row0=a[0], b_row0=b[0], q_head=q[:16])
    # --- pyramid: one cv::resize chain level
    src = synth_image(60, 90, 99, "natural")
    np.savez_compressed(os.path.join(HERE, "resize_90x60_to_75x50.npz"), src=src, dst=O.resize_linear_u8(src, 75, 50))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
