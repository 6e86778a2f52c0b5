"""Committed golden vectors (tests/golden/make_golden.py) vs the oracle (CPU) and vs the HIP path (GPU)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
sys.path.insert(0, G)


def _inputs():
    rng = np.random.default_rng(11)
    a = rng.standard_normal((96, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
    perm = np.random.default_rng(12).permutation(96)[:80]
    b = a[perm] + 0.03 * rng.standard_normal((80, 256)).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
    dbm = rng.standard_normal((40, 4096)).astype(np.float32); dbm /= np.linalg.norm(dbm, axis=1, keepdims=True)
    q = dbm[17] + 0.004 * rng.standard_normal(4096).astype(np.float32); q = (q / np.linalg.norm(q)).astype(np.float32)
    return a.astype(np.float32), b.astype(np.float32), dbm.astype(np.float32), q


def _check_weights(f, weights_path):
    from hfnet_slam_amd import weights
    from make_golden import weights_digest
    if str(f["weights_digest"]) != weights_digest(weights.load(weights_path)):
        pytest.skip("numpy's RNG produced different synthetic weights than when the fixture was made")


def test_oracle_reproduces_extract_fixture(oracle_model, weights_path):
    f = np.load(os.path.join(G, "extract_96x72.npz"))
    _check_weights(f, weights_path)
    n, kps, desc, g, npl = oracle_model.extract(f["image"], 60, 0.01, 2, 1.2)
    assert n == int(f["n"]) and np.array_equal(npl, f["n_per_level"])
    assert np.array_equal(kps, f["kps"]) and np.array_equal(desc, f["desc"]) and np.array_equal(g, f["glob"])
    r = oracle_model.run_local(f["image"])
    nz = np.argwhere(r["scores_nms"] > 0).astype(np.int16)
    assert np.array_equal(nz, f["scores_nms_nonzero"]) and np.array_equal(r["scores_nms"][r["scores_nms"] > 0], f["scores_nms_values"])


def test_oracle_reproduces_match_db_and_resize_fixtures():
    from oracle import oracle as O
    f = np.load(os.path.join(G, "match_db.npz"))
    a, b, dbm, q = _inputs()
    assert np.array_equal(a[0], f["a_row0"]) and np.array_equal(b[0], f["b_row0"]) and np.array_equal(q[:16], f["q_head"])
    n, m, d = O.search_by_bow(a, b, 0.6)
    assert n == int(f["bow_n"]) and np.array_equal(m, f["bow_match"]) and np.array_equal(d, f["bow_dist"])
    n, m = O.search_for_triangulation(a, b, 0.75)
    assert n == int(f["tri_n"]) and np.array_equal(m, f["tri_match"])
    sc = O.db_scores(q, dbm)
    assert np.array_equal(sc, f["db_scores"])
    c, best = O.db_candidates(sc, 0)
    assert np.array_equal(c, f["db_cand"]) and np.float32(best) == f["db_best"]
    r = np.load(os.path.join(G, "resize_90x60_to_75x50.npz"))
    assert np.array_equal(O.resize_linear_u8(r["src"], 75, 50), r["dst"])


@pytest.mark.gpu
def test_hip_reproduces_fixtures(engine, weights_path):
    """data-only target: no oracle call on this path"""
    from hfnet_slam_amd import capi
    f = np.load(os.path.join(G, "extract_96x72.npz"))
    _check_weights(f, weights_path)
    x = capi.Extractor(engine, 96, 72, 60, 0.01, 1.2, 2, max_batch=1)
    n, kps, desc, g, npl = x.extract(f["image"])
    assert n == int(f["n"]) and np.array_equal(npl, f["n_per_level"])
    assert np.array_equal(kps, f["kps"]) and np.array_equal(desc, f["desc"]) and np.array_equal(g, f["glob"])
    x.close()
    f = np.load(os.path.join(G, "match_db.npz"))
    a, b, dbm, q = _inputs()
    n, m, d = engine.search_by_bow(a, b, 0.6)
    assert n == int(f["bow_n"]) and np.array_equal(m, f["bow_match"]) and np.array_equal(d, f["bow_dist"])
    n, m = engine.search_for_triangulation(a, b, 0.75)
    assert n == int(f["tri_n"]) and np.array_equal(m, f["tri_match"])
    db = capi.Database(engine, 40, 4096)
    for i in range(40):
        db.add(i, dbm[i])
    cs, sc, best, scores = db.query(q, 0, want_scores=True)
    assert np.array_equal(scores, f["db_scores"]) and np.array_equal(cs, f["db_cand"]) and np.float32(best) == f["db_best"]
    db.close()
