import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hfnet_slam_amd import capi, weights
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
wpath = os.path.join(tempfile.gettempdir(), "hfnet_soakdb.hfw")
weights.save(wpath, weights.synthetic_weights(101))
eng = capi.Engine(wpath, 0)
def unit(n, d):
    a = rng.standard_normal((n, d)).astype(np.float32)
    return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
bad = 0
for it in range(400):
    n, dim = int(rng.integers(1, 1500)), 4096
    rows = unit(n, dim)
    db = capi.Database(eng, n + 5, dim)
    for i in range(n):
        db.add(i, rows[i])
    nq = int(rng.choice([1, 3, 8, 20, 64]))
    noise = float(rng.choice([0.0, 0.003, 0.02]))
    qs = rows[rng.integers(0, n, nq)] + noise * rng.standard_normal((nq, dim)).astype(np.float32)
    qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
    mode = int(rng.integers(0, 2))
    if nq == 1:
        slots, sc, best, allsc = db.query(qs[0], mode, want_scores=True)
        ref = O.db_scores(qs[0], rows)
        ridx, rbest = O.db_candidates(ref, mode)
        e_sc = not np.array_equal(allsc[:n], ref); e_b = best != rbest; e_s = not np.array_equal(np.sort(slots), np.sort(ridx))
        if e_sc or e_b or e_s:
            bad += 1
            d = np.flatnonzero(allsc[:n] != ref)
            print("q1", it, n, mode, noise, "scores", e_sc, len(d), d[:5], allsc[d[:3]], ref[d[:3]], "best", best, rbest, "slots", len(slots), len(ridx))
    else:
        res, best, allsc = db.query_batch(qs, mode, want_scores=True)
        ref = O.db_scores_gemm(qs, rows) if nq >= 8 else np.stack([O.db_scores(q, rows) for q in qs])
        if not np.array_equal(allsc[:, :n], ref):
            bad += 1
            d = np.argwhere(allsc[:, :n] != ref)
            print("batch", it, n, nq, mode, noise, "diff", len(d), d[:4].tolist(), [float(allsc[i, j]) for i, j in d[:3]], [float(ref[i, j]) for i, j in d[:3]])
    db.close()
print("bad", bad)
