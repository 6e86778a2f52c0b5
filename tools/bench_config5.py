#!/usr/bin/env python3
"""BASELINE config 5 (loop-closure stress) on one MI355X: 10 000 x 4096 keyframe database resident in HBM, Q = 1 and
Q = 64 queries, plus the 1000 x 1000 x 256 SearchByBoW match.  Kernel times are the HIP-event times of the library's
own profiler (hfnet_profile_*), i.e. without the host <-> device copies of the host-pointer entry points.

    python tools/bench_config5.py            # prints one JSON line
"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hfnet_slam_amd import capi, weights

N, DIM, REP = 10000, 4096, 10
rng = np.random.default_rng(13)
wp = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_cfg5.hfw")
weights.save(wp, weights.synthetic_weights(7))
eng = capi.Engine(wp, 0)
rows = rng.standard_normal((N, DIM)).astype(np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
db = capi.Database(eng, N, DIM)
for i in range(N):
    db.add(i, rows[i])
qs = rows[rng.integers(0, N, 64)] + 0.003 * rng.standard_normal((64, DIM)).astype(np.float32)
qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
a = rng.standard_normal((1000, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
b = a[rng.permutation(1000)] + 0.02 * rng.standard_normal((1000, 256)).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
sets = np.stack([a, b]).astype(np.float32)
# warm-up
db.query(qs[0]); db.query_batch(qs); eng.search_by_bow_batch(sets, np.array([1000, 1000], np.int32), [(0, 1)] * 32, 0.6)
eng.profile_reset(); eng.profile_filter(None); eng.profile_enable(True)
t0 = time.perf_counter()
for _ in range(REP):
    db.query(qs[0])
t_q1_wall = (time.perf_counter() - t0) / REP
for _ in range(REP):
    db.query_batch(qs)
for _ in range(REP):
    eng.search_by_bow_batch(sets, np.array([1000, 1000], np.int32), [(0, 1)] * 32, 0.6)
eng.synchronize()
prof = eng.profile(); eng.profile_enable(False)
ms = lambda k: prof[k][1] / max(prof[k][0], 1)
db_bytes = N * DIM * 4
out = {
    "config": "loop-closure stress: 10000 x 4096 f32 database, 1000 x 1000 x 256 match, 1 MI355X",
    "db_q1_us": ms("db_scores") * 1e3, "db_q1_GBps": db_bytes / (ms("db_scores") * 1e-3) / 1e9, "db_q1_frac_hbm": db_bytes / (ms("db_scores") * 1e-3) / 8e12,
    "db_q1_call_us_incl_copies": t_q1_wall * 1e6,
    "db_q64_us": ms("db_scores_batch") * 1e3, "db_q64_queries_per_s": 64 / (ms("db_scores_batch") * 1e-3),
    "db_q64_GFLOPs_algorithmic": 64 * N * DIM * 2 / (ms("db_scores_batch") * 1e-3) / 1e9,
    "match_32_pairs_us": ms("match_bow") * 1e3, "match_TFLOPs": 32 * 2 * 1000 * 1000 * 256 / (ms("match_bow") * 1e-3) / 1e12,
    "match_frac_mfma_f32": 32 * 2 * 1000 * 1000 * 256 / (ms("match_bow") * 1e-3) / 157.3e12,
}
print(json.dumps(out))
