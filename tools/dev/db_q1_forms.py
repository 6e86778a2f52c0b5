"""hfnet_db_query (ONE query) as the exact f32 scan and in the screened form, by database size: wall-clock per call (host pointers in and out) and the
engine's per-launch times -- where engine option db_screen_min_rows should sit"""
import numpy as np, sys, os, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hfnet_slam_amd import capi, weights
wpath = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
DIM = 4096
rng = np.random.default_rng(13)
blk = rng.standard_normal((2048, DIM)).astype(np.float32); blk /= np.linalg.norm(blk, axis=1, keepdims=True)
for N in (1000, 2500, 5000, 10000, 20000, 40000):
    db = capi.Database(eng, N, DIM)
    for i in range(N):
        db.add(i, blk[i & 2047])
    q = (blk[5] + 0.003 * rng.standard_normal(DIM)).astype(np.float32); q /= np.linalg.norm(q)
    out = []
    for form in (0, 1):
        eng.set_option("db_screen_min_rows", form)
        for _ in range(5): db.query(q)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): db.query(q)
        wall = (time.perf_counter() - t0) / 50
        eng.profile_reset(); eng.profile_enable(True)
        for _ in range(20): db.query(q)
        eng.synchronize()
        p = eng.profile(); eng.profile_enable(False)
        dev = sum(v[1] / max(v[0], 1) for k, v in p.items() if k.startswith("db_")) * 1e3
        out.append((wall * 1e6, dev))
    print("N %6d  scan: call %.1f us, launches %.1f us   screened: call %.1f us, launches %.1f us" % (N, out[0][0], out[0][1], out[1][0], out[1][1]))
    db.close()
eng.set_option("db_screen_min_rows", 6144)
# the batched query's call time (64 queries, 10 000 slots)
N = 10000
db = capi.Database(eng, N, DIM)
for i in range(N):
    db.add(i, blk[i & 2047] if i >= 2048 else blk[i])
qs = blk[rng.integers(0, 2048, 64)] + 0.003 * rng.standard_normal((64, DIM)).astype(np.float32)
qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
import ctypes as C
from hfnet_slam_amd.capi import lib, _p, _chk
slot = np.zeros((64, N), np.int32); score = np.zeros((64, N), np.float32); n = np.zeros((64,), np.int32); best = np.zeros((64,), np.float32)
for _ in range(5): _chk(lib().hfnet_db_query_batch(db.h, 64, _p(qs), 0, _p(slot), _p(score), _p(n), _p(best), None))
t0 = time.perf_counter()
for _ in range(50): _chk(lib().hfnet_db_query_batch(db.h, 64, _p(qs), 0, _p(slot), _p(score), _p(n), _p(best), None))
print("64 queries x %d slots: call %.1f us (host pointers in and out, no scores_all); candidates per query %.1f" % (N, (time.perf_counter() - t0) / 50 * 1e6, n.mean()))
db.close()
