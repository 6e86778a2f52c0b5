"""config 5's batched database query alone (10 000 x 4096, Q = 64), repeated: for tools/gpu_kt_py.sh (per-kernel times)"""
import numpy as np, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hfnet_slam_amd import capi, weights
wpath = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
N, DIM, Q = 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(13)
rows = rng.standard_normal((N, DIM)).astype(np.float32); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
db = capi.Database(eng, N, DIM)
for i in range(N):
    db.add(i, rows[i])
qs = rows[rng.integers(0, N, Q)] + 0.003 * rng.standard_normal((Q, DIM)).astype(np.float32)
qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
for _ in range(30):
    db.query_batch(qs)
eng.synchronize()
