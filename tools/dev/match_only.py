"""N (second argument, default 32) SearchByBoW pairs of 1000 x 1000 x 256 (planted matches), repeated: run under rocprofv3 --kernel-trace --stats for isolated kernel times"""
import numpy as np, sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfnet_slam_amd import capi, weights
wpath = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
rng = np.random.default_rng(13)
a = rng.standard_normal((1000, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
b = a[rng.permutation(1000)] + 0.02 * rng.standard_normal((1000, 256)).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
sets = np.stack([a, b]).astype(np.float32)
nr = np.array([1000, 1000], np.int32)
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    c, m, d = eng.search_by_bow_batch(sets, nr, [(0, 1)] * NP, 0.6)
    c2, m2 = eng.search_for_triangulation_batch(sets, nr, [(0, 1)] * NP, 0.75)
print("matches", c[:3], c2[:3])
