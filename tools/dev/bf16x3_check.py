"""development: the split-bf16 options against the exact path -- keypoints must be identical, descriptors / global descriptor within tolerance
   python tools/dev/bf16x3_check.py (GPU box)"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import bench
from hfnet_slam_amd import capi, weights

wpath = os.path.join(tempfile.gettempdir(), "hfnet_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
opts = [o for o in ("desc_bf16x3", "global_bf16x3") if o in capi.Engine.OPTIONS]
for (w, h, nf, B) in ((752, 480, 1000, 16), (512, 512, 850, 8)):
    imgs = np.concatenate([bench.make_frames(B // 2, 0, "uniform", w, h), bench.make_frames(B // 2, 100, "natural", w, h)])
    res = {}
    for mode in (0, 1):
        for o in opts: eng.set_option(o, mode)
        x = capi.Extractor(eng, w, h, nf, 0.01, 1.2, 4, max_batch=B)
        res[mode] = x.extract_batch(imgs)
        x.close()
    n0, k0, d0, g0 = res[0]; n1, k1, d1, g1 = res[1]
    assert np.array_equal(n0, n1) and np.array_equal(k0, k1), "keypoints differ"
    dd = max(float(np.abs(d0[f, :n0[f]] - d1[f, :n0[f]]).max()) for f in range(B))
    rn = max(float(np.abs(np.linalg.norm(d1[f, :n0[f]].astype(np.float64), axis=1) - 1).max()) for f in range(B))
    print(f"{w}x{h}: keypoints identical; descriptors max |d| {dd:.3e} (row norms within {rn:.1e} of 1); global max |d| {float(np.abs(g0 - g1).max()):.3e}")
    # match sets on consecutive frames: exact descriptors vs bf16x3 descriptors
    same = tot = 0
    for f in range(1, B):
        c0, m0, _ = eng.search_by_bow(d0[f - 1, :n0[f - 1]], d0[f, :n0[f]], 0.6)
        c1, m1, _ = eng.search_by_bow(d1[f - 1, :n0[f - 1]], d1[f, :n0[f]], 0.6)
        same += int(np.sum(m0 == m1)); tot += len(m0)
    print(f"   SearchByBoW on consecutive frames: {same} of {tot} query rows get the same answer")
