"""development: layer-8..18 taps, exact vs global_bf16x3 with the fused forms forced (fuse_min_wgs = 0), several sizes
   python tools/dev/l8_bf16x3_check.py (GPU box)"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hfnet_slam_amd import capi, weights
from conftest import synth_image

wpath = os.path.join(tempfile.gettempdir(), "hfnet_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
for (W, H) in [(200, 152), (131, 121), (752, 480)]:
    img = synth_image(H, W, 6300 + W, "natural")
    res = {}
    for mode in (0, 1):
        eng.set_option("global_bf16x3", mode); eng.set_option("fuse_min_wgs", 0); eng.set_option("tail_fuse", 0)
        m = capi.Model(eng, capi.MODE_LOCAL_AND_GLOBAL, H, W, 300)
        st, kps, desc, g = m.detect(img, 300, 0.01)
        assert st == 0
        res[mode] = {t: m.tap(t) for t in range(6, 18)}
        res[mode]["g"] = g
        m.close()
    print(W, H)
    for t in range(6, 18):
        a, b = res[0][t].astype(np.float64), res[1][t].astype(np.float64)
        d = np.abs(a - b)
        print(f"  layer_{t + 1:2d} n {a.size:8d} rms {np.sqrt((a * a).mean()):8.4f} max|d| {d.max():.3e} rms d {np.sqrt((d * d).mean()):.3e} at {int(d.argmax())}")
    print("  global max|d|", float(np.abs(res[0]["g"].astype(np.float64) - res[1]["g"]).max()))
