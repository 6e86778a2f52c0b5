"""development: per-layer deviation of the scores_bf16x3 path from the exact path (one level, one frame; Model taps)
   python tools/dev/scores_bf16x3_layers.py [W H] (GPU box)"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfnet_slam_amd import capi, weights
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import synth_image

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (752, 480)
wpath = os.path.join(tempfile.gettempdir(), "hfnet_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
img = synth_image(H, W, 6200, "natural")
taps = list(range(1, 7)) + [20, 21, 22]
names = {**{i: f"layer_{i + 1}" for i in range(1, 7)}, 20: "det_hidden", 21: "logits", 22: "scores_dense"}
res = {}
for mode in (0, 1):
    eng.set_option("scores_bf16x3", mode)
    m = capi.Model(eng, capi.MODE_LOCAL, H, W, 1000)
    st, kps, desc, _ = m.detect(img, 1000, 0.01)
    assert st == 0
    res[mode] = {t: m.tap(t) for t in taps}
    res[mode]["kps"] = kps
    m.close()
for t in taps:
    a, b = res[0][t].astype(np.float64), res[1][t].astype(np.float64)
    d = np.abs(a - b)
    print(f"{names[t]:13s} n {a.size:9d}  max|x| {np.abs(a).max():9.4f}  rms {np.sqrt((a * a).mean()):9.5f}  max|d| {d.max():.3e}  rms d {np.sqrt((d * d).mean()):.3e}  at {int(d.argmax())}")
k0, k1 = res[0]["kps"], res[1]["kps"]
s0 = {(float(a), float(b)) for a, b in zip(k0["x"], k0["y"])}; s1 = {(float(a), float(b)) for a, b in zip(k1["x"], k1["y"])}
print("keypoints", len(k0), len(k1), "common", len(s0 & s1))
