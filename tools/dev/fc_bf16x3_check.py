"""development: the FC on split-bf16 operands (calls of >= 64 frames with global_bf16x3) against the exact mode and, per frame, against the same
   frames in calls below the threshold (the f32 FC on the same split-bf16 activations): isolates the FC's own deviation
   python tools/dev/fc_bf16x3_check.py (GPU box)"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hfnet_slam_amd import capi, weights
from conftest import synth_image

for seed in (7, 11):
    wpath = os.path.join(tempfile.gettempdir(), "hfnet_dev.hfw")
    weights.save(wpath, weights.synthetic_weights(seed))
    eng = capi.Engine(wpath, 0)
    for (W, H, B) in [(200, 152, 64), (131, 121, 200), (376, 240, 130)]:
        imgs = np.stack([synth_image(H, W, 7100 + i, "natural" if i % 3 else "uniform") for i in range(B)])
        out = {}
        for mode in (0, 1):
            eng.set_option("global_bf16x3", mode)
            x = capi.Extractor(eng, W, H, 200, 0.01, 1.2, 2, max_batch=B)
            out[mode] = x.extract_batch(imgs)[3].astype(np.float64)
            if mode:
                small = np.stack([x.extract_batch(imgs[i:i + 8])[3] for i in range(0, 16, 8)]).reshape(16, -1).astype(np.float64)
            x.close()
        print(f"seed {seed} {W}x{H} x{B}: |g| {np.linalg.norm(out[1], axis=1).min():.7f}..{np.linalg.norm(out[1], axis=1).max():.7f}"
              f"  max|d| vs exact {np.abs(out[1] - out[0]).max():.3e}  FC alone (vs calls of 8) {np.abs(out[1][:16] - small).max():.3e}")
    eng.close()
