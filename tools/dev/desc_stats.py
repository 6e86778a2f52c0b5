"""descriptor statistics of the synthetic bench workload: how close are the descriptors of consecutive synthetic frames?"""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tempfile
from hfnet_slam_amd import capi, weights
wpath = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
ext = capi.Extractor(eng, 752, 480, 1000, 0.01, 1.2, 4)
rng = np.random.default_rng(5)
fr = [rng.integers(0, 256, (480, 752), dtype=np.uint8) for _ in range(2)]
out = [ext.extract(f) for f in fr]
d0, d1 = out[0][2], out[1][2]
print("rows", d0.shape, d1.shape)
S = d0 @ d1.T
D2 = np.maximum(2 - 2 * S, 0)
srt = np.sort(D2, axis=1)
print("nearest d^2: min %.3g med %.3g max %.3g" % (srt[:, 0].min(), np.median(srt[:, 0]), srt[:, 0].max()))
print("gap 2nd-1st d^2: min %.3g med %.3g" % ((srt[:, 1] - srt[:, 0]).min(), np.median(srt[:, 1] - srt[:, 0])))
print("mean d^2 %.3g  spread (std of all d^2) %.3g" % (D2.mean(), D2.std()))
for band in (1e-5, 6e-5, 1e-4, 1e-3, 1.3e-3, 4e-3, 1.6e-2, 3.2e-2):
    print("band", band, "avg candidates per row within band of the row minimum:", ((D2 <= srt[:, :1] + band).sum(1)).mean())
