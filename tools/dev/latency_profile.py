"""per-launch times of ONE 752x480 frame per call (HIP events around every launch, one stream: each kernel alone on the GPU).
latency_profile.py [reps] [name=value engine options ...]"""
import numpy as np, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hfnet_slam_amd import capi, weights
from conftest import synth_image
wpath = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
for o in sys.argv[2:]:
    k, v = o.split("=")
    eng.set_option(k, int(v))
ext = capi.Extractor(eng, 752, 480, 1000, 0.01, 1.2, 4, max_batch=1)
imgs = [synth_image(480, 752, 100 + i) for i in range(4)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bufs = ext.output_buffers()
for i in range(10):
    ext.extract(imgs[i % 4], bufs)
eng.profile_enable(True); eng.profile_reset()
for i in range(reps):
    ext.extract(imgs[i % 4], bufs)
rows = eng.profile()
eng.profile_enable(False)
tot = 0.0
for name, (n, ms) in rows.items():
    print("%-26s %4d launches  %7.1f us" % (name, n, ms / n * 1e3)); tot += ms / reps * 1e3
print("sum per frame %.1f us" % tot)
