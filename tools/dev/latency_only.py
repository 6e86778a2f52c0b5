"""one 752x480 frame per host-pointer call, repeated: run under rocprofv3 --kernel-trace --stats (tools/gpu_kt_py.sh) for the per-frame kernel list"""
import numpy as np, sys, os, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hfnet_slam_amd import capi, weights
from conftest import synth_image
wpath = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
ext = capi.Extractor(eng, 752, 480, 1000, 0.01, 1.2, 4, max_batch=1)
imgs = [synth_image(480, 752, 100 + i) for i in range(4)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for i in range(10):
    ext.extract(imgs[i % 4])
t0 = time.perf_counter()
for i in range(n):
    ext.extract(imgs[i % 4])
print("ms per frame", (time.perf_counter() - t0) / n * 1e3)
