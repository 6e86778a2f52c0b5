"""one 752x480 frame per host-pointer call, repeated: run under rocprofv3 --kernel-trace (tools/gpu_timeline.sh, tools/gpu_kt_py.sh) for the
per-frame kernel list.   latency_only.py [frames] [name=value engine options ...]"""
import numpy as np, sys, os, tempfile, time
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bufs = ext.output_buffers()
for i in range(10):
    ext.extract(imgs[i % 4], bufs)
ts, st = [], []
for i in range(n):
    t0 = time.perf_counter(); ext.extract(imgs[i % 4], bufs); ts.append(time.perf_counter() - t0)
    st.append(ext.last_timing())
print("host stamps (us, median): staged %.1f  enqueued %.1f  local seen %.1f  unpacked %.1f  drained %.1f  return %.1f" % tuple(np.median(np.array(st), axis=0)))
print("ms per frame: median %.4f  mean %.4f" % (float(np.median(ts)) * 1e3, float(np.mean(ts)) * 1e3), sys.argv[2:])
