import numpy as np, sys, os, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from hfnet_slam_amd import capi, weights
from conftest import synth_image
wpath = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
imgs = [synth_image(480, 752, 100 + i) for i in range(4)]
for ts in (3, 1, 2, 0):
    eng = capi.Engine(wpath, 0)
    eng.set_option("two_streams", ts)
    ext = capi.Extractor(eng, 752, 480, 1000, 0.01, 1.2, 4, max_batch=1)
    for i in range(20):
        ext.extract(imgs[i % 4])
    ts_ = []
    for i in range(300):
        t0 = time.perf_counter(); ext.extract(imgs[i % 4]); ts_.append(time.perf_counter() - t0)
    print("two_streams", ts, "median ms", float(np.median(ts_)) * 1e3)
    ext.close(); eng.close()
