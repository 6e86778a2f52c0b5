"""development: the host-fed batch path by itself -- pageable / registered arrays, one thread / matcher thread, frames per call
   python tools/dev/host_io_only.py [chunk] (GPU box)"""
import os, sys, time, threading, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import bench
from hfnet_slam_amd import capi, weights

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 128
wpath = os.path.join(tempfile.gettempdir(), "hfnet_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
for cpc in (8, 16):
    n = chunk * cpc
    ext = capi.Extractor(eng, bench.W_IMG, bench.H_IMG, bench.N_FEAT, bench.THRESH, bench.SCALE, bench.N_LEVELS, max_batch=chunk)
    store = capi.Store(eng, 2 * n, bench.N_FEAT)
    imgs = np.concatenate([bench.make_frames(256, 0)] * (n // 256))
    ext.attach_store(store, 0)
    out = ext.extract_batch(imgs)
    for reg in (0, 1):
        bufs = [imgs, out[0], out[1], out[2], out[3]]
        if reg:
            for b in bufs: capi.host_register(b)
        for mode in ("extract", "seq", "thread"):
            def match_call(i):
                base = (i & 1) * n
                pr = [((base + f - 1) % (2 * n), base + f) for f in range(0 if i else 1, n)]
                for p0 in range(0, len(pr), chunk):
                    store.search_by_bow(pr[p0:p0 + chunk], bench.TH_LOW)
            res = []
            for rep in range(2):
                worker = None
                t0 = time.perf_counter()
                for i in range(3):
                    ext.attach_store(store, (i & 1) * n)
                    ext.extract_batch(imgs, out)
                    if mode == "seq": match_call(i)
                    elif mode == "thread":
                        if worker: worker.join()
                        worker = threading.Thread(target=match_call, args=(i,)); worker.start()
                if worker: worker.join()
                res.append(3 * n / (time.perf_counter() - t0))
            print(f"frames/call {n} registered {reg} {mode:8s}: {res[-1]:8.0f} frames/s", flush=True)
        if reg:
            for b in bufs: capi.host_unregister(b)
    ext.attach_store(None); store.close(); ext.close()
