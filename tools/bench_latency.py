#!/usr/bin/env python3
"""BASELINE config 2, per-frame view: latency of one HFextractor call through the host-pointer entry point (upload, 4-level
extraction incl. global descriptor, download, host synchronisation) and of extract + SearchByBoW against the previous
frame, 752x480 / 1000 keypoints, one frame at a time.    python tools/bench_latency.py"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hfnet_slam_amd import capi, weights

wp = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_lat.hfw")
weights.save(wp, weights.synthetic_weights(7))
eng = capi.Engine(wp, 0)
ext = capi.Extractor(eng, 752, 480, 1000, 0.01, 1.2, 4, max_batch=1)
frames = [np.random.default_rng(1000 + i).integers(0, 256, (480, 752), dtype=np.uint8) for i in range(8)]
for f in frames[:3]:
    ext.extract(f)
N = 60
t_ext, t_both, prev = [], [], None
for i in range(N):
    f = frames[i % len(frames)]
    t0 = time.perf_counter()
    n, kps, desc, g, _ = ext.extract(f)
    t1 = time.perf_counter()
    if prev is not None:
        eng.search_by_bow(prev, desc, 0.6)
    t2 = time.perf_counter()
    t_ext.append(t1 - t0); t_both.append(t2 - t0); prev = desc
# the same with the descriptors kept on the GPU: the frame's block goes device-to-device into a two-slot store and the match
# names the slots (hfnet_store_put_extracted / hfnet_store_search_by_bow) -- only the 8 KB of matches come down
store = capi.Store(eng, 2, 1000)
t_dev = []
for i in range(N):
    f = frames[i % len(frames)]
    t0 = time.perf_counter()
    ext.extract(f)
    store.put_extracted(i & 1, ext, 0)
    if i:
        store.search_by_bow([(1 - (i & 1), i & 1)], 0.6)
    t_dev.append(time.perf_counter() - t0)
med = lambda v: float(np.median(v)) * 1e3
print(json.dumps({"config": "752x480, 4 levels, 1000 keypoints, one frame per call, host pointers, 1 MI355X",
                  "extract_ms_median": med(t_ext), "extract_plus_match_ms_median": med(t_both[1:]),
                  "extract_plus_store_match_ms_median": med(t_dev[1:]), "keypoints": int(n),
                  "frames_per_s_unpipelined": 1e3 / med(t_both[1:])}))
