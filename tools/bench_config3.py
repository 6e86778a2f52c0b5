#!/usr/bin/env python3
"""BASELINE config 3 (TUM-VI corridor-style tracking loop) on one MI355X, one frame at a time through the host-pointer
entry points, as Tracking / LocalMapping / LoopClosing would call them:

  every frame     HFextractor (512x512, 4 levels, nFeatures keypoints, global descriptor) + SearchByBoW vs the last frame
  every 5th frame "keyframe": database add + DetectNBestCandidates scan over all previous keyframes
                  + SearchForTriangulation against the 30 most recent keyframes (one batched call)

    python tools/bench_config3.py [nFeatures=1000] [frames=300] [store=1]

store=1 keeps every keyframe's descriptor block in a device-resident hfnet_store (uploaded once when the keyframe is made);
store=0 re-packs and re-uploads the 31 blocks on every keyframe through the host-pointer batch call.
"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hfnet_slam_amd import capi, weights

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 300
STORE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W = H = 512
wp = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_cfg3.hfw")
weights.save(wp, weights.synthetic_weights(7))
eng = capi.Engine(wp, 0)
ext = capi.Extractor(eng, W, H, NF, 0.01, 1.2, 4, max_batch=1)
db = capi.Database(eng, FRAMES // 5 + 8, eng.global_dim)
store = capi.Store(eng, FRAMES // 5 + 8, NF) if STORE else None       # keyframe slots; the last two slots hold the current / previous frame
F0 = FRAMES // 5 + 6
imgs = [np.random.default_rng(1000 + i).integers(0, 256, (H, W), dtype=np.uint8) for i in range(16)]
for i in range(3):
    ext.extract(imgs[i])
kf_desc, prev = [], None
t_frame, t_kf = [], []
n_kf = 0
t_all0 = time.perf_counter()
for i in range(FRAMES):
    t0 = time.perf_counter()
    n, kps, desc, g, _ = ext.extract(imgs[i % len(imgs)])
    if STORE:                                        # frame-to-frame match by slot: the descriptors stay on the GPU
        store.put_extracted(F0 + (i & 1), ext, 0)
        if i:
            store.search_by_bow([(F0 + 1 - (i & 1), F0 + (i & 1))], 0.6)
    elif prev is not None:
        eng.search_by_bow(prev, desc, 0.6)
    prev = desc
    t1 = time.perf_counter()
    t_frame.append(t1 - t0)
    if i % 5 == 0:
        if n_kf and STORE:
            db.query(g, 0)
            store.put_extracted(n_kf, ext, 0)
            store.search_for_triangulation([(n_kf, j) for j in range(max(0, n_kf - 30), n_kf)], 0.75)
        elif n_kf:
            db.query(g, 0)
            nb = kf_desc[-30:]
            mr = max(d.shape[0] for d in nb + [desc])
            sets = np.zeros((len(nb) + 1, mr, 256), np.float32)
            rows = np.zeros((len(nb) + 1,), np.int32)
            for j, d in enumerate([desc] + nb):
                sets[j, :d.shape[0]] = d; rows[j] = d.shape[0]
            eng.search_for_triangulation_batch(sets, rows, [(0, j + 1) for j in range(len(nb))], 0.75)
        if STORE and not n_kf:
            store.put_extracted(0, ext, 0)
        db.add(n_kf, g); kf_desc.append(desc); n_kf += 1
        t_kf.append(time.perf_counter() - t1)
wall = time.perf_counter() - t_all0
med = lambda v: float(np.median(v)) * 1e3
print(json.dumps({"config": f"tracking loop, 512x512, 4 levels, {NF} keypoints, {FRAMES} frames, keyframe every 5th, 1 MI355X, " + ("device-resident keyframe store" if STORE else "host-pointer calls"),
                  "frame_ms_median": med(t_frame), "keyframe_extra_ms_median": med(t_kf[31:] if len(t_kf) > 40 else t_kf[1:]),
                  "keyframes": n_kf, "frames_per_s_whole_loop": FRAMES / wall}))
