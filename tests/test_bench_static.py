"""bench.py pieces that do not need a GPU: workload tables, the committed traffic file, synthetic frames"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layer_work_covers_the_launch_names():
    import bench
    w = bench.layer_work(32)
    for name in ["stem", "block_L02", "block_L03", "block_L07", "block_L08", "conv3x3_det", "pointwise_det", "nms", "conv3x3_desc_taps", "pointwise_desc_taps",
                 "fc_l2", "match_bow"]:
        assert name in w, name
        flop, byts = w[name]
        assert flop > 0 and byts > 0
    # the detector 3x3 conv is 221 kFLOP per cell, 14041 cells per 752x480 frame (4 levels)
    flop, _ = w["conv3x3_det"]
    assert abs(flop / 32 / 14041 - 2 * 9 * 96 * 128) < 1e-6 * flop


def test_traffic_file_matches_bench_lookup():
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic_b32.json")))
    assert t["batch"] == 32 and "conv3x3_det" in t["kernels"]
    k = t["kernels"]["conv3x3_det"]
    assert bench.hbm_traffic("conv3x3_det", 32) == (2.0 * k["fetch_kb"] + k["write_kb"]) * 1024.0
    assert bench.hbm_traffic("conv3x3_det", 8) is None and bench.hbm_traffic("no_such_launch", 32) is None


def test_synthetic_frames_are_seeded():
    import bench
    for kind in ("uniform", "natural"):
        a = bench.make_frames(2, 5, kind); b = bench.make_frames(1, 6, kind)
        assert a.shape == (2, bench.H_IMG, bench.W_IMG) and a.dtype == np.uint8
        assert np.array_equal(a[1], b[0])                       # frame index -> seed, independent of the call
        assert a.std() > 20                                     # not degenerate
