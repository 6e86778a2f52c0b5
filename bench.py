#!/usr/bin/env python3
"""bench.py -- frames/sec of HF-Net extract + brute-force match, 752x480, 1000 keypoints (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Headline = BASELINE config 2.  One "step" = one batch of `--batch` (768) synthetic 752x480 frames through the whole
front end on one GPU, in chunks of `--chunk` (128) frames: 4-level pyramid (x1.2), HF-Net (MobileNetV2 backbone,
detector + descriptor heads, NetVLAD on level 0), NMS, per-level top-K (budget 322/268/224/186), bilinear descriptor
sampling, then one SearchByBoW-style brute-force match (1000 x 1000 x 256, L2 cross-check, < 0.6) of every frame
against its predecessor.  Inputs are resident in HBM before the timed region; outputs stay in HBM.  Frames are
independent, so N GPUs run N replicas on disjoint frames (weak scaling, no collective on the data path); the timed
region is bracketed by barrier + device synchronise, MAX over ranks.

The same run also measures, as sub-records of the one JSON line (`configs`), the other BASELINE configs:
  2-latency  one frame per call through the host-pointer entry points (upload, extract, download, match)
  2-host-io  the batch path with host buffers on both sides (images up, keypoints / descriptors / global down)
  3          TUM-VI 512x512 tracking loop: extract + match per frame, every 5th frame a keyframe (database scan +
             SearchForTriangulation against 30 neighbours)
  4          all 11 EuRoC sequences (27 049 frames) assigned to the ranks longest-first, per-sequence frame order
  5          loop-closure stress: 10 000 x 4096 database scans (Q = 1 warm / cold, Q = 64) and 32 1000 x 1000 x 256 matches
and a time-weighted per-launch roofline table from a dedicated single-stream profiling pass (`roofline.table`).
Rank 0 prints ONE JSON line (contract: the task description / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_IMG, H_IMG, N_FEAT, N_LEVELS, SCALE, THRESH, TH_LOW, TH_HIGH = 752, 480, 1000, 4, 1.2, 0.01, 0.6, 0.75
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4)
MFMA_BF16_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense bf16 MFMA (16 x the f32 rate); only the matcher's screening GEMM runs there
TRAFFIC_FILES = [os.path.join("profiles", r + "_traffic_b{batch}.json") for r in ("r06", "r05", "r04", "r03", "r02")]     # newest first; one file per frames-per-call value
ALL_CONFIGS = ["2-latency", "2-host-io", "2-bf16x3", "2-sparse", "3", "4", "5"]
DEFAULT_CHUNK = 256            # frames per extract / match call of the headline (tests/test_gpu_fullsize.py checks THIS size against the oracle);
                               # measured 96 / 128 / 160 / 192 / 256 frames per call: 7012 / 7131 / 7124 / 7229 / 7245 frames/s on one box (NOTEBOOK.md R5.6);
                               # the gain flattens there (tests cover calls up to this size)
DEFAULT_BATCH = 768            # frames per step and GPU


# ------------------------------------------------------------------------------------------------ work model
def layer_work(batch: int, width: int = W_IMG, height: int = H_IMG, n_feat: int = N_FEAT):
    """Per launch name: (algorithmic FLOP, algorithmic bytes, executed FLOP) for one chunk of `batch` frames.
    Algorithmic = fp32, every layer reads its input once, writes its output once, reads its weights once (SURVEY.md
    8d); executed = what the kernels really issue (halo recompute and M-tile padding of the fused blocks)."""
    from hfnet_slam_amd import spec as S
    sp = S.net_spec()
    sizes = S.level_sizes(width, height, N_LEVELS, SCALE)
    work = {}

    def add(name, flop, byts, executed=None):
        f, b, x = work.get(name, (0.0, 0.0, 0.0))
        work[name] = (f + flop, b + byts, x + (flop if executed is None else executed))

    V4_SHAPES = {(1, 3, 1), (1, 3, 2), (1, 6, 2), (1, 6, 3), (1, 9, 3), (2, 2, 1), (2, 3, 1), (2, 12, 2)}   # (stride, cin / 8, cout tiles): kernels_block.hip
    V6_SHAPES = {(1, 6, 3), (1, 12, 3), (1, 12, 5), (1, 18, 5)}                                  # (stride, cin / 4, 16-column tiles): k_block_fused6
    # (k_block_fused8 -- layers 3, 5, 7: the expanded tensor in registers -- issues k_block_fused4's MFMAs: same tiles, same M tiles)

    def fused_executed(b, oh, ow):
        """FLOP the fused block kernel issues for one image.  k_block_fused4 (wave tiles of 4 x 8 outputs, expansion of
        the (3 s + 3) x (7 s + 3) halo in M-tiles of 32 positions, 32-channel chunks) where it is instantiated, otherwise
        k_block_fused2 (workgroup tiles of 8 x 8 (stride 2) / 8 x 16 outputs, halo shared by the four waves)"""
        s = b.stride
        chunks = -(-b.expand // 32)
        nto = -(-b.cout // 32)
        if (s, b.cin // 4, -(-b.cout // 16)) in V6_SHAPES:
            # k_block_fused6: 6 x 8 outputs, the 8 x 10 halo as five 16-row M tiles, 16-column tiles (v_mfma_f32_16x16x4_f32: 2048 FLOP)
            tiles = -(-oh // 6) * -(-ow // 8)
            n16e, n16p = -(-b.expand // 16), -(-b.cout // 16)
            mfma16 = tiles * (5 * n16e * (b.cin // 4) + 3 * n16p * (b.expand // 4))
            return mfma16 * 2048.0 + 2.0 * 9 * b.expand * tiles * 48
        if (s, b.cin // 8, nto) in V4_SHAPES:
            th, tw = 4, 8
        else:
            th, tw = 8, (8 if s == 2 else 16)
        tiles = -(-oh // th) * -(-ow // tw)
        mt_in = -(-(((th - 1) * s + 3) * ((tw - 1) * s + 3)) // 32)
        mt_out = th * tw // 32
        mfma = tiles * (chunks * mt_in * (b.cin // 2) + (b.expand // 8) * 4 * nto * mt_out)       # v_mfma_f32_32x32x2_f32: 4096 FLOP
        return mfma * 4096.0 + 2.0 * 9 * b.expand * tiles * th * tw

    for lvl, (w, h) in enumerate(sizes):
        hc, wc = S.cropped(h), S.cropped(w)
        ph, pw = S.same_pad(hc, 3, 2)[0], S.same_pad(wc, 3, 2)[0]
        px = ph * pw * batch
        add("stem", 2.0 * 9 * sp.stem_out * px, hc * wc * batch + 4.0 * px * sp.stem_out)
        if lvl:
            add("pyramid_resize", 8.0 * h * w * batch, (1.0 + 1.44) * h * w * batch)
        for b in sp.blocks:
            if b.index > 7 and lvl > 0:
                break
            oh, ow = S.same_pad(ph, 3, b.stride)[0], S.same_pad(pw, 3, b.stride)[0]
            pin, pout = ph * pw * batch, oh * ow * batch
            if b.expand > b.cin:
                add(f"expand_L{b.index:02d}", 2.0 * pin * b.cin * b.expand, 4.0 * (pin * (b.cin + b.expand) + b.cin * b.expand))
            add(f"depthwise_L{b.index:02d}", 2.0 * 9 * pout * b.expand, 4.0 * (pin * b.expand + pout * b.expand + 9 * b.expand))
            add(f"project_L{b.index:02d}", 2.0 * pout * b.expand * b.cout,
                4.0 * (pout * (b.expand + b.cout * (2 if b.residual else 1)) + b.expand * b.cout))
            # the same block as ONE fused launch: all three layers' FLOP, but only the block input / output cross HBM
            alg = (2.0 * pin * b.cin * b.expand if b.expand > b.cin else 0.0) + 2.0 * 9 * pout * b.expand + 2.0 * pout * b.expand * b.cout
            add(f"block_L{b.index:02d}", alg,
                4.0 * (pin * b.cin + pout * b.cout * (2 if b.residual else 1) + b.cin * b.expand + 9 * b.expand + b.expand * b.cout),
                batch * fused_executed(b, oh, ow) if b.expand > b.cin else alg)
            if b.index == 2:
                # stem + layer_2 as ONE launch (the default): u8 image in, layer_2 output out; the stem is recomputed on
                # the 18 x 18 halo of every 16 x 16 tile
                tiles = -(-oh // 16) * -(-ow // 16) * batch
                add("stem_block_L02", 2.0 * 9 * sp.stem_out * px + 2.0 * 9 * pout * b.expand + 2.0 * pout * b.expand * b.cout,
                    hc * wc * batch + 4.0 * (pout * b.cout + 9 * sp.stem_out + 9 * b.expand + b.expand * b.cout),
                    tiles * (324 * 2.0 * 9 * sp.stem_out + 256 * (2.0 * 9 * b.expand + 2.0 * b.expand * b.cout)))
            ph, pw = oh, ow
            if b.index == 7:
                cells = ph * pw * batch
                c7 = b.cout
                add("conv3x3_desc", 2.0 * 9 * c7 * 256 * cells, 4.0 * (cells * (c7 + 256) + 9 * c7 * 256))
                add("pointwise_desc", 2.0 * 256 * 256 * cells, 4.0 * (cells * 512 + 256 * 256))
                add("l2norm_desc", 3.0 * 256 * cells, 4.0 * cells * 512)
                add("conv3x3_det", 2.0 * 9 * c7 * 128 * cells, 4.0 * (cells * (c7 + 128) + 9 * c7 * 128))
                add("pointwise_det", 2.0 * 128 * 65 * cells, 4.0 * (cells * (128 + 65) + 128 * 65), 2.0 * 128 * 96 * cells)
                add("softmax_d2s", 4.0 * 65 * cells, 4.0 * cells * (65 + 64))
                # the two as ONE launch (the default): hidden map in, score map out, the logits stay in LDS; executed = two 32-column
                # MFMA tiles + the dustbin column and the softmax on the vector ALU
                add("det_tail", (2.0 * 128 * 65 + 4.0 * 65) * cells, 4.0 * (cells * (128 + 64) + 128 * 65), (2.0 * 128 * 65 + 4.0 * 65) * cells)
                add("nms", 2.0 * 3 * 18 * hc * wc * batch, 4.0 * 2 * hc * wc * batch)
        if lvl == 0:
            pg = ph * pw * batch
            add("pointwise_memberships", 2.0 * pg * sp.global_channels * sp.n_clusters, 4.0 * pg * (sp.global_channels + sp.n_clusters))
            add("softmax_memberships", 4.0 * pg * sp.n_clusters, 8.0 * pg * sp.n_clusters)
            add("vlad", 3.0 * pg * sp.vlad_dim, 4.0 * (pg * (sp.global_channels + sp.n_clusters) + batch * sp.vlad_dim))
            add("vlad_norm", 6.0 * batch * sp.vlad_dim, 4.0 * 2 * batch * sp.vlad_dim)
            add("fc_l2", 2.0 * batch * sp.vlad_dim * sp.global_dim, 4.0 * (sp.vlad_dim * sp.global_dim + batch * (sp.vlad_dim + 2 * sp.global_dim)),
                2.0 * (-(-batch // 16) * 16) * sp.vlad_dim * sp.global_dim)
    # sparse descriptor head: 4 bilinear taps per keypoint
    rows = 4.0 * n_feat * batch
    work["conv3x3_desc_taps"] = (2.0 * 9 * sp.local_channels * 256 * rows, 4.0 * (rows * (9 * sp.local_channels + 256) + 9 * sp.local_channels * 256),
                                 2.0 * 9 * sp.local_channels * 256 * rows)
    work["pointwise_desc_taps"] = (2.0 * 256 * 256 * rows, 4.0 * (rows * 512 + 256 * 256), 2.0 * 256 * 256 * rows)
    work["sample"] = (8.0 * 256 * n_feat * batch, 4.0 * (rows * 256 + n_feat * batch * 260), 8.0 * 256 * n_feat * batch)
    work["topk"] = (0.0, 8.0 * 4 * n_feat * batch * 16, 0.0)
    # matcher: all frame pairs of a chunk in one batched call (prep + GEMM with candidate epilogue + exact refinement + finalize);
    # no similarity matrix: two descriptor sets in, one (match, distance) pair per query out
    mm = batch * 2.0 * n_feat * n_feat * 256
    work["match_bow"] = (mm, batch * (4.0 * 2 * n_feat * 256 + 8.0 * n_feat), batch * 2.0 * 1024 * 1024 * 256)
    return work


def roofline_entry(flop, byts, seconds):
    """{bound, achieved, peak, unit, frac} of one launch: the MFMA roof when the arithmetic intensity is above the
    machine balance (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B), the HBM roof otherwise."""
    if byts > 0 and flop / byts >= MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        r = {"bound": "mfma", "achieved": flop / seconds / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s"}
    else:
        r = {"bound": "hbm", "achieved": byts / seconds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    r["frac"] = r["achieved"] / r["peak"]
    return r


def hbm_traffic(launch_name: str, batch: int):
    """(HBM bytes per launch, file) from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate
    runs of this command, folded by tools/make_traffic.py), with the gfx950 correction of MI355X_MICROARCH.md
    (FETCH_SIZE counts 128-byte requests as 64 bytes -> doubled).  (None, None) when no pass exists for this kernel / chunk
    size.  NOT a measurement of the present run: rocprofv3 cannot run inside the timed process."""
    for pat in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, pat.format(batch=batch))) as f:
                t = json.load(f)
            k = t["kernels"][launch_name]
            if t["batch"] != batch:
                continue
            return (2.0 * k["fetch_kb"] + k["write_kb"]) * 1024.0, pat.format(batch=batch)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


# ------------------------------------------------------------------------------------------------ synthetic data
def host(t):
    """device tensor -> numpy through PINNED host memory: a plain `.cpu()` has torch's HIP runtime pin the pageable destination in place -- the
    mechanism behind the rare "Memory access fault ... <host heap page>" aborts of NOTEBOOK.md R5.4 (read-backs only: outside every timed region)"""
    import torch
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.synchronize()
    return h.numpy().copy()


def to_device(torch, a, dev):
    """numpy -> device tensor through pinned host memory (see host())"""
    src = torch.from_numpy(np.ascontiguousarray(a))
    h = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    h.copy_(src)
    d = h.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    return d


def make_frames(count: int, first_index: int, kind: str = "uniform", w: int = W_IMG, h: int = H_IMG) -> np.ndarray:
    """seed 1000 + frame index; "uniform" = iid uniform u8, "natural" = sum of 6 octaves of bilinearly
    up-sampled uniform noise, clipped (smooth structures at several scales)"""
    out = np.empty((count, h, w), np.uint8)
    for i in range(count):
        rng = np.random.default_rng(1000 + first_index + i)
        if kind == "uniform":
            out[i] = rng.integers(0, 256, (h, w), dtype=np.uint8)
            continue
        acc = np.zeros((h, w), np.float64)
        for o in range(6):
            gh, gw = 2 + (h >> (6 - o)), 2 + (w >> (6 - o))
            g = rng.random((gh, gw))
            ys = np.linspace(0, gh - 1.001, h); xs = np.linspace(0, gw - 1.001, w)
            y0 = ys.astype(int); x0 = xs.astype(int); fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
            up = (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0][:, x0 + 1] * (1 - fy) * fx + g[y0 + 1][:, x0] * fy * (1 - fx) + g[y0 + 1][:, x0 + 1] * fy * fx)
            acc += up * 0.5 ** (5 - o)
        acc = (acc - acc.min()) / (acc.max() - acc.min())
        out[i] = np.clip(acc * 1.2 * 255.0 - 25.0, 0, 255).astype(np.uint8)
    return out


def make_natural_frames_device(torch, dev, count: int, first_index: int, w: int = W_IMG, h: int = H_IMG):
    """make_frames(..., "natural") evaluated on the GPU (the numpy form takes 80 ms per frame on the host): the same seeded
    noise grids, the same bilinear up-sampling in float64 -- equal to the host form up to the rounding of the interpolation.
    Used for `value_natural` only; the frames that are compared with the oracle come from make_frames."""
    out = torch.empty((count, h, w), dtype=torch.uint8, device=dev)
    for i in range(count):
        rng = np.random.default_rng(1000 + first_index + i)
        acc = torch.zeros((h, w), dtype=torch.float64, device=dev)
        for o in range(6):
            gh, gw = 2 + (h >> (6 - o)), 2 + (w >> (6 - o))
            g = to_device(torch, rng.random((gh, gw)), dev)
            ys = torch.linspace(0, gh - 1.001, h, dtype=torch.float64, device=dev); xs = torch.linspace(0, gw - 1.001, w, dtype=torch.float64, device=dev)
            y0 = ys.long(); x0 = xs.long(); fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
            up = (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0][:, x0 + 1] * (1 - fy) * fx + g[y0 + 1][:, x0] * fy * (1 - fx) + g[y0 + 1][:, x0 + 1] * fy * fx)
            acc += up * 0.5 ** (5 - o)
        acc = (acc - acc.min()) / (acc.max() - acc.min())
        out[i] = torch.clamp(acc * 1.2 * 255.0 - 25.0, 0, 255).to(torch.uint8)
    return out


def unit_rows(rng, n, d):
    a = rng.standard_normal((n, d)).astype(np.float32)
    return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(weights_path: str, budget_s: float = 24.0):
    """Oracle (CPU restatement, kind 'port') on a bounded sample of the same workload, host cores of this box: all
    usable cores (the headline `value`), 4 threads (the reference runs one OpenCV worker per pyramid level,
    HFextractor.cc:265), and the matcher / database scan on half the cores (the reference gives Eigen nbThreads / 2,
    System.cc:49).  Reported, never optimised against."""
    from oracle import oracle as O
    O.build()
    cores = O.usable_cpus(32)      # affinity and cgroup quota, at most 32
    m = O.Model(weights_path)
    frames = make_frames(6, 0)
    O.set_threads(cores)
    m.extract(frames[0], N_FEAT, THRESH, N_LEVELS, SCALE)       # warm-up (page-in, FC transpose)

    def run(threads, n_max, seconds):
        O.set_threads(threads)
        done, prev, t0 = 0, None, time.perf_counter()
        for i in range(1, 1 + n_max):
            _, _, desc, _, _ = m.extract(frames[i % len(frames)], N_FEAT, THRESH, N_LEVELS, SCALE)
            if prev is not None:
                O.search_by_bow(prev, desc, TH_LOW)
            prev = desc
            done += 1
            if time.perf_counter() - t0 > seconds:
                break
        return done, time.perf_counter() - t0

    n_all, t_all = run(cores, 16, budget_s * 0.4)             # ~20 core-seconds on a 16-core box
    n_4, t_4 = run(min(4, cores), 4, budget_s * 0.4)
    half = max(cores // 2, 1)
    O.set_threads(half)
    rng = np.random.default_rng(11)
    a = unit_rows(rng, 1000, 256); b = unit_rows(rng, 1000, 256)
    t0 = time.perf_counter(); O.search_by_bow(a, b, TH_LOW); t_match = time.perf_counter() - t0
    t0 = time.perf_counter(); O.search_for_triangulation(a, b, TH_HIGH); t_tri = time.perf_counter() - t0
    db = unit_rows(rng, 10000, 4096)                              # 164 MB: does not fit the host's last-level cache, like the real scan
    O.db_scores(db[0], db[:64])
    t0 = time.perf_counter(); O.db_scores(db[0], db); t_db = time.perf_counter() - t0
    wq = windowed_inputs()
    O.set_threads(min(8, cores))
    O.match_candidates(*wq)
    t_w = []
    for _ in range(20):
        t0 = time.perf_counter(); O.match_candidates(*wq); t_w.append(time.perf_counter() - t0)
    return {"value": n_all / t_all, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_all} frames 752x480 extract (4 levels, 1000 kpts) + {max(n_all - 1, 0)} SearchByBoW matches, oracle/libhfnet_oracle.so, {cores} OpenMP threads",
            "variants": {"extract_match_4_threads_frames_per_s": n_4 / t_4, "threads_4": min(4, cores),
                         "search_by_bow_1000x1000_ms": t_match * 1e3, "search_for_triangulation_1000x1000_ms": t_tri * 1e3,
                         "db_scan_10000x4096_ms": t_db * 1e3, "matcher_db_threads": half,
                         "windowed_candidates_8_threads_us": float(np.median(t_w)) * 1e6, "windowed_candidates_threads": min(8, cores)}}


def verify_last_chunk(torch, pipe, weights_path, cur_frames, prev_frames, which, threads):
    """The checker, outside the timed region: outputs of the LAST TIMED chunk as the pipeline left them in HBM -- keypoints,
    descriptors, global descriptor, and the SearchByBoW matches / distances / count against the predecessor frame -- compared bit
    for bit with the oracle (oracle/, the CPU restatement) for the chunk frames `which`.  Frame 0's predecessor is the last frame
    of the chunk before.  Returns {"frames", "equal", "checked", "mismatch"}; bench.py exits non-zero when equal is false."""
    from oracle import oracle as O
    O.build()
    O.set_threads(threads)
    m = O.Model(weights_path)
    B, nb = pipe.B, pipe.n_buf
    s0 = ((pipe.cur - 1) % nb) * B                       # buffer block the last run_chunk wrote
    imgs = host(cur_frames)
    prev_last = host(prev_frames[B - 1])
    cache = {}

    def ref(f):                                            # f = -1: the previous chunk's last frame
        if f not in cache:
            cache[f] = m.extract(prev_last if f < 0 else imgs[f], N_FEAT, THRESH, N_LEVELS, SCALE)
        return cache[f]

    bad = []
    for f in which:
        rn, rk, rd, rg, _ = ref(f)
        slot = s0 + f
        n = int(pipe.n_rows[slot].item())
        k = host(pipe.kps[slot]); d = host(pipe.desc[slot]); g = host(pipe.glob[f])
        ok = {"count": n == rn}
        if n == rn:
            ok["kps_xy_response"] = all(np.array_equal(k[:n, j], rk[name]) for j, name in enumerate(("x", "y", "response")))
            ok["kps_octave"] = np.array_equal(k[:n, 3].view(np.int32), rk["octave"])
            ok["descriptors"] = np.array_equal(d[:n], rd)
        ok["global"] = np.array_equal(g, rg)
        qn, _, qd, _, _ = ref(f - 1)                       # query = the predecessor frame (Pipeline._default_pairs)
        rc, rm, rdist = O.search_by_bow(qd, rd, TH_LOW)
        ok["match_count"] = int(pipe.mcnt[f].item()) == rc
        ok["matches"] = np.array_equal(host(pipe.match[f])[:qn], rm)
        ok["distances"] = np.array_equal(host(pipe.mdist[f])[:qn], rdist)
        bad += [f"frame {f}: {name}" for name, v in ok.items() if not v]
    return {"frames": list(which), "equal": not bad, "mismatch": bad,
            "checked": "keypoints (x, y, response, octave), descriptors, global descriptor, SearchByBoW matches + distances + count vs the predecessor "
                       "frame, np.array_equal against oracle/libhfnet_oracle.so, read from the device buffers of the last timed chunk"}


# ------------------------------------------------------------------------------------------------ headline pipeline
class Pipeline:
    """Device-resident extract + match of frame chunks: one hfnet_extractor_extract_batch(on_device) and one
    hfnet_match_search_by_bow_batch(on_device) call per chunk, no host synchronisation in between (keypoint counts stay on
    the device; the matcher stream waits for the extraction by event, hfnet_engine_fence orders buffer reuse)."""

    def __init__(self, torch, capi, eng, dev, width, height, chunk, n_buf=3):
        self.torch, self.capi, self.eng, self.dev, self.B, self.n_buf = torch, capi, eng, dev, chunk, n_buf
        self.w, self.h = width, height
        self.ext = capi.Extractor(eng, width, height, N_FEAT, THRESH, SCALE, N_LEVELS, max_batch=chunk)
        B = chunk
        self.kps = torch.zeros((n_buf * B, N_FEAT, 4), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((n_buf * B, N_FEAT, 256), dtype=torch.float32, device=dev)
        self.n_rows = torch.zeros((n_buf * B,), dtype=torch.int32, device=dev)      # keypoints per set, written by the extractor
        self.glob = torch.zeros((B, eng.global_dim), dtype=torch.float32, device=dev)
        self.match = torch.zeros((B, N_FEAT), dtype=torch.int32, device=dev)
        self.mdist = torch.zeros((B, N_FEAT), dtype=torch.float32, device=dev)
        self.mcnt = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.cur = 0
        self._pairs = {}
        self.L = capi.lib()

    def run_chunk(self, d_images, n_frames, qset=None, tset=None, n_pairs=None, isolate=False):
        """d_images: device tensor [>= n_frames, h, w] u8.  qset / tset: device int32 slot lists of the match pairs
        (default: every frame against its predecessor, the first one against the previous chunk's last frame).
        isolate: host-synchronise between the extraction and the match and after it (profiling pass only: the matcher has
        its own stream and would otherwise share the GPU with the next chunk's first kernels)."""
        B, s0 = self.B, self.cur * self.B
        self.ext.extract_batch_device(n_frames, d_images.data_ptr(), self.w, self.w * self.h, self.kps[s0].data_ptr(), self.desc[s0].data_ptr(),
                                      self.glob.data_ptr(), self.n_rows[s0:].data_ptr())
        # the next extraction overwrites the buffer the PREVIOUS chunk's matches still read: fence it behind them
        self.eng.fence()
        if isolate:
            self.eng.synchronize()
        if qset is None:
            qset, tset, n_pairs = self._default_pairs(s0, n_frames)
        if n_pairs and not os.environ.get("BENCH_DEV_NO_MATCH"):      # (development switch: such a line is marked invalid below)
            st = self.L.hfnet_match_search_by_bow_batch(self.eng.h, n_pairs, C.c_void_p(self.desc.data_ptr()), C.c_size_t(N_FEAT * 256),
                                                        C.c_void_p(self.n_rows.data_ptr()), self.n_buf * B, C.c_void_p(qset.data_ptr()),
                                                        C.c_void_p(tset.data_ptr()), N_FEAT, 256, C.c_float(TH_LOW), C.c_void_p(self.match.data_ptr()),
                                                        C.c_void_p(self.mdist.data_ptr()), C.c_void_p(self.mcnt.data_ptr()), 1)
            if st != 0:
                raise RuntimeError(self.capi.last_error())
        if isolate:
            self.eng.synchronize()
        self.cur = (self.cur + 1) % self.n_buf
        return self.n_rows[s0:s0 + n_frames]

    def _default_pairs(self, s0, n):
        if (s0, n) not in self._pairs:
            t = self.torch.arange(s0, s0 + n, dtype=self.torch.int32, device=self.dev)
            q = ((t.to(self.torch.int64) - 1) % (self.n_buf * self.B)).to(self.torch.int32)
            self.torch.cuda.synchronize()                 # (torch's stream is not ordered with the library's: finish the lists first)
            self._pairs[(s0, n)] = (q, t, n)
        return self._pairs[(s0, n)]

    def close(self):
        self.ext.close()


def profile_pass(eng, pipe, frames, chunk, reps=10):
    """dedicated profiling pass: HIP events around EVERY launch, one stream (the engine serialises the global branch while
    an unfiltered profile is on) and a host synchronisation around the matcher call (it has its own stream), so no kernel
    shares the GPU with another.  Returns {name: (launches, total_ms)} per chunk."""
    eng.synchronize()
    eng.profile_reset(); eng.profile_filter(None); eng.profile_enable(True)
    for i in range(reps):
        pipe.run_chunk(frames[i % len(frames)], chunk, isolate=True)
    eng.synchronize()
    prof = eng.profile()
    eng.profile_enable(False)
    return {k: (v[0] / reps, v[1] / reps) for k, v in prof.items()}


def launch_class(name: str) -> str:
    """what bounds a launch by construction: the classes of roofline.classes"""
    if name.startswith("block_L") or name == "stem_block_L02":
        return "fused_block"
    if name.startswith(("conv3x3_", "pointwise_de", "det_tail")):
        return "head_gemm"
    if name.startswith(("expand_L", "project_L", "pointwise_memberships", "fc_l2")):
        return "global_gemm"
    if name.startswith("match_"):
        return "match"
    return "hbm_stream"      # depthwise_L15-18, nms, softmax_d2s, sample, pyramid_resize, topk, vlad, softmax_memberships, stem


def roofline_table(prof, work, chunk_seconds_sum, match_bf16=False):
    """(table, classes).  classes: per launch class its share of the profiled chunk, time, algorithmic FLOP and bytes and the
    fraction of the roof that bounds the class (hbm_stream: bytes once / time / 8 TB/s; the others: FLOP / time / f32 MFMA peak).
    table: time-weighted rows of every launch in a class that carries >= 2 % of the chunk (own share >= 0.25 %), so the
    memory-bound launches, each of which is small, are listed as well."""
    cls = {}
    for name, (launches, ms) in prof.items():
        if name not in work or launches <= 0:
            continue
        c = cls.setdefault(launch_class(name), {"us": 0.0, "flop": 0.0, "bytes": 0.0, "executed": 0.0, "launches": 0.0})
        flop, byts, executed = work[name]
        c["us"] += ms * 1e3; c["flop"] += flop; c["bytes"] += byts; c["executed"] += executed; c["launches"] += launches
    for k, c in cls.items():
        sec = c["us"] * 1e-6
        c["share"] = sec / chunk_seconds_sum
        if k == "hbm_stream":
            c.update({"bound": "hbm", "achieved": c["bytes"] / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"})
        elif k == "match" and match_bf16:
            # SearchByBoW screens on the bf16 matrix pipe: three bf16 products per f32 product (split operands), so the executed
            # work is 3 x the padded f32 count, priced against the bf16 roof; "achieved" stays the algorithmic f32-equivalent rate
            c.update({"bound": "mfma_bf16", "achieved": c["flop"] / sec / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac_executed": 3.0 * c["executed"] / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS})
        else:
            c.update({"bound": "mfma", "achieved": c["flop"] / sec / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac_executed": c["executed"] / sec / 1e12 / MFMA_F32_PEAK_TFLOPS})
        c["frac"] = c["achieved"] / c["peak"]
    rows = []
    for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        share = ms * 1e-3 / chunk_seconds_sum
        if name not in work or launches <= 0 or share < 0.0025 or cls[launch_class(name)]["share"] < 0.02:
            continue
        flop, byts, executed = work[name]
        sec = ms * 1e-3 / launches
        e = roofline_entry(flop / launches, byts / launches, sec)
        if launch_class(name) == "hbm_stream" and e["bound"] != "hbm":     # priced against the roof of its class
            e = {"bound": "hbm", "achieved": byts / launches / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            e["frac"] = e["achieved"] / e["peak"]
        if launch_class(name) == "match" and match_bf16:
            e = {"bound": "mfma_bf16", "achieved": flop / launches / sec / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s"}
            e["frac"] = e["achieved"] / e["peak"]
            fx = 3.0 * (executed / launches) / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS
        else:
            fx = (executed / launches) / sec / 1e12 / MFMA_F32_PEAK_TFLOPS if e["bound"] == "mfma" else None
        e.update({"name": name, "class": launch_class(name), "us": sec * 1e6, "launches_per_chunk": launches, "share": share, "flop": flop / launches,
                  "bytes": byts / launches, "frac_executed": fx})
        rows.append(e)
    return rows, cls


LAYER_GRANULAR = ("stem", "pyramid_resize", "expand_L", "depthwise_L", "project_L", "conv3x3_desc", "pointwise_desc", "l2norm_desc", "conv3x3_det",
                  "pointwise_det", "softmax_d2s", "nms", "topk", "sample", "pointwise_memberships", "softmax_memberships", "vlad", "vlad_norm", "fc_l2")


def extractor_algorithmic_bytes(work):
    """bytes of the extractor when every LAYER reads its input once, writes its output once and reads its
    weights once (the unfused, dense-head network; ~1 GB per 752x480 frame) -- the numerator of the extractor-level HBM fraction"""
    return sum(v[1] for k, v in work.items() if k.startswith(LAYER_GRANULAR) and not k.endswith("_taps"))


# ------------------------------------------------------------------------------------------------ sub-configs
def config_latency(capi, eng, n=100):
    """config 2, per-frame view: one HFextractor call through the host-pointer entry point (upload, 4 levels + global
    descriptor, download, host sync), then SearchByBoW against the previous frame.  The caller owns the result buffers and
    reuses them from frame to frame, the way a C++ caller of the ABI holds its cv::Mat headers (two sets: the previous
    frame's descriptors are the matcher's query)."""
    ext = capi.Extractor(eng, W_IMG, H_IMG, N_FEAT, THRESH, SCALE, N_LEVELS, max_batch=1)
    frames = make_frames(8, 0)
    bufs = [ext.output_buffers(), ext.output_buffers()]
    for i, f in enumerate(frames[:4]):
        ext.extract(f, bufs[i & 1])
    t_ext, t_both, prev = [], [], None
    for i in range(n):
        f = frames[i % len(frames)]
        t0 = time.perf_counter()
        nk, _, desc, _, _ = ext.extract(f, bufs[i & 1])
        t1 = time.perf_counter()
        if prev is not None:
            eng.search_by_bow(prev, desc, TH_LOW)
        t2 = time.perf_counter()
        t_ext.append(t1 - t0); t_both.append(t2 - t0); prev = desc
    store = capi.Store(eng, 2, N_FEAT)       # descriptors kept on the GPU: device-to-device into a two-slot store, match by slot
    t_dev = []
    for i in range(n):
        t0 = time.perf_counter()
        ext.extract(frames[i % len(frames)], bufs[i & 1])
        store.put_extracted(i & 1, ext, 0)
        if i:
            store.search_by_bow([(1 - (i & 1), i & 1)], TH_LOW)
        t_dev.append(time.perf_counter() - t0)
    store.close(); ext.close()
    med = lambda v: float(np.median(v)) * 1e3
    return {"workload": "752x480, 4 levels, 1000 keypoints, one frame per call, host pointers, caller-owned result buffers", "extract_ms_median": med(t_ext),
            "extract_plus_match_ms_median": med(t_both[1:]), "extract_plus_store_match_ms_median": med(t_dev[1:]), "keypoints": int(nk),
            "frames_per_s_unpipelined": 1e3 / med(t_dev[1:])}


def config_host_io(capi, eng, chunk, chunks_per_call=16, reps=3):
    """the batch path with host buffers on both sides: images go up, keypoints + descriptors + global descriptors come
    down (what the reference's extraction time includes, HFNetRTModel.cc:128,134).  A call of several chunks runs as a
    double-buffered pipeline (copies overlap the compute).  The frame-to-frame match runs on device copies the extractor
    leaves in an attached hfnet_store; only the matches come down.  Three legs:
      sequential   one host thread: extract_batch, then the match calls of its frames (pageable numpy arrays, pinned staging inside the library)
      pageable     two host threads on the one engine, as the SLAM system drives it (Tracking extracts while LocalMapping matches): the
                   matches of call i run while call i + 1 extracts into the other half of the store
      registered   the same with the caller's arrays registered for DMA (hfnet_host_register): no staging copies on the host"""
    import threading
    n = chunk * chunks_per_call
    ext = capi.Extractor(eng, W_IMG, H_IMG, N_FEAT, THRESH, SCALE, N_LEVELS, max_batch=chunk)
    store = capi.Store(eng, 2 * n, N_FEAT)
    imgs = np.concatenate([make_frames(min(n, 512), 0)] * -(-n // 512))[:n]      # (512 different frames, repeated: the path is data-independent)
    ext.attach_store(store, 0)
    out = ext.extract_batch(imgs)                              # (the result arrays are reused: a caller's buffers are paged in)
    ext.extract_batch(imgs, out)                               # (warm: clocks, page tables of the 2.4 GB of result arrays)
    pairs = [(f - 1, f) for f in range(1, n)]
    t_e, t_all = [], []
    for r in range(reps):
        t0 = time.perf_counter()
        ext.extract_batch(imgs, out)
        t1 = time.perf_counter()
        for p0 in range(0, len(pairs), chunk):
            store.search_by_bow(pairs[p0:p0 + chunk], TH_LOW)
        t_all.append(time.perf_counter() - t0); t_e.append(t1 - t0)

    def overlapped(calls):
        """`calls` extract_batch calls of n frames; a second thread matches call i's frames (each against its predecessor, the first one
        against the previous call's last frame) while call i + 1 extracts"""
        done = []

        def match_call(i):
            base = (i & 1) * n
            pr = [((base + f - 1) % (2 * n), base + f) for f in range(0 if i else 1, n)]
            for p0 in range(0, len(pr), 4 * chunk):              # (512 pairs per call: a call's uploads, downloads and host synchronisation amortise)
                store.search_by_bow(pr[p0:p0 + 4 * chunk], TH_LOW)
            done.append(i)

        worker = None
        t0 = time.perf_counter()
        for i in range(calls):
            ext.attach_store(store, (i & 1) * n)
            ext.extract_batch(imgs, out)
            if worker is not None:
                worker.join()
            worker = threading.Thread(target=match_call, args=(i,))
            worker.start()
        worker.join()
        assert len(done) == calls
        return calls * n / (time.perf_counter() - t0)

    overlapped(1)
    fps_pageable = overlapped(reps)
    bufs = [imgs, out[0], out[1], out[2], out[3]]
    t0 = time.perf_counter()
    for b in bufs:
        capi.host_register(b)
    t_reg = time.perf_counter() - t0
    try:
        overlapped(1)
        fps_registered = overlapped(reps)
    finally:
        for b in bufs:
            capi.host_unregister(b)
    ext.attach_store(None)
    store.close(); ext.close()
    return {"workload": f"752x480, {n} frames per call in chunks of {chunk}, host buffers in and out (numpy arrays), matches by slot on device-resident copies",
            "extract_frames_per_s": n / float(np.median(t_e)), "extract_plus_match_frames_per_s_one_thread": n / float(np.median(t_all[1:])),
            "extract_plus_match_frames_per_s": fps_pageable, "extract_plus_match_frames_per_s_registered": fps_registered,
            "legs": "one_thread: extract_batch then its match calls, pageable arrays (round 3's definition); the default: a second host thread matches call "
                    "i's frames while call i + 1 extracts (two SLAM threads on one engine), pageable arrays staged through pinned blocks inside the library; "
                    "registered: the same with the arrays registered for DMA (hfnet_host_register, no staging copies)",
            "register_seconds": t_reg, "registered_bytes": int(sum(b.nbytes for b in bufs))}


TOLERANCE_OPTIONS = ("scores_bf16x3", "desc_bf16x3", "global_bf16x3")
BF16X3_HBM_FILES = ["profiles/r06_chunk_traffic_bf16x3_b{batch}.json"]


def _timed_steps(torch, pipe, eng, frames, chunks_per_step, steps):
    n_sets = len(frames)
    for c in range(chunks_per_step):
        pipe.run_chunk(frames[c % n_sets], pipe.B)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    i = 0
    for _ in range(steps):
        for _ in range(chunks_per_step):
            pipe.run_chunk(frames[i % n_sets], pipe.B); i += 1
    eng.synchronize(); torch.cuda.synchronize()
    return time.perf_counter() - t0, i


def config_bf16x3(torch, capi, eng, dev, frames, B, chunks_per_step, steps, weights_path, work):
    """The headline workload in TOLERANCE MODE: engine options scores_bf16x3 + desc_bf16x3 + global_bf16x3 -- every GEMM-shaped stage that has a
    split-bf16 form (fused blocks 3-7 and 9-14, the detector head's 3x3 conv, the descriptor head at the tap cells, the 1x1 convolutions of
    layers 15-18) on the bf16 matrix pipe, two bf16 pieces per f32 operand, three products.  `value` stays the exact path.  The score map is a
    tolerance tensor in this mode, so the check is: NMS / threshold scan / top-K EXACT on the score map the device produced (the oracle's
    hfo_simple_nms + hfo_select_keypoints on the dense scores read back from the device == the device's keypoints, array_equal), keypoint-set
    overlap with the oracle's exact selection, descriptors of the common keypoints and the global descriptor within the stated tolerance.
    Also measured: the two options that touch no index alone (desc + global; keypoints then equal the oracle's bit for bit)."""
    from oracle import oracle as O
    TOL, TOL_G, TOL_S = 2e-5, 1e-4, 5e-4            # include/hfnet_hip.h (full tolerance mode): descriptors / global descriptor / dense scores
    saved = {o: eng.get_option(o) for o in TOLERANCE_OPTIONS}
    n_sets = len(frames)
    res = {}
    try:
        # ---- (i) desc + global only: every index exact
        eng.set_option("desc_bf16x3", 1); eng.set_option("global_bf16x3", 1); eng.set_option("scores_bf16x3", 0)
        pipe = Pipeline(torch, capi, eng, dev, W_IMG, H_IMG, B)
        try:
            for c in range(2 * chunks_per_step):
                pipe.run_chunk(frames[c % n_sets], B)
            elapsed, _ = _timed_steps(torch, pipe, eng, frames, chunks_per_step, steps)
            res["frames_per_s_indices_exact"] = B * chunks_per_step * steps / elapsed
        finally:
            pipe.close()
        # ---- (ii) the whole tolerance pipeline
        eng.set_option("scores_bf16x3", 1)
        pipe = Pipeline(torch, capi, eng, dev, W_IMG, H_IMG, B)
        try:
            for c in range(2 * chunks_per_step):
                pipe.run_chunk(frames[c % n_sets], B)
            eng.synchronize()
            prof = profile_pass(eng, pipe, frames, B, reps=4)
            chunk_ms = sum(v[1] for v in prof.values())
            rows = []
            for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                if not name.endswith("_bf16x3") or name[:-7] not in work or launches <= 0:
                    continue
                flop = work[name[:-7]][0]
                rows.append({"name": name, "us": ms * 1e3, "share": ms / chunk_ms, "f32_equivalent_TFLOPs": flop / (ms * 1e-3) / 1e12,
                             "frac_bf16_roof_3_products": 3.0 * flop / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS})
            elapsed, i = _timed_steps(torch, pipe, eng, frames, chunks_per_step, steps)
            # the dominant split-bf16 launch, timed live over a second timed region (HIP events on its stream)
            dom = max(rows, key=lambda r: r["us"])["name"] if rows else None
            live = None
            if dom:
                eng.profile_reset(); eng.profile_filter(dom); eng.profile_enable(True)
                _, i = _timed_steps(torch, pipe, eng, frames, chunks_per_step, max(1, steps // 2))
                pr = eng.profile().get(dom, (0, 0.0))
                eng.profile_enable(False); eng.profile_filter(None)
                if pr[0]:
                    live = pr[1] / pr[0] * 1e-3
            # ---- the checker (outside the timed regions): frames spread over the LAST chunk
            m = O.Model(weights_path)
            s0 = ((pipe.cur - 1) % pipe.n_buf) * B
            imgs = host(frames[(i - 1) % n_sets])
            dense = pipe.ext.tap(22, B)
            sf = pipe.ext.tables()[0]
            import hfnet_slam_amd.spec as S
            budget = S.features_per_level(N_FEAT, N_LEVELS, SCALE)
            checked = sorted(set([0, 1, B - 1] + list(range(0, B, max(1, B // 8)))))
            nms_topk_exact, overlap, total, dmax, gmax, smax = True, 0, 0, 0.0, 0.0, 0.0
            for f in checked:
                n = int(pipe.n_rows[s0 + f].item())
                k = host(pipe.kps[s0 + f]); d = host(pipe.desc[s0 + f]); g = host(pipe.glob[f])
                want = []
                for l, kb in enumerate(budget):
                    kp = O.select_keypoints(O.simple_nms(dense[l][f], 4, 2), THRESH, kb)
                    want.append(np.stack([kp["x"] * np.float32(sf[l]), kp["y"] * np.float32(sf[l]), kp["response"],
                                          np.full(len(kp), l, np.int32).view(np.float32)], axis=1) if len(kp) else np.zeros((0, 4), np.float32))
                want = np.concatenate(want)
                nms_topk_exact = nms_topk_exact and n == len(want) and np.array_equal(k[:n].view(np.int32), want.view(np.int32))
                rn, rk, rd, rg, _ = m.extract(imgs[f], N_FEAT, THRESH, N_LEVELS, SCALE)
                pos = {(int(o), float(a), float(b)): j for j, (o, a, b) in enumerate(zip(rk["octave"], rk["x"], rk["y"]))}
                total += rn
                for j in range(n):
                    r = pos.get((int(k[j, 3:4].view(np.int32)[0]), float(k[j, 0]), float(k[j, 1])))
                    if r is not None:
                        overlap += 1
                        dmax = max(dmax, float(np.abs(d[j].astype(np.float64) - rd[r]).max()))
                gmax = max(gmax, float(np.abs(g.astype(np.float64) - rg).max()))
                if f == checked[0]:
                    # dense scores of this frame's level 0 against the oracle's own tap
                    hc, wc = H_IMG // 8 * 8, W_IMG // 8 * 8
                    ref_dense = m.run_local(imgs[f], taps=(O.TAP_SCORES_DENSE,))["taps"][O.TAP_SCORES_DENSE]
                    smax = float(np.abs(dense[0][f].astype(np.float64) - ref_dense.reshape(hc, wc)).max())
            ok = bool(nms_topk_exact and overlap >= 0.99 * total and dmax <= TOL and gmax <= TOL_G and smax <= TOL_S)
            res.update({
                "workload": "the headline workload in tolerance mode: engine options scores_bf16x3 = desc_bf16x3 = global_bf16x3 = 1 (fused blocks 3-7 / 9-14, "
                            "detector 3x3, descriptor head at the tap cells, 1x1 convolutions of layers 15-18 on split-bf16 operands, three products, bf16 matrix "
                            "pipe); NMS / top-K exact on the device's score map, float outputs within the stated tolerances",
                "frames_per_s": B * chunks_per_step * steps / elapsed, "steps": steps, "profiled_chunk_ms_single_stream": chunk_ms,
                "bf16x3_launches": rows, "bf16x3_share_of_chunk": sum(r["share"] for r in rows),
                "verified": {"frames": checked, "nms_topk_equal_oracle_on_device_scores": bool(nms_topk_exact), "keypoint_overlap_with_exact": overlap / max(total, 1),
                             "descriptor_max_abs_dev_common_keypoints": dmax, "global_max_abs_dev": gmax, "dense_score_max_abs_dev_level0_frame0": smax,
                             "tolerance": TOL, "tolerance_global": TOL_G, "tolerance_scores": TOL_S, "within_tolerance": ok}})
            if dom:
                flop = work[dom[:-7]][0]
                sec = live if live else next(r["us"] for r in rows if r["name"] == dom) * 1e-6
                hb, hf = None, None
                for pat in BF16X3_HBM_FILES:
                    try:
                        with open(os.path.join(ROOT, pat.format(batch=B))) as fh:
                            t = json.load(fh)
                        hb, hf = t["chunk_hbm_bytes"], pat.format(batch=B)
                        break
                    except (OSError, KeyError, ValueError):
                        continue
                gemms = [r for r in rows if r["name"].startswith(("conv3x3_", "pointwise_", "det_tail"))]
                lg = max(gemms, key=lambda r: r["us"]) if gemms else None
                res["roofline_bf16x3"] = {"bound": "mfma_bf16", "kernel": dom, "achieved": 3.0 * flop / sec / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "largest_gemm": ({"kernel": lg["name"], "us": lg["us"], "frac": lg["frac_bf16_roof_3_products"]} if lg else None),
                                          "kernel_note": "the largest launch of the mode is a fused inverted-residual block: its 1x1 convolutions are on the bf16 matrix pipe, what bounds it is the "
                                                         "depthwise / ReLU6 / operand-split work on the vector ALU (profiles/: VALU-busy 0.9); largest_gemm is the largest GEMM-shaped launch",
                                          "frac": 3.0 * flop / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, "avg_launch_us": sec * 1e6,
                                          "timing": "HIP events on the kernel's stream over a timed region of this mode" if live else "single-stream profiling pass",
                                          "note": "achieved = 3 bf16 products per f32 product x the launch's algorithmic FLOP / its average duration",
                                          "chunk_hbm_gbs": (hb / (chunk_ms * 1e-3) / 1e9) if hb else None, "chunk_hbm_frac": (hb / (chunk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hb else None,
                                          "chunk_hbm_source": (hf + ": sum of (2 x FETCH_SIZE + WRITE_SIZE) over the launches of one call, rocprofv3 --pmc passes of this mode "
                                                               "(committed; not measured in this run), over the single-stream chunk time of this run") if hf else None}
        finally:
            pipe.close()
        return res
    finally:
        for o, v in saved.items():
            eng.set_option(o, v)


def config_sparse(torch, capi, dev, B, chunks_per_step, steps, rank=0):
    """The headline workload on weights whose detector lets FEW cells through (weights.synthetic_weights(dustbin_bias=15)): the coarse pyramid
    levels fall short of their budget and the candidates cluster -- the regime trained weights live in; with the default seeded weights every
    cell is a candidate and top-K is saturated at every level.  Exact mode; sampled frames of the last chunk against the oracle, bit for bit."""
    from hfnet_slam_amd import weights
    wp = os.path.join(tempfile.gettempdir(), f"hfnet_bench_seed7_dustbin15_rank{rank}.hfw")
    weights.save(wp, weights.synthetic_weights(7, dustbin_bias=15.0))
    eng = capi.Engine(wp, dev.index or 0)
    try:
        n_sets = 2
        frames = [to_device(torch, np.concatenate([make_frames(B // 2, s * B, "uniform"), make_frames(B - B // 2, s * B + B // 2, "natural")]), dev) for s in range(n_sets)]
        pipe = Pipeline(torch, capi, eng, dev, W_IMG, H_IMG, B)
        try:
            for c in range(chunks_per_step):
                pipe.run_chunk(frames[c % n_sets], B)
            elapsed, i = _timed_steps(torch, pipe, eng, frames, chunks_per_step, steps)
            s0 = ((pipe.cur - 1) % pipe.n_buf) * B
            n = host(pipe.n_rows[s0:s0 + B])
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            ver = verify_last_chunk(torch, pipe, wp, frames[(i - 1) % n_sets], frames[(i - 2) % n_sets], sorted({0, 1, B // 2, B - 1}), max(1, min(32, cores)))
            return {"workload": "the headline workload (exact mode) on seeded weights with dustbin bias 15: half uniform, half natural-ish frames per call; "
                                "the coarse levels are short of their budget, candidates cluster",
                    "frames_per_s": B * chunks_per_step * steps / elapsed, "steps": steps,
                    "keypoints_per_frame": {"min": int(n.min()), "mean": float(n.mean()), "max": int(n.max()), "budget": N_FEAT},
                    "verified": {k: ver[k] for k in ("equal", "frames", "mismatch") if k in ver}}
        finally:
            pipe.close()
    finally:
        eng.close()


def config_tracking(capi, eng, n_feat, frames_n=400):
    """config 3 (TUM-VI corridor-style loop) one frame at a time through the host-pointer entry points, keyframes'
    descriptor blocks resident in an hfnet_store: every frame extract + SearchByBoW vs the last frame; every 5th frame a
    keyframe: database add + DetectNBestCandidates scan + SearchForTriangulation against the 30 most recent keyframes."""
    W = H = 512
    ext = capi.Extractor(eng, W, H, n_feat, THRESH, SCALE, N_LEVELS, max_batch=1)
    db = capi.Database(eng, frames_n // 5 + 8, eng.global_dim)
    store = capi.Store(eng, frames_n // 5 + 8, n_feat)       # keyframe slots; the last two slots hold the current / previous frame
    F0 = frames_n // 5 + 6
    imgs = make_frames(16, 0, w=W, h=H)
    bufs = [ext.output_buffers(), ext.output_buffers()]
    for i in range(3):
        ext.extract(imgs[i], bufs[i & 1])
    t_frame, t_kf, n_kf = [], [], 0
    t_all0 = time.perf_counter()
    for i in range(frames_n):
        t0 = time.perf_counter()
        _, _, _, g, _ = ext.extract(imgs[i % len(imgs)], bufs[i & 1])
        store.put_extracted(F0 + (i & 1), ext, 0)
        if i:
            store.search_by_bow([(F0 + 1 - (i & 1), F0 + (i & 1))], TH_LOW)
        t1 = time.perf_counter()
        t_frame.append(t1 - t0)
        if i % 5 == 0:
            if n_kf:
                db.query(g, 0)
                store.put_extracted(n_kf, ext, 0)
                store.search_for_triangulation([(n_kf, j) for j in range(max(0, n_kf - 30), n_kf)], TH_HIGH)
            else:
                store.put_extracted(0, ext, 0)
            db.add(n_kf, g); n_kf += 1
            t_kf.append(time.perf_counter() - t1)
    wall = time.perf_counter() - t_all0
    store.close(); db.close(); ext.close()
    med = lambda v: float(np.median(v)) * 1e3
    return {"frame_ms_median": med(t_frame), "keyframe_extra_ms_median": med(t_kf[31:] if len(t_kf) > 40 else t_kf[1:]), "keyframes": n_kf,
            "frames": frames_n, "frames_per_s_whole_loop": frames_n / wall}


def windowed_inputs():
    """Tracking-sized input of the windowed matchers' candidate loop: ~1000 MapPoint queries, each against the 5-40 keypoints
    its grid lookup returned (Matcher.cc:40-210); seeded, shared by the device leg (configs["3"]) and the CPU leg (cpu_baseline)"""
    rng = np.random.default_rng(21)
    nq, nt = 1000, 1000
    train = unit_rows(rng, nt, 256)
    query = train[rng.integers(0, nt, nq)] + 0.05 * rng.standard_normal((nq, 256)).astype(np.float32)
    counts = rng.integers(5, 41, nq)
    off = np.zeros(nq + 1, np.int32); off[1:] = np.cumsum(counts)
    idx = rng.integers(0, nt, int(off[-1])).astype(np.int32)
    lvl = rng.integers(0, 4, nt).astype(np.int32)
    return query, train, lvl, off, idx


def config_windowed_candidates(capi, eng, reps=30):
    """SURVEY.md 8f rank 4, the question NOTEBOOK.md 4.3 answers with this number: is the candidate loop of the windowed
    matchers worth a device call?  hfnet_match_candidates through the host-pointer entry point (queries, train descriptors and
    the ragged candidate lists go up, five result arrays come down); the CPU side of the comparison -- the oracle's restatement
    of the reference loop on 8 host threads, same inputs -- is cpu_baseline.variants.windowed_candidates_8_threads_us."""
    query, train, lvl, off, idx = windowed_inputs()
    eng.match_candidates(query, train, lvl, off, idx)
    t_dev = []
    for _ in range(reps):
        t0 = time.perf_counter(); eng.match_candidates(query, train, lvl, off, idx); t_dev.append(time.perf_counter() - t0)
    return {"workload": f"{len(query)} queries x 5-40 candidates of {len(train)} train rows x 256 ({int(off[-1])} distances), host pointers in and out",
            "device_call_us_median": float(np.median(t_dev)) * 1e6}


def config_sequences(torch, pipe, dev, rank, world, dist, pool_frames=64, dry=False):
    """config 4: the 11 EuRoC sequences (27 049 frames) assigned to the ranks longest-first (hfnet_slam_amd/shard.py), each
    rank walks its sequences in frame order in chunks of the pipeline's chunk size; frame i is matched against frame i - 1
    of the same sequence.  Frames come from a pool of synthetic frames resident in HBM."""
    from hfnet_slam_amd import shard
    B = pipe.B
    plan = shard.assign_sequences(shard.EUROC_SEQUENCES, world)
    chunks = shard.sequence_chunks(plan[rank], shard.EUROC_SEQUENCES, B)
    pool = None if dry else to_device(torch, make_frames(pool_frames + B, 10_000 * (rank + 1)), dev)
    # pair lists of every chunk, built before the timed region (slots rotate with the pipeline's buffers)
    cur, todo = pipe.cur, []
    for ci, (name, f0, n) in enumerate(chunks):
        s0 = cur * B
        q, t = shard.chunk_pairs(f0, n, s0, (s0 - 1) % (pipe.n_buf * B))
        todo.append((n, torch.tensor(q, dtype=torch.int32, device=dev), torch.tensor(t, dtype=torch.int32, device=dev), len(q), (ci * B) % pool_frames))
        cur = (cur + 1) % pipe.n_buf
    if not dry:
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for n, q, t, npairs, off in todo:
        pipe.run_chunk(None if dry else pool[off:], n, q, t, npairs)
    pipe.eng.synchronize()
    if not dry:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mine = sum(c[2] for c in chunks)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([mine], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt)
        total = int(cnt.item())
    else:
        total = mine
    return {"workload": "11 EuRoC sequences, 27049 synthetic 752x480 frames, sequences assigned to GPUs longest-first, frame order kept, "
                        "extract + SearchByBoW vs the previous frame of the sequence", "frames": total, "sequences": sum(len(p) for p in plan),
            "gpus": world, "seconds": elapsed, "frames_per_s": total / elapsed, "rank0_sequences": plan[0], "scaling": "strong"}


def config_loop_closure(capi, eng, reps=10):
    """config 5: keyframe-database scans at 10 000 x 4096 (Q = 1 and Q = 64; warm: the 164 MB database stays in the 256 MB
    Infinity Cache between scans; cold: a 65 536-row = 1 GB database) and 32 SearchByBoW matches of 1000 x 1000 x 256.
    Kernel times are HIP-event times of the library's profiler."""
    eng.set_option("tri_screen_bf16", eng.get_option("tri_screen_bf16"))      # (forget what config 3's descriptor sets taught the screened path)
    N, DIM = 10000, 4096
    rng = np.random.default_rng(13)
    rows = unit_rows(rng, N, DIM)
    db = capi.Database(eng, N, DIM)
    for i in range(N):
        db.add(i, rows[i])
    qs = rows[rng.integers(0, N, 64)] + 0.003 * rng.standard_normal((64, DIM)).astype(np.float32)
    qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
    a = unit_rows(rng, 1000, 256)
    b = a[rng.permutation(1000)] + 0.02 * rng.standard_normal((1000, 256)).astype(np.float32)
    b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    sets = np.stack([a, b]).astype(np.float32)
    nr = np.array([1000, 1000], np.int32)
    db.query(qs[0]); db.query_batch(qs); eng.search_by_bow_batch(sets, nr, [(0, 1)] * 32, TH_LOW)       # warm-up
    eng.search_for_triangulation_batch(sets, nr, [(0, 1)] * 32, 0.75)
    eng.synchronize()
    ms = lambda p, k: p[k][1] / max(p[k][0], 1) if k in p else float("nan")
    # ONE query (its own profiler window: the screened form shares its launch names with the batched query): the default form for a database of this
    # size, and the exact f32 scan beside it
    def q1_window():
        db.query(qs[0]); eng.synchronize()
        eng.profile_reset(); eng.profile_filter(None); eng.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            db.query(qs[0])
        call = (time.perf_counter() - t0) / reps
        eng.synchronize()
        p = eng.profile(); eng.profile_enable(False)
        return p, call
    prof_q1, t_q1_call = q1_window()
    q1_screened = "db_screen" in prof_q1
    t_q1 = (ms(prof_q1, "db_qnorm") + ms(prof_q1, "db_screen")) if q1_screened else ms(prof_q1, "db_scores")
    min_rows = eng.get_option("db_screen_min_rows")
    eng.set_option("db_screen_min_rows", 0)
    prof_q1s, t_q1s_call = q1_window()
    eng.set_option("db_screen_min_rows", min_rows)
    eng.profile_reset(); eng.profile_filter(None); eng.profile_enable(True)
    for _ in range(reps):
        db.query_batch(qs)
    for _ in range(reps):
        eng.search_by_bow_batch(sets, nr, [(0, 1)] * 32, TH_LOW)
    for _ in range(reps):
        eng.search_for_triangulation_batch(sets, nr, [(0, 1)] * 32, 0.75)
    eng.synchronize()
    prof = eng.profile(); eng.profile_enable(False)
    ms = lambda p, k: p[k][1] / max(p[k][0], 1) if k in p else float("nan")
    q64 = next((k for k in ("db_screen", "db_scores_batch") if k in prof), "db_scores_batch")
    # the screened query = ONE launch, "db_screen" (k_db_sweep: the 8-bit copy streamed once against all queries, the bound test and the exact chain
    # for what is left in the same kernel); the queries' own preparation (db_qnorm) and the candidate filter (db_filter) are reported beside it
    t_q64 = ms(prof, q64)
    db.close()
    out = {"workload": "10000 x 4096 f32 database resident in HBM; 1000 x 1000 x 256 SearchByBoW and SearchForTriangulation x 32 pairs",
           "db_q1_form": "screened (the query's 8-bit fragments + one pass over the database's 8-bit copy + the exact chain)" if q1_screened else "exact f32 scan",
           "db_q1_warm_us": t_q1 * 1e3, "db_q1_call_us_incl_copies": t_q1_call * 1e6,
           "db_q1_scan_warm_us": ms(prof_q1s, "db_scores") * 1e3, "db_q1_scan_warm_GBps": N * DIM * 4 / (ms(prof_q1s, "db_scores") * 1e-3) / 1e9,
           "db_q1_scan_call_us_incl_copies": t_q1s_call * 1e6,
           "db_q64_kernel": q64, "db_q64_us": t_q64 * 1e3,
           "db_q64_prep_us": ms(prof, "db_qnorm") * 1e3 if "db_qnorm" in prof else None, "db_q64_filter_us": ms(prof, "db_filter") * 1e3 if "db_filter" in prof else None,
           "db_q64_screening": "8-bit steps of every vector at its own scale, one exact int32 product on v_mfma_i32_32x32x32_i8, a rigorous bound of the quantisation "
                               "error from the rows' scales and 1-norms: rules out every slot at distance >= 1 (score exactly 0); the others take the exact chain: "
                               "all outputs equal the exact scan's bits",
           # the roof the sweep is on: the HBM stream of the database's 8-bit copy (1 byte per element, read once per <= 64 queries) -- beside it the
           # launch moves the queries' fragments to every CU (256 KB each from L2) and writes 64 x 10 000 scores
           "db_q64_GBps_of_i8_copy": N * DIM / (t_q64 * 1e-3) / 1e9,
           "db_q64_frac_hbm_i8_copy": N * DIM / (t_q64 * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "match_32_pairs_us": ms(prof, "match_bow") * 1e3, "match_TFLOPs": 32 * 2 * 1000 * 1000 * 256 / (ms(prof, "match_bow") * 1e-3) / 1e12,
           "match_f32_equivalent_over_f32_roof": 32 * 2 * 1000 * 1000 * 256 / (ms(prof, "match_bow") * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
           # three bf16 products per f32-equivalent one, over the dense bf16 MFMA peak: the roof the screened matcher is on
           "match_frac_bf16_roof_executed": (3 if eng.options().get("match_screen_bf16") else 1) * 32 * 2 * 1000 * 1000 * 256 / (ms(prof, "match_bow") * 1e-3) / 1e12
                                            / (MFMA_BF16_PEAK_TFLOPS if eng.options().get("match_screen_bf16") else MFMA_F32_PEAK_TFLOPS),
           "match_screening": "split bf16 x 3 on v_mfma_f32_32x32x16_bf16 (matches exact)" if eng.options().get("match_screen_bf16") else "f32 MFMA",
           "triangulation_32_pairs_us": ms(prof, "match_tri") * 1e3,
           "triangulation_f32_equivalent_over_f32_roof": 32 * 2 * 1000 * 1000 * 256 / (ms(prof, "match_tri") * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
           "triangulation_screening": "threshold screen, split bf16 x 3 on the bf16 matrix pipe + exact chains for the listed products (matches exact)"
                                      if eng.options().get("tri_screen_bf16") else "none (f32 MFMA)"}
    # cold: a 1 GB database (4x the Infinity Cache): every scan streams it from HBM
    NC = 65536
    dbc = capi.Database(eng, NC, DIM)
    blk = unit_rows(rng, 2048, DIM)
    for i in range(NC):
        dbc.add(i, blk[i & 2047])
    def cold(reps_=6):
        dbc.query(qs[0])
        eng.synchronize()
        eng.profile_reset(); eng.profile_enable(True)
        for i in range(reps_):
            dbc.query(qs[i])
        eng.synchronize()
        p = eng.profile(); eng.profile_enable(False)
        return p
    # the default for a database of this size: the screened form (the query's fragments + ONE pass over the 8-bit copy); beside it the exact f32 scan
    prof = cold()
    screened_cold = "db_screen" in prof
    t_cold = (ms(prof, "db_qnorm") + ms(prof, "db_screen")) if screened_cold else ms(prof, "db_scores")
    eng.set_option("db_screen_min_rows", 0)
    prof_scan = cold()
    eng.set_option("db_screen_min_rows", min_rows)
    dbc.close()
    # the matcher at the headline call's size: 255 pairs (a 256-frame call matches every frame against its predecessor) -- the sweep form
    eng.search_by_bow_batch(sets, nr, [(0, 1)] * 255, TH_LOW); eng.synchronize()
    eng.profile_reset(); eng.profile_enable(True)
    for _ in range(reps):
        eng.search_by_bow_batch(sets, nr, [(0, 1)] * 255, TH_LOW)
    eng.synchronize()
    prof255 = eng.profile(); eng.profile_enable(False)
    out["match_255_pairs_us"] = ms(prof255, "match_bow") * 1e3
    out["match_255_frac_bf16_roof_executed"] = ((3 if eng.options().get("match_screen_bf16") else 1) * 255 * 2 * 1000 * 1000 * 256 / (ms(prof255, "match_bow") * 1e-3) / 1e12
                                                / (MFMA_BF16_PEAK_TFLOPS if eng.options().get("match_screen_bf16") else MFMA_F32_PEAK_TFLOPS))
    out["db_q1_cold_rows"] = NC
    out["db_q1_cold_us"] = t_cold * 1e3
    out["db_q1_cold_form"] = "screened (k_db_rowstat + k_db_quant of the query, k_db_sweep over the 8-bit copy)" if screened_cold else "exact f32 scan"
    out["db_q1_cold_GBps"] = NC * DIM * (1 if screened_cold else 4) / (t_cold * 1e-3) / 1e9
    out["db_q1_cold_frac_hbm"] = out["db_q1_cold_GBps"] / HBM_PEAK_GBS       # (of the bytes its form has to read: the 8-bit copy / the f32 rows)
    out["db_q1_cold_scan_us"] = ms(prof_scan, "db_scores") * 1e3
    out["db_q1_cold_scan_frac_hbm"] = NC * DIM * 4 / (ms(prof_scan, "db_scores") * 1e-3) / 1e9 / HBM_PEAK_GBS
    return out


# ------------------------------------------------------------------------------------------------ main
class DryPipeline:
    """--dry-ranks: the pipeline's interface with a timed no-op instead of the GPU work, so that the multi-rank control flow
    of this file (rendezvous, per-rank frame blocks, config 4's sequence plan and pair lists, barriers, the MAX over ranks,
    rank 0's JSON line) can be executed on a box without GPUs (tests/test_distributed.py).  Its numbers mean nothing."""

    class _Eng:
        def synchronize(self):
            pass

        def fence(self):
            pass

    def __init__(self, chunk, n_buf=3, seconds_per_frame=2e-5):
        self.B, self.n_buf, self.cur, self.eng, self.spf = chunk, n_buf, 0, DryPipeline._Eng(), seconds_per_frame
        self.frames_done = 0

    def run_chunk(self, d_images, n_frames, qset=None, tset=None, n_pairs=None, isolate=False):
        time.sleep(self.spf * n_frames)
        self.frames_done += n_frames
        self.cur = (self.cur + 1) % self.n_buf
        return None

    def close(self):
        pass


LINE_LIMIT = 8000      # bytes: the driver keeps an ~8 KB tail of stdout and parses the LAST line of it (BENCH_r04.parsed was null at 21 KB)


def compact_line(out: dict) -> dict:
    """the one stdout line: the contract's fields + roofline + cpu_baseline + one number per config.  Notes, per-launch tables and the
    sub-records in full are in bench_detail.json (and on stderr)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "build_id", "invalid", "value_natural", "value_bf16x3", "value_sparse", "value_host_io", "value_host_io_registered",
            "per_rank_frames_per_s", "configs_not_run")
    line = {k: out[k] for k in keep if k in out}
    r3 = lambda v: round(v, 3) if isinstance(v, float) else v     # noqa: E731
    if "verified" in out:
        v = out["verified"]
        line["verified"] = {k: v[k] for k in ("equal", "frames", "equal_all_ranks", "ranks", "skipped") if k in v}
    if "roofline" in out:
        r = out["roofline"]
        roof = {k: r3(r[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "launches",
                                      "step_frac_algorithmic", "step_frac_executed", "step_frac_hbm_algorithmic", "profiled_chunk_ms_single_stream") if k in r}
        roof["traffic_file"] = (r.get("traffic_source") or "").split(":")[0] or None
        # share of the single-stream chunk and achieved fraction of the class's own roof (f32 MFMA peak / 8 TB/s; *_executed counts
        # padded tiles and halo recomputation as work)
        roof["classes"] = {k: {kk: r3(c[kk]) for kk in ("share", "bound", "frac", "frac_executed") if kk in c} for k, c in r.get("classes", {}).items()}
        line["roofline"] = roof
    if "cpu_baseline" in out:
        c = out["cpu_baseline"]
        line["cpu_baseline"] = {k: r3(c[k]) for k in ("value", "unit", "cores", "kind", "sample") if k in c}
    cf = out.get("configs", {})
    pick = {}
    rb = cf.get("2-bf16x3", {}).get("roofline_bf16x3")
    if rb:
        line["roofline_bf16x3"] = {k: r3(rb[k]) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "chunk_hbm_gbs", "chunk_hbm_frac") if k in rb}
        if rb.get("largest_gemm"):
            line["roofline_bf16x3"]["largest_gemm"] = {k: r3(v) for k, v in rb["largest_gemm"].items()}
        line["roofline_bf16x3"]["within_tolerance"] = cf["2-bf16x3"].get("verified", {}).get("within_tolerance")

    def take(name, cfg, *path):
        v = cf.get(cfg)
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
        if v is not None:
            pick[name] = r3(v)
    take("2_extract_ms_median", "2-latency", "extract_ms_median")
    take("2_extract_plus_match_ms_median", "2-latency", "extract_plus_match_ms_median")
    take("3_tracking_frames_per_s_1000", "3", "nFeatures_1000", "frames_per_s_whole_loop")
    take("3_tracking_frames_per_s_850", "3", "nFeatures_850", "frames_per_s_whole_loop")
    take("4_frames_per_s", "4", "frames_per_s")
    take("4_seconds", "4", "seconds")
    take("5_db_q1_cold_us", "5", "db_q1_cold_us")
    take("5_db_q1_cold_frac_hbm", "5", "db_q1_cold_frac_hbm")
    take("5_db_q1_cold_scan_us", "5", "db_q1_cold_scan_us")
    take("5_db_q1_cold_scan_frac_hbm", "5", "db_q1_cold_scan_frac_hbm")
    take("5_db_q64_us", "5", "db_q64_us")
    take("5_db_q64_frac_hbm_i8_copy", "5", "db_q64_frac_hbm_i8_copy")
    take("5_match_32_pairs_us", "5", "match_32_pairs_us")
    take("5_match_frac_bf16_roof_executed", "5", "match_frac_bf16_roof_executed")
    take("5_match_255_pairs_us", "5", "match_255_pairs_us")
    take("5_match_255_frac_bf16_roof_executed", "5", "match_255_frac_bf16_roof_executed")
    take("5_triangulation_32_pairs_us", "5", "triangulation_32_pairs_us")
    line["configs"] = pick
    line["detail"] = "bench_detail.json"
    return line


def emit(out: dict) -> None:
    """full record -> bench_detail.json (next to this file; BENCH_DETAIL overrides) and stderr; the compact line -> stdout, LAST"""
    detail = os.environ.get("BENCH_DETAIL") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_detail.json")
    try:
        with open(detail, "w") as fh:
            json.dump(out, fh, indent=1)
    except OSError as e:
        print(f"bench.py: could not write {detail}: {e}", file=sys.stderr)
    print(json.dumps(out), file=sys.stderr, flush=True)
    line = json.dumps(compact_line(out))
    if len(line.encode()) >= LINE_LIMIT:
        raise SystemExit(f"bench.py: the stdout line is {len(line.encode())} bytes (limit {LINE_LIMIT}): move fields to bench_detail.json")
    print(line, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="frames per step and GPU")
    ap.add_argument("--chunk", type=int, default=None, help=f"frames per extract / match call (the extractor's batch); default {DEFAULT_CHUNK}, or --batch when that is smaller")
    ap.add_argument("--frames", choices=["uniform", "natural"], default="uniform", help="synthetic frame distribution (SURVEY.md 8d)")
    ap.add_argument("--configs", default="all", help="comma list of sub-records to measure besides the headline: " + ",".join(ALL_CONFIGS) + " | all | none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-natural", action="store_true", help="skip value_natural (the headline workload on natural-ish frames)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the last timed chunk's outputs (the line then says so)")
    ap.add_argument("--profile-all", action="store_true", help="also print the per-launch timing table to stderr")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option (hfnet_engine_set_option), e.g. fused_variant=2")
    ap.add_argument("--dry-ranks", type=int, default=0, metavar="N",
                    help="no GPU: run the N-rank control flow (gloo, rendezvous from the environment like the real run) with a timed no-op "
                         "instead of the kernels; the line is marked invalid")
    args = ap.parse_args()
    want = ALL_CONFIGS if args.configs == "all" else [] if args.configs == "none" else [c.strip() for c in args.configs.split(",")]
    if args.chunk is None:
        args.chunk = min(DEFAULT_CHUNK, args.batch)
    if args.chunk < 1 or args.batch % args.chunk:
        raise SystemExit("--batch must be a multiple of --chunk")
    dry = args.dry_ranks > 0
    if dry and args.dry_ranks != args.gpus:
        raise SystemExit("--dry-ranks N goes with --gpus N")

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if dry:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP front end has no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} visible device(s)")
        torch.cuda.set_device(local_rank)
        # torch bundles its own HIP runtime: it has to be initialised on this rank's device BEFORE libhfnet_hip.so pulls in the
        # system one (capi.lib() below), or the library sees no devices afterwards (same order as tests/conftest.py)
        torch.cuda.init()
        dev = torch.device("cuda", local_rank)
    dist = None
    # BENCH_FORCE_DIST=1: take the process-group path with one rank too (the only way to execute the RCCL init, barrier and
    # MAX-reduce of the N > 1 run on a one-GPU box)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if dry:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod

    def dev_sync():
        if not dry:
            torch.cuda.synchronize()

    B, chunks_per_step = args.chunk, args.batch // args.chunk
    n_sets = max(chunks_per_step, 2)      # every frame of a step is a different image (the same step is replayed K times)
    eng = capi = None
    if dry:
        pipe = DryPipeline(B)
        frames = [None] * n_sets
        wpath = None
    else:
        from hfnet_slam_amd import capi, weights
        wpath = os.path.join(tempfile.gettempdir(), f"hfnet_synth_seed7_rank{rank}.hfw")
        weights.save(wpath, weights.synthetic_weights(7))
        eng = capi.Engine(wpath, local_rank)
        # N replicas share one host: each takes its share of the cores for the staging copies of host-pointer batch calls
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        eng.set_option("copy_threads", max(0, min(3, cores // world - 1)))
        for o in args.opt:
            k, v = o.split("=")
            eng.set_option(k, int(v))
        pipe = Pipeline(torch, capi, eng, dev, W_IMG, H_IMG, B)
        # rank r's frames: block r of the global frame index space (hfnet_slam_amd/shard.py: disjoint blocks, no collective)
        frames = [to_device(torch, make_frames(B, (rank * n_sets + s) * B, args.frames), dev) for s in range(n_sets)]
        dev_sync()
    state = {"i": 0}

    def step():
        n = None
        for _ in range(chunks_per_step):
            n = pipe.run_chunk(frames[state["i"] % n_sets], B)
            state["i"] += 1
        return n

    def sync_all():
        pipe.eng.synchronize()
        dev_sync()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank(value):
        """[value of rank 0, ..., value of rank N-1] on every rank (one SUM all-reduce of a one-hot vector: bookkeeping, not data path)"""
        if dist is None:
            return [float(value)]
        t = torch.zeros((world,), dtype=torch.float64, device=dev)
        t[rank] = float(value)
        dist.all_reduce(t)
        return [float(v) for v in t.cpu()]

    # ---- warm-up, budget check; then the per-launch profile (single stream, every kernel alone) on a WARM chip -----
    work = layer_work(B)
    prof, table, classes, dominant, prof_sum_s = {}, [], {}, None, 0.0
    if not dry:
        n = pipe.run_chunk(frames[0], B)
        eng.synchronize()
        n = host(n)
        if int(n.min()) < N_FEAT and not os.environ.get("BENCH_NO_KP_CHECK"):
            raise SystemExit(f"synthetic frames gave only {int(n.min())} keypoints (< {N_FEAT}): budget not exercised")
    for _ in range(args.warmup):
        step()
    if not dry:
        prof = profile_pass(eng, pipe, frames, B)
        prof_sum_s = sum(v[1] for v in prof.values()) * 1e-3
        if args.profile_all and rank == 0:
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                print(f"  {k:26s} launches {v[0]:5.1f}  avg {v[1] / max(v[0], 1e-9) * 1e3:9.1f} us  share {v[1] * 1e-3 / prof_sum_s * 100:5.1f}%", file=sys.stderr)
        match_bf16 = bool(eng.options().get("match_screen_bf16"))
        table, classes = roofline_table(prof, work, prof_sum_s, match_bf16)
        dominant = max((k for k in prof if k in work), key=lambda k: prof[k][1])      # largest by TIME in the single-stream pass
        step()                                                                        # (back to the two-stream steady state)
        # ---- timed region: the dominant kernel is timed live with HIP events on its own stream ------
        eng.profile_reset(); eng.profile_filter(dominant); eng.profile_enable(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.eng.synchronize()
    dev_sync()
    own_elapsed = time.perf_counter() - t0               # this rank's own work, before it waits for the others
    sync_all()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    rank_fps = [args.batch * args.steps / max(t, 1e-12) for t in per_rank(own_elapsed)]
    dom = (0, 0.0)
    verified = natural = None
    if not dry:
        dom = eng.profile().get(dominant, (0, 0.0))
        eng.profile_enable(False); eng.profile_filter(None)
        if not args.no_verify and args.steps > 0:
            # ---- outside the timed region: sampled outputs of the LAST TIMED chunk against the oracle, on every rank ----
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            last, before = frames[(state["i"] - 1) % n_sets], frames[(state["i"] - 2) % n_sets]
            which = sorted({0, 1, B // 3, (2 * B) // 3, B - 1})
            if world >= 8:
                # eight ranks on one host each run the CPU oracle here: at most min(2, cores // world) frames per rank (>= 1), so that the first
                # real SCALE run cannot time out on a thin host
                which = which[:max(1, min(2, cores // world))]
            verified = verify_last_chunk(torch, pipe, wpath, last, before, which, max(1, min(32, cores // world)))
            if dist is not None:
                okt = torch.tensor([1 if verified["equal"] else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                verified["equal_all_ranks"] = bool(okt.item())
                verified["ranks"] = world
        if args.frames == "uniform" and not args.no_natural:
            # ---- the headline workload on the second synthetic distribution of SURVEY.md 8(d) ("natural-ish" frames) -------
            nat = make_natural_frames_device(torch, dev, n_sets * B, rank * n_sets * B).view(n_sets, B, H_IMG, W_IMG)
            dev_sync()
            uni, frames = frames, [nat[s_] for s_ in range(n_sets)]
            nat_steps = max(1, min(args.steps, 6))
            i_keep = state["i"]
            step(); sync_all()
            tn = time.perf_counter()
            for _ in range(nat_steps):
                n_last = step()
            sync_all()
            natural = {"frames_per_s": world * args.batch * nat_steps / max_over_ranks(time.perf_counter() - tn), "steps": nat_steps,
                       "min_keypoints": int(n_last.min().item())}
            frames = uni
            state["i"] = i_keep

    out = None
    if rank == 0:
        frames_total = world * args.batch * args.steps
        out = {
            "metric": "frames/sec HF-Net extract+match, 752x480, 1000 kpts",
            "value": frames_total / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "every result (keypoints, descriptors, global descriptors, matches, distances) is the f32 fma chain's, bit-identical to the "
                          "oracle; with engine option match_screen_bf16 (default) SearchByBoW's PRE-SELECTION -- not a result -- runs on split-bf16 "
                          "products with a rigorous band, and every candidate inside it is re-evaluated exactly",
            "data": f"synthetic (seeded {args.frames} u8 frames, seeded random-init weights of the reference architecture)",
            "config": {"workload": "EuRoC-size 752x480 mono, HF-Net extract (4 levels x1.2, budget 322/268/224/186, thr 0.01, "
                                   "level 0 incl. NetVLAD 4096-D) + SearchByBoW brute-force match vs previous frame",
                       "frames_per_step_per_gpu": args.batch, "frames_per_call": B, "parallelism": f"replicas x{world} (no collective)"},
        }
        if dry:
            out["invalid"] = f"--dry-ranks {args.dry_ranks}: control flow only, a timed no-op stands in for the GPU work"
        else:
            flop, byts, executed = work[dominant]
            launches_per_chunk = max(prof[dominant][0], 1.0)
            avg_s = dom[1] / max(dom[0], 1) * 1e-3
            roof = roofline_entry(flop / launches_per_chunk, byts / launches_per_chunk, avg_s)
            traffic, traffic_file = hbm_traffic(dominant, B)
            roof.update({"traffic": traffic,
                         "traffic_source": f"{traffic_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at chunk {B} (committed; not measured in this run)" if traffic_file else None,
                         "kernel": dominant, "avg_launch_us": avg_s * 1e6, "launches": dom[0],
                         "profiled_us_single_stream": prof[dominant][1] / launches_per_chunk * 1e3,
                         "selection": "largest launch name by time in the single-stream profiling pass (after the warm-up steps, 10 chunks); timed live (HIP events on its stream) over the timed region"})
            chunk_s = elapsed / args.steps / chunks_per_step
            # (f32 pipe only: with the bf16 screening the matcher's GEMM is not on it)
            on_f32 = [k for k in prof if k in work and not (match_bf16 and launch_class(k) == "match")]
            alg = sum(work[k][0] for k in on_f32)
            exe = sum(work[k][2] for k in on_f32)
            roof["step_frac_algorithmic"] = alg / chunk_s / 1e12 / MFMA_F32_PEAK_TFLOPS
            roof["step_frac_executed"] = exe / chunk_s / 1e12 / MFMA_F32_PEAK_TFLOPS
            # extractor-level HBM fraction: layer-granular algorithmic bytes x frames/s over the HBM peak -- what
            # the unfused network would have to move; the fused kernels move far less (roofline.classes, profiles/*traffic*)
            roof["step_frac_hbm_algorithmic"] = extractor_algorithmic_bytes(work) / chunk_s / 1e9 / HBM_PEAK_GBS
            roof["profiled_chunk_ms_single_stream"] = prof_sum_s * 1e3
            rnd = lambda d: {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()}
            roof["classes"] = {k: rnd(v) for k, v in sorted(classes.items(), key=lambda kv: -kv[1]["us"])}
            roof["table"] = [rnd(r) for r in table]
            out.update({"roofline": roof, "build_id": capi.build_id(), "options": eng.options()})
            if natural is not None:
                out["value_natural"] = natural["frames_per_s"]
                out["value_natural_min_keypoints"] = natural["min_keypoints"]
                out["value_natural_note"] = (f"the headline workload on the 'natural-ish' frames of SURVEY.md 8(d) (six octaves of up-sampled noise), "
                                             f"{natural['steps']} steps after one warm-up step; `value` is on iid uniform frames")
            out["verified"] = verified if verified is not None else {"frames": [], "equal": None, "skipped": "--no-verify"}
        if world > 1:
            out["per_rank_frames_per_s"] = {"min": min(rank_fps), "max": max(rank_fps), "all": [round(v, 1) for v in rank_fps]}
        if os.environ.get("BENCH_DEV_NO_MATCH"):
            out["invalid"] = "development run: the matcher was skipped (BENCH_DEV_NO_MATCH)"

    # ---- the other BASELINE configs (sub-records) ------------------------------------------------
    configs = {}
    if "4" in want:
        r = config_sequences(torch, pipe, dev, rank, world, dist, dry=dry)
        if rank == 0:
            configs["4"] = r
    if world > 1 and not dry and "2-host-io" in want:
        # N replicas feed from ONE host: the host-fed rate is the number the ranks contend for (pageable staging)
        dist.barrier()
        r = config_host_io(capi, eng, B)
        hio = per_rank(r["extract_plus_match_frames_per_s"])
        if rank == 0:
            r["per_rank_extract_plus_match_frames_per_s"] = {"min": min(hio), "max": max(hio), "all": [round(v, 1) for v in hio]}
            r["note"] = "all ranks run this leg at the same time; value_host_io is the sum over ranks"
            configs["2-host-io"] = r
            out["value_host_io"] = sum(hio)
    if rank == 0 and world == 1 and not dry:
        if "2-bf16x3" in want:
            configs["2-bf16x3"] = config_bf16x3(torch, capi, eng, dev, frames, B, chunks_per_step, max(1, min(args.steps, 8)), wpath, work)
            out["value_bf16x3"] = configs["2-bf16x3"]["frames_per_s"]
        if "2-sparse" in want:
            configs["2-sparse"] = config_sparse(torch, capi, dev, B, chunks_per_step, max(1, min(args.steps, 6)))
            out["value_sparse"] = configs["2-sparse"]["frames_per_s"]
        if "2-latency" in want:
            configs["2-latency"] = config_latency(capi, eng)
        if "2-host-io" in want:
            configs["2-host-io"] = config_host_io(capi, eng, B)
            out["value_host_io"] = configs["2-host-io"]["extract_plus_match_frames_per_s"]
            out["value_host_io_registered"] = configs["2-host-io"]["extract_plus_match_frames_per_s_registered"]
        if "3" in want:
            configs["3"] = {"workload": "TUM-VI-size 512x512 tracking loop, 4 levels, one frame per call, keyframe every 5th (database scan + 30 "
                                        "SearchForTriangulation pairs), device-resident keyframe store",
                            "nFeatures_1000": config_tracking(capi, eng, 1000), "nFeatures_850": config_tracking(capi, eng, 850),
                            "windowed_candidates": config_windowed_candidates(capi, eng)}
        if "5" in want:
            configs["5"] = config_loop_closure(capi, eng)
    if rank == 0:
        out["configs"] = configs
        out["configs_not_run"] = [c for c in ALL_CONFIGS if c not in configs]
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(wpath)
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    pipe.close()
    if eng is not None:
        eng.close()
    if verified is not None and not (verified["equal"] and verified.get("equal_all_ranks", True)):
        raise SystemExit("bench.py: outputs of the timed run differ from the oracle: " + "; ".join(verified["mismatch"]))


if __name__ == "__main__":
    main()
