#!/usr/bin/env python3
"""bench.py -- frames/sec of HF-Net extract + brute-force match, 752x480, 1000 keypoints.

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batch of `--batch` synthetic 752x480 frames through the whole front end on one
GPU: 4-level pyramid (x1.2), HF-Net (MobileNetV2 backbone, detector + descriptor heads, NetVLAD
on level 0), NMS, per-level top-K (budget 322/268/224/186), bilinear descriptor sampling, then one
SearchByBoW-style brute-force match (1000 x 1000 x 256, L2 cross-check, < 0.6) of every frame
against its predecessor.  Inputs are resident in HBM before the timed region; outputs stay in HBM.
Frames are independent, so N GPUs run N replicas on disjoint frames (weak scaling, no collective
on the data path); the timed region is bracketed by barrier + device synchronise, MAX over ranks.

Rank 0 prints ONE JSON line (contract: see the task description / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_IMG, H_IMG, N_FEAT, N_LEVELS, SCALE, THRESH, TH_LOW = 752, 480, 1000, 4, 1.2, 0.01, 0.6
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA (v_mfma_f32_32x32x2_f32)


def layer_work(batch: int):
    """Algorithmic FLOP and bytes (fp32, each layer reads its input once, writes its output once,
    reads its weights once -- SURVEY.md 8d) per launch, keyed by the profiler's launch names."""
    from hfnet_slam_amd import spec as S
    sp = S.net_spec()
    sizes = S.level_sizes(W_IMG, H_IMG, N_LEVELS, SCALE)
    work = {}

    def add(name, flop, byts):
        f, b = work.get(name, (0.0, 0.0))
        work[name] = (f + flop, b + byts)

    for lvl, (w, h) in enumerate(sizes):
        hc, wc = S.cropped(h), S.cropped(w)
        ph, pw = S.same_pad(hc, 3, 2)[0], S.same_pad(wc, 3, 2)[0]
        px = ph * pw * batch
        add("stem", 2.0 * 9 * sp.stem_out * px, hc * wc * batch + 4.0 * px * sp.stem_out)
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
            add(f"block_L{b.index:02d}",
                (2.0 * pin * b.cin * b.expand if b.expand > b.cin else 0.0) + 2.0 * 9 * pout * b.expand + 2.0 * pout * b.expand * b.cout,
                4.0 * (pin * b.cin + pout * b.cout * (2 if b.residual else 1) + b.cin * b.expand + 9 * b.expand + b.expand * b.cout))
            if b.index == 2:
                # stem + layer_2 as ONE launch (the default): u8 image in, layer_2 output out
                add("stem_block_L02", 2.0 * 9 * sp.stem_out * px + 2.0 * 9 * pout * b.expand + 2.0 * pout * b.expand * b.cout,
                    hc * wc * batch + 4.0 * (pout * b.cout + 9 * sp.stem_out + 9 * b.expand + b.expand * b.cout))
            ph, pw = oh, ow
            if b.index == 7:
                cells = ph * pw * batch
                c7 = b.cout
                add("conv3x3_desc", 2.0 * 9 * c7 * 256 * cells, 4.0 * (cells * (c7 + 256) + 9 * c7 * 256))
                add("pointwise_desc", 2.0 * 256 * 256 * cells, 4.0 * (cells * 512 + 256 * 256))
                add("l2norm_desc", 3.0 * 256 * cells, 4.0 * cells * 512)
                add("conv3x3_det", 2.0 * 9 * c7 * 128 * cells, 4.0 * (cells * (c7 + 128) + 9 * c7 * 128))
                add("pointwise_det", 2.0 * 128 * 65 * cells, 4.0 * (cells * (128 + 65) + 128 * 65))
                add("softmax_d2s", 4.0 * 65 * cells, 4.0 * cells * (65 + 64))
                add("nms", 2.0 * 3 * 18 * hc * wc * batch, 4.0 * 2 * hc * wc * batch)
        if lvl == 0:
            pg = ph * pw * batch
            add("pointwise_memberships", 2.0 * pg * sp.global_channels * sp.n_clusters, 4.0 * pg * (sp.global_channels + sp.n_clusters))
            add("vlad", 3.0 * pg * sp.vlad_dim / batch * batch, 4.0 * (pg * (sp.global_channels + sp.n_clusters) + 3 * batch * sp.vlad_dim))
            add("fc_l2", 2.0 * batch * sp.vlad_dim * sp.global_dim, 4.0 * (sp.vlad_dim * sp.global_dim + batch * (sp.vlad_dim + 2 * sp.global_dim)))
    # sparse descriptor head: 4 bilinear taps per keypoint (N_FEAT keypoints per frame)
    rows = 4.0 * N_FEAT * batch
    work["conv3x3_desc_taps"] = (2.0 * 9 * sp.local_channels * 256 * rows, 4.0 * (rows * (9 * sp.local_channels + 256) + 9 * sp.local_channels * 256))
    work["pointwise_desc_taps"] = (2.0 * 256 * 256 * rows, 4.0 * (rows * 512 + 256 * 256))
    work["l2norm_desc_taps"] = (3.0 * 256 * rows, 4.0 * rows * 512)
    # matcher: all frame pairs of a step in one batched call (prep + GEMM + train pass + finalize)
    work["match_bow"] = (batch * 2.0 * N_FEAT * N_FEAT * 256, batch * 4.0 * (2 * N_FEAT * 256 + 2 * N_FEAT * N_FEAT))
    return work


def hbm_traffic(launch_name: str, batch: int):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_traffic_b32.json:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this same command), with the gfx950
    correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte requests as 64 bytes -> doubled).  None when
    no pass exists for this kernel / batch size."""
    path = os.path.join(ROOT, "profiles", "r01_traffic_b32.json")
    try:
        with open(path) as f:
            t = json.load(f)
        k = t["kernels"][launch_name]
        if t["batch"] != batch:
            return None
        return (2.0 * k["fetch_kb"] + k["write_kb"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def make_frames(count: int, first_index: int, kind: str = "uniform") -> np.ndarray:
    """seed 1000 + frame index; "uniform" = iid uniform u8, "natural" = sum of 6 octaves of bilinearly
    up-sampled uniform noise, clipped (smooth structures at several scales)"""
    out = np.empty((count, H_IMG, W_IMG), np.uint8)
    for i in range(count):
        rng = np.random.default_rng(1000 + first_index + i)
        if kind == "uniform":
            out[i] = rng.integers(0, 256, (H_IMG, W_IMG), dtype=np.uint8)
            continue
        acc = np.zeros((H_IMG, W_IMG), np.float64)
        for o in range(6):
            gh, gw = 2 + (H_IMG >> (6 - o)), 2 + (W_IMG >> (6 - o))
            g = rng.random((gh, gw))
            ys = np.linspace(0, gh - 1.001, H_IMG); xs = np.linspace(0, gw - 1.001, W_IMG)
            y0 = ys.astype(int); x0 = xs.astype(int); fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
            up = (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0][:, x0 + 1] * (1 - fy) * fx + g[y0 + 1][:, x0] * fy * (1 - fx) + g[y0 + 1][:, x0 + 1] * fy * fx)
            acc += up * 0.5 ** (5 - o)
        acc = (acc - acc.min()) / (acc.max() - acc.min())
        out[i] = np.clip(acc * 1.2 * 255.0 - 25.0, 0, 255).astype(np.uint8)
    return out


def cpu_baseline(weights_path: str, max_seconds: float = 25.0):
    """Oracle (CPU restatement, kind 'port') on a bounded sample of the same workload, host cores
    of this box.  Reported, never optimised against."""
    from oracle import oracle as O
    O.build()
    threads = O.usable_cpus(32)      # affinity and cgroup quota, at most 32
    O.set_threads(threads)
    m = O.Model(weights_path)
    frames = make_frames(5, 0)
    m.extract(frames[0], N_FEAT, THRESH, N_LEVELS, SCALE)       # warm-up (page-in, FC transpose)
    done, t_total, prev = 0, 0.0, None
    t0 = time.perf_counter()
    for i in range(1, len(frames)):
        n, kps, desc, g, _ = m.extract(frames[i], N_FEAT, THRESH, N_LEVELS, SCALE)
        if prev is not None:
            O.search_by_bow(prev, desc, TH_LOW)
        prev = desc
        done += 1
        t_total = time.perf_counter() - t0
        if t_total > max_seconds:
            break
    return {"value": done / t_total, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{done} frames 752x480 extract (4 levels, 1000 kpts) + {max(done - 1, 0)} SearchByBoW matches, oracle/libhfnet_oracle.so, {threads} OpenMP threads"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frames per step and GPU")
    ap.add_argument("--frames", choices=["uniform", "natural"], default="uniform", help="synthetic frame distribution (SURVEY.md 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="also print the per-launch timing table to stderr")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option (hfnet_engine_set_option), e.g. fused_variant=2")
    args = ap.parse_args()

    import torch
    from hfnet_slam_amd import capi, weights

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP front end has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod

    wpath = os.path.join(tempfile.gettempdir(), f"hfnet_synth_seed7_rank{rank}.hfw")
    weights.save(wpath, weights.synthetic_weights(7))
    eng = capi.Engine(wpath, local_rank)
    for o in args.opt:
        k, v = o.split("=")
        eng.set_option(k, int(v))
    B = args.batch
    ext = capi.Extractor(eng, W_IMG, H_IMG, N_FEAT, THRESH, SCALE, N_LEVELS, max_batch=B)

    n_sets = 2
    frames = [torch.from_numpy(make_frames(B, (rank * n_sets + s) * B, args.frames)).to(dev) for s in range(n_sets)]
    # descriptor sets of the last n_buf steps live in one rotating store: set id = buffer * B + frame
    n_buf = 3
    kps = torch.zeros((n_buf * B, N_FEAT, 4), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_buf * B, N_FEAT, 256), dtype=torch.float32, device=dev)
    n_rows = torch.zeros((n_buf * B,), dtype=torch.int32, device=dev)          # keypoints per set, written by the extractor
    glob = torch.zeros((B, eng.global_dim), dtype=torch.float32, device=dev)
    match = torch.zeros((B, N_FEAT), dtype=torch.int32, device=dev)
    mdist = torch.zeros((B, N_FEAT), dtype=torch.float32, device=dev)
    mcnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    # frame j of buffer c is matched against its predecessor (query = previous frame, train = this frame)
    tset = [torch.arange(c * B, (c + 1) * B, dtype=torch.int32, device=dev) for c in range(n_buf)]
    qset = [((t.to(torch.int64) - 1) % (n_buf * B)).to(torch.int32) for t in tset]
    torch.cuda.synchronize()
    L = capi.lib()
    import ctypes as C

    state = {"step": 0}

    def step():
        i = state["step"]
        cur = i % n_buf
        f = frames[i % n_sets]
        ext.extract_batch_device(B, f.data_ptr(), W_IMG, W_IMG * H_IMG, kps[cur * B].data_ptr(), desc[cur * B].data_ptr(), glob.data_ptr(),
                                 n_rows[cur * B:].data_ptr())
        # the next extraction overwrites the buffer the PREVIOUS step's matches still read: fence it behind them
        eng.fence()
        # all B SearchByBoW pairs of the step in one call; keypoint counts stay on the device (no host sync)
        st = L.hfnet_match_search_by_bow_batch(eng.h, B, C.c_void_p(desc.data_ptr()), C.c_size_t(N_FEAT * 256), C.c_void_p(n_rows.data_ptr()),
                                               n_buf * B, C.c_void_p(qset[cur].data_ptr()), C.c_void_p(tset[cur].data_ptr()), N_FEAT, 256,
                                               C.c_float(TH_LOW), C.c_void_p(match.data_ptr()), C.c_void_p(mdist.data_ptr()),
                                               C.c_void_p(mcnt.data_ptr()), 1)
        if st != 0:
            raise RuntimeError(capi.last_error())
        state["step"] = i + 1
        return n_rows[cur * B:(cur + 1) * B]

    def sync_all():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- warm-up (full per-launch profile -> dominant kernel) -----------------------------------
    eng.profile_reset(); eng.profile_filter(None); eng.profile_enable(True)
    for _ in range(max(args.warmup, 1)):
        n = step()
        eng.synchronize()
        n = n.cpu().numpy()
        if int(n.min()) < N_FEAT and not os.environ.get("BENCH_NO_KP_CHECK"):
            raise SystemExit(f"synthetic frames gave only {int(n.min())} keypoints (< {N_FEAT}): budget not exercised")
    eng.synchronize()
    prof = eng.profile()
    eng.profile_enable(False)
    if args.profile_all and rank == 0:
        tot = sum(v[1] for v in prof.values())
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k:26s} launches {v[0]:5d}  avg {v[1] / max(v[0], 1) * 1e3:9.1f} us  share {v[1] / tot * 100:5.1f}%", file=sys.stderr)
    work = layer_work(B)
    dominant = max((k for k in prof if k in work), key=lambda k: prof[k][1])
    eng.profile_reset(); eng.profile_filter(dominant); eng.profile_enable(True)

    # ---- timed region ----------------------------------------------------------------------------
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dom = eng.profile().get(dominant, (0, 0.0))
    eng.profile_enable(False)

    if rank == 0:
        frames_total = world * B * args.steps
        value = frames_total / elapsed
        flop, byts = work[dominant]
        avg_s = dom[1] / max(dom[0], 1) * 1e-3
        intensity = flop / byts
        if intensity >= MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
            roof = {"bound": "mfma", "achieved": flop / avg_s / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": byts / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["traffic"] = hbm_traffic(dominant, B)
        roof["kernel"] = dominant
        roof["avg_launch_us"] = avg_s * 1e6
        roof["launches"] = dom[0]
        out = {
            "metric": "frames/sec HF-Net extract+match, 752x480, 1000 kpts",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": f"synthetic (seeded {args.frames} u8 frames, seeded random-init weights of the reference architecture)",
            "config": {"workload": "EuRoC-size 752x480 mono, HF-Net extract (4 levels x1.2, budget 322/268/224/186, thr 0.01, "
                                   "level 0 incl. NetVLAD 4096-D) + SearchByBoW brute-force match vs previous frame",
                       "frames_per_step_per_gpu": B, "parallelism": f"replicas x{world} (no collective)"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wpath)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ext.close(); eng.close()


if __name__ == "__main__":
    main()
