"""development experiment: N independent extract + match pipelines (own extractor, own streams) fed alternately, frames/s
against one pipeline.  usage: two_pipes.py <chunk> <n_pipes> [chunks_per_pipe]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import bench
from hfnet_slam_amd import capi, weights

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_pipes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
dev = torch.device("cuda", 0)
wpath = os.path.join(tempfile.gettempdir(), "hfnet_synth_seed7_dev.hfw")
weights.save(wpath, weights.synthetic_weights(7))
eng = capi.Engine(wpath, 0)
for o in sys.argv[4:]:
    k, v = o.split("=")
    eng.set_option(k, int(v))
pipes = [bench.Pipeline(torch, capi, eng, dev, bench.W_IMG, bench.H_IMG, chunk) for _ in range(n_pipes)]
frames = [torch.from_numpy(bench.make_frames(chunk, s * chunk, "uniform")).to(dev) for s in range(4)]
torch.cuda.synchronize()


def run(n):
    for i in range(n):
        pipes[i % n_pipes].run_chunk(frames[i % 4], chunk)


run(2 * n_pipes)
eng.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    run(reps)
    eng.synchronize()
    dt = time.perf_counter() - t0
    print(f"chunk {chunk} pipes {n_pipes}: {reps * chunk / dt:8.0f} frames/s  ({dt / reps * 1e3 * 64 / chunk:.3f} ms per 64 frames)")
