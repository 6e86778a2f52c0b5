"""N > 1 path on CPU: world_size 2, gloo backend.  The data path has no collective (replicas);
what crosses ranks is the shard plan, the barrier and the MAX-over-ranks timing."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from hfnet_slam_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.frame_block(rank, world, 6)
    mine = torch.arange(lo, hi, dtype=torch.int64)
    got = [torch.zeros(6, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(got, mine)
    allf = torch.cat(got)
    assert torch.equal(allf, torch.arange(0, 6 * world)), "frame shards overlap or leave gaps"
    seqs = shard.assign_sequences(shard.EUROC_SEQUENCES, world)
    objs = [None] * world
    dist.all_gather_object(objs, seqs)
    assert all(o == seqs for o in objs), "ranks disagree on the sequence plan"
    dist.barrier()
    t = shard.max_over_ranks(dist, 1.0 + rank)
    assert t == float(world)
    ret[rank] = (lo, hi, seqs[rank], t)
    dist.destroy_process_group()


def test_two_rank_shard_barrier_and_max_reduce():
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0][:2] == (0, 6) and ret[1][:2] == (6, 12)
    assert sorted(ret[0][2] + ret[1][2]) == sorted(__import__("hfnet_slam_amd.shard", fromlist=["x"]).EUROC_SEQUENCES)


def test_sequence_assignment_is_balanced():
    from hfnet_slam_amd import shard
    total = sum(shard.EUROC_SEQUENCES.values())
    assert total == 27049
    for world in (1, 2, 4, 8):
        plan = shard.assign_sequences(shard.EUROC_SEQUENCES, world)
        flat = [s for p in plan for s in p]
        assert sorted(flat) == sorted(shard.EUROC_SEQUENCES)
        loads = [sum(shard.EUROC_SEQUENCES[s] for s in p) for p in plan]
        assert max(loads) <= total / world + max(shard.EUROC_SEQUENCES.values())
    assert shard.split_even(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        shard.frame_block(2, 2, 4)


def test_bench_dry_ranks_runs_the_two_rank_control_flow(tmp_path):
    """bench.py --gpus 2 --dry-ranks 2: the same main() the driver launches with torchrun -- rendezvous from the environment,
    per-rank frame blocks, config 4's sequence plan, barriers, MAX over ranks, one JSON line from rank 0 -- with a timed no-op
    in place of the GPU work (this box has no GPU; the N > 1 path cannot run otherwise)."""
    import json
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_DETAIL=str(tmp_path / "detail.json"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks", "2", "--steps", "2", "--warmup", "1",
                                       "--batch", "128", "--configs", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    json_lines = lambda text: [l for l in text.splitlines() if l.startswith("{")]           # (gloo logs its rendezvous to stdout)
    assert json_lines(outs[1][0]) == [] and len(json_lines(outs[0][0])) == 1, "rank 0 prints the one line"
    line = json.loads(json_lines(outs[0][0])[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and "invalid" in line and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 2 * 128 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    pr = line["per_rank_frames_per_s"]                                                      # the first real SCALE run explains itself
    assert len(pr["all"]) == 2 and 0 < pr["min"] <= pr["max"] and pr["min"] * 2 >= line["value"] * 0.5
    assert len(json_lines(outs[0][0])[0].encode()) < 8000 and line["configs"]["4_frames_per_s"] > 0      # the compact line; the record in full:
    c4 = json.load(open(tmp_path / "detail.json"))["configs"]["4"]
    assert c4["frames"] == 27049 and c4["gpus"] == 2 and c4["sequences"] == 11 and c4["frames_per_s"] > 0
    # a world size that does not match --gpus is refused before anything is initialised
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_rccl_path_with_one_rank():
    """the driver's launch line (torch.distributed.run, one rank per GPU) with the RCCL process group forced on for a
    single rank: init_process_group("nccl"), the barriers, the MAX all-reduce and config 4's reductions execute on the GPU
    (two ranks cannot share the one device of the test box)"""
    import json
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(), "detail.json")
    env = dict(os.environ, BENCH_FORCE_DIST="1", BENCH_DETAIL=detail)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "256",
                        "--configs", "4", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["configs"]["4_frames_per_s"] > 0 and len(lines[0].encode()) < 8000
    assert json.load(open(detail))["configs"]["4"]["frames"] == 27049
    v = line["verified"]                                       # the timed run checks its own last chunk against the oracle
    import bench
    B = min(bench.DEFAULT_CHUNK, 256)                          # frames per call of this run (--batch 256)
    assert v["equal"] is True and v["equal_all_ranks"] is True and v["frames"] == sorted({0, 1, B // 3, (2 * B) // 3, B - 1}) and line["value_natural"] > 0
