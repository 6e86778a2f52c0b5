"""Frame sharding for the multi-GPU replica driver (host plumbing only).

Frames are independent (BaseModel::Detect is a pure function of image + weights), so N GPUs run
N replicas of the front end on disjoint frames: no collective on the data path.
The only cross-rank exchange is the timing barrier + MAX over ranks in bench.py."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

# EuRoC sequence lengths (lines of Examples/Monocular/EuRoC_TimeStamps/*.txt in the reference)
EUROC_SEQUENCES: Dict[str, int] = {
    "MH01": 3682, "MH02": 3040, "MH03": 2700, "MH04": 2033, "MH05": 2273, "V101": 2912,
    "V102": 1710, "V103": 2149, "V201": 2280, "V202": 2348, "V203": 1922,
}


def assign_sequences(lengths: Dict[str, int], world: int) -> List[List[str]]:
    """BASELINE config 4: whole sequences to GPUs, longest first onto the least-loaded GPU."""
    loads = [0] * world
    out: List[List[str]] = [[] for _ in range(world)]
    for name, n in sorted(lengths.items(), key=lambda kv: (-kv[1], kv[0])):
        r = min(range(world), key=lambda i: (loads[i], i))
        out[r].append(name)
        loads[r] += n
    return out


def sequence_chunks(names: Sequence[str], lengths: Dict[str, int], chunk: int) -> List[Tuple[str, int, int]]:
    """BASELINE config 4, one rank: its sequences in order, each cut into chunks of at most `chunk` consecutive frames
    (a chunk never spans two sequences: the frame-to-frame match stops at a sequence boundary).
    Returns [(sequence, first_frame, n_frames), ...]."""
    if chunk < 1:
        raise ValueError("chunk < 1")
    out: List[Tuple[str, int, int]] = []
    for name in names:
        n = lengths[name]
        for f0 in range(0, n, chunk):
            out.append((name, f0, min(chunk, n - f0)))
    return out


def chunk_pairs(first_frame: int, n_frames: int, slot0: int, prev_slot: int) -> Tuple[List[int], List[int]]:
    """Frame-vs-previous match pairs of one chunk whose frames land in descriptor slots slot0 .. slot0 + n_frames - 1;
    `prev_slot` holds the last frame of the previous chunk of the same sequence.  The first frame of a sequence has no
    predecessor.  Returns (query_slots, train_slots) -- query = the earlier frame."""
    q: List[int] = []
    t: List[int] = []
    for i in range(n_frames):
        if first_frame + i == 0:
            continue
        q.append(slot0 + i - 1 if i > 0 else prev_slot)
        t.append(slot0 + i)
    return q, t


def frame_block(rank: int, world: int, frames_per_rank: int) -> Tuple[int, int]:
    """Weak-scaling shard used by bench.py: rank r owns frames [r*F, (r+1)*F)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return rank * frames_per_rank, (rank + 1) * frames_per_rank


def split_even(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Strong-scaling shard of n_items over `world` ranks (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    out, start = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        out.append((start, start + size))
        start += size
    return out


def max_over_ranks(dist, seconds: float, device=None) -> float:
    """MAX of the per-rank elapsed time (the number bench.py divides by)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
