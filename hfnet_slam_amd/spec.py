"""HF-Net graph geometry (host-side plumbing, pure Python).

Restates the channel / stride plan of the reference network definition so that the
weight container, the oracle and the tests agree on tensor shapes:

* backbone spec            /root/reference hfnet/models/hf_net.py:13-52
* channel rounding         hfnet/models/backbones/utils/mobilenet.py:62-69,96-106
* expansion size           hfnet/models/backbones/utils/conv_blocks.py:50-57,158-159
* residual rule            hfnet/models/backbones/utils/conv_blocks.py:304-311
* endpoints                hf_net.py:160-161 (layer_7 local, layer_18 global)
* per-level input sizes    src/Extractors/BaseModel.cc:33-65, src/Extractors/HFextractor.cc:159-173
* per-level budget         src/Extractors/HFextractor.cc:108-119

Nothing here runs on the hot path; the HIP library derives the same numbers from the
tensor shapes it finds in the weight container.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

# (stride, num_outputs at depth multiplier 1.0) -- hf_net.py:29-51; entry 0 is the stem conv.
_BACKBONE = [
    (2, 32),   # layer_1  slim.conv2d 3x3
    (1, 16),   # layer_2  expanded_conv, expansion factor 1 (no expand conv is built)
    (2, 24),   # layer_3
    (1, 24),   # layer_4
    (2, 32),   # layer_5
    (1, 64),   # layer_6
    (1, 128),  # layer_7  <- local endpoint
    (2, 64),   # layer_8
    (1, 64),
    (1, 64),
    (1, 64),
    (1, 96),
    (1, 96),
    (1, 96),
    (2, 160),
    (1, 160),
    (1, 160),
    (1, 320),  # layer_18 <- global endpoint; the trailing 1x1->1280 is never built
]

LOCAL_ENDPOINT = 7    # 1-based layer index
GLOBAL_ENDPOINT = 18
DESC_DIM = 256        # hf_net.py:170
DET_GRID = 8          # hf_net.py:171
DET_HIDDEN = 128      # hf_net.py:85
BN_EPS = 1e-3         # slim.batch_norm default epsilon
NMS_RADIUS = 4        # hfnet/export_model.py:35
NMS_ITERS = 2         # hfnet/export_model.py:37, hfnet/README.md:48


def make_divisible(v: float, divisor: int, min_value: int | None = None) -> int:
    """mobilenet.py:62-69 / conv_blocks.py:50-57."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


@dataclass(frozen=True)
class Block:
    index: int        # 1-based layer_N
    scope: str        # TF variable scope (inferred from slim auto-numbering)
    cin: int
    expand: int       # inner size; == cin means "no expand conv"
    stride: int
    cout: int
    residual: bool


@dataclass(frozen=True)
class NetSpec:
    depth_multiplier: float
    stem_out: int
    blocks: Tuple[Block, ...]
    local_channels: int
    global_channels: int
    n_clusters: int
    global_dim: int

    @property
    def vlad_dim(self) -> int:
        return self.n_clusters * self.global_channels


def net_spec(depth_multiplier: float = 0.75, n_clusters: int = 32, global_dim: int = 4096) -> NetSpec:
    """Defaults: mult 0.75 is forced by the hard-coded {1,H/8,W/8,96} intermediate
    (src/Extractors/BaseModel.cc:70); K=32 / 4096 are the upstream distillation config
    (4096 is hard-coded at src/Extractors/HFNetTFModelV2.cc:173)."""
    outs = [make_divisible(n * depth_multiplier, 8, 8) for _, n in _BACKBONE]
    blocks: List[Block] = []
    cin = outs[0]
    for i in range(1, len(_BACKBONE)):
        stride, _ = _BACKBONE[i]
        cout = outs[i]
        if i == 1:
            inner = make_divisible(cin * 1, 1)          # expand_input_by_factor(1, divisible_by=1)
        else:
            inner = make_divisible(cin * 6, 8)          # expand_input_by_factor(6)
        scope = "MobilenetV2/expanded_conv" + ("" if i == 1 else f"_{i - 1}")
        blocks.append(Block(i + 1, scope, cin, inner, stride, cout, stride == 1 and cin == cout))
        cin = cout
    return NetSpec(depth_multiplier, outs[0], tuple(blocks),
                   blocks[LOCAL_ENDPOINT - 2].cout, blocks[GLOBAL_ENDPOINT - 2].cout,
                   n_clusters, global_dim)


def same_pad(in_size: int, k: int, stride: int) -> Tuple[int, int, int]:
    """TensorFlow 'SAME': returns (out, pad_before, pad_after)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2, total - total // 2


def cv_round(x: float) -> int:
    """cvRound == lrint: round half to even."""
    return int(np.rint(x))


def level_sizes(width: int, height: int, n_levels: int, scale_factor: float) -> List[Tuple[int, int]]:
    """(W, H) of every pyramid level as HFextractor::ComputePyramid computes them
    (HFextractor.cc:93-106,159-166): float tables, cvRound((float)cols * inv[l])."""
    sf = np.float32(scale_factor)
    scales = [np.float32(1.0)]
    for _ in range(1, n_levels):
        scales.append(np.float32(scales[-1] * sf))
    out = []
    for lvl in range(n_levels):
        inv = np.float32(1.0) / scales[lvl]
        out.append((cv_round(np.float32(np.float32(width) * inv)), cv_round(np.float32(np.float32(height) * inv))))
    return out


def model_level_sizes(width: int, height: int, n_levels: int, scale_factor: float) -> List[Tuple[int, int]]:
    """(W, H) the per-level models are built for (BaseModel.cc:33-63: scale /= scaleFactor)."""
    sf = np.float32(scale_factor)
    scale = np.float32(1.0)
    out = []
    for _ in range(n_levels):
        out.append((cv_round(np.float32(np.float32(width) * scale)), cv_round(np.float32(np.float32(height) * scale))))
        scale = np.float32(scale / sf)
    return out


def features_per_level(n_features: int, n_levels: int, scale_factor: float) -> List[int]:
    """HFextractor.cc:108-119 (float math, pow in double, cvRound)."""
    if n_levels == 1:
        return [n_features]
    factor = np.float32(1.0) / np.float32(scale_factor)
    denom = np.float32(1.0) - np.float32(math.pow(float(factor), float(n_levels)))
    desired = np.float32(np.float32(n_features) * (np.float32(1.0) - factor)) / denom
    desired = np.float32(desired)
    res, total = [], 0
    for _ in range(n_levels - 1):
        n = cv_round(desired)
        res.append(n)
        total += n
        desired = np.float32(desired * factor)
    res.append(max(n_features - total, 0))
    return res


def cropped(size: int) -> int:
    """In-graph crop to a multiple of 8 (hf_net.py:188-190)."""
    return (size // 8) * 8
