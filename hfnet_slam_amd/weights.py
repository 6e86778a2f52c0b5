"""Weight container ("HFNETW1") + seeded synthetic weights for the HF-Net graph.

The reference ships no weights (README.md:62,86 point at external downloads), so parity and
benchmarks run on seeded random-init weights of the reference architecture.  Tensor names and
layouts follow the TF-slim variables the reference's exporter would read
(hfnet/export_model.py:32-50; scopes inferred from slim auto-numbering, one point confirmed by
src/Extractors/HFNetTFModelV2.cc:41) so a real checkpoint can be dropped in later:

    conv weights      HWIO   [kh, kw, Cin, Cout]
    depthwise weights        [kh, kw, C, 1]
    batch norm               gamma / beta / moving_mean / moving_variance  [C]
    fully connected          [in, out] + biases [out]
    VLAD clusters            [K, D]   (TF stores [1,1,1,K,D])

Container layout (little endian), read independently by oracle/hfnet_oracle.c and
hfnet_slam_amd/csrc/weights.cpp:

    char     magic[8]  = "HFNETW1\\0"
    uint32   n_tensors
    uint32   reserved (0)
    n_tensors x { char name[96]; uint32 ndim; uint32 dims[4]; uint32 pad; uint64 offset; uint64 nbytes }
    float32 payloads, each 64-byte aligned, `offset` from the start of the file
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from typing import Dict

import numpy as np

from .spec import DESC_DIM, DET_GRID, DET_HIDDEN, NetSpec, net_spec

MAGIC = b"HFNETW1\0"
_ENTRY = struct.Struct("<96sI4I4xQQ")  # 136 bytes, natural C alignment


def tensor_shapes(spec: NetSpec) -> "OrderedDict[str, tuple]":
    """Every tensor of the graph in a fixed order (name -> shape)."""
    t: "OrderedDict[str, tuple]" = OrderedDict()

    def bn(scope: str, c: int) -> None:
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            t[f"{scope}/BatchNorm/{n}"] = (c,)

    t["MobilenetV2/Conv/weights"] = (3, 3, 1, spec.stem_out)
    bn("MobilenetV2/Conv", spec.stem_out)
    for b in spec.blocks:
        if b.expand > b.cin:
            t[f"{b.scope}/expand/weights"] = (1, 1, b.cin, b.expand)
            bn(f"{b.scope}/expand", b.expand)
        t[f"{b.scope}/depthwise/depthwise_weights"] = (3, 3, b.expand, 1)
        bn(f"{b.scope}/depthwise", b.expand)
        t[f"{b.scope}/project/weights"] = (1, 1, b.expand, b.cout)
        bn(f"{b.scope}/project", b.cout)
    c = spec.local_channels
    t["local_head/descriptor/Conv/weights"] = (3, 3, c, DESC_DIM)
    bn("local_head/descriptor/Conv", DESC_DIM)
    t["local_head/descriptor/Conv_1/weights"] = (1, 1, DESC_DIM, DESC_DIM)
    t["local_head/descriptor/Conv_1/biases"] = (DESC_DIM,)
    t["local_head/detector/Conv/weights"] = (3, 3, c, DET_HIDDEN)
    bn("local_head/detector/Conv", DET_HIDDEN)
    t["local_head/detector/Conv_1/weights"] = (1, 1, DET_HIDDEN, DET_GRID * DET_GRID + 1)
    t["local_head/detector/Conv_1/biases"] = (DET_GRID * DET_GRID + 1,)
    g, k = spec.global_channels, spec.n_clusters
    t["global_head/vlad/memberships/weights"] = (1, 1, g, k)
    bn("global_head/vlad/memberships", k)
    t["global_head/vlad/clusters"] = (k, g)
    t["global_head/dimensionality_reduction/weights"] = (k * g, spec.global_dim)
    t["global_head/dimensionality_reduction/biases"] = (spec.global_dim,)
    return t


def synthetic_weights(seed: int = 7, spec: NetSpec | None = None, detector_gain: float = 1.0, dustbin_bias: float = 0.0
                      ) -> "OrderedDict[str, np.ndarray]":
    """Seeded weights: He-normal convs, BN gamma~U[0.5,1.5], beta~N(0,0.1),
    mean~N(0,0.1), var~U[0.5,1.5], biases~N(0,0.01), Xavier for FC / memberships / clusters.
    `detector_gain` scales the last detector conv so the softmax is not flat (scores spread
    around 1/65 and the threshold / top-K logic is genuinely exercised).
    `dustbin_bias` is added to the bias of the detector's 65th ("no keypoint") channel: with the default 0 every cell of a random-weight
    network has candidates above the reference's threshold 0.01 and top-K is always saturated; 15 leaves the coarser pyramid levels of a
    752x480 / 512x512 frame SHORT of their budget with the survivors clustered where the logits peak, 16 makes every level short on
    smooth frames, 18 leaves whole levels without a single candidate -- the nearest stand-in for trained weights the data-dependent
    kernels (candidate emission, top-K, tap de-duplication, the sparse descriptor head's row counts) can be given here."""
    spec = spec or net_spec()
    rng = np.random.Generator(np.random.PCG64(seed))
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in tensor_shapes(spec).items():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "gamma":
            a = rng.uniform(0.5, 1.5, shape)
        elif leaf == "beta":
            a = rng.normal(0.0, 0.1, shape)
        elif leaf == "moving_mean":
            a = rng.normal(0.0, 0.1, shape)
        elif leaf == "moving_variance":
            a = rng.uniform(0.5, 1.5, shape)
        elif leaf == "biases":
            a = rng.normal(0.0, 0.01, shape)
        elif leaf == "depthwise_weights":
            a = rng.normal(0.0, np.sqrt(2.0 / 9.0), shape)
        elif leaf == "clusters":
            a = rng.uniform(-1.0, 1.0, shape) * np.sqrt(6.0 / (shape[0] + shape[1]))
        elif name.startswith("global_head"):
            fan_in = int(np.prod(shape[:-1]))
            a = rng.uniform(-1.0, 1.0, shape) * np.sqrt(6.0 / (fan_in + shape[-1]))
        else:  # conv weights, He normal on fan-in
            fan_in = int(np.prod(shape[:-1]))
            a = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
            if name == "local_head/detector/Conv_1/weights":
                a = a * detector_gain
        if name == "local_head/detector/Conv_1/biases" and dustbin_bias:
            a = a.copy()
            a[-1] += dustbin_bias
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def save(path: str, tensors: Dict[str, np.ndarray]) -> None:
    names = list(tensors)
    header = 16 + _ENTRY.size * len(names)
    off = (header + 63) // 64 * 64
    entries, blobs = [], []
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype="<f4")
        if a.ndim > 4 or len(n.encode()) > 95:
            raise ValueError(f"tensor {n}: unsupported rank/name")
        dims = list(a.shape) + [1] * (4 - a.ndim)
        entries.append(_ENTRY.pack(n.encode(), a.ndim, *dims, off, a.nbytes))
        blobs.append((off, a))
        off = (off + a.nbytes + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<II", len(names), 0))
        for e in entries:
            f.write(e)
        for o, a in blobs:
            f.seek(o)
            f.write(a.tobytes())
        f.truncate(off)


def load(path: str) -> "OrderedDict[str, np.ndarray]":
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC:
        raise ValueError("not an HFNETW1 container")
    n, _ = struct.unpack_from("<II", buf, 8)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i in range(n):
        raw = _ENTRY.unpack_from(buf, 16 + i * _ENTRY.size)
        name = raw[0].split(b"\0", 1)[0].decode()
        ndim, dims, off, nbytes = raw[1], raw[2:6], raw[6], raw[7]
        out[name] = np.frombuffer(buf, dtype="<f4", count=nbytes // 4, offset=off).reshape(dims[:ndim]).copy()
    return out


def spec_from_tensors(tensors: Dict[str, np.ndarray]) -> NetSpec:
    """Recover the hyper-parameters from tensor shapes (the same derivation the C sides do)."""
    k, g = tensors["global_head/vlad/clusters"].shape
    gd = tensors["global_head/dimensionality_reduction/biases"].shape[0]
    stem = tensors["MobilenetV2/Conv/weights"].shape[-1]
    for mult in (0.35, 0.5, 0.75, 1.0, 1.3, 1.4):
        s = net_spec(mult, k, gd)
        if s.stem_out == stem and s.global_channels == g and all(
                tuple(tensors[n].shape) == sh for n, sh in tensor_shapes(s).items() if n in tensors):
            return s
    raise ValueError("tensor shapes do not match any known depth multiplier")
