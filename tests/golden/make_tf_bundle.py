#!/usr/bin/env python3
"""Builds tests/golden/tf_bundle/model.ckpt-7.{index,data-00000-of-00001}: a small TensorFlow "tensor bundle" checkpoint
written WITHOUT hfnet_slam_amd/tf_checkpoint.py (nothing of it is imported), straight from the published formats, so that
the reader is checked against a second implementation and not against its own writer:

  * tensorflow/core/lib/io/table_builder.cc, format.cc, block_builder.cc (the LevelDB table): data blocks of
    (shared, non_shared, value_len) varint32 triples + key delta + value, restart points every 16 entries, a restart array
    and its length (fixed32) at the end of the block, 1 type byte (0 = no compression) + fixed32 masked CRC-32C after every
    block; index block with SHORTENED separator keys (FindShortestSeparator between blocks, FindShortSuccessor after the
    last), restart interval 1; empty metaindex block; 48-byte footer = two block handles (varint64 offset, size), zero
    padding to 40 bytes, magic 0xdb4775248b80fb57 little endian.
  * tensorflow/core/protobuf/tensor_bundle.proto: key "" -> BundleHeaderProto {num_shards = 1, endianness = 2 (LITTLE = 0,
    omitted), version = 3 {producer = 1}}; every other key -> BundleEntryProto {dtype = 1, shape = 2, shard_id = 3,
    offset = 4, size = 5, crc32c = 6 (fixed32, masked)}.  proto3: zero-valued fields are NOT serialised (the first tensor
    has no offset field, no entry has a shard_id field), a scalar has an empty shape message.
  * the data file: tensors back to back in key order, little endian, row-major; a DT_STRING tensor (the object graph a
    tf.train.Checkpoint adds) is stored as varint lengths + a fixed32 CRC of the lengths + the bytes.

Contents: a few variables with the names / ranks the HF-Net importer meets (HWIO conv weights, BatchNorm vectors, FC,
the [1,1,1,K,D] cluster tensor), optimizer slots, an int64 global_step and a string entry.  Values are f(index) so that
the test can recompute them.

    python tests/golden/make_tf_bundle.py        (rewrites the two fixture files)
"""
import os
import struct

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_bundle")
PREFIX = os.path.join(HERE, "model.ckpt-7")


def crc32c(data: bytes) -> int:
    """bitwise CRC-32C (Castagnoli, reflected polynomial 0x82F63B78)"""
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


assert crc32c(b"123456789") == 0xE3069283 and crc32c(bytes(32)) == 0x8A9136AA          # RFC 3720 B.4


def masked(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def vint(v: int) -> bytes:
    out = b""
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def pb_varint(field: int, v: int) -> bytes:
    return b"" if v == 0 else vint(field << 3) + vint(v)           # proto3: defaults are not written


def pb_bytes(field: int, payload: bytes) -> bytes:
    return vint((field << 3) | 2) + vint(len(payload)) + payload


def shape_proto(shape) -> bytes:
    return b"".join(pb_bytes(2, pb_varint(1, int(d))) for d in shape)


DT_FLOAT, DT_STRING, DT_INT64 = 1, 7, 9


def tensors():
    def ramp(shape, k):
        n = int(np.prod(shape)) if shape else 1
        return ((np.arange(n, dtype=np.float64) * 0.37 + k) % 5.0 - 2.5).astype("<f4").reshape(shape)
    t = {
        "MobilenetV2/Conv/weights": ramp((3, 3, 1, 24), 1),
        "MobilenetV2/Conv/BatchNorm/gamma": ramp((24,), 2),
        "MobilenetV2/Conv/BatchNorm/beta": ramp((24,), 3),
        "MobilenetV2/Conv/BatchNorm/moving_mean": ramp((24,), 4),
        "MobilenetV2/Conv/BatchNorm/moving_variance": np.abs(ramp((24,), 5)) + np.float32(0.5),
        "MobilenetV2/Conv/weights/Adam": ramp((3, 3, 1, 24), 6),
        "MobilenetV2/Conv/weights/Adam_1": ramp((3, 3, 1, 24), 7),
        "MobilenetV2/expanded_conv/depthwise/depthwise_weights": ramp((3, 3, 24, 1), 8),
        "MobilenetV2/expanded_conv_1/expand/weights": ramp((1, 1, 16, 96), 9),
        "global_head/dimensionality_reduction/biases": ramp((8,), 10),
        "global_head/dimensionality_reduction/weights": ramp((64, 8), 11),
        "global_head/vlad/clusters": ramp((1, 1, 1, 2, 32), 12),
        "global_head/vlad/memberships/BatchNorm/beta": ramp((2,), 13),          # (no gamma: slim.batch_norm scale=False)
        "global_head/vlad/memberships/BatchNorm/moving_mean": ramp((2,), 14),
        "global_head/vlad/memberships/BatchNorm/moving_variance": np.abs(ramp((2,), 15)) + np.float32(0.5),
        "global_head/vlad/memberships/weights": ramp((1, 1, 32, 2), 16),
        "beta1_power": np.float32(0.9).reshape(()),
        "global_step": np.array(83096, "<i8"),
    }
    return t


def main():
    os.makedirs(HERE, exist_ok=True)
    t = tensors()
    graph = [b"checkpointable object graph stand-in", b"", b"x" * 200]             # a DT_STRING tensor of 3 elements
    keys = sorted([k.encode() for k in t] + [b"_CHECKPOINTABLE_OBJECT_GRAPH"])
    entries = [(b"", pb_varint(1, 1) + pb_bytes(3, pb_varint(1, 1)))]              # header: num_shards 1, version {producer 1}
    data = b""
    for k in keys:
        if k == b"_CHECKPOINTABLE_OBJECT_GRAPH":
            lens = b"".join(vint(len(s)) for s in graph)
            raw = lens + struct.pack("<I", masked(crc32c(lens))) + b"".join(graph)
            dtype, shape = DT_STRING, (3,)
        else:
            a = t[k.decode()]
            raw, shape = a.tobytes(), a.shape
            dtype = DT_INT64 if a.dtype.kind == "i" else DT_FLOAT
        e = (pb_varint(1, dtype) + pb_bytes(2, shape_proto(shape)) + pb_varint(4, len(data)) + pb_varint(5, len(raw))
             + vint((6 << 3) | 5) + struct.pack("<I", masked(crc32c(raw))))
        entries.append((k, e))
        data += raw
    with open(PREFIX + ".data-00000-of-00001", "wb") as f:
        f.write(data)

    out = b""

    def block(items, interval):
        b, restarts, prev = b"", [], b""
        for i, (k, v) in enumerate(items):
            if i % interval == 0:
                restarts.append(len(b)); shared = 0
            else:
                shared = 0
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            b += vint(shared) + vint(len(k) - shared) + vint(len(v)) + k[shared:] + v
            prev = k
        restarts = restarts or [0]
        return b + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))

    def emit(b):
        nonlocal out
        handle = vint(len(out)) + vint(len(b))
        out += b + b"\0" + struct.pack("<I", masked(crc32c(b + b"\0")))
        return handle

    def separator(a, b):
        """shortest key k with a <= k < b (format.cc FindShortestSeparator); successor of a when b is None"""
        if b is None:
            for i, c in enumerate(a):
                if c != 0xFF:
                    return a[:i] + bytes([c + 1])
            return a
        n = 0
        while n < min(len(a), len(b)) and a[n] == b[n]:
            n += 1
        if n < min(len(a), len(b)) and a[n] < 0xFF and a[n] + 1 < b[n]:
            return a[:n] + bytes([a[n] + 1])
        return a

    # data blocks of ~700 bytes (TensorFlow flushes at 256 KiB; small blocks here so that there are several), 17 entries at
    # most per block so that one block has a second restart point
    blocks, cur, size = [], [], 0
    for k, v in entries:
        cur.append((k, v)); size += len(k) + len(v)
        if size >= 700 or len(cur) == 17:
            blocks.append(cur); cur, size = [], 0
    if cur:
        blocks.append(cur)
    index = []
    for i, b in enumerate(blocks):
        h = emit(block(b, 16))
        nxt = blocks[i + 1][0][0] if i + 1 < len(blocks) else None
        index.append((separator(b[-1][0], nxt), h))
    meta = emit(block([], 16))
    idx = emit(block(index, 1))
    footer = meta + idx
    out += footer + bytes(40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    with open(PREFIX + ".index", "wb") as f:
        f.write(out)
    print(f"{len(blocks)} data blocks, index {len(out)} bytes, data {len(data)} bytes")


if __name__ == "__main__":
    main()
