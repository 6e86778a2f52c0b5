"""Reader for TensorFlow "tensor bundle" checkpoints (`model.ckpt-N.index` + `model.ckpt-N.data-?????-of-?????`),
without TensorFlow: what `hfnet/export_model.py:32` restores (`model.ckpt-83096`) and what a SavedModel keeps
under `variables/`.  Restated from the published formats:

  * the `.index` file is a LevelDB-style sorted table: data blocks of prefix-compressed (key, value) entries with a
    restart array, each followed by a 5-byte trailer (compression type, masked CRC-32C); an index block maps
    separator keys to block handles; a 48-byte footer holds the metaindex and index handles and the magic
    0xdb4775248b80fb57.  Bundles are written uncompressed.
  * key "" holds a BundleHeaderProto (num_shards = 1, endianness = 2, version = 3); every other key is a variable
    name whose value is a BundleEntryProto {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6}.
  * the bytes of a tensor are `size` bytes at `offset` of shard `shard_id`, row-major, little endian.

`write_bundle` produces the same layout (one data block per 4 KiB of entries, one shard) so the importer can be
exercised end to end without TensorFlow.  No real checkpoint is available in this build environment (no network); the
reader is additionally checked against a bundle written by an independent implementation of the formats
(tests/golden/make_tf_bundle.py -> tests/golden/tf_bundle/: shortened index keys, several data blocks, proto3 default
omission, a scalar, a string tensor, optimizer slots), which is as close to TensorFlow's own files as this image allows.
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict
from typing import Dict, Iterator, List, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: "<f4", 2: "<f8", 3: "<i4", 9: "<i8", 19: "<f2"}          # DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64, DT_HALF
_DTYPE_IDS = {np.dtype(v).str: k for k, v in _DTYPES.items()}

# ---------------------------------------------------------------------------------------------- CRC-32C
_CRC_TABLE = None


def _crc_table() -> np.ndarray:
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data: bytes, crc: int = 0) -> int:
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- varints / protobuf
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one protobuf message; length-delimited values as bytes."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, v


def _parse_entry(buf: bytes) -> dict:
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for num, _, v in _fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:                                   # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
            for n2, _, d in _fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _, s in _fields(d):
                        if n3 == 1:
                            size = s
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["sliced"] = True
    return e


# ---------------------------------------------------------------------------------------------- table
def _block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    if ctype != 0:
        raise ValueError("compressed table block (type %d): tensor bundles are written uncompressed" % ctype)
    if verify:
        stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if _mask(crc32c(buf[offset:offset + size + 1])) != stored:
            raise ValueError("table block checksum mismatch")
    return raw


def _entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_index(index_path: str, verify: bool = True) -> "OrderedDict[str, dict]":
    """name -> {dtype, shape, shard_id, offset, size, crc32c}; the header entry is returned under ''."""
    with open(index_path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{index_path}: not a tensor-bundle index (bad table magic)")
    footer = buf[-48:]
    _, p = _varint(footer, 0)
    _, p = _varint(footer, p)                              # metaindex handle, unused
    idx_off, p = _varint(footer, p)
    idx_size, p = _varint(footer, p)
    out: "OrderedDict[str, dict]" = OrderedDict()
    for _, handle in _entries(_block(buf, idx_off, idx_size, verify)):
        off, q = _varint(handle, 0)
        size, q = _varint(handle, q)
        for key, value in _entries(_block(buf, off, size, verify)):
            if key == b"":
                hdr = {"num_shards": 1, "endianness": 0}
                for num, _, v in _fields(value):
                    if num == 1:
                        hdr["num_shards"] = v
                    elif num == 2:
                        hdr["endianness"] = v
                if hdr["endianness"] != 0:
                    raise ValueError("big-endian bundle")
                out[""] = hdr
            else:
                out[key.decode()] = _parse_entry(value)
    return out


def read_checkpoint(prefix: str, names=None, verify_data_crc: bool = False) -> "OrderedDict[str, np.ndarray]":
    """All (or the named) variables of the bundle `prefix` (+ '.index', '.data-…')."""
    index = read_index(prefix + ".index")
    n_shards = index.get("", {"num_shards": 1})["num_shards"]
    shards: Dict[int, np.memmap] = {}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, e in index.items():
        if name == "" or (names is not None and name not in names):
            continue
        if e["sliced"]:
            raise ValueError(f"{name}: partitioned variables are not supported")
        if e["dtype"] not in _DTYPES:
            continue                                      # strings, resources: not weights
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap(f"{prefix}.data-{sid:05d}-of-{n_shards:05d}", dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify_data_crc and e["crc32c"] is not None and _mask(crc32c(raw.tobytes())) != e["crc32c"]:
            raise ValueError(f"{name}: data checksum mismatch")
        out[name] = np.frombuffer(raw.tobytes(), dtype=_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    return out


# ---------------------------------------------------------------------------------------------- writer (tests)
def _table_block(entries: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _emit(f, block: bytes) -> bytes:
    off = f.tell()
    trailer = b"\0"
    f.write(block + trailer + struct.pack("<I", _mask(crc32c(block + trailer))))
    return _put_varint(off) + _put_varint(len(block))


def _msg(num: int, payload: bytes) -> bytes:
    return _put_varint((num << 3) | 2) + _put_varint(len(payload)) + payload


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], block_bytes: int = 4096, data_crc: bool = True) -> None:
    """Single-shard bundle with the layout `read_checkpoint` expects (keys sorted, like BundleWriter)."""
    items = sorted(((k.encode(), np.asarray(v, order="C")) for k, v in tensors.items()), key=lambda kv: kv[0])
    entries: List[Tuple[bytes, bytes]] = [(b"", _put_varint(1 << 3) + _put_varint(1) + _msg(3, _put_varint(1 << 3) + _put_varint(1)))]
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for k, a in items:
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            raw = a.tobytes()
            shape = b"".join(_msg(2, _put_varint(1 << 3) + _put_varint(d)) for d in a.shape)
            e = (_put_varint(1 << 3) + _put_varint(_DTYPE_IDS[a.dtype.str]) + _msg(2, shape) + _put_varint(4 << 3) + _put_varint(f.tell())
                 + _put_varint(5 << 3) + _put_varint(len(raw)) + (_put_varint((6 << 3) | 5) + struct.pack("<I", _mask(crc32c(raw))) if data_crc else b""))
            entries.append((k, e))
            f.write(raw)
    with open(prefix + ".index", "wb") as f:
        index_entries, cur, size = [], [], 0
        for k, v in entries:
            cur.append((k, v))
            size += len(k) + len(v)
            if size >= block_bytes:
                index_entries.append((cur[-1][0], _emit(f, _table_block(cur))))
                cur, size = [], 0
        if cur:
            index_entries.append((cur[-1][0], _emit(f, _table_block(cur))))
        meta = _emit(f, _table_block([]))
        idx = _emit(f, _table_block(index_entries, restart_interval=1))
        footer = meta + idx
        f.write(footer + b"\0" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))


# ---------------------------------------------------------------------------------------------- HF-Net mapping
def import_hfnet(prefix: str, scope: str = "") -> "OrderedDict[str, np.ndarray]":
    """Variables of an HF-Net checkpoint -> the tensors of `weights.tensor_shapes` (depth multiplier, clusters and
    output width recovered from the shapes).  A checkpoint variable matches a tensor when it is named
    `<scope><tensor name>` or ends with `/<tensor name>`; optimizer slots and moving-average shadows are ignored.
    `global_head/vlad/clusters` is stored as [1,1,1,K,D] (hfnet/models/utils/layers.py:78-80) and is flattened."""
    from . import weights as W
    from .spec import net_spec
    index = read_index(prefix + ".index")
    keys = [k for k in index if k and not k.endswith(("/Adam", "/Adam_1", "/Momentum", "/RMSProp", "/RMSProp_1", "/ExponentialMovingAverage"))]

    def find(name: str):
        if scope + name in index:
            return scope + name
        hits = [k for k in keys if k.endswith("/" + name)]
        if len(hits) > 1:
            raise ValueError(f"{name}: ambiguous in checkpoint ({hits}); pass scope=")
        return hits[0] if hits else None

    probe = {n: find(n) for n in ("MobilenetV2/Conv/weights", "global_head/vlad/clusters", "global_head/dimensionality_reduction/biases")}
    missing = [n for n, k in probe.items() if k is None]
    if missing:
        raise ValueError(f"checkpoint has no variable for {missing}")
    stem = index[probe["MobilenetV2/Conv/weights"]]["shape"][-1]
    k_clusters, g = index[probe["global_head/vlad/clusters"]]["shape"][-2:]
    gdim = index[probe["global_head/dimensionality_reduction/biases"]]["shape"][0]
    spec = next((s for s in (net_spec(m, k_clusters, gdim) for m in (0.35, 0.5, 0.75, 1.0, 1.3, 1.4)) if s.stem_out == stem and s.global_channels == g), None)
    if spec is None:
        raise ValueError("checkpoint shapes do not match a MobileNetV2 depth multiplier")
    want = W.tensor_shapes(spec)
    src = {n: find(n) for n in want}
    # BatchNorm gamma is optional: slim.batch_norm defaults to scale=False, and the NetVLAD memberships conv is built outside
    # the mobilenet arg_scope that sets scale=True (hfnet/models/utils/layers.py:71-76) -- the checkpoint has no such
    # variable there.  The HIP library and the oracle read that one missing gamma as 1; a MobileNet gamma that is missing
    # means a truncated or mis-scoped checkpoint and stays an error (it falls into `missing` below).
    for n in [n for n, k in src.items() if k is None and n == "global_head/vlad/memberships/BatchNorm/gamma"]:
        del src[n], want[n]
    missing = [n for n, k in src.items() if k is None]
    if missing:
        raise ValueError(f"{len(missing)} tensors missing from the checkpoint, first: {missing[:5]}")
    raw = read_checkpoint(prefix, names=set(src.values()))
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for n, shape in want.items():
        a = raw[src[n]]
        if a.size != int(np.prod(shape)) or (a.shape != tuple(shape) and tuple(d for d in a.shape if d != 1) != tuple(d for d in shape if d != 1)):
            raise ValueError(f"{n}: checkpoint shape {a.shape}, expected {shape}")
        out[n] = np.ascontiguousarray(a.reshape(shape), dtype=np.float32)
    return out


def main(argv=None) -> int:
    import argparse
    from . import weights as W
    ap = argparse.ArgumentParser(description="TensorFlow HF-Net checkpoint -> HFNETW1 weight container")
    ap.add_argument("checkpoint", help="bundle prefix, e.g. …/model.ckpt-83096 or saved_model/variables/variables")
    ap.add_argument("output", nargs="?", help="container to write (omit with --list)")
    ap.add_argument("--scope", default="", help="prefix of the variable names inside the checkpoint")
    ap.add_argument("--list", action="store_true", help="print the variables of the checkpoint and exit")
    a = ap.parse_args(argv)
    if a.list:
        for k, e in read_index(a.checkpoint + ".index").items():
            if k:
                print(f"{k}\t{e['shape']}\tdtype={e['dtype']}")
        return 0
    if not a.output:
        ap.error("output path required")
    t = import_hfnet(a.checkpoint, a.scope)
    W.save(a.output, t)
    print(f"{os.path.basename(a.output)}: {len(t)} tensors, {sum(v.nbytes for v in t.values()) / 1e6:.1f} MB")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
