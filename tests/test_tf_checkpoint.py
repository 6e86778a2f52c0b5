"""TensorFlow checkpoint (tensor bundle) reader and the HF-Net variable mapping, without TensorFlow: the writer in the
same module lays files out the way BundleWriter does, so this pins the reader against the documented format only
(no real checkpoint can be fetched here -- see DESIGN.md)."""
import numpy as np
import pytest

from hfnet_slam_amd import tf_checkpoint as T
from hfnet_slam_amd import weights as W
from hfnet_slam_amd.spec import net_spec


def test_crc32c_known_answers():
    assert T.crc32c(b"123456789") == 0xE3069283                    # the CRC-32C check value (RFC 3720 B.4)
    assert T.crc32c(b"\0" * 32) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43


def test_bundle_round_trip_many_blocks(tmp_path):
    rng = np.random.default_rng(5)
    tensors = {f"scope_{i // 7}/layer_{i}/weights": rng.standard_normal((i % 5 + 1, 3)).astype(np.float32) for i in range(300)}
    tensors["global_step"] = np.array(83096, np.int64)
    tensors["half"] = rng.standard_normal((4,)).astype(np.float16)
    prefix = str(tmp_path / "model.ckpt-1")
    T.write_bundle(prefix, tensors, block_bytes=512)
    index = T.read_index(prefix + ".index")
    assert index[""]["num_shards"] == 1 and len(index) == len(tensors) + 1
    assert list(index)[1:] == sorted(tensors)
    got = T.read_checkpoint(prefix, verify_data_crc=True)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v)
    # a flipped bit in a table block is caught by the block checksum, one in the data by the entry checksum
    raw = bytearray(open(prefix + ".index", "rb").read()); raw[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        T.read_index(prefix + ".index")
    raw[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(raw))
    d = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); d[3] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(d))
    with pytest.raises(ValueError):
        T.read_checkpoint(prefix, verify_data_crc=True)
    with pytest.raises(ValueError):
        T.read_index(prefix + ".data-00000-of-00001")


def test_import_hfnet_checkpoint(tmp_path):
    spec = net_spec(0.75, 8, 64)
    ref = W.synthetic_weights(3, spec)
    ck = {}
    for n, a in ref.items():
        ck[n] = a.reshape(1, 1, 1, *a.shape) if n == "global_head/vlad/clusters" else a
        if n.endswith("weights"):
            ck[n + "/Adam"] = np.zeros_like(a); ck[n + "/Adam_1"] = np.zeros_like(a)
    ck["global_step"] = np.array(83096, np.int64)
    ck["beta1_power"] = np.array(0.9, np.float32)
    prefix = str(tmp_path / "model.ckpt-83096")
    T.write_bundle(prefix, ck, data_crc=False)
    got = T.import_hfnet(prefix)
    assert list(got) == list(ref)
    for n in ref:
        assert got[n].shape == ref[n].shape and np.array_equal(got[n], ref[n])
    out = str(tmp_path / "w.hfw")
    assert T.main([prefix, out]) == 0
    back = W.load(out)
    assert W.spec_from_tensors(back).global_dim == 64 and all(np.array_equal(back[n], ref[n]) for n in ref)
    # variables under an outer scope resolve by suffix; a missing variable is an error, not a silent default
    T.write_bundle(prefix, {"tower_0/" + k: v for k, v in ck.items()}, data_crc=False)
    assert np.array_equal(T.import_hfnet(prefix)["MobilenetV2/Conv/weights"], ref["MobilenetV2/Conv/weights"])
    del ck["local_head/detector/Conv_1/biases"]
    T.write_bundle(prefix, ck, data_crc=False)
    with pytest.raises(ValueError, match="missing"):
        T.import_hfnet(prefix)



def test_reader_on_an_independently_written_bundle():
    """tests/golden/tf_bundle/ was written by tests/golden/make_tf_bundle.py, a second implementation of the table / bundle
    formats that shares no code with tf_checkpoint.py (shortened index separators, several data blocks with a second
    restart point, proto3 default omission, a scalar, a DT_STRING tensor, optimizer slots): the reader must decode the
    committed bytes to the values the generator derives from the tensor index."""
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    try:
        import make_tf_bundle as M
    finally:
        sys.path.remove(here)
    prefix = os.path.join(here, "tf_bundle", "model.ckpt-7")
    index = T.read_index(prefix + ".index")
    want = M.tensors()
    assert index[""]["num_shards"] == 1
    assert sorted(k for k in index if k) == sorted(list(want) + ["_CHECKPOINTABLE_OBJECT_GRAPH"])
    assert index["_CHECKPOINTABLE_OBJECT_GRAPH"]["dtype"] == 7 and index["global_step"]["shape"] == []
    first = sorted(want)[0]
    assert index[first]["offset"] == 0 and index[first]["shard_id"] == 0          # fields the writer omitted (proto3 defaults)
    got = T.read_checkpoint(prefix, verify_data_crc=True)
    assert "_CHECKPOINTABLE_OBJECT_GRAPH" not in got                              # strings are not weights
    for k, v in want.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert got["global_head/vlad/clusters"].shape == (1, 1, 1, 2, 32) and int(got["global_step"]) == 83096


def test_checkpoint_without_memberships_gamma(tmp_path):
    """slim.batch_norm defaults to scale=False and the NetVLAD memberships conv is built outside the mobilenet arg_scope
    (hfnet/models/utils/layers.py:71-76): a real checkpoint has no gamma there.  The importer passes the tensor set on
    without it and the oracle (like the HIP library, tests/test_gpu_parity.py) reads the missing gamma as 1."""
    from oracle import oracle as O
    spec = net_spec(0.75, 8, 64)
    ref = W.synthetic_weights(3, spec)
    gname = "global_head/vlad/memberships/BatchNorm/gamma"
    ck = {n: (a.reshape(1, 1, 1, *a.shape) if n == "global_head/vlad/clusters" else a) for n, a in ref.items() if n != gname}
    prefix = str(tmp_path / "model.ckpt-83096")
    T.write_bundle(prefix, ck, data_crc=False)
    got = T.import_hfnet(prefix)
    assert gname not in got and len(got) == len(ref) - 1
    p_without, p_ones = str(tmp_path / "a.hfw"), str(tmp_path / "b.hfw")
    W.save(p_without, got)
    ones = dict(ref); ones[gname] = np.ones_like(ref[gname])
    W.save(p_ones, ones)
    O.build()
    img = np.random.default_rng(4).integers(0, 256, (64, 96), dtype=np.uint8)
    a = O.Model(p_without).run_local(img, want_global=True)
    b = O.Model(p_ones).run_local(img, want_global=True)
    assert np.array_equal(a["global"], b["global"])
    # any other missing BatchNorm tensor is still an error
    del ck["global_head/vlad/memberships/BatchNorm/beta"]
    T.write_bundle(prefix, ck, data_crc=False)
    with pytest.raises(ValueError, match="missing"):
        T.import_hfnet(prefix)
    # ... and so is a missing gamma of any OTHER scope (a truncated / mis-scoped checkpoint must not load as gamma = 1):
    # the importer refuses it, and so does the oracle's container reader
    mb = "MobilenetV2/expanded_conv_3/depthwise/BatchNorm/gamma"
    ck2 = {n: (a.reshape(1, 1, 1, *a.shape) if n == "global_head/vlad/clusters" else a) for n, a in ref.items() if n not in (gname, mb)}
    T.write_bundle(prefix, ck2, data_crc=False)
    with pytest.raises(ValueError, match="missing"):
        T.import_hfnet(prefix)
    p_bad = str(tmp_path / "c.hfw")
    W.save(p_bad, {n: a for n, a in ref.items() if n != mb})
    with pytest.raises(Exception):
        O.Model(p_bad)
