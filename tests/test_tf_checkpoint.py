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
