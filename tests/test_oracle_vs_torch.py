"""C oracle vs the independent PyTorch-CPU fp64 restatement (oracle/torch_ref.py): every
intermediate of the graph, two input sizes (one needs the crop and has odd strides)."""
import numpy as np
import pytest

from conftest import synth_image


@pytest.mark.parametrize("hw", [(64, 96), (93, 131)])
def test_every_intermediate(oracle_model, weights_path, hw):
    from hfnet_slam_amd import spec, weights
    from oracle import oracle as O
    from oracle.torch_ref import TorchHFNet
    img = synth_image(hw[0], hw[1], 7)
    taps = [O.TAP_STEM] + [O.TAP_BLOCK0 + i for i in range(17)] + [O.TAP_LOGITS, O.TAP_SCORES_DENSE, O.TAP_MEMBERSHIPS, O.TAP_VLAD]
    r = oracle_model.run_local(img, want_global=True, want_intermediate=True, taps=taps)
    q = TorchHFNet(weights.load(weights_path), spec.net_spec()).run(img)

    def close(name, a, b, tol):
        b = b.numpy() if hasattr(b, "numpy") else np.asarray(b)
        err = np.abs(np.asarray(a, np.float64) - b).max()
        assert err <= tol * max(1.0, np.abs(b).max()), f"{name}: {err:.3e}"

    close("stem", r["taps"][O.TAP_STEM], q["feats"][1][0].permute(1, 2, 0), 1e-5)
    for i in range(17):
        close(f"layer_{i + 2}", r["taps"][O.TAP_BLOCK0 + i], q["feats"][i + 2][0].permute(1, 2, 0), 2e-5)
    close("intermediate", r["intermediate"], q["feats"][7][0].permute(1, 2, 0), 2e-5)
    close("logits", r["taps"][O.TAP_LOGITS], q["logits"], 2e-5)
    close("scores_dense", r["taps"][O.TAP_SCORES_DENSE], q["scores_dense"], 1e-4)
    close("desc_map", r["desc_map"], q["desc_map"], 1e-5)
    close("memberships", r["taps"][O.TAP_MEMBERSHIPS].reshape(-1, 32), q["memberships"], 2e-4)
    close("vlad", r["taps"][O.TAP_VLAD], q["vlad"], 5e-5)
    close("global", r["global"], q["global"], 1e-5)
    assert abs(np.linalg.norm(r["global"]) - 1) < 1e-6
    # NMS on the oracle's own dense map must agree exactly with the library-op NMS (same input)
    import torch
    dense = r["taps"][O.TAP_SCORES_DENSE]
    ref = TorchHFNet.simple_nms(torch.from_numpy(dense.astype(np.float64))).numpy().astype(np.float32)
    assert np.array_equal(O.simple_nms(dense), ref)
    assert np.array_equal(r["scores_nms"], ref)


def test_global_from_intermediate_equals_fused(oracle_model):
    from oracle import oracle as O
    img = synth_image(72, 88, 9)
    r = oracle_model.run_local(img, want_global=True, want_intermediate=True)
    g = oracle_model.run_global(r["intermediate"])
    assert np.array_equal(g, r["global"])
    ok, g2 = oracle_model.detect_global(r["intermediate"])
    assert ok and np.array_equal(g2, g)
    ok, _ = oracle_model.detect_global(r["intermediate"], mode=O.MODE_LOCAL)     # wrong mode -> false
    assert not ok
    ok, *_ = oracle_model.detect(img, O.MODE_INTERMEDIATE_TO_GLOBAL, 10, 0.01)
    assert not ok
