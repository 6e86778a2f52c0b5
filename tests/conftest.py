import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights_path(tmp_path_factory):
    from hfnet_slam_amd import weights
    p = str(tmp_path_factory.mktemp("w") / "synthetic_seed7.hfw")
    weights.save(p, weights.synthetic_weights(7))
    return p


@pytest.fixture(scope="session")
def weights_ties_path(tmp_path_factory):
    """detector gain 4: the softmax saturates to exactly 1.0 in many cells -> response ties"""
    from hfnet_slam_amd import weights
    p = str(tmp_path_factory.mktemp("w") / "synthetic_seed7_gain4.hfw")
    weights.save(p, weights.synthetic_weights(7, detector_gain=4.0))
    return p


@pytest.fixture(scope="session", params=[15.0, 16.0, 18.0], ids=["dustbin15", "dustbin16", "dustbin18"])
def sparse_pair(request, tmp_path_factory):
    """(engine, oracle model) on weights whose detector leaves pyramid levels short of their budget (weights.synthetic_weights' dustbin_bias:
    15 -- the coarse levels are short; 16 -- every level on smooth frames; 18 -- whole levels without a candidate)"""
    from hfnet_slam_amd import capi, weights
    from oracle import oracle as O
    p = str(tmp_path_factory.mktemp("w") / f"synthetic_seed7_dustbin{int(request.param)}.hfw")
    weights.save(p, weights.synthetic_weights(7, dustbin_bias=request.param))
    O.build()
    e = capi.Engine(p, 0)
    yield e, O.Model(p), request.param
    e.close()


@pytest.fixture(scope="session")
def oracle_model(weights_path):
    from oracle import oracle as O
    O.build()
    return O.Model(weights_path)


@pytest.fixture(scope="session")
def engine(weights_path):
    # torch (used by one device-resident test for its device buffers) bundles its own HIP runtime: it has to initialise
    # before libhfnet_hip.so pulls in the system one, or it sees no devices afterwards (bench.py has the same order)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    from hfnet_slam_amd import capi
    if capi.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible: " + capi.last_error())
    e = capi.Engine(weights_path, 0)
    yield e
    e.close()


@pytest.fixture
def engine_options(engine):
    """set engine options for the objects a test creates; the defaults come back afterwards"""
    saved = engine.options()

    def apply(opts):
        for k, v in opts.items():
            engine.set_option(k, v)
    yield apply
    for k, v in saved.items():
        engine.set_option(k, v)


def synth_image(h, w, seed, kind="uniform"):
    """(i) iid uniform; (ii) 'natural-ish' = 6 octaves of bilinear-upsampled noise"""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    acc = np.zeros((h, w), np.float64)
    for o in range(6):
        gh, gw = max(2, h >> (6 - o)), max(2, w >> (6 - o))
        g = rng.uniform(0, 1, (gh + 1, gw + 1))
        ys = np.linspace(0, gh - 1e-6, h); xs = np.linspace(0, gw - 1e-6, w)
        y0 = ys.astype(int); x0 = xs.astype(int); fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        acc += (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx) * (0.5 ** (5 - o))
    acc = (acc - acc.min()) / (acc.max() - acc.min())
    return np.clip(acc * 255.0, 0, 255).astype(np.uint8)


def torch_to_host(t):
    """device tensor -> numpy array through PINNED host memory (torch's own HIP runtime would otherwise pin the pageable destination in place:
    the same mechanism behind the rare "Memory access fault ... on address <host heap page>" of NOTEBOOK.md R5.4 -- one abort of the GPU
    suite happened inside a plain `.cpu()` of this file's callers)"""
    import torch
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.synchronize()
    return h.numpy().copy()


def torch_to_device(a, dev):
    """numpy array -> device tensor through pinned host memory (see torch_to_host)"""
    import torch
    src = torch.from_numpy(np.ascontiguousarray(a))
    h = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    h.copy_(src)
    d = h.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    return d
