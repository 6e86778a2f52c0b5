"""A short run of the randomised soak (tools/dev/soak.py): random extractor geometries / budgets / call sizes / engine
options, single-frame graph calls, both matchers, the descriptor store with row filters and the database scans, all
bit-exact against the oracle (the batched database query against its own oracle function).  Longer runs:
`python tools/dev/soak.py 600 <seed>` on the GPU box (rounds of ~7000 cases; this is how the null-stream race of
hfnet_db_create / hfnet_store_create was found)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _load(name):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dev", name + ".py")
    spec = importlib.util.spec_from_file_location("hfnet_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_randomised_soak_20s(engine):           # (the fixture initialises torch's HIP runtime before the library's: conftest.py)
    cases, fails = _load("soak").run(20.0, 20260928)
    assert cases >= 20, cases
    assert not fails, fails[:10]


def test_device_pipeline_soak_12s(engine):
    """device-resident calls enqueued back to back without host synchronisation (random call sizes spanning several
    chunks, pairs across call boundaries, random stream options): extraction, global descriptors and matches == oracle"""
    pytest.importorskip("torch")
    rounds, fails = _load("soak_pipeline").run(12.0, 20260929)
    assert rounds >= 3, rounds
    assert not fails, fails[:10]


def test_three_host_threads_on_one_engine_10s(engine):
    """Tracking (extractor + frame-to-frame match), LocalMapping (triangulation matches on the keyframe store) and
    LoopClosing (database add / erase / query) as three host threads on ONE engine, every result against the oracle"""
    counts, fails = _load("soak_threads").run(10.0, 20260930)
    assert min(counts.values()) >= 5, counts
    assert not fails, fails[:10]


def test_tolerance_mode_soak_20s(engine):
    """the three split-bf16 options in random subsets beside random dispatch options, random pyramids / budgets / call sizes, default and sparse-score
    weights: keypoints == the oracle's NMS + top-K on the device's score map (or == the oracle's, scores exact), floats within the stated tolerances"""
    cases, fails, stats = _load("soak_tolerance").run(20.0, 20261001)
    assert cases >= 10, cases
    assert not fails, fails[:10]
    assert stats["overlap"] >= 0.985 * stats["total"], stats
