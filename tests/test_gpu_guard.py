"""The diagnostic allocator modes of the library (hfnet_slam_amd/csrc/devmem.cpp): HFNET_GUARD_ALLOC=1 / 2 put every device
allocation at the end / start of its own mapping with unmapped address space behind / in front of it, HFNET_GUARD_FILL poisons
fresh memory.  A kernel that reads or writes outside a buffer then faults in the first test that has the defect instead of once
in a thousand runs on somebody else's box (GPUTEST_r04).  Here: the modes really fault where they should (a self-test program in
its own process), and a slice of the suite -- geometries whose tiles hang over every edge, the BaseModel overloads, both
matchers, the database -- runs clean under each of them.  The whole suite and the soaks under the modes: tools/gpu_accept.sh."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def selftest(tmp_path_factory):
    from hfnet_slam_amd import build
    exe = str(tmp_path_factory.mktemp("guard") / "guard_selftest")
    r = subprocess.run([build._hipcc(), "--offload-arch=gfx950", "-O2", "-o", exe, os.path.join(ROOT, "tools", "dev", "guard_selftest.hip"),
                        os.path.join(build.CSRC, "devmem.cpp")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    return exe


def _run(exe, what, **env):
    e = dict(os.environ); e.pop("HFNET_GUARD_ALLOC", None); e.pop("HFNET_GUARD_FILL", None); e.update(env)
    return subprocess.run([exe, what], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)


def test_guarded_allocations_fault_one_element_outside(engine, selftest):
    r = _run(selftest, "inside", HFNET_GUARD_ALLOC="1")
    assert r.returncode == 0 and "inside ok (guard mode 1)" in r.stdout, r.stdout
    r = _run(selftest, "after", HFNET_GUARD_ALLOC="1")
    assert r.returncode != 0 and "Memory access fault" in r.stdout, r.stdout
    r = _run(selftest, "before", HFNET_GUARD_ALLOC="2")
    assert r.returncode != 0 and "Memory access fault" in r.stdout, r.stdout
    r = _run(selftest, "inside", HFNET_GUARD_ALLOC="2")
    assert r.returncode == 0 and "inside ok (guard mode 2)" in r.stdout, r.stdout


SLICE = ["tests/test_gpu_parity.py::test_random_geometries_match_oracle", "tests/test_gpu_parity.py::test_modes_and_overloads",
         "tests/test_gpu_parity.py::test_database", "tests/test_gpu_parity.py::test_database_batched_screen_geometries",
         "tests/test_gpu_parity.py::test_bow_sweep_many_pairs", "tests/test_gpu_soak.py::test_randomised_soak_20s"]


@pytest.mark.parametrize("mode", [{"HFNET_GUARD_ALLOC": "1"}, {"HFNET_GUARD_ALLOC": "2"}, {"HFNET_GUARD_FILL": "ff"}],
                         ids=["guard_end", "guard_start", "poison_ff"])
def test_suite_slice_is_clean_under_the_allocator_modes(engine, mode):
    e = dict(os.environ); e.update(mode); e["HFNET_SOAK_LOG"] = os.devnull
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + SLICE, cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


def test_bad_inputs_and_healthy_runs_report_status(engine, tmp_path):
    """a directory as weight container used to reach blob.resize((size_t)-1) -- std::length_error, std::terminate in the host process; a healthy
    model / extractor reports no device-side fault bits"""
    import numpy as np
    from conftest import synth_image
    from hfnet_slam_amd import capi
    with pytest.raises(capi.HfnetError) as ei:
        capi.Engine(str(tmp_path), 0)
    assert ei.value.status in (capi.ERR_IO, capi.ERR_INTERNAL)
    m = capi.Model(engine, capi.MODE_LOCAL_AND_GLOBAL, 88, 80, max_keypoints=2)          # (few keypoints on a large cell grid: the R5.1 shape)
    st, k, d, g = m.detect(synth_image(88, 80, 3), 2, 0.0)
    assert st == capi.OK and len(k) == 2 and m.device_faults() == 0
    m.close()
    x = capi.Extractor(engine, 160, 120, 50, 0.0, 1.2, 2, max_batch=2)
    x.extract_batch(np.stack([synth_image(120, 160, 5), synth_image(120, 160, 6)]))
    assert x.device_faults() == 0
    x.close()
