"""C++ mirror of the reference interface (hfnet_slam_amd/csrc/host/hfnet_host.hpp), driven by
tests/cpp/test_host_mirror.cpp the way the reference's Examples/Utility test programs drive the originals."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import synth_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mirror_exe(tmp_path_factory):
    from hfnet_slam_amd import build
    build.build()
    exe = str(tmp_path_factory.mktemp("cpp") / "test_host_mirror")
    lib_dir = os.path.join(ROOT, "hfnet_slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"),
                           "-L", lib_dir, "-lhfnet_hip", f"-Wl,-rpath,{lib_dir}"])
    return exe


def _write_input(path, img, nfeat, nlev):
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", img.shape[1], img.shape[0], nfeat, nlev) + np.ascontiguousarray(img).tobytes())


def test_mirror_compiles_and_fails_loudly_without_gpu(mirror_exe, weights_path, tmp_path):
    from hfnet_slam_amd import capi
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    _write_input(str(tmp_path / "in.bin"), synth_image(64, 96, 1), 50, 2)
    r = subprocess.run([mirror_exe, weights_path, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "Failed to load HFNet model" in r.stderr      # the reference prints the same words and exit(-1)s


@pytest.mark.gpu
def test_mirror_matches_oracle(mirror_exe, weights_path, oracle_model, tmp_path):
    from oracle import oracle as O
    w, h, nfeat, nlev = 160, 120, 120, 3
    img = synth_image(h, w, 31)
    _write_input(str(tmp_path / "in.bin"), img, nfeat, nlev)
    r = subprocess.run([mirror_exe, weights_path, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    buf = open(str(tmp_path / "out.bin"), "rb").read()
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a

    flags, n = take(np.int32, 2)
    assert flags == (1 | 4), "Detect ok, 5-arg overload on a LocalAndGlobal model false, IsValid true"
    ok, rk, rd, rg = oracle_model.detect(img, O.MODE_LOCAL_AND_GLOBAL, nfeat, 0.01)
    assert n == len(rk)
    k = take(np.float32, 3 * n).reshape(n, 3)
    assert np.array_equal(k[:, 0], rk["x"]) and np.array_equal(k[:, 1], rk["y"]) and np.array_equal(k[:, 2], rk["response"])
    local = take(np.float32, n * 256).reshape(n, 256)
    assert np.array_equal(local, rd)
    assert np.array_equal(take(np.float32, 4096), rg)
    en = int(take(np.int32, 1)[0])
    rn, rek, red, reg, _ = oracle_model.extract(img, nfeat, 0.01, nlev, 1.2)
    assert en == rn
    ek = take(np.float32, 4 * en).reshape(en, 4)
    assert np.array_equal(ek[:, 0], rek["x"]) and np.array_equal(ek[:, 1], rek["y"]) and np.array_equal(ek[:, 3], rek["octave"].astype(np.float32))
    edesc = take(np.float32, en * 256).reshape(en, 256)
    assert np.array_equal(edesc, red)
    assert take(np.int32, 1)[0] == -1                      # empty image -> -1 (HFextractor.cc:145)
    nb = int(take(np.int32, 1)[0]); m1 = take(np.int32, n); d1 = take(np.float32, n)
    rnb, rm1, rd1 = O.search_by_bow(rd, red, 0.6)
    assert nb == rnb and np.array_equal(m1, rm1) and np.array_equal(d1, rd1)
    nt = int(take(np.int32, 1)[0]); m2 = take(np.int32, n)
    rnt, rm2 = O.search_for_triangulation(rd, red, 0.75)
    assert nt == rnt and np.array_equal(m2, rm2)
    assert take(np.float32, 1)[0] == np.float32(O.descriptor_distance(rd[0], rd[1]))
    nc = int(take(np.int32, 1)[0]); best = take(np.float32, 1)[0]; slots = take(np.int32, nc)
    assert nc >= 1 and 3 in slots and best == np.float32(1.0)
    assert take(np.int32, 1)[0] == 1                       # KeyFrameDescriptorStore calls all succeeded
    sn = take(np.int32, 2); s0 = take(np.int32, n); sd0 = take(np.float32, n); s1 = take(np.int32, en)
    rn1, rs1, _ = O.search_by_bow(red, rd, 0.6)
    assert sn[0] == rnb and np.array_equal(s0, rm1) and np.array_equal(sd0, rd1)
    assert sn[1] == rn1 and np.array_equal(s1, rs1)
    assert take(np.int32, 1)[0] == rnt and np.array_equal(take(np.int32, n), rm2)
