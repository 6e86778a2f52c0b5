"""integration/: the adapter header and the patch a maintainer of the reference applies (no GPU, no OpenCV needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _abi_symbols():
    hdr = open(os.path.join(ROOT, "include", "hfnet_hip.h")).read()
    return set(re.findall(r"\b(hfnet_[a-z0-9_]+)\s*\(", hdr)), set(re.findall(r"\b(HFNET_[A-Z0-9_]+)\b", hdr))


def test_adapter_uses_only_declared_abi():
    fns, consts = _abi_symbols()
    for f in ("HFNetHIPModel.h", "hfnet_slam_hip.patch"):
        src = open(os.path.join(ROOT, "integration", f)).read()
        used = set(re.findall(r"\b(hfnet_[a-z0-9_]+)\s*\(", src))
        assert used, f
        assert used <= fns, f"{f} calls undeclared {sorted(used - fns)}"
        for c in set(re.findall(r"\b(HFNET_[A-Z0-9_]+)\b", src)) - {"HFNET_HIP_ROOT"}:
            assert c in consts, f"{f} uses undeclared {c}"
    src = open(os.path.join(ROOT, "integration", "HFNetHIPModel.h")).read()
    # the three Detect overloads, IsValid and Type of include/Extractors/BaseModel.h:38-54 -- in both the real class and the stub
    assert src.count("bool Detect(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors, cv::Mat &globalDescriptors,") == 2
    assert src.count("bool Detect(const cv::Mat &intermediate, cv::Mat &globalDescriptors) override") == 2
    assert src.count("ModelType Type(void) override { return kHFNetHIPModel; }") == 2


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="needs the reference tree and patch(1)")
def test_patch_applies_to_the_reference(tmp_path):
    patch = os.path.join(ROOT, "integration", "hfnet_slam_hip.patch")
    files = re.findall(r"^--- a/(\S+)", open(patch).read(), re.M)
    assert len(files) >= 5
    for f in files:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), tmp_path / f)
    r = subprocess.run(["patch", "-p1", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "kHFNetHIPModel" in open(tmp_path / "include/Extractors/BaseModel.h").read()
    assert '"HFNetHIP"' in open(tmp_path / "src/Settings.cc").read()
