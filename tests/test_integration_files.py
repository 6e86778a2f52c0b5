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
    hdr = open(os.path.join(ROOT, "integration", "HFNetHIPModel.h")).read()
    used = set(re.findall(r"\b(hfnet_[a-z0-9_]+)\s*\(", hdr))
    assert used and used <= fns, f"HFNetHIPModel.h calls undeclared {sorted(used - fns)}"
    for c in set(re.findall(r"\b(HFNET_[A-Z0-9_]+)\b", hdr)) - {"HFNET_HIP_ROOT"}:
        assert c in consts, f"HFNetHIPModel.h uses undeclared {c}"
    # the three Detect overloads, IsValid and Type of include/Extractors/BaseModel.h:38-54 -- in both the real class and the stub
    assert hdr.count("bool Detect(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors, cv::Mat &globalDescriptors,") == 2
    assert hdr.count("bool Detect(const cv::Mat &intermediate, cv::Mat &globalDescriptors) override") == 2
    assert hdr.count("ModelType Type(void) override { return kHFNetHIPModel; }") == 2
    # the patch reaches the library only through the helpers of that header: every HIP* name it uses is defined there, and it
    # never calls the C ABI behind their back
    patch = open(os.path.join(ROOT, "integration", "hfnet_slam_hip.patch")).read()
    added = "\n".join(l[1:] for l in patch.split("\n") if l.startswith("+") and not l.startswith("+++"))
    assert not re.findall(r"\bhfnet_[a-z0-9_]+\s*\(", added)
    helpers = set(re.findall(r"\b(HIP[A-Z][A-Za-z]+)\b", added))
    assert {"HIPSearchByBoW", "HIPSearchForTriangulation", "HIPGlobalDatabase", "HIPKeyFrameStore"} <= helpers
    for h in helpers:
        assert re.search(r"\b(inline bool|class) %s\b" % h, hdr), f"patch uses {h}, which HFNetHIPModel.h does not define"
    # every call site falls back to the reference's CPU code when the helper reports false (no unconditional #else)
    assert "#else" not in added


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
    # the five call sites of SURVEY.md 8b: both SearchByBoW bodies, SearchForTriangulation, the database (add / erase / clear /
    # both scans) and LocalMapping's neighbour loop
    m = open(tmp_path / "src/Matcher.cc").read()
    assert m.count("HIPSearchByBoW(") == 2 and m.count("HIPSearchForTriangulation(") == 1 and "PrecomputeTriangulation" in m
    k = open(tmp_path / "src/KeyFrameDatabase.cc").read()
    assert k.count("mHipDatabase.Query(") == 2 and k.count("mHipDatabase.Erase(") == 2 and "mHipDatabase.Add(" in k and "mHipDatabase.Clear(" in k
    assert "PrecomputeTriangulation(mpCurrentKeyFrame, vpNeighKFs)" in open(tmp_path / "src/LocalMapping.cc").read()
