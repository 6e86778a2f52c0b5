"""Builds hfnet_slam_amd/libhfnet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m hfnet_slam_amd.build [--force]

-ffp-contract=off: the kernels spell every fused multiply-add out (fmaf / MFMA), so that results
are bit-identical to the oracle's accumulation order; the compiler must not fuse anything else.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libhfnet_hip.so")
SOURCES = ["weights.cpp", "kernels_conv.hip", "kernels_detect.hip", "kernels_global.hip", "kernels_match.hip", "engine.hip"]
HEADERS = ["common.hpp", "kernels.hpp", "engine.hpp", os.path.join("..", "..", "include", "hfnet_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-result", "-x", "hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout)
        return r.stdout

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
