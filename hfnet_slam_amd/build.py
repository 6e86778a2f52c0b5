"""Builds hfnet_slam_amd/libhfnet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m hfnet_slam_amd.build [--force]

-ffp-contract=off: the kernels spell every fused multiply-add out (fmaf / MFMA), so that results
are bit-identical to the oracle's accumulation order; the compiler must not fuse anything else.
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libhfnet_hip.so")
SOURCES = ["weights.cpp", "devmem.cpp", "kernels_conv.hip", "kernels_block.hip", "kernels_detect.hip", "kernels_global.hip", "kernels_match.hip", "kernels_tail.hip", "engine.hip", "api_extract.hip", "api_match.hip", "api_db.hip"]
HEADERS = ["common.hpp", "kernels.hpp", "engine.hpp", "device_util.hpp", os.path.join("..", "..", "include", "hfnet_hip.h")]
# -fno-slp-vectorize: left on, clang packs neighbouring scalar f32 multiplies / adds into v_pk_mul_f32 / v_pk_add_f32, and on gfx950 those
# returned WRONG values in lanes 48-63 of a wave that shared its SIMD with the split-bf16 fused-block kernels (NOTEBOOK.md R4.8:
# tools/dev/xq_repro3.hip reproduces it outside the engine; without the packed instructions 8 000 iterations are clean).  Same bits
# otherwise (the packed forms are two IEEE operations), and beside MFMAs they are slower than two plain ones anyway.
# -target-feature -packed-fp32-ops takes the packed forms away from instruction selection altogether (explicit f32x2 / f32x4 arithmetic would
# still become v_pk_*); the host pass of the compile answers "not a recognized feature for this target (ignoring feature)": filtered below.
FLAGS = ["--offload-arch=gfx950", "--no-offload-compress", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
         "-fPIC", "-Wall", "-Wno-unused-result", "-x", "hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_id() -> str:
    """sha256 (16 hex digits) over every source and header of the library, in a fixed order"""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + [os.path.join("host", "hfnet_host.hpp")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def library_id():
    """the build id compiled into libhfnet_hip.so (HFNET_BUILD_ID), read from the file without loading it; None if absent"""
    try:
        with open(LIB, "rb") as fh:
            m = re.search(rb"hfnet-build-id:([0-9a-f]{16})", fh.read())
        return m.group(1).decode() if m else None
    except OSError:
        return None


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _flags_id() -> str:
    return hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()[:16]


def device_code_objects(lib: str = LIB):
    """[(triple, ELF bytes)] of every gfx950 code object embedded in the library: its .hip_fatbin section is a sequence of
    uncompressed clang offload bundles (magic, u64 entry count, entries {u64 offset, u64 size, u64 triple length, triple})"""
    import struct
    data = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    pos = data.find(magic)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(magic))
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append((triple, data[pos + off:pos + off + size]))
        pos = data.find(magic, pos + 1)
    return out


def packed_f32_instructions(lib: str = LIB):
    """disassembles every device code object of the BUILT library -> (number of code objects, number of instructions,
    [the v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 lines found]).  NOTEBOOK.md R4.8: those instructions returned wrong values on
    gfx950 beside the split-bf16 kernels; FLAGS is meant to keep every one of them out, this checks that it did."""
    import tempfile
    objdump = os.environ.get("HFNET_OBJDUMP") or os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib", "llvm", "bin", "llvm-objdump")
    if not os.path.exists(objdump):
        objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        raise RuntimeError("build: llvm-objdump not found next to hipcc nor under /opt/rocm/lib/llvm/bin (set HFNET_OBJDUMP to its path, or "
                           "HFNET_SKIP_PK_CHECK=1 to link without the packed-f32 check -- tests/test_abi.py still runs it)")
    objs = device_code_objects(lib)
    found, total = [], 0
    pat = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")
    for i, (_, elf) in enumerate(objs):
        with tempfile.NamedTemporaryFile(suffix=".co") as fh:
            fh.write(elf); fh.flush()
            r = subprocess.run([objdump, "-d", "--mcpu=gfx950", fh.name], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("llvm-objdump failed on code object %d:\n%s" % (i, r.stdout[-2000:]))
        for line in r.stdout.splitlines():
            if "\t" in line and "//" in line:
                total += 1
                if pat.search(line):
                    found.append("object %d: %s" % (i, line.strip()))
    return len(objs), total, found


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    sid = source_id()
    if not force and library_id() == sid:
        return LIB
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    # the object cache is keyed on the compile flags as well as on the sources' time stamps: after a flag change (the packed-f32 switch of
    # round 4 was one) every object is rebuilt, not only those whose source happens to be newer
    stamp = os.path.join(OBJ, "flags.id")
    try:
        flags_changed = open(stamp).read().strip() != _flags_id()
    except OSError:
        flags_changed = True
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        extra = [f'-DHFNET_BUILD_ID="hfnet-build-id:{sid}"'] if s == "engine.hip" else []     # (engine.hip exports hfnet_build_id)
        if force or flags_changed or extra or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout)
        return "\n".join(l for l in r.stdout.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    with open(stamp, "w") as fh:
        fh.write(_flags_id() + "\n")
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        tmp = LIB + ".tmp"
        run([hipcc, "--offload-arch=gfx950", "--no-offload-compress", "-shared", "-fPIC", "-o", tmp] + objs)
        if os.environ.get("HFNET_SKIP_PK_CHECK") == "1":      # (the check needs llvm-objdump and disassembles > 100k instructions: the override for boxes without it)
            os.replace(tmp, LIB)
            return LIB
        # a compiler that ignores -packed-fp32-ops on the device pass must not produce a library silently
        n_obj, n_inst, packed = packed_f32_instructions(tmp)
        if n_obj == 0 or n_inst == 0:
            os.remove(tmp)
            raise RuntimeError("build: no gfx950 code object found in the linked library (a compressed offload bundle? the link passes --no-offload-compress)")
        if packed:
            os.remove(tmp)
            raise RuntimeError("build: the device code contains %d packed f32 instructions (NOTEBOOK.md R4.8) although FLAGS disables them; "
                               "this hipcc ignores -packed-fp32-ops?  first: %s" % (len(packed), packed[0]))
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
