"""C-ABI surface (no compute calls: runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "hfnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hfnet_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from hfnet_slam_amd import build, capi
    build.build()
    lib = capi.lib()
    declared = _declared()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hfnet_hip.h but not exported"
    assert sorted(capi.SYMBOLS) == declared, "capi.SYMBOLS out of sync with the header"
    assert lib.hfnet_abi_version() == 2


def test_header_cites_reference_for_every_group():
    text = open(os.path.join(ROOT, "include", "hfnet_hip.h")).read()
    for cite in ("BaseModel.h:38-54", "HFextractor.cc:82-284", "Matcher.cc:229-260", "Matcher.cc:845-889",
                 "KeyFrameDatabase.cc:75-104", "Matcher.cc:1893-1900"):
        assert cite in text


def test_no_silent_cpu_fallback(tmp_path):
    """without a GPU the product path must fail loudly, never compute on the CPU"""
    from hfnet_slam_amd import capi, weights
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    p = str(tmp_path / "w.hfw")
    weights.save(p, {k: v for k, v in list(weights.synthetic_weights(7).items())[:5]})
    with pytest.raises(capi.HfnetError) as ei:
        capi.Engine(p, 0)
    assert ei.value.status == capi.ERR_DEVICE
    assert "GPU" in str(ei.value) or "device" in str(ei.value)


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's checker legs (cpu_baseline: the reported CPU number; verify_last_chunk and
    config_bf16x3's tail: the comparison of a run's own outputs, after its timed region) may touch oracle/ -- never the thing measured"""
    pkg = os.path.join(ROOT, "hfnet_slam_amd")
    pat = re.compile(r"libhfnet_oracle|from\s+oracle|import\s+oracle|#include\s*[<\"][^>\"]*hfnet_oracle\.h|dlopen")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) in ("build", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(src), f"{f} reaches into oracle/"
    bench = open(os.path.join(ROOT, "bench.py")).read()
    lines = bench.splitlines()
    uses = [i for i, l in enumerate(lines) if re.search(r"from\s+oracle|import\s+oracle", l)]
    assert len(uses) == 3
    enclosing = sorted([l for l in lines[:u] if l.startswith("def ")][-1].split("(")[0] for u in uses)
    assert enclosing == ["def config_bf16x3", "def cpu_baseline", "def verify_last_chunk"]      # (config_bf16x3: the tolerance check of its own outputs)
    # ... and the timed region of main() sits between two sync_all() calls that no oracle call is near
    timed = bench[bench.index("    t0 = time.perf_counter()\n    for _ in range(args.steps):"):bench.index("    elapsed = max_over_ranks(")]
    assert "oracle" not in timed and "verify" not in timed
    assert "from oracle" not in open(os.path.join(ROOT, "hfnet_slam_amd", "shard.py")).read()


def test_library_is_built_without_packed_f32_instructions():
    """gfx950: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 returned wrong values in lanes 48-63 of a wave that shared its SIMD with the split-bf16
    fused-block kernels (NOTEBOOK.md R4.8; tools/micro/pk_f32_hazard_standalone.hip) -- clang forms them from neighbouring scalar operations
    (SLP) and from f32x2 / f32x4 arithmetic.  The build flags must keep them out of EVERY kernel: all gfx950 code objects of the BUILT
    libhfnet_hip.so are disassembled (build() runs the same check on every link and refuses to produce a library otherwise)."""
    from hfnet_slam_amd import build
    assert "-fno-slp-vectorize" in build.FLAGS and "-packed-fp32-ops" in build.FLAGS
    build.build()
    n_obj, n_inst, packed = build.packed_f32_instructions()
    assert n_obj >= 6, n_obj                    # one code object per .hip source that has kernels
    assert n_inst > 100000, n_inst
    assert not packed, packed[:3]


def test_object_cache_is_keyed_on_the_compile_flags(tmp_path, monkeypatch):
    """after a flag change only engine.o used to be rebuilt (time stamps alone) while the library was stamped with the new id"""
    from hfnet_slam_amd import build
    build.build()
    stamp = os.path.join(build.OBJ, "flags.id")
    assert open(stamp).read().strip() == build._flags_id()
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DHFNET_SOME_NEW_FLAG=1"])
    assert open(stamp).read().strip() != build._flags_id()          # -> build() would schedule every source


def test_no_exception_crosses_the_c_abi():
    """include/hfnet_hip.h: "nothing here throws or exits".  Every extern "C" entry point is a function-try-block whose handler turns a
    C++ exception into HFNET_ERR_INTERNAL (an exception that left an extern "C" function would std::terminate the host process -- the
    SLAM system).  (a) the sources: every `int hfnet_*` / `void hfnet_*` definition with a body of its own ends in that handler;
    (b) the built library: the test hook of hfnet_engine_set_option throws std::bad_alloc / std::runtime_error behind the boundary
    (no engine, no GPU needed) and the call returns status 7 with the text in hfnet_last_error()."""
    import ctypes as C
    from hfnet_slam_amd import build, capi
    n = 0
    for f in ("api_db.hip", "api_extract.hip", "api_match.hip", "engine.hip"):
        lines = open(os.path.join(build.CSRC, f)).read().split("\n")
        for i, l in enumerate(lines):
            m = re.match(r"^(int|void) (hfnet_\w+)\(", l)
            if not m or l.rstrip().endswith("}"):                 # (one-line accessors: no call that can throw)
                continue
            j = i
            while not lines[j].rstrip().endswith("{"):
                j += 1
            assert lines[j].rstrip().endswith(") try {"), (f, m.group(2))
            k = j + 1
            while not lines[k].startswith("}"):
                k += 1
            assert lines[k].startswith("} catch (...) {") and "api_exception()" in lines[k], (f, m.group(2))
            assert ("return" in lines[k]) == (m.group(1) == "int"), (f, m.group(2))
            n += 1
    assert n >= 50, n
    L = capi.lib()
    L.hfnet_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    # the throwing hook exists only with HFNET_TEST_HOOKS=1: without it the name is an ordinary (null-engine) call
    os.environ.pop("HFNET_TEST_HOOKS", None)
    assert L.hfnet_engine_set_option(None, b"debug_throw", 1) == capi.ERR_INVALID_ARG
    os.environ["HFNET_TEST_HOOKS"] = "1"
    assert L.hfnet_engine_set_option(None, b"debug_throw", 1) == capi.ERR_INTERNAL == 7
    assert "out of host memory" in capi.last_error()
    assert L.hfnet_engine_set_option(None, b"debug_throw", 2) == capi.ERR_INTERNAL
    assert "debug_throw" in capi.last_error()
    os.environ.pop("HFNET_TEST_HOOKS", None)
    assert L.hfnet_engine_set_option(None, b"fuse_blocks", 1) == capi.ERR_INVALID_ARG       # (the ordinary null-engine answer)
