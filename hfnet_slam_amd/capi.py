"""ctypes binding of the C ABI in include/hfnet_hip.h (libhfnet_hip.so).

Thin plumbing for tests / bench: every call goes through the same `extern "C"` entry points a
C++ SLAM build would bind.  There is no CPU fallback: without the built library
or without a GPU these calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhfnet_hip.so")

DESC_DIM = 256
OK, ERR_INVALID_ARG, ERR_WRONG_MODE, ERR_SHAPE, ERR_DEVICE, ERR_IO, ERR_CAPACITY, ERR_INTERNAL = range(8)
MODE_LOCAL_AND_GLOBAL, MODE_LOCAL, MODE_LOCAL_AND_INTERMEDIATE, MODE_INTERMEDIATE_TO_GLOBAL = 0, 1, 2, 3
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("octave", "<i4")])

# every symbol include/hfnet_hip.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "hfnet_last_error", "hfnet_abi_version", "hfnet_build_id", "hfnet_device_count",
    "hfnet_engine_create", "hfnet_engine_destroy", "hfnet_engine_info", "hfnet_engine_set_option", "hfnet_engine_get_option",
    "hfnet_engine_synchronize", "hfnet_engine_fence",
    "hfnet_model_create", "hfnet_model_destroy", "hfnet_model_is_valid", "hfnet_model_mode",
    "hfnet_model_detect", "hfnet_model_detect_global", "hfnet_model_tap", "hfnet_model_device_faults",
    "hfnet_extractor_create", "hfnet_extractor_destroy", "hfnet_extractor_tables",
    "hfnet_extractor_extract", "hfnet_extractor_extract_batch", "hfnet_extractor_last_timing", "hfnet_extractor_device_faults", "hfnet_extractor_tap", "hfnet_host_register", "hfnet_host_unregister",
    "hfnet_descriptor_distance", "hfnet_resampler", "hfnet_match_search_by_bow", "hfnet_match_search_by_bow_batch", "hfnet_match_search_for_triangulation", "hfnet_match_search_for_triangulation_batch",
    "hfnet_match_candidates", "hfnet_distinctive_descriptors",
    "hfnet_extractor_attach_store", "hfnet_store_create", "hfnet_store_destroy", "hfnet_store_put", "hfnet_store_put_extracted", "hfnet_store_rows", "hfnet_store_set_flags", "hfnet_store_search_by_bow",
    "hfnet_store_search_for_triangulation",
    "hfnet_db_create", "hfnet_db_destroy", "hfnet_db_add", "hfnet_db_erase", "hfnet_db_clear", "hfnet_db_query", "hfnet_db_query_batch",
    "hfnet_profile_enable", "hfnet_profile_reset", "hfnet_profile_filter", "hfnet_profile_count", "hfnet_profile_get",
]


class HfnetError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"hfnet status {status}: {msg}")
        self.status = status


_lib = None


def lib() -> C.CDLL:
    """loads libhfnet_hip.so -- after checking that it was built from the sources in this tree (the build id compiled
    into the library is the hash of csrc/ + include/; a stale or missing library is rebuilt when hipcc is here)"""
    global _lib
    if _lib is None:
        from . import build as _build
        want = _build.source_id()
        have = _build.library_id()
        if have != want:
            try:
                _build.build()
            except Exception as ex:
                raise RuntimeError(f"{LIB_PATH} is {'missing' if have is None else 'stale (built from ' + have + ', tree is ' + want + ')'} "
                                   f"and cannot be rebuilt here: {ex} (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.hfnet_last_error.restype = C.c_char_p
        L.hfnet_build_id.restype = C.c_char_p
        if L.hfnet_build_id().decode() != want:
            raise RuntimeError(f"{LIB_PATH} reports build id {L.hfnet_build_id().decode()}, the sources hash to {want}")
        for s in SYMBOLS:
            getattr(L, s)  # AttributeError if the library does not export it
        for s in ("hfnet_engine_destroy", "hfnet_model_destroy", "hfnet_extractor_destroy", "hfnet_db_destroy", "hfnet_store_destroy"):
            getattr(L, s).restype = None
            getattr(L, s).argtypes = [C.c_void_p]
        _lib = L
    return _lib


def last_error() -> str:
    return lib().hfnet_last_error().decode()


def _chk(status: int) -> None:
    if status != OK:
        raise HfnetError(status, last_error())


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def host_register(a: np.ndarray) -> None:
    """page-lock a (contiguous) array for DMA: host-pointer batch calls whose buffers are all registered skip the staging copies"""
    assert a.flags["C_CONTIGUOUS"]
    _chk(lib().hfnet_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)))


def host_unregister(a: np.ndarray) -> None:
    _chk(lib().hfnet_host_unregister(C.c_void_p(a.ctypes.data)))


def build_id() -> str:
    return lib().hfnet_build_id().decode()


def device_count() -> int:
    return int(lib().hfnet_device_count())


class Engine:
    def __init__(self, weights_path: str, device: int = 0):
        self.h = C.c_void_p()
        _chk(lib().hfnet_engine_create(int(device), weights_path.encode(), C.byref(self.h)))
        q = lambda w: lib().hfnet_engine_info(self.h, w)
        self.stem_out, self.c_local, self.c_global, self.n_clusters, self.global_dim, self.device = (q(i) for i in range(6))

    def close(self):
        if self.h:
            lib().hfnet_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        _chk(lib().hfnet_engine_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        _chk(lib().hfnet_engine_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    OPTIONS = ("fuse_blocks", "fuse_max_layer", "fused_variant", "fuse_stem", "dense_desc", "conv_wlds", "two_streams", "graph", "pinned_frames", "db_gemm_min_queries", "db_screen_min_rows", "fuse_min_wgs", "copy_threads", "tail_fuse", "dedupe_taps", "pyramid_fuse", "resize_band", "fc_tile", "interleave", "host_global", "det_fuse", "match_screen_bf16", "tri_screen_bf16", "desc_bf16x3", "global_bf16x3", "scores_bf16x3", "join_fused_branch", "match_stats")

    def options(self) -> dict:
        return {n: self.get_option(n) for n in self.OPTIONS}

    def synchronize(self):
        _chk(lib().hfnet_engine_synchronize(self.h))

    def fence(self):
        _chk(lib().hfnet_engine_fence(self.h))

    # ---- Matcher ---------------------------------------------------------------------------
    def descriptor_distance(self, a, b) -> float:
        a = np.ascontiguousarray(a, np.float32).ravel(); b = np.ascontiguousarray(b, np.float32).ravel()
        out = C.c_float(0)
        _chk(lib().hfnet_descriptor_distance(self.h, _p(a), _p(b), a.size, C.byref(out)))
        return out.value

    def search_by_bow(self, q, t, th_low: float = 0.6):
        q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
        dim = q.shape[1] if q.ndim == 2 and q.shape[0] else t.shape[1]
        match = np.full((q.shape[0],), -2, np.int32); dist = np.zeros((q.shape[0],), np.float32)
        n = C.c_int(-1)
        _chk(lib().hfnet_match_search_by_bow(self.h, _p(q), q.shape[0], _p(t), t.shape[0], dim, C.c_float(th_low),
                                             _p(match), _p(dist), C.byref(n), 0))
        return n.value, match, dist

    def search_by_bow_batch(self, sets, n_rows, pairs, th_low: float = 0.6):
        """sets: [S, max_rows, dim] float32 (host); n_rows: [S]; pairs: list of (query_set, train_set)."""
        sets = np.ascontiguousarray(sets, np.float32)
        n_rows = np.ascontiguousarray(n_rows, np.int32)
        qs = np.ascontiguousarray([p[0] for p in pairs], np.int32); ts = np.ascontiguousarray([p[1] for p in pairs], np.int32)
        S, mr, dim = sets.shape
        match = np.full((len(pairs), mr), -2, np.int32); dist = np.zeros((len(pairs), mr), np.float32); cnt = np.full((len(pairs),), -1, np.int32)
        _chk(lib().hfnet_match_search_by_bow_batch(self.h, len(pairs), _p(sets), C.c_size_t(mr * dim), _p(n_rows), S, _p(qs), _p(ts), mr, dim,
                                                   C.c_float(th_low), _p(match), _p(dist), _p(cnt), 0))
        return cnt, match, dist

    def search_for_triangulation_batch(self, sets, n_rows, pairs, th_high: float = 0.75):
        """sets: [S, max_rows, dim] float32 (host); n_rows: [S]; pairs: list of (set1, set2)."""
        sets = np.ascontiguousarray(sets, np.float32)
        n_rows = np.ascontiguousarray(n_rows, np.int32)
        s1 = np.ascontiguousarray([p[0] for p in pairs], np.int32); s2 = np.ascontiguousarray([p[1] for p in pairs], np.int32)
        S, mr, dim = sets.shape
        match = np.full((len(pairs), mr), -2, np.int32); cnt = np.full((len(pairs),), -1, np.int32)
        _chk(lib().hfnet_match_search_for_triangulation_batch(self.h, len(pairs), _p(sets), C.c_size_t(mr * dim), _p(n_rows), S, _p(s1), _p(s2),
                                                              mr, dim, C.c_float(th_high), _p(match), _p(cnt), 0))
        return cnt, match

    def search_for_triangulation(self, d1, d2, th_high: float = 0.75):
        d1 = np.ascontiguousarray(d1, np.float32); d2 = np.ascontiguousarray(d2, np.float32)
        dim = d1.shape[1] if d1.ndim == 2 and d1.shape[0] else d2.shape[1]
        match = np.full((d1.shape[0],), -2, np.int32)
        n = C.c_int(-1)
        _chk(lib().hfnet_match_search_for_triangulation(self.h, _p(d1), d1.shape[0], _p(d2), d2.shape[0], dim,
                                                        C.c_float(th_high), _p(match), C.byref(n), 0))
        return n.value, match

    def match_candidates(self, query, train, train_level, cand_offsets, cand_index):
        """the candidate loop of the windowed matchers; returns (best_idx, best_dist, best_level, second_dist, second_level)"""
        q = np.ascontiguousarray(query, np.float32); t = np.ascontiguousarray(train, np.float32)
        lv = None if train_level is None else np.ascontiguousarray(train_level, np.int32)
        off = np.ascontiguousarray(cand_offsets, np.int32); idx = np.ascontiguousarray(cand_index, np.int32)
        n = q.shape[0]
        dim = q.shape[1] if q.ndim == 2 and n else (t.shape[1] if t.ndim == 2 else 256)
        bi = np.full(n, -7, np.int32); bd = np.zeros(n, np.float32); bl = np.full(n, -7, np.int32); sd = np.zeros(n, np.float32); sl = np.full(n, -7, np.int32)
        _chk(lib().hfnet_match_candidates(self.h, _p(q), n, _p(t), t.shape[0], _p(lv), dim, _p(off), _p(idx) if idx.size else None, _p(bi), _p(bd), _p(bl),
                                          _p(sd), _p(sl), 0))
        return bi, bd, bl, sd, sl

    def distinctive_descriptors(self, desc, set_offsets):
        d = np.ascontiguousarray(desc, np.float32); off = np.ascontiguousarray(set_offsets, np.int32)
        best = np.full(len(off) - 1, -7, np.int32)
        _chk(lib().hfnet_distinctive_descriptors(self.h, _p(d), _p(off), len(off) - 1, d.shape[1], _p(best)))
        return best

    def resampler(self, data, warp):
        """Resampler(data [B,H,W,C], warp [B,N,2]) -> [B,N,C]  (BaseModel.cc:491-562)"""
        d = np.ascontiguousarray(data, np.float32); w = np.ascontiguousarray(warp, np.float32)
        b, dh, dw, c = d.shape
        out = np.zeros((b, w.shape[1], c), np.float32)
        _chk(lib().hfnet_resampler(self.h, _p(d), _p(w), _p(out), b, dh, dw, c, w.shape[1]))
        return out

    # ---- profiling -------------------------------------------------------------------------
    def profile_enable(self, on: bool):
        _chk(lib().hfnet_profile_enable(self.h, int(on)))

    def profile_filter(self, name):
        _chk(lib().hfnet_profile_filter(self.h, name.encode() if name else None))

    def profile_reset(self):
        _chk(lib().hfnet_profile_reset(self.h))

    def profile(self):
        """{name: (launches, total_ms)}"""
        out = {}
        for i in range(lib().hfnet_profile_count(self.h)):
            name = C.create_string_buffer(64); n = C.c_int(0); ms = C.c_double(0)
            _chk(lib().hfnet_profile_get(self.h, i, name, 64, C.byref(n), C.byref(ms)))
            out[name.value.decode()] = (n.value, ms.value)
        return out


class Model:
    """== one BaseModel instance (include/Extractors/BaseModel.h:38-54)."""

    def __init__(self, engine: Engine, mode: int, height: int, width: int, max_keypoints: int = 1000):
        self.engine, self.mode, self.height, self.width, self.max_keypoints = engine, mode, height, width, max_keypoints
        self.h = C.c_void_p()
        _chk(lib().hfnet_model_create(engine.h, mode, height, width, max_keypoints, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().hfnet_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def is_valid(self) -> bool:
        return bool(lib().hfnet_model_is_valid(self.h))

    def detect(self, img: np.ndarray, n_keypoints: int, threshold: float, with_aux=None):
        """Returns (status, kps, desc, aux).  status != OK mirrors the reference's `return false`."""
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
            img = np.ascontiguousarray(img, np.uint8)
        kps = np.zeros((max(n_keypoints, 1),), KP_DTYPE)
        desc = np.zeros((max(n_keypoints, 1), DESC_DIM), np.float32)
        if with_aux is None:
            with_aux = self.mode != MODE_LOCAL
        aux = None
        if with_aux:
            if self.mode == MODE_LOCAL_AND_INTERMEDIATE:
                aux = np.zeros((self.height // 8, self.width // 8, self.engine.c_local), np.float32)
            else:
                aux = np.zeros((self.engine.global_dim,), np.float32)
        n = C.c_int(0)
        st = lib().hfnet_model_detect(self.h, _p(img), img.strides[0], n_keypoints, C.c_float(threshold), _p(kps), _p(desc),
                                      _p(aux), C.byref(n))
        return st, kps[:n.value].copy(), desc[:n.value].copy(), aux

    def detect_global(self, intermediate: np.ndarray):
        x = np.ascontiguousarray(intermediate, np.float32)
        if self.mode == MODE_INTERMEDIATE_TO_GLOBAL and x.size != self.height * self.width * self.engine.c_local:
            raise ValueError(f"intermediate map of {x.shape} given to a {self.height}x{self.width}x{self.engine.c_local} model")
        g = np.zeros((self.engine.global_dim,), np.float32)
        st = lib().hfnet_model_detect_global(self.h, _p(x), _p(g))
        return st, g

    def device_faults(self) -> int:
        """HFNET_DEVICE_FAULT_* bits (0 in a healthy run): a kernel had to bound an index it read from device memory"""
        b = C.c_uint(0)
        _chk(lib().hfnet_model_device_faults(self.h, C.byref(b)))
        return int(b.value)

    def tap(self, tap_id: int, shape=None) -> np.ndarray:
        cap = 1 << 26
        buf = np.empty((cap,), np.float32)
        cnt = C.c_size_t(0)
        _chk(lib().hfnet_model_tap(self.h, tap_id, _p(buf), C.c_size_t(cap), C.byref(cnt)))
        out = buf[:cnt.value].copy()
        return out.reshape(shape) if shape is not None else out


class Extractor:
    """== HFextractor (include/Extractors/HFextractor.h) over the engine's per-level models."""

    def __init__(self, engine: Engine, width: int, height: int, n_features=1000, threshold=0.01, scale_factor=1.2,
                 n_levels=4, max_batch=1):
        self.engine, self.width, self.height, self.n_features, self.n_levels, self.max_batch = \
            engine, width, height, n_features, n_levels, max_batch
        self.h = C.c_void_p()
        _chk(lib().hfnet_extractor_create(engine.h, width, height, n_features, C.c_float(threshold), C.c_float(scale_factor),
                                          n_levels, max_batch, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().hfnet_extractor_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tables(self):
        sf = np.zeros(self.n_levels, np.float32)
        fpl, lw, lh = (np.zeros(self.n_levels, np.int32) for _ in range(3))
        _chk(lib().hfnet_extractor_tables(self.h, _p(sf), _p(fpl), _p(lw), _p(lh)))
        return sf, fpl, lw, lh

    def device_faults(self) -> int:
        """see Model.device_faults"""
        b = C.c_uint(0)
        _chk(lib().hfnet_extractor_device_faults(self.h, C.byref(b)))
        return int(b.value)

    def tap(self, tap_id: int, n_frames: int):
        """Diagnostics: a tensor of the last call (of n_frames frames), per level: list of arrays [n_frames, H_l, W_l(, C)].  Only taps whose
        per-level shapes the wrapper knows: 22 (dense scores) and 25 (scores after NMS), both at the cropped level size."""
        assert tap_id in (22, 25)
        cap = 1 << 28
        buf = np.empty((cap,), np.float32)
        cnt = C.c_size_t(0)
        _chk(lib().hfnet_extractor_tap(self.h, tap_id, _p(buf), C.c_size_t(cap), C.byref(cnt)))
        _, _, lw, lh = self.tables()
        out, off = [], 0
        for l in range(self.n_levels):
            hc, wc = int(lh[l]) // 8 * 8, int(lw[l]) // 8 * 8
            n = n_frames * hc * wc
            out.append(buf[off:off + n].reshape(n_frames, hc, wc).copy())
            off += n
        assert off == cnt.value, (off, cnt.value)
        return out

    def last_timing(self):
        """host-side stamps (us since entry) of the last latency-path call: image staged, enqueued, local results seen,
        unpacked, stream drained, return"""
        t = np.zeros(6, np.float64)
        _chk(lib().hfnet_extractor_last_timing(self.h, _p(t), 6))
        return t

    def extract(self, img: np.ndarray, out=None):
        """HFextractor::operator().  Returns (n, kps, desc, global, n_per_level).
        out: (kps[n_features], desc[n_features, 256], global[G], n_per_level[n_levels]) caller-owned buffers to fill (what a C++
        caller of the ABI passes); the returned kps / desc are then views of their first n rows instead of fresh copies."""
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
            img = np.ascontiguousarray(img, np.uint8)
        if out is None:
            kps = np.zeros((self.n_features,), KP_DTYPE)
            desc = np.zeros((self.n_features, DESC_DIM), np.float32)
            g = np.zeros((self.engine.global_dim,), np.float32)
            npl = np.zeros((self.n_levels,), np.int32)
        else:
            kps, desc, g, npl = out
            assert kps.shape == (self.n_features,) and desc.shape == (self.n_features, DESC_DIM) and desc.dtype == np.float32
        n = C.c_int(0)
        _chk(lib().hfnet_extractor_extract(self.h, _p(img), img.strides[0], _p(kps), _p(desc), _p(g), C.byref(n), _p(npl)))
        k = max(n.value, 0)
        if out is None:
            return n.value, kps[:k].copy(), desc[:k].copy(), g, npl
        return n.value, kps[:k], desc[:k], g, npl

    def output_buffers(self):
        """caller-owned result buffers for extract(img, out=...)"""
        return (np.zeros((self.n_features,), KP_DTYPE), np.zeros((self.n_features, DESC_DIM), np.float32),
                np.zeros((self.engine.global_dim,), np.float32), np.zeros((self.n_levels,), np.int32))

    def extract_batch(self, imgs: np.ndarray, out=None):
        """imgs: [F, H, W] uint8 (host).  Returns (n[F], kps[F, n_features], desc[F, n_features, 256], global[F, G]).
        out: the tuple a previous call returned, to write into the same (already paged-in) arrays again."""
        imgs = np.ascontiguousarray(imgs, np.uint8)
        f = imgs.shape[0]
        if out is not None:
            n, kps, desc, g = out
            assert kps.shape == (f, self.n_features) and desc.shape == (f, self.n_features, DESC_DIM)
        else:
            kps = np.zeros((f, self.n_features), KP_DTYPE)
            desc = np.zeros((f, self.n_features, DESC_DIM), np.float32)
            g = np.zeros((f, self.engine.global_dim), np.float32)
            n = np.zeros((f,), np.int32)
        _chk(lib().hfnet_extractor_extract_batch(self.h, f, _p(imgs), imgs.strides[1], C.c_size_t(imgs.strides[0]), _p(kps), _p(desc),
                                                 _p(g), _p(n), 0))
        return n, kps, desc, g

    def attach_store(self, store, first_slot: int = 0):
        """every following host-pointer extraction also leaves frame f in slot (first_slot + f) % n_sets of `store`"""
        _chk(lib().hfnet_extractor_attach_store(self.h, store.h if store is not None else None, int(first_slot)))

    def extract_batch_device(self, n_frames, d_images, row_stride, frame_stride, d_kps, d_desc, d_global, d_n):
        """All pointers are raw device addresses (ints); only enqueues work on the engine's GPU."""
        _chk(lib().hfnet_extractor_extract_batch(self.h, int(n_frames), C.c_void_p(d_images), int(row_stride), C.c_size_t(frame_stride),
                                                 C.c_void_p(d_kps), C.c_void_p(d_desc), C.c_void_p(d_global), C.c_void_p(d_n), 1))


ROWS_ALL, ROWS_FLAGGED, ROWS_UNFLAGGED = 0, 1, 2


class Store:
    """Device-resident descriptor sets (one slot per keyframe) for the batched matchers."""

    def __init__(self, engine: Engine, n_sets: int, max_rows: int, dim: int = DESC_DIM):
        self.engine, self.n_sets, self.max_rows, self.dim = engine, n_sets, max_rows, dim
        self.h = C.c_void_p()
        _chk(lib().hfnet_store_create(engine.h, n_sets, max_rows, dim, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().hfnet_store_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def put(self, slot: int, rows: np.ndarray):
        r = np.ascontiguousarray(rows, np.float32)
        _chk(lib().hfnet_store_put(self.h, int(slot), _p(r), int(r.shape[0])))

    def put_extracted(self, slot: int, extractor, frame: int = 0):
        _chk(lib().hfnet_store_put_extracted(self.h, int(slot), extractor.h, int(frame)))

    def rows(self, slot: int) -> int:
        return lib().hfnet_store_rows(self.h, int(slot))

    def set_flags(self, slot: int, flags: np.ndarray):
        f = np.ascontiguousarray(flags, np.uint8)
        _chk(lib().hfnet_store_set_flags(self.h, int(slot), _p(f), int(f.shape[0])))

    def _pairs(self, pairs):
        return (np.ascontiguousarray([p[0] for p in pairs], np.int32), np.ascontiguousarray([p[1] for p in pairs], np.int32))

    def search_by_bow(self, pairs, th_low: float = 0.6, query_rows: int = ROWS_ALL, train_rows: int = ROWS_ALL):
        a, b = self._pairs(pairs)
        match = np.full((len(pairs), self.max_rows), -2, np.int32); dist = np.zeros((len(pairs), self.max_rows), np.float32)
        cnt = np.full((len(pairs),), -1, np.int32)
        _chk(lib().hfnet_store_search_by_bow(self.h, len(pairs), _p(a), _p(b), int(query_rows), int(train_rows), C.c_float(th_low), _p(match), _p(dist), _p(cnt)))
        return cnt, match, dist

    def search_for_triangulation(self, pairs, th_high: float = 0.75, rows1: int = ROWS_ALL, rows2: int = ROWS_ALL):
        a, b = self._pairs(pairs)
        match = np.full((len(pairs), self.max_rows), -2, np.int32); cnt = np.full((len(pairs),), -1, np.int32)
        _chk(lib().hfnet_store_search_for_triangulation(self.h, len(pairs), _p(a), _p(b), int(rows1), int(rows2), C.c_float(th_high), _p(match), _p(cnt)))
        return cnt, match


class Database:
    """== the descriptor store + linear scans of KeyFrameDatabase (src/KeyFrameDatabase.cc:75-104,170-197)."""

    def __init__(self, engine: Engine, capacity: int, dim: int = 4096):
        self.engine, self.capacity, self.dim = engine, capacity, dim
        self.h = C.c_void_p()
        _chk(lib().hfnet_db_create(engine.h, capacity, dim, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().hfnet_db_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, slot: int, desc: np.ndarray):
        d = np.ascontiguousarray(desc, np.float32).ravel()
        assert d.size == self.dim
        _chk(lib().hfnet_db_add(self.h, int(slot), _p(d)))

    def erase(self, slot: int):
        _chk(lib().hfnet_db_erase(self.h, int(slot)))

    def clear(self):
        _chk(lib().hfnet_db_clear(self.h))

    def query(self, q: np.ndarray, mode: int = 0, want_scores=False):
        q = np.ascontiguousarray(q, np.float32).ravel()
        assert q.size == self.dim
        slot = np.zeros((self.capacity,), np.int32); score = np.zeros((self.capacity,), np.float32)
        n = C.c_int(0); best = C.c_float(0)
        scores = np.zeros((self.capacity,), np.float32) if want_scores else None
        _chk(lib().hfnet_db_query(self.h, _p(q), mode, _p(slot), _p(score), C.byref(n), C.byref(best), _p(scores)))
        return slot[:n.value].copy(), score[:n.value].copy(), best.value, scores

    def query_batch(self, qs: np.ndarray, mode: int = 0, want_scores=False):
        """qs: [Q, dim]; returns per-query lists of (slots, scores), best[Q], scores_all[Q, capacity] or None"""
        qs = np.ascontiguousarray(qs, np.float32)
        Q = qs.shape[0]
        assert qs.shape[1] == self.dim
        slot = np.zeros((Q, self.capacity), np.int32); score = np.zeros((Q, self.capacity), np.float32)
        n = np.zeros((Q,), np.int32); best = np.zeros((Q,), np.float32)
        scores = np.zeros((Q, self.capacity), np.float32) if want_scores else None
        _chk(lib().hfnet_db_query_batch(self.h, Q, _p(qs), mode, _p(slot), _p(score), _p(n), _p(best), _p(scores)))
        return [(slot[i, :n[i]].copy(), score[i, :n[i]].copy()) for i in range(Q)], best, scores
