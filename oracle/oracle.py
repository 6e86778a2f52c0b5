"""ctypes binding of oracle/libhfnet_oracle.so (TEST INFRASTRUCTURE -- see hfnet_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhfnet_oracle.so")

DESC_DIM = 256
MODE_LOCAL_AND_GLOBAL, MODE_LOCAL, MODE_LOCAL_AND_INTERMEDIATE, MODE_INTERMEDIATE_TO_GLOBAL = 0, 1, 2, 3
TAP_STEM, TAP_BLOCK0, TAP_DESC_HIDDEN, TAP_DESC_RAW, TAP_DET_HIDDEN, TAP_LOGITS = 0, 1, 18, 19, 20, 21
TAP_SCORES_DENSE, TAP_MEMBERSHIPS, TAP_VLAD, N_TAPS = 22, 23, 24, 32

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("octave", "<i4")])


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("hfnet_oracle.c", "hfnet_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def usable_cpus(cap: int = 32) -> int:
    """CPUs this process may really use: the scheduler affinity AND the cgroup quota (a container that sees 256
    CPUs but is throttled to 16 makes an OpenMP team of 256 crawl), capped at `cap`."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(parts[0]) // int(parts[1])))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(_LIB_PATH)
        L.hfo_set_threads(usable_cpus())
        fp, u8p, i32p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_void_p
        L.hfo_model_load.restype = vp
        L.hfo_model_load.argtypes = [C.c_char_p]
        L.hfo_model_free.argtypes = [vp]
        L.hfo_model_info.argtypes = [vp, C.c_int]
        L.hfo_expf.restype = C.c_float
        L.hfo_expf.argtypes = [C.c_float]
        L.hfo_sumsq_tree256.restype = C.c_float
        L.hfo_sumsq_tree256.argtypes = [vp, C.c_int]
        L.hfo_descriptor_distance.restype = C.c_float
        L.hfo_descriptor_distance.argtypes = [vp, vp, C.c_int]
        for name in ("hfo_run_local", "hfo_run_global", "hfo_detect", "hfo_detect_global", "hfo_extract",
                     "hfo_select_keypoints", "hfo_nms_points", "hfo_search_by_bow",
                     "hfo_search_for_triangulation", "hfo_db_candidates"):
            getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Model:
    def __init__(self, path: str):
        self.h = lib().hfo_model_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load weights {path}")
        q = lambda w: lib().hfo_model_info(C.c_void_p(self.h), w)
        self.stem_out, self.c_local, self.c_global, self.n_clusters, self.global_dim = (q(i) for i in range(5))

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().hfo_model_free(C.c_void_p(self.h))
            except TypeError:          # interpreter shutdown: the module globals are already gone
                pass
            self.h = None

    # -- network -----------------------------------------------------------------------------
    def run_local(self, img: np.ndarray, want_global=False, want_intermediate=False, taps=()):
        """Returns dict with scores_nms, desc_map, (intermediate), (global), taps{id: array}."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        hc, wc = h // 8 * 8, w // 8 * 8
        hd, wd = hc // 8, wc // 8
        out = {"scores_nms": np.empty((hc, wc), np.float32), "desc_map": np.empty((hd, wd, DESC_DIM), np.float32)}
        inter = np.empty((hd, wd, self.c_local), np.float32) if want_intermediate else None
        glob = np.empty((self.global_dim,), np.float32) if want_global else None
        tap_arr = (C.c_void_p * N_TAPS)()
        tap_np = {}
        for t in taps:
            tap_np[t] = np.empty(self._tap_shape(t, hc, wc), np.float32)
            tap_arr[t] = tap_np[t].ctypes.data
        ok = lib().hfo_run_local(C.c_void_p(self.h), _p(img), h, w, img.strides[0], _p(out["scores_nms"]),
                                 _p(out["desc_map"]), _p(inter), _p(glob), tap_arr if taps else None)
        if not ok:
            raise RuntimeError("hfo_run_local failed")
        if inter is not None:
            out["intermediate"] = inter
        if glob is not None:
            out["global"] = glob
        out["taps"] = tap_np
        return out

    def _tap_shape(self, t, hc, wc):
        from hfnet_slam_amd.spec import net_spec, same_pad
        spec = net_spec(n_clusters=self.n_clusters, global_dim=self.global_dim)
        h, w = same_pad(hc, 3, 2)[0], same_pad(wc, 3, 2)[0]
        if t == TAP_STEM:
            return (h, w, spec.stem_out)
        shapes = []
        for b in spec.blocks:
            h, w = same_pad(h, 3, b.stride)[0], same_pad(w, 3, b.stride)[0]
            shapes.append((h, w, b.cout))
        if TAP_BLOCK0 <= t < TAP_BLOCK0 + 17:
            return shapes[t - TAP_BLOCK0]
        hd, wd = hc // 8, wc // 8
        hg, wg = shapes[-1][0], shapes[-1][1]
        return {TAP_DESC_HIDDEN: (hd, wd, 256), TAP_DESC_RAW: (hd, wd, 256), TAP_DET_HIDDEN: (hd, wd, 128),
                TAP_LOGITS: (hd, wd, 65), TAP_SCORES_DENSE: (hc, wc), TAP_MEMBERSHIPS: (hg, wg, self.n_clusters),
                TAP_VLAD: (self.n_clusters * self.c_global,)}[t]

    def run_global(self, intermediate: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(intermediate, np.float32)
        g = np.empty((self.global_dim,), np.float32)
        lib().hfo_run_global(C.c_void_p(self.h), _p(x), x.shape[0], x.shape[1], _p(g), None)
        return g

    def detect(self, img: np.ndarray, mode: int, nkeypoints: int, threshold: float):
        """BaseModel::Detect(image, ...).  Returns (ok, kps, desc, global_or_intermediate)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        hd, wd = h // 8, w // 8
        kps = np.zeros((max(nkeypoints, 1),), KP_DTYPE)
        desc = np.zeros((max(nkeypoints, 1), DESC_DIM), np.float32)
        extra = None
        if mode == MODE_LOCAL_AND_GLOBAL:
            extra = np.zeros((self.global_dim,), np.float32)
        elif mode == MODE_LOCAL_AND_INTERMEDIATE:
            extra = np.zeros((hd, wd, self.c_local), np.float32)
        n = C.c_int(0)
        ok = lib().hfo_detect(C.c_void_p(self.h), mode, _p(img), h, w, img.strides[0], nkeypoints,
                              C.c_float(threshold), _p(kps), _p(desc), _p(extra), C.byref(n))
        return bool(ok), kps[:n.value].copy(), desc[:n.value].copy(), extra

    def detect_global(self, intermediate: np.ndarray, mode: int = MODE_INTERMEDIATE_TO_GLOBAL):
        x = np.ascontiguousarray(intermediate, np.float32)
        g = np.zeros((self.global_dim,), np.float32)
        ok = lib().hfo_detect_global(C.c_void_p(self.h), mode, _p(x), x.shape[0], x.shape[1], _p(g))
        return bool(ok), g

    def extract(self, img: np.ndarray, nfeatures=1000, threshold=0.01, nlevels=4, scale_factor=1.2):
        """HFextractor::operator().  Returns (n, kps, desc, global, n_per_level)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        kps = np.zeros((max(nfeatures, 1),), KP_DTYPE)
        desc = np.zeros((max(nfeatures, 1), DESC_DIM), np.float32)
        g = np.zeros((self.global_dim,), np.float32)
        npl = np.zeros((nlevels,), np.int32)
        n = lib().hfo_extract(C.c_void_p(self.h), _p(img), h, w, img.strides[0], nfeatures, C.c_float(threshold),
                              nlevels, C.c_float(scale_factor), _p(kps), _p(desc), _p(g), _p(npl))
        return n, kps[:max(n, 0)].copy(), desc[:max(n, 0)].copy(), g, npl


# -- free functions --------------------------------------------------------------------------

def set_threads(n: int) -> None:
    lib().hfo_set_threads(int(n))


def expf(x: float) -> float:
    return float(lib().hfo_expf(C.c_float(x)))


def resize_linear_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.empty((dh, dw), np.uint8)
    lib().hfo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def extractor_tables(nfeatures, nlevels, scale_factor, width, height):
    sf = np.zeros(nlevels, np.float32)
    fpl, lw, lh = (np.zeros(nlevels, np.int32) for _ in range(3))
    lib().hfo_extractor_tables(nfeatures, nlevels, C.c_float(scale_factor), width, height, _p(sf), _p(fpl), _p(lw), _p(lh))
    return sf, fpl, lw, lh


def simple_nms(scores: np.ndarray, radius=4, iterations=2) -> np.ndarray:
    s = np.ascontiguousarray(scores, np.float32)
    out = np.empty_like(s)
    lib().hfo_simple_nms(_p(s), s.shape[0], s.shape[1], radius, iterations, _p(out))
    return out


def resampler(data: np.ndarray, warp: np.ndarray) -> np.ndarray:
    d = np.ascontiguousarray(data, np.float32)
    wp = np.ascontiguousarray(warp, np.float32)
    b, dh, dw, c = d.shape
    n = wp.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().hfo_resampler(_p(d), _p(wp), _p(out), b, dh, dw, c, n)
    return out


def nms_points(kps: np.ndarray, width: int, height: int, radius: int) -> np.ndarray:
    k = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros_like(k)
    n = lib().hfo_nms_points(_p(k), len(k), width, height, radius, _p(out))
    return out[:n].copy()


def select_keypoints(scores_nms: np.ndarray, threshold: float, kmax: int) -> np.ndarray:
    s = np.ascontiguousarray(scores_nms, np.float32)
    kps = np.zeros((max(kmax, 1),), KP_DTYPE)
    n = lib().hfo_select_keypoints(_p(s), s.shape[0], s.shape[1], C.c_float(threshold), kmax, _p(kps))
    return kps[:n].copy()


def sample_descriptors(desc_map: np.ndarray, kps: np.ndarray, h: int, w: int) -> np.ndarray:
    d = np.ascontiguousarray(desc_map, np.float32)
    k = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros((len(k), d.shape[2]), np.float32)
    lib().hfo_sample_descriptors(_p(d), d.shape[0], d.shape[1], d.shape[2], _p(k), len(k), h, w, _p(out))
    return out


def descriptor_distance(a: np.ndarray, b: np.ndarray) -> float:
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    return float(lib().hfo_descriptor_distance(_p(a), _p(b), a.size))


def bfmatch_l2_crosscheck(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    idx = np.empty((q.shape[0],), np.int32); dist = np.empty((q.shape[0],), np.float32)
    lib().hfo_bfmatch_l2_crosscheck(_p(q), q.shape[0], _p(t), t.shape[0], q.shape[1] if q.size else t.shape[1], _p(idx), _p(dist))
    return idx, dist


def search_by_bow(q: np.ndarray, t: np.ndarray, th_low: float = 0.6):
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    idx = np.empty((q.shape[0],), np.int32); dist = np.empty((q.shape[0],), np.float32)
    n = lib().hfo_search_by_bow(_p(q), q.shape[0], _p(t), t.shape[0], q.shape[1], C.c_float(th_low), _p(idx), _p(dist))
    return n, idx, dist


def search_for_triangulation(d1: np.ndarray, d2: np.ndarray, th_high: float = 0.75, want_sim=False):
    d1 = np.ascontiguousarray(d1, np.float32); d2 = np.ascontiguousarray(d2, np.float32)
    m = np.empty((d1.shape[0],), np.int32)
    sim = np.empty((d1.shape[0], d2.shape[0]), np.float32) if want_sim else None
    n = lib().hfo_search_for_triangulation(_p(d1), d1.shape[0], _p(d2), d2.shape[0], d1.shape[1], C.c_float(th_high), _p(m), _p(sim))
    return (n, m, sim) if want_sim else (n, m)


def match_candidates(query: np.ndarray, train: np.ndarray, train_level, cand_offsets: np.ndarray, cand_index: np.ndarray):
    """best / second-best loop of the windowed matchers; returns (best_idx, best_dist, best_level, second_dist, second_level)"""
    q = np.ascontiguousarray(query, np.float32); t = np.ascontiguousarray(train, np.float32)
    lv = None if train_level is None else np.ascontiguousarray(train_level, np.int32)
    off = np.ascontiguousarray(cand_offsets, np.int32); idx = np.ascontiguousarray(cand_index, np.int32)
    n = q.shape[0]
    bi = np.empty(n, np.int32); bd = np.empty(n, np.float32); bl = np.empty(n, np.int32); sd = np.empty(n, np.float32); sl = np.empty(n, np.int32)
    lib().hfo_match_candidates(_p(q), n, _p(t), _p(lv), q.shape[1], _p(off), _p(idx), _p(bi), _p(bd), _p(bl), _p(sd), _p(sl))
    return bi, bd, bl, sd, sl


def distinctive_descriptors(desc: np.ndarray, set_offsets: np.ndarray) -> np.ndarray:
    d = np.ascontiguousarray(desc, np.float32); off = np.ascontiguousarray(set_offsets, np.int32)
    best = np.empty(len(off) - 1, np.int32)
    lib().hfo_distinctive_descriptors(_p(d), _p(off), len(off) - 1, d.shape[1], _p(best))
    return best


def db_scores(query: np.ndarray, db: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(query, np.float32).ravel(); d = np.ascontiguousarray(db, np.float32)
    s = np.empty((d.shape[0],), np.float32)
    lib().hfo_db_scores(_p(q), _p(d), d.shape[0], d.shape[1], _p(s))
    return s


def db_candidates(scores: np.ndarray, mode: int = 0):
    s = np.ascontiguousarray(scores, np.float32)
    idx = np.empty((max(len(s), 1),), np.int32)
    best = C.c_float(0)
    n = lib().hfo_db_candidates(_p(s), len(s), mode, _p(idx), C.byref(best))
    return idx[:n].copy(), best.value
