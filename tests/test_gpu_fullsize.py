"""BASELINE.json's full-size configurations on the GPU: oracle parity where the oracle finishes in seconds,
plus size-independent properties (determinism, batch invariance, budget, ordering, unit norms, self-match)."""
import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu


def _props(kps, desc, glob, n_per_level, budget, sizes, thr):
    assert list(n_per_level) == list(budget) and len(kps) == sum(budget)
    assert np.allclose(np.linalg.norm(desc.astype(np.float64), axis=1), 1.0, atol=2e-6)
    assert abs(np.linalg.norm(glob.astype(np.float64)) - 1.0) < 2e-6
    off = 0
    for lvl, n in enumerate(budget):
        k = kps[off:off + n]
        off += n
        assert np.all(k["octave"] == lvl)
        assert np.all(k["response"] >= thr)
        assert np.all(np.diff(k["response"]) <= 0), "canonical order: response descending within a level"
        sf = np.float32(1.2) ** lvl
        w, h = sizes[lvl]
        assert np.all(k["x"] >= 0) and np.all(k["x"] <= (w // 8 * 8 - 1) * sf + 1e-3)
        assert np.all(k["y"] >= 0) and np.all(k["y"] <= (h // 8 * 8 - 1) * sf + 1e-3)
        assert len({(float(a), float(b)) for a, b in zip(k["x"], k["y"])}) == n, "duplicate keypoint"


@pytest.mark.parametrize("cfg", [(752, 480, 1000), (512, 512, 850), (512, 512, 1000)])     # EuRoC.yaml / TUM-VI.yaml sizes (+ config 3's second budget)
def test_full_size_frame_vs_oracle_and_properties(engine, oracle_model, cfg):
    from hfnet_slam_amd import capi, spec
    w, h, nf = cfg
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 4, max_batch=3)
    img = synth_image(h, w, 1000)
    n, kps, desc, g, npl = x.extract(img)
    budget = spec.features_per_level(nf, 4, 1.2)
    _props(kps, desc, g, npl, budget, spec.level_sizes(w, h, 4, 1.2), 0.01)
    rn, rk, rd, rg, rnpl = oracle_model.extract(img, nf, 0.01, 4, 1.2)
    assert n == rn and np.array_equal(npl, rnpl)
    assert np.array_equal(kps, rk), "keypoints must be bit-exact at full size"
    assert np.array_equal(desc, rd) and np.array_equal(g, rg)
    # determinism and batch invariance: alone == inside a batch, at any position, twice
    imgs = np.stack([synth_image(h, w, 1001, "natural"), img, synth_image(h, w, 1002)])
    for _ in range(2):
        nb, kb, db, gb = x.extract_batch(imgs)
        assert nb[1] == n and np.array_equal(kb[1, :n], kps) and np.array_equal(db[1, :n], desc) and np.array_equal(gb[1], g)
    # strided input (ROI of a larger buffer) gives the same result
    big = np.zeros((h + 6, w + 40), np.uint8)
    big[3:3 + h, 16:16 + w] = img
    n2, k2, d2, g2, _ = x.extract(big[3:3 + h, 16:16 + w])
    assert n2 == n and np.array_equal(k2, kps) and np.array_equal(d2, desc) and np.array_equal(g2, g)
    x.close()


def _bench_call_sizes():
    import bench                                     # the measured call size is read from bench.py so this test cannot go stale
    return sorted({64, bench.DEFAULT_CHUNK})


@pytest.mark.parametrize("B", _bench_call_sizes())
def test_bench_sized_call_vs_oracle(engine, oracle_model, B):
    """A full call of bench.py's size (`bench.DEFAULT_CHUNK` frames, and 64) in ONE extract_batch -- where the GEMM-shaped layers
    switch kernels (weight slabs through LDS for the 120 -> 720 expansions, skipped keypoint-slot tiles in the descriptor head), the
    tile lists of the fused blocks double and the XCD slot arithmetic runs at its largest -- against the oracle on the first two
    frames, the frames either side of the 64- and 128-frame boundaries and the last one; natural and uniform frames mixed."""
    from hfnet_slam_amd import capi
    w, h, nf = 752, 480, 1000
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 4, max_batch=B)
    imgs = np.stack([synth_image(h, w, 3000 + i, "natural" if i % 5 == 0 else "uniform") for i in range(B)])
    nb, kb, db, gb = x.extract_batch(imgs)
    for i in sorted({0, 1, 21, 63, 64, 127, 128, B - 1} & set(range(B))):
        rn, rk, rd, rg, _ = oracle_model.extract(imgs[i], nf, 0.01, 4, 1.2)
        assert nb[i] == rn, i
        assert np.array_equal(kb[i, :rn], rk) and np.array_equal(db[i, :rn], rd) and np.array_equal(gb[i], rg), i
    x.close()


@pytest.mark.parametrize("nf", [1000, 850])
def test_full_size_keyframe_step_vs_oracle(engine, oracle_model, nf):
    """BASELINE config 3 at its real size (512 x 512, the yaml's 850 features and 1000): 31 keyframes go through the extractor
    into the device-resident store (hfnet_store_put_extracted) and the database; the 32nd frame's keyframe step -- frame-to-frame
    SearchByBoW, hfnet_db_query against all previous keyframes, SearchForTriangulation against its 30 most recent neighbours in
    one batched call (LocalMapping.cc:516-520) -- is compared with the oracle, every pair."""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    W = H = 512
    n_kf = 31
    x = capi.Extractor(engine, W, H, nf, 0.01, 1.2, 4, max_batch=1)
    db = capi.Database(engine, n_kf + 1, engine.global_dim)
    store = capi.Store(engine, n_kf + 1, nf)
    ref = []
    for k in range(n_kf + 1):
        img = synth_image(H, W, 5000 + k, "natural" if k % 3 == 0 else "uniform")
        n, kps, desc, g, _ = x.extract(img)
        rn, rk, rd, rg, _ = oracle_model.extract(img, nf, 0.01, 4, 1.2)
        assert n == rn and np.array_equal(kps, rk) and np.array_equal(desc, rd) and np.array_equal(g, rg), k
        store.put_extracted(k, x, 0)
        assert store.rows(k) == rn
        if k < n_kf:
            db.add(k, g)
        ref.append((rd, rg))
    last = n_kf
    # frame-to-frame match on the store's copies (query = previous frame, as bench.py's config 3 calls it)
    cnt, match, dist = store.search_by_bow([(last - 1, last)], 0.6)
    rc, rm, rdist = O.search_by_bow(ref[last - 1][0], ref[last][0], 0.6)
    assert cnt[0] == rc and np.array_equal(match[0, :len(rm)], rm) and np.array_equal(dist[0, :len(rm)], rdist)
    # place recognition against all previous keyframes
    for mode in (0, 1):
        cs, sc, best, scores = db.query(ref[last][1], mode, want_scores=True)
        rs = O.db_scores(ref[last][1], np.stack([r[1] for r in ref[:n_kf]]))
        ridx, rbest = O.db_candidates(rs, mode)
        assert np.array_equal(scores[:n_kf], rs) and best == rbest and np.array_equal(cs, ridx) and np.array_equal(sc, rs[ridx])
    # 30 neighbours in one call, with the threshold screen on the bf16 pipe and without
    pairs = [(last, j) for j in range(last - 30, last)]
    want = [O.search_for_triangulation(ref[last][0], ref[j][0], 0.75) for _, j in pairs]
    for screen in (1, 0):
        engine.set_option("tri_screen_bf16", screen)
        try:
            for _ in range(2):                                    # (the second call sees the adaptive state the first one left)
                cnt, match = store.search_for_triangulation(pairs, 0.75)
                for p, (rc, rm) in enumerate(want):
                    assert cnt[p] == rc, (screen, p, cnt[p], rc)
                    assert np.array_equal(match[p, :len(rm)], rm), (screen, p)
        finally:
            engine.set_option("tri_screen_bf16", 1)
    store.close(); db.close(); x.close()


@pytest.mark.parametrize("size", [(752, 480), (640, 360)])
def test_mid_sized_call_every_frame_vs_oracle(engine, oracle_model, size):
    """9 frames per call: enough 128-row tiles for the heads to take four column tiles per wave (the LDS-weight kernels), few
    enough to compare EVERY frame with the oracle; the second size has level widths that are not multiples of 8 (ragged
    tiles in every kernel)."""
    from hfnet_slam_amd import capi
    w, h = size
    nf, B = 1000, 9
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 4, max_batch=B)
    imgs = np.stack([synth_image(h, w, 4000 + i, "natural" if i % 2 else "uniform") for i in range(B)])
    nb, kb, db, gb = x.extract_batch(imgs)
    for i in range(B):
        rn, rk, rd, rg, _ = oracle_model.extract(imgs[i], nf, 0.01, 4, 1.2)
        assert nb[i] == rn, i
        assert np.array_equal(kb[i, :rn], rk) and np.array_equal(db[i, :rn], rd) and np.array_equal(gb[i], rg), i
    x.close()


@pytest.mark.parametrize("cfg", [(752, 480, 1000), (512, 512, 850)])
def test_sparse_score_regime_full_size(sparse_pair, cfg):
    """The regime real weights live in and seeded random weights never reach (their scores sit near 1/65 > the threshold 0.01, so every NMS
    survivor is a candidate and top-K is saturated at every level): weights whose detector lets FEW cells through -- pyramid levels that fall
    short of their budget, candidates clustered where the logits peak (tap cells shared by neighbouring keypoints), levels with no candidate at
    all.  Every frame of a call against the oracle, bit for bit; the call mixes frames with full and short levels."""
    from hfnet_slam_amd import capi, spec
    engine, om, bias = sparse_pair
    w, h, nf = cfg
    B = 6
    imgs = np.stack([synth_image(h, w, 6400 + i, "natural" if i % 2 else "uniform") for i in range(B)])
    budget = spec.features_per_level(nf, 4, 1.2)
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 4, max_batch=B)
    nb, kb, db, gb = x.extract_batch(imgs)
    short = empty = 0
    for f in range(B):
        rn, rk, rd, rg, rnpl = om.extract(imgs[f], nf, 0.01, 4, 1.2)
        assert nb[f] == rn, (bias, f, nb[f], rn)
        assert np.array_equal(kb[f, :rn], rk) and np.array_equal(db[f, :rn], rd) and np.array_equal(gb[f], rg), (bias, f)
        short += sum(int(a) < b for a, b in zip(rnpl, budget)); empty += sum(int(a) == 0 for a in rnpl)
    assert short >= B, "the fixture must produce short levels"
    if bias >= 18:
        assert empty > 0, "bias 18 must leave levels without a candidate"
    # a single frame through the latency path (graph, two branches) sees the same short levels
    n1, k1, d1, g1, npl1 = x.extract(imgs[1])
    rn, rk, rd, rg, rnpl = om.extract(imgs[1], nf, 0.01, 4, 1.2)
    assert n1 == rn and np.array_equal(npl1, rnpl) and np.array_equal(k1, rk) and np.array_equal(d1, rd) and np.array_equal(g1, rg)
    assert x.device_faults() == 0
    x.close()


BF16X3_DESC_TOL = 1e-5        # abs, on unit-norm 256-D rows (include/hfnet_hip.h; observed <= 2e-6)
BF16X3_GLOBAL_TOL = 2e-5      # abs, on the unit-norm global descriptor (observed <= 6e-6: ten layers of split-bf16 1x1 convolutions deep)


@pytest.mark.parametrize("cfg", [(752, 480, 1000), (512, 512, 850)])
def test_split_bf16_options_keep_indices_exact_and_floats_within_tolerance(engine, oracle_model, cfg):
    """engine options desc_bf16x3 / global_bf16x3 (default off): the stages that decide no index -- the descriptor head at the tap
    cells, the 1x1 convolutions of layers 9-18 -- on split-bf16 operands (two pieces, three products on the bf16 matrix pipe).
    north_star's contract: keypoint indices bit-exact, descriptor tensors within a stated tolerance.  Against the ORACLE: counts,
    keypoints (x, y, response, octave) array_equal; descriptors and global descriptor within the stated tolerance; unit norms.  And the
    matcher's answers on the produced descriptors agree with its answers on the exact path's descriptors (frames that really match:
    shifted copies of one scene)."""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    w, h, nf = cfg
    B = 6
    base = synth_image(h, w + 64, 6100, "natural")
    imgs = np.stack([np.ascontiguousarray(base[:, 8 * f:8 * f + w]) for f in range(B - 1)] + [synth_image(h, w, 6101)])
    res = {}
    saved = {o: engine.get_option(o) for o in ("desc_bf16x3", "global_bf16x3")}
    try:
        for mode in (0, 1):
            engine.set_option("desc_bf16x3", mode); engine.set_option("global_bf16x3", mode)
            x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 4, max_batch=B)
            res[mode] = x.extract_batch(imgs)
            x.close()
    finally:
        for o, v in saved.items():
            engine.set_option(o, v)
    n0, k0, d0, g0 = res[0]
    n1, k1, d1, g1 = res[1]
    assert np.array_equal(n0, n1) and np.array_equal(k0, k1), "the options must not move a keypoint"
    worst_d = worst_g = 0.0
    for f in range(B):
        rn, rk, rd, rg, _ = oracle_model.extract(imgs[f], nf, 0.01, 4, 1.2)
        assert n1[f] == rn and np.array_equal(k1[f, :rn], rk), f
        assert np.array_equal(d0[f, :rn], rd) and np.array_equal(g0[f], rg), f         # (options off: the oracle's bits)
        worst_d = max(worst_d, float(np.abs(d1[f, :rn].astype(np.float64) - rd).max()))
        worst_g = max(worst_g, float(np.abs(g1[f].astype(np.float64) - rg).max()))
        assert np.allclose(np.linalg.norm(d1[f, :rn].astype(np.float64), axis=1), 1.0, atol=2e-6)
        assert abs(np.linalg.norm(g1[f].astype(np.float64)) - 1.0) < 2e-6
    assert 0 < worst_d <= BF16X3_DESC_TOL, worst_d          # (> 0: the option really took the other pipe)
    assert 0 < worst_g <= BF16X3_GLOBAL_TOL, worst_g
    agree = total = matched = 0
    for f in range(1, B):
        c0, m0, _ = engine.search_by_bow(d0[f - 1, :n0[f - 1]], d0[f, :n0[f]], 0.6)
        c1, m1, _ = engine.search_by_bow(d1[f - 1, :n0[f - 1]], d1[f, :n0[f]], 0.6)
        agree += int(np.sum(m0 == m1)); total += len(m0); matched += int(c0)
    assert matched > 0.1 * total, "the shifted frames must really match"
    assert agree >= total - 2, (agree, total)               # (a distance within 1e-5 of TH_LOW or of a runner-up may flip: none observed)


def test_monocular_initialisation_extractor_5x_features(engine, oracle_model):
    """Tracking.cc:693 builds the initialisation extractor with 5 * nFeatures on the same models; with the 8-level pyramid
    of the monocular yaml files the small levels run out of candidates before their budget is met."""
    from hfnet_slam_amd import capi, spec
    w, h, nf, nl = 752, 480, 5000, 8
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, nl, max_batch=1)
    img = synth_image(h, w, 1003, "natural")
    n, kps, desc, g, npl = x.extract(img)
    rn, rk, rd, rg, rnpl = oracle_model.extract(img, nf, 0.01, nl, 1.2)
    budget = spec.features_per_level(nf, nl, 1.2)
    assert n == rn and np.array_equal(npl, rnpl) and all(a <= b for a, b in zip(npl, budget)) and n > 1000
    assert np.array_equal(kps, rk) and np.array_equal(desc, rd) and np.array_equal(g, rg)
    x.close()


def test_full_size_match_1000x1000(engine):
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    a = rng.standard_normal((1000, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
    perm = np.random.default_rng(12).permutation(1000)
    b = a[perm] + 0.02 * rng.standard_normal((1000, 256)).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
    a = a.astype(np.float32); b = b.astype(np.float32)
    n, m, d = engine.search_by_bow(a, b, 0.6)
    assert n == 1000 and np.array_equal(m, np.argsort(perm))                     # planted permutation recovered
    rn, rm, rd = O.search_by_bow(a, b, 0.6)
    assert n == rn and np.array_equal(m, rm) and np.array_equal(d, rd)
    n, m = engine.search_for_triangulation(a, b, 0.75)
    rn, rm = O.search_for_triangulation(a, b, 0.75)
    assert n == rn == 1000 and np.array_equal(m, rm)
    # self match: every row matches itself at distance 0; symmetric under swapping the roles
    n, m, d = engine.search_by_bow(a, a, 0.6)
    assert n == 1000 and np.array_equal(m, np.arange(1000)) and np.all(d == 0)
    n_ab, m_ab, _ = engine.search_by_bow(a, b, 0.6)
    n_ba, m_ba, _ = engine.search_by_bow(b, a, 0.6)
    assert n_ab == n_ba and np.array_equal(m_ba[m_ab], np.arange(1000))


def test_loop_closure_stress_10k_database(engine):
    """BASELINE config 5: 10 000 x 4096 unit rows resident in HBM, planted neighbours"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    rng = np.random.default_rng(13)
    n, dim = 10000, 4096
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    rows = rows.astype(np.float32)
    db = capi.Database(engine, n, dim)
    for i in range(n):
        db.add(i, rows[i])
    for planted in (0, 4321, 9999):
        q = rows[planted] + 0.003 * rng.standard_normal(dim).astype(np.float32)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        cs, sc, best, scores = db.query(q, 0, want_scores=True)
        assert cs.tolist() == [planted] and best == scores[planted] and 0.7 < best < 0.9
        ref = O.db_scores(q, rows)
        assert np.array_equal(scores, ref)
        ridx, rbest = O.db_candidates(ref, 0)
        assert np.array_equal(cs, ridx) and best == rbest
        cs1, _, _, _ = db.query(q, 1)
        assert cs1.tolist() == [planted]
    # config 5, Q = 67 (not a multiple of any tile + an erased slot)
    db.erase(17)
    planted = rng.integers(0, n, 67)
    qs = rows[planted] + 0.003 * rng.standard_normal((67, dim)).astype(np.float32)
    qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
    # (a) the exact batched scan (engine option db_gemm_min_queries above Q): batched == single-query == oracle, bit for bit
    engine.set_option("db_gemm_min_queries", 1 << 20); engine.set_option("db_screen_min_rows", 0)
    for mode in (0, 1):
        cands, best, scores = db.query_batch(qs, mode, want_scores=True)
        for i in (0, 7, 8, 33, 66):
            cs, sc, b1, s1 = db.query(qs[i], mode, want_scores=True)
            assert np.array_equal(scores[i], s1) and best[i] == b1
            assert np.array_equal(cands[i][0], cs) and np.array_equal(cands[i][1], sc)
        ref = O.db_scores(qs[5], rows)
        ref[17] = -1.0
        assert np.array_equal(scores[5], ref)
    # (b) the default for >= 8 queries: a crude product on the integer matrix pipe screens every slot (8-bit steps of every vector at its own
    #     scale, exact int32 sums, a rigorous bound of the quantisation error: a slot it rules out is at distance >= 1, score exactly 0), every
    #     other occupied slot is scored with the exact chain.
    #     Contract: EVERY output equals the exact scan's (== the oracle's) bit for bit -- the scores of all slots (-1 for the erased one),
    #     best, candidate set, candidate scores.
    engine.set_option("db_gemm_min_queries", 8); engine.set_option("db_screen_min_rows", 6144)
    exact_all = np.stack([O.db_scores(qs[i], rows) for i in range(67)])
    exact_all[:, 17] = -1.0
    for mode in (0, 1):
        cands, best, scores = db.query_batch(qs, mode, want_scores=True)
        assert np.array_equal(scores, exact_all)
        for i in range(67):
            eidx, ebest = O.db_candidates(exact_all[i], mode)
            assert best[i] == ebest, (i, best[i], ebest)
            assert np.array_equal(cands[i][0], eidx) and np.array_equal(cands[i][1], exact_all[i][eidx])
    # the revisit case (queries that ARE database rows, near-duplicates a few ulp apart), rows planted right at the 0.8 * best candidate
    # threshold, rows planted around distance 1 and around the end of the screen's band -- where it decides between "exactly 0" and the exact
    # chain: 0.99 .. 1.06, the band ends near d^2 = 1.055 (distance 1.027) for these unit vectors --, scaled (non-unit) rows, and a row of zeros
    rows2 = rows[:2048].copy()
    rows2[100] = rows2[7]; rows2[101] = np.nextafter(rows2[7], np.float32(1)); rows2[102] = rows2[7] * np.float32(1.0 + 2e-7)
    base = rows2[300]
    plant = [(400 + k, eps) for k, eps in enumerate((0.1995, 0.19999, 0.2, 0.20001, 0.2005))]          # distance ~eps from row 300: score ~ 1 - eps
    plant += [(420 + k, eps) for k, eps in enumerate((0.99, 0.999, 0.9999, 1.0, 1.0001, 1.001, 1.005, 1.0085, 1.0095, 1.02, 1.024, 1.027, 1.03, 1.04, 1.06))]
    for slot, eps in plant:
        v = rng.standard_normal(dim).astype(np.float32); v -= v.dot(base) * base; v /= np.linalg.norm(v)
        c = 1.0 - eps * eps / 2.0                                                       # unit vector at distance eps: cos = 1 - eps^2 / 2
        rows2[slot] = (base * np.float32(c) + v * np.float32(np.sqrt(max(1 - c * c, 0.0)))).astype(np.float32)
    rows2[500] = rows2[300] * np.float32(1.7); rows2[501] = rows2[300] * np.float32(0.45); rows2[502] = 0.0
    db2 = capi.Database(engine, 2048, dim)
    for i in range(2048):
        db2.add(i, rows2[i])
    qs2 = np.stack([rows2[7], rows2[300], rows2[101], rows2[1500], rows2[300] * np.float32(1.3), rows2[502]] + [rows2[i] for i in range(600, 610)]).astype(np.float32)
    ex2 = np.stack([O.db_scores(q, rows2) for q in qs2])
    for mode in (0, 1):
        cands, best, scores = db2.query_batch(qs2, mode, want_scores=True)
        assert np.array_equal(scores, ex2), np.argwhere(scores != ex2)[:8]
        for i in range(len(qs2)):
            eidx, ebest = O.db_candidates(ex2[i], mode)
            assert best[i] == ebest, (i, best[i], ebest)
            assert np.array_equal(cands[i][0], eidx) and np.array_equal(cands[i][1], ex2[i][eidx]), (mode, i)
        assert set(cands[0][0].tolist()) >= {7, 100, 101, 102} and 300 in cands[1][0].tolist()
    assert 0 < ex2[1][421] < 2e-3 and ex2[1][425] == 0.0                               # (the plants straddle distance 1)
    db2.close()
    db.close()
