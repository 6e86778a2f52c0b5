"""HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs -- `-m gpu`.

Bar: keypoints / indices bit-exact; float tensors bit-exact where the HIP kernels follow the
oracle's accumulation order (everything here), reported with np.array_equal so that any drift
shows up as a failure rather than hiding inside a tolerance."""
import numpy as np
import pytest

from conftest import synth_image, torch_to_device, torch_to_host

pytestmark = pytest.mark.gpu


def _eq(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, f"{name}: shape {a.shape} vs {b.shape}"
    if not np.array_equal(a, b):
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        raise AssertionError(f"{name}: {np.count_nonzero(a != b)} of {a.size} elements differ, max |d| = {d.max():.3e} "
                             f"at {np.unravel_index(d.argmax(), d.shape)} (ref scale {np.abs(b).max():.3e})")


SIZES = [(64, 96), (120, 160), (133, 171)]   # the last one exercises the crop to multiples of 8 and odd strides


# engine options (hfnet_engine_set_option) of the kernel variants: the default (wave-autonomous fused blocks, sparse
# descriptor head), the reference variant (every block as three launches, dense descriptor head), and the barrier-phased
# fused kernel for every block shape it covers
# (fuse_min_wgs 0: layers 8-14 take their fused kernels even for the few tiles of a single small frame)
VARIANTS = {"default": {}, "unfused_dense": {"fuse_blocks": 0, "dense_desc": 1}, "fused_v2": {"fused_variant": 2},
            "no_tail_fuse": {"tail_fuse": 0}, "no_dedupe": {"dedupe_taps": 0}, "dedupe_two_launch": {"dedupe_taps": 2}, "fused_all": {"fuse_min_wgs": 0}, "fused_occ3_all": {"fused_variant": 3, "fuse_min_wgs": 0}, "fused_v2_all": {"fused_variant": 2, "fuse_min_wgs": 0},
            "separate_det_tail": {"det_fuse": 0}, "fused_v6": {"fused_variant": 6, "fuse_min_wgs": 0}, "fused_v6_all": {"fused_variant": 7, "fuse_min_wgs": 0},
            "fused_v4_all": {"fused_variant": 5, "fuse_min_wgs": 0}, "fused_v8": {"fused_variant": 8, "fuse_min_wgs": 0}}


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("hw", SIZES)
def test_layer_taps_bit_exact(engine, oracle_model, hw, variant, engine_options):
    """every kernel variant (fused / unfused blocks, sparse / dense descriptor head) must give the same bits"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    engine_options(VARIANTS[variant])
    h, w = hw
    img = synth_image(h, w, 1000 + h)
    m = capi.Model(engine, capi.MODE_LOCAL_AND_GLOBAL, h, w, 500)
    st, kps, desc, glob = m.detect(img, 300, 0.01)
    assert st == capi.OK, capi.last_error()
    taps = [O.TAP_STEM] + [O.TAP_BLOCK0 + i for i in range(17)] + [O.TAP_DESC_HIDDEN, O.TAP_DESC_RAW, O.TAP_DET_HIDDEN,
                                                                    O.TAP_LOGITS, O.TAP_SCORES_DENSE, O.TAP_MEMBERSHIPS, O.TAP_VLAD]
    ref = oracle_model.run_local(img, want_global=True, taps=taps)
    for t in taps:
        _eq(f"tap {t}", m.tap(t, ref["taps"][t].shape), ref["taps"][t])
    _eq("scores_nms", m.tap(25, ref["scores_nms"].shape), ref["scores_nms"])
    _eq("desc_map", m.tap(26, ref["desc_map"].shape), ref["desc_map"])
    _eq("global", glob, ref["global"])
    ok, rk, rd, rg = oracle_model.detect(img, O.MODE_LOCAL_AND_GLOBAL, 300, 0.01)
    assert ok
    assert len(kps) == len(rk)
    for f in ("x", "y", "response", "octave"):
        _eq(f"kps.{f}", kps[f], rk[f])
    _eq("local descriptors", desc, rd)
    _eq("global (detect)", glob, rg)
    m.close()


def test_modes_and_overloads(engine, oracle_model):
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    h, w = 96, 128
    img = synth_image(h, w, 5)
    # kImageToLocal: 5-argument overload only
    m = capi.Model(engine, capi.MODE_LOCAL, h, w, 200)
    assert m.is_valid()
    st, kps, desc, _ = m.detect(img, 150, 0.01)
    assert st == capi.OK
    ok, rk, rd, _ = oracle_model.detect(img, O.MODE_LOCAL, 150, 0.01)
    _eq("kps", kps, rk); _eq("desc", desc, rd)
    st, *_ = m.detect(img, 150, 0.01, with_aux=True)       # 6-argument overload on a kImageToLocal model -> false
    assert st == capi.ERR_WRONG_MODE
    st, _ = m.detect_global(np.zeros((h // 8, w // 8, engine.c_local), np.float32))
    assert st == capi.ERR_WRONG_MODE
    m.close()
    # kImageToLocalAndIntermediate + kIntermediateToGlobal == the TF wiring (BaseModel.cc:42-44,73-77)
    mi = capi.Model(engine, capi.MODE_LOCAL_AND_INTERMEDIATE, h, w, 200)
    st, kps, desc, inter = mi.detect(img, 150, 0.01)
    assert st == capi.OK
    ok, rk, rd, rinter = oracle_model.detect(img, O.MODE_LOCAL_AND_INTERMEDIATE, 150, 0.01)
    _eq("kps", kps, rk); _eq("desc", desc, rd); _eq("intermediate", inter, rinter)
    st, *_ = mi.detect(img, 150, 0.01, with_aux=False)
    assert st == capi.ERR_WRONG_MODE
    mg = capi.Model(engine, capi.MODE_INTERMEDIATE_TO_GLOBAL, h // 8, w // 8, 1)
    st, g = mg.detect_global(inter)
    assert st == capi.OK
    ok, rg = oracle_model.detect_global(rinter)
    _eq("global from intermediate", g, rg)
    st, *_ = mg.detect(img, 10, 0.01)
    assert st == capi.ERR_WRONG_MODE
    # and it equals the one-shot LocalAndGlobal descriptor
    ml = capi.Model(engine, capi.MODE_LOCAL_AND_GLOBAL, h, w, 200)
    st, _, _, g2 = ml.detect(img, 150, 0.01)
    _eq("global split == fused", g, g2)
    for x in (mi, mg, ml):
        x.close()


def test_detect_edge_cases(engine, oracle_model):
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    h, w = 72, 104
    m = capi.Model(engine, capi.MODE_LOCAL, h, w, 400)
    # fewer candidates than requested (scan order is kept), zero requested, huge threshold, strided ROI input
    img = synth_image(h, w, 77)
    for nk, thr in ((400, 0.01), (0, 0.01), (50, 0.999999), (400, 0.2), (1, 0.0)):
        st, kps, desc, _ = m.detect(img, nk, thr)
        assert st == capi.OK, capi.last_error()
        ok, rk, rd, _ = oracle_model.detect(img, O.MODE_LOCAL, nk, thr)
        assert len(kps) == len(rk), (nk, thr, len(kps), len(rk))
        _eq(f"kps n={nk} thr={thr}", kps, rk); _eq("desc", desc, rd)
    big = synth_image(h + 10, w + 24, 78)
    roi = big[5:5 + h, 8:8 + w]
    st, kps, desc, _ = m.detect(roi, 100, 0.01)
    ok, rk, rd, _ = oracle_model.detect(np.ascontiguousarray(roi), O.MODE_LOCAL, 100, 0.01)
    _eq("roi kps", kps, rk); _eq("roi desc", desc, rd)
    # constant image: every score equal -> nothing is suppressed, all H'*W' pixels are candidates (tie-break only)
    const = np.full((h, w), 131, np.uint8)
    st, kps, desc, _ = m.detect(const, 300, 0.01)
    ok, rk, rd, _ = oracle_model.detect(const, O.MODE_LOCAL, 300, 0.01)
    _eq("const kps", kps, rk); _eq("const desc", desc, rd)
    # capacity / shape errors
    st, *_ = m.detect(img, 401, 0.01)
    assert st == capi.ERR_CAPACITY
    m.close()


def test_response_ties(weights_ties_path):
    """saturated detector: many responses are exactly 1.0, selection is decided by the index tie-break"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    e = capi.Engine(weights_ties_path, 0)
    om = O.Model(weights_ties_path)
    h, w = 160, 200
    img = synth_image(h, w, 2024)
    m = capi.Model(e, capi.MODE_LOCAL, h, w, 300)
    st, kps, desc, _ = m.detect(img, 120, 0.01)
    ok, rk, rd, _ = om.detect(img, O.MODE_LOCAL, 120, 0.01)
    assert np.sum(rk["response"] == rk["response"][0]) > 1, "fixture no longer produces ties"
    _eq("kps", kps, rk); _eq("desc", desc, rd)
    m.close(); e.close()


# the single-frame path's own pieces on and off: pyramid chain in one launch, distinct tap cells, one launch per block for layers
# 8-18, and the caller-owned result buffers of the wrapper
EXTRACTOR_VARIANTS = {"default": {}, "unfused_stem": {"fuse_stem": 0}, "separate_launches": {"pyramid_fuse": 0, "dedupe_taps": 0, "tail_fuse": 0},
                      "resize_gather": {"pyramid_fuse": 0, "resize_band": 0},
                      "no_graph": {"graph": 0, "pinned_frames": 0}, "plain_launches": {"graph": 0},
                      "branch_after_heads": {"interleave": 0, "host_global": 0}}


@pytest.mark.parametrize("variant", list(EXTRACTOR_VARIANTS))
@pytest.mark.parametrize("cfg", [(160, 120, 300, 3), (200, 152, 500, 4), (96, 96, 64, 1)])
def test_extractor_matches_oracle(engine, oracle_model, cfg, variant, engine_options):
    from hfnet_slam_amd import capi
    engine_options(EXTRACTOR_VARIANTS[variant])
    w, h, nf, nl = cfg
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, nl, max_batch=2)
    sf, fpl, lw, lh = x.tables()
    from oracle import oracle as O
    rsf, rfpl, rlw, rlh = O.extractor_tables(nf, nl, 1.2, w, h)
    _eq("scale factors", sf, rsf); _eq("budget", fpl, rfpl); _eq("level w", lw, rlw); _eq("level h", lh, rlh)
    imgs = np.stack([synth_image(h, w, 300 + i, "natural" if i else "uniform") for i in range(3)])
    n, kps, desc, g, npl = x.extract(imgs[0])
    rn, rk, rd, rg, rnpl = oracle_model.extract(imgs[0], nf, 0.01, nl, 1.2)
    assert n == rn
    _eq("n per level", npl, rnpl); _eq("kps", kps, rk); _eq("desc", desc, rd); _eq("global", g, rg)
    # the same call into caller-owned buffers, twice (the second call's results must not depend on what the first left behind)
    bufs = x.output_buffers()
    for img in (imgs[1], imgs[0]):
        n2, k2, d2, g2, npl2 = x.extract(img, bufs)
    assert n2 == rn and d2.base is bufs[1]
    if EXTRACTOR_VARIANTS[variant].get("pinned_frames", 1):
        t = x.last_timing()                       # host stamps of the latency path: monotone, all set
        assert (t >= 0).all() and (np.diff(t) >= 0).all(), t
    _eq("n per level (own buffers)", npl2, rnpl); _eq("kps (own buffers)", k2, rk); _eq("desc (own buffers)", d2, rd); _eq("global (own buffers)", g2, rg)
    # batched (3 frames through a max_batch=2 extractor -> two chunks)
    nb, kb, db, gb = x.extract_batch(imgs)
    for i in range(3):
        rn, rk, rd, rg, _ = oracle_model.extract(imgs[i], nf, 0.01, nl, 1.2)
        assert nb[i] == rn
        _eq(f"batch kps {i}", kb[i, :rn], rk); _eq(f"batch desc {i}", db[i, :rn], rd); _eq(f"batch global {i}", gb[i], rg)
    x.close()


@pytest.mark.parametrize("fc_tile,B", [(0, 21), (2, 21), (4, 21), (4, 70), (2, 130)])
def test_many_frames_per_call_global_descriptor(engine, oracle_model, engine_options, fc_tile, B):
    """calls of more than 16 frames take the FC form that walks the 16 input ranges as one stream (k_fc_mfma_seq, fc_tile = 0), or the
    blocked form (k_fc_mfma_tile: weights shared through LDS, the range partials merged as a binary counter in registers; fc_tile = 2 / 4
    force it with 32 / 64 columns per workgroup at sizes the default dispatch gives to the other kernel) -- with a partial last 16-frame
    row tile, a second / third 64-frame row group and waves without frames here: every frame's global descriptor (and keypoints) must
    equal the oracle's"""
    from hfnet_slam_amd import capi
    engine_options({"fc_tile": fc_tile})
    w, h, nf, nl = 96, 96, 64, 2
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, nl, max_batch=B)
    imgs = np.stack([synth_image(h, w, 4100 + i, "natural" if i % 3 else "uniform") for i in range(B)])
    nb, kb, db, gb = x.extract_batch(imgs)
    for i in sorted(set(range(B)) if B <= 21 else {0, 1, 15, 16, 63, 64, 65, B - 2, B - 1}):
        rn, rk, rd, rg, _ = oracle_model.extract(imgs[i], nf, 0.01, nl, 1.2)
        assert nb[i] == rn
        _eq(f"kps {i}", kb[i, :rn], rk); _eq(f"desc {i}", db[i, :rn], rd); _eq(f"global {i}", gb[i], rg)
    x.close()


def _unit_rows(rng, n, d=256):
    a = rng.standard_normal((n, d)).astype(np.float32)
    return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("n1,n2,sigma", [(300, 280, 0.02), (257, 129, 0.045), (33, 1000, 0.02), (1, 1, 0.0), (40, 0, 0.0), (1100, 700, 0.03)])
def test_matchers(engine, n1, n2, sigma):
    """planted permutation: B = normalise(A[pi] + sigma * g)"""
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    a = _unit_rows(rng, n1)
    if n2:
        perm = np.random.default_rng(12).permutation(max(n1, n2))[:n2] % n1
        b = a[perm] + sigma * rng.standard_normal((n2, 256)).astype(np.float32)
        b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    else:
        b = np.zeros((0, 256), np.float32)
    n, m, d = engine.search_by_bow(a, b, 0.6)
    rn, rm, rd = O.search_by_bow(a, b, 0.6)
    assert n == rn
    _eq("bow match", m, rm); _eq("bow dist", d, rd)
    n, m = engine.search_for_triangulation(a, b, 0.75)
    rn, rm = O.search_for_triangulation(a, b, 0.75)
    assert n == rn
    _eq("triangulation match", m, rm)


def test_matcher_duplicates_and_ties(engine):
    """exact duplicates -> equal distances: first-minimum rules decide"""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    a = _unit_rows(rng, 64)
    a[10] = a[3]; a[40] = a[3]
    b = np.concatenate([a[:32], a[3:4], a[3:4]]).copy()
    n, m, d = engine.search_by_bow(a, b, 0.6)
    rn, rm, rd = O.search_by_bow(a, b, 0.6)
    assert n == rn
    _eq("bow match", m, rm); _eq("bow dist", d, rd)
    n, m = engine.search_for_triangulation(a, b, 0.75)
    rn, rm = O.search_for_triangulation(a, b, 0.75)
    assert n == rn
    _eq("tri match", m, rm)
    x, y = a[0], a[1]
    assert engine.descriptor_distance(x, y) == O.descriptor_distance(x, y)


@pytest.fixture(params=[1, 0], ids=["screen_bf16", "screen_f32"])
def screen(request, engine):
    """engine option match_screen_bf16 for one test (the engine fixture lives for the session)"""
    engine.set_option("match_screen_bf16", request.param)
    yield request.param
    engine.set_option("match_screen_bf16", 1)


@pytest.mark.parametrize("seed,spread,scale", [(21, 3e-4, 1.0), (22, 3e-5, 1.0), (23, 2e-6, 1.0), (24, 1e-4, 7.5), (25, 1e-4, 0.05),
                                               (26, 1e-3, 1.0), (27, 3e-3, 1.0), (28, 1e-3, 30.0)])
def test_matcher_near_ties_inside_the_rounding_band(engine, seed, spread, scale, screen):
    """Adversarial for the pre-selection of SearchByBoW: clusters of descriptors whose mutual distances differ by less than
    the rounding band of the screening form of the distance (kernels_match.hip: band = 1.25e-6 * dim relative to |q|^2 + |t|^2
    for the split-bf16 products on the bf16 matrix pipe, 5e-7 * dim for the f32 MFMA form -- engine option match_screen_bf16),
    from differences of a few ulp up to several bands, with non-unit norms as well -- the exact re-evaluation must still pick
    the oracle's match and distance bit for bit, whichever pipe screened."""
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    centres = _unit_rows(rng, 12)
    q = np.repeat(centres, 40, axis=0) + spread * rng.standard_normal((480, 256)).astype(np.float32)
    t = np.repeat(centres, 35, axis=0) + spread * rng.standard_normal((420, 256)).astype(np.float32)
    q = (scale * q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    t = (scale * t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float32)
    t[7] = q[3]; t[100] = q[3]; q[200] = q[3]                       # exact duplicates on both sides
    th = 0.6 * scale
    n, m, d = engine.search_by_bow(q, t, th)
    rn, rm, rd = O.search_by_bow(q, t, th)
    assert n == rn
    _eq("near-tie bow match", m, rm); _eq("near-tie bow dist", d, rd)
    cnt, mb, db = engine.search_by_bow_batch(np.stack([q, np.pad(t, ((0, 60), (0, 0)))]), np.array([480, 420], np.int32),
                                            [(0, 1), (1, 0)], th)
    _eq("batched near-tie match", mb[0][:480], rm); _eq("batched near-tie dist", db[0][:480], rd)
    rn2, rm2, rd2 = O.search_by_bow(t, q, th)
    assert cnt[1] == rn2
    _eq("batched reverse match", mb[1][:420], rm2); _eq("batched reverse dist", db[1][:420], rd2)


@pytest.mark.parametrize("screen_min_rows", [0, 1], ids=["scan", "screened"])
def test_database(engine, screen_min_rows):
    """hfnet_db_query (KeyFrameDatabase.cc:86-104, 178-197) vs the oracle: as the exact f32 scan, and in the screened form large databases take
    (engine option db_screen_min_rows: the 8-bit copy + the exact chain, hfnet_db_query_batch's kernel with one query) -- the same bits"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    engine.set_option("db_screen_min_rows", screen_min_rows)
    rng = np.random.default_rng(13)
    cap, dim, n = 700, 4096, 520
    rows = _unit_rows(rng, n, dim)
    db = capi.Database(engine, cap, dim)
    slots = np.random.default_rng(14).permutation(cap)[:n]
    for s, r in zip(slots, rows):
        db.add(int(s), r)
    dense = np.zeros((cap, dim), np.float32); dense[slots] = rows
    occ = np.zeros(cap, bool); occ[slots] = True
    for trial, sig in enumerate((0.002, 0.01, 0.05)):
        q = rows[7 * trial + 1] + sig * rng.standard_normal(dim).astype(np.float32)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        for mode in (0, 1):
            cs, sc, best, scores = db.query(q, mode, want_scores=True)
            ref = O.db_scores(q, dense)
            _eq("scores", scores[occ], ref[occ])
            assert np.all(scores[~occ] == -1.0)
            ridx, rbest = O.db_candidates(np.where(occ, ref, -1.0).astype(np.float32), mode)
            assert best == rbest
            _eq("candidates", cs, ridx); _eq("candidate scores", sc, ref[ridx])
    db.erase(int(slots[1]))
    cs, sc, best, scores = db.query(rows[1], 0, want_scores=True)
    assert scores[slots[1]] == -1.0 and slots[1] not in cs
    db.clear()
    cs, sc, best, _ = db.query(rows[1], 0)
    assert len(cs) == 0 and best == 0.0
    db.close()
    engine.set_option("db_screen_min_rows", 6144)


@pytest.mark.parametrize("cap,dim,n_q", [(700, 4096, 9), (33, 256, 8), (257, 512, 31), (1000, 768, 33), (95, 1024, 64), (3001, 4096, 97),
                                          (1500, 2048, 129), (64, 4096, 200), (31, 256, 40), (700, 4096, 3), (90, 512, 1)])
def test_database_batched_screen_geometries(engine, cap, dim, n_q):
    """hfnet_db_query_batch, >= 8 queries: the screen on the integer matrix pipe (k_db_sweep: the database's 8-bit copy in fragment order, whole
    32-row tiles; one or two query tiles of 32 in registers, every wave a quarter of k, a launch per 64 queries; descriptor lengths other than
    4096 take the guarded form) + the exact chain in the same kernel -- every output equals the exact batched scan's bits
    (KeyFrameDatabase.cc:86-104), for capacities / descriptor lengths / query counts that fill no tile, with empty slots, near-duplicates and
    rows around distance 1 from a query."""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    rng = np.random.default_rng(cap * 7 + n_q)
    n = max(1, cap - cap // 5)
    rows = _unit_rows(rng, n, dim)
    slots = np.random.default_rng(cap).permutation(cap)[:n]
    base = rows[0]
    for k, eps in enumerate((0.2, 0.99, 0.9999, 1.0, 1.0001, 1.005, 1.0085, 1.0095, 1.02)):
        if 1 + k >= n: break
        v = rng.standard_normal(dim).astype(np.float32); v -= v.dot(base) * base; v /= np.linalg.norm(v)
        c = 1.0 - eps * eps / 2.0
        rows[1 + k] = (base * np.float32(c) + v * np.float32(np.sqrt(max(1 - c * c, 0.0)))).astype(np.float32)
    if n > 12: rows[11] = rows[0]; rows[12] = rows[0] * np.float32(1.5)
    db = capi.Database(engine, cap, dim)
    for s, r in zip(slots, rows):
        db.add(int(s), r)
    qs = rows[rng.integers(0, n, n_q)] + 0.01 * rng.standard_normal((n_q, dim)).astype(np.float32)
    qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
    qs[0] = rows[0]
    if n_q > 2: qs[2] = 0.0
    try:
        for mode in (0, 1):
            engine.set_option("db_gemm_min_queries", 1 << 20); engine.set_option("db_screen_min_rows", 0)
            ce, be, se = db.query_batch(qs, mode, want_scores=True)
            engine.set_option("db_gemm_min_queries", 8); engine.set_option("db_screen_min_rows", 1)      # (the size rule on: a burst of fewer than 8 is screened too)
            engine.set_option("match_stats", 1); engine.get_option("stat_db_exact")                 # (reads and clears)
            cs, bs, ss = db.query_batch(qs, mode, want_scores=True)
            evals = engine.get_option("stat_db_exact")
            # what the screen let through: every pair with a positive score, and only a small multiple of them (unrelated unit vectors are at
            # d^2 ~ 2; the plants around distance 1 and the duplicates are the rest, and the all-zero query is at distance exactly 1 from every
            # unit row: n pairs) -- a screen that rules out nothing would score n x n_q pairs
            assert (se > 0).sum() <= evals <= (se > 0).sum() + 16 * n_q + n, (evals, int((se > 0).sum()))
            assert n_q < 16 or evals < n * n_q // 4
            assert np.array_equal(ss, se), np.argwhere(ss != se)[:8]
            assert np.array_equal(bs, be)
            for i in range(n_q):
                assert np.array_equal(cs[i][0], ce[i][0]) and np.array_equal(cs[i][1], ce[i][1]), (mode, i)
        dense = np.zeros((cap, dim), np.float32); dense[slots] = rows
        occ = np.zeros(cap, bool); occ[slots] = True
        for i in (0, n_q - 1):
            ref = np.where(occ, O.db_scores(qs[i], dense), -1.0).astype(np.float32)
            assert np.array_equal(ss[i], ref)
    finally:
        engine.set_option("db_gemm_min_queries", 8); engine.set_option("db_screen_min_rows", 6144); engine.set_option("match_stats", 0)
        db.close()


def test_database_batched_query_between_adds(engine):
    """the 8-bit copy and the row statistics of the screened batched query follow hfnet_db_add incrementally (only the 32-row tiles the added slots
    fall into are prepared again): queries between adds -- new slots, overwritten slots, a slot erased and re-added, tiles far apart -- equal the
    exact batched scan's bits every time"""
    from hfnet_slam_amd import capi
    rng = np.random.default_rng(77)
    cap, dim, n_q = 333, 4096, 24
    rows = _unit_rows(rng, cap, dim)
    db = capi.Database(engine, cap, dim)
    live = {}

    def check():
        qs = np.stack([rows[s] for s in list(live)[:n_q // 2]] + [r for r in _unit_rows(rng, n_q - min(len(live), n_q // 2), dim)]).astype(np.float32)
        qs = qs + 0.003 * rng.standard_normal(qs.shape).astype(np.float32)              # (distance ~0.19 from its row: score ~0.8)
        qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
        engine.set_option("db_gemm_min_queries", 1 << 20); engine.set_option("db_screen_min_rows", 0)
        ce, be, se = db.query_batch(qs, 0, want_scores=True)
        engine.set_option("db_gemm_min_queries", 8); engine.set_option("db_screen_min_rows", 6144)
        cs, bs, ss = db.query_batch(qs, 0, want_scores=True)
        assert np.array_equal(ss, se), np.argwhere(ss != se)[:8]
        assert np.array_equal(bs, be) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(cs, ce))
        assert (se > 0.5).sum() >= min(len(live), n_q // 2)                 # (the planted neighbours are found)

    try:
        for s in range(0, 40):
            db.add(s, rows[s]); live[s] = True
        check()
        for s in (40, 41, 200, 332):                                      # the same tile again, two tiles far apart
            db.add(s, rows[s]); live[s] = True
        check()
        rows[5] = rows[300]; db.add(5, rows[5])                             # overwritten
        db.erase(7); live.pop(7); check()
        db.add(7, rows[7]); live[7] = True
        for s in range(100, 131):
            db.add(s, rows[s]); live[s] = True
        check()
    finally:
        engine.set_option("db_gemm_min_queries", 8); engine.set_option("db_screen_min_rows", 6144)
        db.close()


def test_resampler_entry_point(engine):
    """free-standing Resampler (BaseModel.h:78-80) vs the oracle, incl. border / outside points and batch > 1"""
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    data = rng.standard_normal((2, 7, 9, 24)).astype(np.float32)
    warp = rng.uniform(-1.5, 10.5, (2, 300, 2)).astype(np.float32)
    warp[:, :6] = [[0, 0], [8, 6], [8.5, 6.5], [-1, 0], [9, 3], [3.25, 2.75]]
    assert np.array_equal(engine.resampler(data, warp), O.resampler(data, warp))


def test_tracking_loop_shape(engine, oracle_model):
    """BASELINE config 3 in miniature: per frame extract + match against the previous frame; every 5th frame is a
    keyframe: global descriptor into the database, place-recognition query against all previous keyframes, and
    SearchForTriangulation against the previous keyframes (LocalMapping.cc:516-520)."""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    w, h, nf = 128, 128, 150
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 4, max_batch=1)
    db = capi.Database(engine, 8, engine.global_dim)
    prev, kfs, ref_prev, ref_kfs = None, [], None, []
    for f in range(11):
        img = synth_image(h, w, 700 + f, "natural" if f % 2 else "uniform")
        n, kps, desc, g, _ = x.extract(img)
        rn, rk, rd, rg, _ = oracle_model.extract(img, nf, 0.01, 4, 1.2)
        _eq("kps", kps, rk); _eq("desc", desc, rd); _eq("global", g, rg)
        if prev is not None:
            c, m, d = engine.search_by_bow(prev, desc, 0.6)
            rc, rm, rdist = O.search_by_bow(ref_prev, rd, 0.6)
            assert c == rc
            _eq("match", m, rm); _eq("dist", d, rdist)
        if f % 5 == 0:
            if kfs:
                cs, sc, best, _ = db.query(g, 0)
                ref = O.db_scores(rg, np.stack([k[1] for k in ref_kfs]))
                ridx, rbest = O.db_candidates(ref, 0)
                assert best == rbest
                _eq("loop candidates", cs, ridx)
                for kd in kfs:
                    c, m = engine.search_for_triangulation(desc, kd[0], 0.75)
                    rc, rm = O.search_for_triangulation(rd, kd[0], 0.75)
                    assert c == rc
                    _eq("triangulation", m, rm)
            db.add(len(kfs), g)
            kfs.append((desc, g)); ref_kfs.append((rd, rg))
        prev, ref_prev = desc, rd
    x.close(); db.close()


def test_batched_bow_matches_per_pair_calls(engine):
    """hfnet_match_search_by_bow_batch == the single-pair entry point == the oracle, ragged row counts incl. an empty set"""
    from oracle import oracle as O
    rng = np.random.default_rng(31)
    S, mr = 6, 90
    n_rows = np.array([90, 64, 1, 0, 77, 90], np.int32)
    sets = np.zeros((S, mr, 256), np.float32)
    base = _unit_rows(rng, mr)
    for s_ in range(S):
        v = base + 0.05 * (s_ + 1) * rng.standard_normal((mr, 256)).astype(np.float32)
        sets[s_] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    pairs = [(0, 1), (1, 0), (2, 5), (3, 4), (4, 3), (5, 5), (0, 4)]
    cnt, match, dist = engine.search_by_bow_batch(sets, n_rows, pairs, 0.6)
    for p, (qs, ts) in enumerate(pairs):
        q, t = sets[qs, :n_rows[qs]], sets[ts, :n_rows[ts]]
        rn, rm, rd = O.search_by_bow(q, t, 0.6)
        assert cnt[p] == rn, (p, cnt[p], rn)
        _eq(f"pair {p} match", match[p, :n_rows[qs]], rm); _eq(f"pair {p} dist", dist[p, :n_rows[qs]], rd)
        if n_rows[qs]:
            n1, m1, d1 = engine.search_by_bow(q, t, 0.6)
            assert n1 == rn
            _eq("single", m1, rm)


@pytest.mark.parametrize("scale", [1.0, 3.0])
def test_bow_sweep_many_pairs(engine, engine_options, scale):
    """launches of >= 16 pairs of 256-D sets take the sweep form of the screening GEMM (kernels_match.hip k_bow_sweep256: query fragments resident
    in registers, train rows through an LDS ring): ragged row counts around its 32 / 64 / 256-row granules incl. an empty and a one-row set, clustered
    descriptors with near-ties and exact duplicates (more candidates than slots in a half tile), unnormalised rows (the band scales with |q|^2):
    every pair == the oracle.  And the screen has to SCREEN: a wrong product only costs exact evaluations, never a wrong match, so the number of
    exact evaluations (engine statistic stat_bow_exact) is checked too."""
    from oracle import oracle as O
    rng = np.random.default_rng(41)
    mr = 300
    n_rows = np.array([300, 257, 256, 255, 64, 65, 63, 1, 0, 200, 129, 300], np.int32)
    S = len(n_rows)
    centres = _unit_rows(rng, 9)
    sets = np.zeros((S, mr, 256), np.float32)
    for s_ in range(S):
        v = np.repeat(centres, 34, axis=0)[:mr] + (0.004 if s_ % 3 == 0 else 0.05) * rng.standard_normal((mr, 256)).astype(np.float32)
        v = v[rng.permutation(mr)]
        sets[s_] = (scale * v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    sets[1, 5] = sets[0, 9]; sets[1, 200] = sets[0, 9]; sets[0, 77] = sets[0, 9]; sets[2, :6] = sets[0, 9]      # duplicates on both sides
    pairs = [(a, (a * 5 + 1) % S) for a in range(S)] + [(0, 0), (1, 0), (0, 1), (8, 0), (0, 8), (7, 7), (11, 2), (2, 11)]
    assert len(pairs) >= 16
    engine_options({"match_stats": 1})
    engine.get_option("stat_bow_exact")                                      # (reads and clears)
    cnt, match, dist = engine.search_by_bow_batch(sets, n_rows, pairs, 0.6 * scale)
    for p, (qs, ts) in enumerate(pairs):
        q, t = sets[qs, :n_rows[qs]], sets[ts, :n_rows[ts]]
        rn, rm, rd = O.search_by_bow(q, t, 0.6 * scale)
        assert cnt[p] == rn, (p, qs, ts, cnt[p], rn)
        _eq(f"pair {p} ({qs}, {ts}) match", match[p, :n_rows[qs]], rm); _eq(f"pair {p} ({qs}, {ts}) dist", dist[p, :n_rows[qs]], rd)
    evals = engine.get_option("stat_bow_exact")
    products = sum(int(n_rows[a]) * int(n_rows[b]) for a, b in pairs)
    print(f"\nclustered sets x {len(pairs)} pairs, scale {scale}: {evals} exact evaluations of {products} products")
    assert 0 < evals < 0.25 * products, (evals, products)               # (nine tight clusters: many true near-ties; a broken screen evaluates everything)


def test_bow_sweep_screens(engine, engine_options):
    """the sweep on distinct descriptors (1000 x 1000 x 256, a planted permutation, 16 pairs): matches == the oracle's for one pair, and about one
    exact evaluation per train row and competitive query tile -- not 1000"""
    from oracle import oracle as O
    rng = np.random.default_rng(42)
    a = _unit_rows(rng, 1000)
    perm = rng.permutation(1000)
    b = a[perm] + 0.02 * rng.standard_normal((1000, 256)).astype(np.float32)
    b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    sets = np.stack([a, b]).astype(np.float32)
    nr = np.array([1000, 1000], np.int32)
    engine_options({"match_stats": 1})
    engine.get_option("stat_bow_exact")
    for n_pairs in (16, 4):                                                 # the sweep / k_bow_gemm_cand (fewer pairs)
        cnt, match, dist = engine.search_by_bow_batch(sets, nr, [(0, 1)] * n_pairs, 0.6)
        evals = engine.get_option("stat_bow_exact")
        rn, rm, rd = O.search_by_bow(a, b, 0.6)
        for p in (0, n_pairs - 1):
            assert cnt[p] == rn
            _eq("planted match", match[p], rm); _eq("planted dist", dist[p], rd)
        print(f"\nplanted permutation x {n_pairs} pairs: {evals / (n_pairs * 1000):.2f} exact evaluations per train row")
        assert n_pairs * 1000 <= evals <= 4 * n_pairs * 1000, evals


def test_batched_triangulation_matches_per_pair_calls(engine):
    """hfnet_match_search_for_triangulation_batch == the single-pair entry point == the oracle (30 neighbours, ragged)"""
    from oracle import oracle as O
    rng = np.random.default_rng(32)
    S, mr = 7, 150
    n_rows = np.array([150, 131, 1, 0, 97, 150, 64], np.int32)
    base = _unit_rows(rng, mr)
    sets = np.zeros((S, mr, 256), np.float32)
    for s_ in range(S):
        v = base[rng.permutation(mr)] + 0.03 * (s_ + 1) * rng.standard_normal((mr, 256)).astype(np.float32)
        sets[s_] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    pairs = [(0, s_) for s_ in range(S)] + [(4, 1), (2, 5), (3, 3), (6, 0)]
    cnt, match = engine.search_for_triangulation_batch(sets, n_rows, pairs, 0.75)
    for p, (a, b) in enumerate(pairs):
        d1, d2 = sets[a, :n_rows[a]], sets[b, :n_rows[b]]
        rn, rm = O.search_for_triangulation(d1, d2, 0.75)
        assert cnt[p] == rn, (p, cnt[p], rn)
        _eq(f"pair {p}", match[p, :n_rows[a]], rm)
        if n_rows[a]:
            n1, m1 = engine.search_for_triangulation(d1, d2, 0.75)
            assert n1 == rn
            _eq("single", m1, rm)


@pytest.mark.parametrize("tri_screen", [1, 0])
def test_triangulation_screened_path(engine, tri_screen):
    """SearchForTriangulation for several pairs with the threshold screen on the bf16 matrix pipe (engine option tri_screen_bf16)
    against the full f32 path and the oracle: planted matches, exact duplicates on both sides (first-maximum ties), products a few
    ulp around the threshold 1 - th^2 / 2, non-unit norms, ragged and empty sets, and a degenerate pair whose candidate list
    overflows (every product above the threshold) and has to come back through the full path -- next to ordinary pairs."""
    from oracle import oracle as O
    engine.set_option("tri_screen_bf16", tri_screen)
    try:
        rng = np.random.default_rng(41)
        mr, th = 320, 0.75
        thr = np.float32(-0.5 * th * th + 1)
        base = _unit_rows(rng, mr)
        sets = np.zeros((8, mr, 256), np.float32)
        n_rows = np.array([320, 301, 320, 7, 0, 320, 320, 200], np.int32)
        for s_ in (0, 1, 2, 3, 7):
            v = base[rng.permutation(mr)] + 0.02 * (s_ + 1) * rng.standard_normal((mr, 256)).astype(np.float32)
            sets[s_] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
        sets[1, 5] = sets[0, 9]; sets[1, 17] = sets[0, 9]; sets[0, 40] = sets[0, 9]        # duplicates: ties on rows and columns
        # set 2: rows at products a few ulp around the threshold with set 0's rows (b = a * c + orthogonal rest)
        for k in range(0, 60):
            a = sets[0, k].astype(np.float64)
            o = rng.standard_normal(256); o -= o.dot(a) * a; o /= np.linalg.norm(o)
            c = float(thr) + (k - 30) * 3e-8
            sets[2, k] = (c * a + np.sqrt(max(1 - c * c, 0.0)) * o).astype(np.float32)
        sets[5] = np.repeat(base[:1], mr, axis=0) + 1e-4 * rng.standard_normal((mr, 256)).astype(np.float32)      # degenerate: all alike
        sets[6] = 3.0 * sets[5]                                                                                   # ... at a norm of 3
        pairs = [(0, 1), (1, 0), (0, 2), (2, 0), (0, 3), (3, 0), (0, 4), (4, 0), (5, 5), (5, 6), (0, 7), (7, 1), (5, 0), (2, 2)]
        cnt, match = engine.search_for_triangulation_batch(sets, n_rows, pairs, th)
        for p, (a, b) in enumerate(pairs):
            d1, d2 = sets[a, :n_rows[a]], sets[b, :n_rows[b]]
            rn, rm = O.search_for_triangulation(d1, d2, th)
            assert cnt[p] == rn, (p, (a, b), cnt[p], rn)
            _eq(f"pair {p} {(a, b)}", match[p, :n_rows[a]], rm)
    finally:
        engine.set_option("tri_screen_bf16", 1)


@pytest.mark.parametrize("band", [1, 0])
def test_pyramid_resize_of_misaligned_device_rois(engine, oracle_model, engine_options, band):
    """Level-to-level resize of calls beyond pyramid_fuse (k_resize_u8_band: a workgroup's source rows through LDS as aligned 16-byte
    pieces, the partial pieces at a row's ends byte by byte; resize_band = 0: the per-thread gathers).  Device-resident frames that are
    ROIs of a larger buffer at odd offsets with an odd row stride give every piece alignment a turn; the buffer ENDS with the last
    ROI row's last byte, so a read past a row's end is a read past the allocation (the guard modes of test_gpu_guard.py fault on it).
    Widths that are not multiples of 4 / 16, one level wider than 1024 columns (the gather form's territory)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    from hfnet_slam_amd import capi
    engine_options({"resize_band": band, "pyramid_fuse": 0})
    dev = torch.device("cuda", 0)
    for (w, h, nl, F, ox, oy, pad) in [(203, 157, 4, 6, 5, 3, 13), (160, 120, 3, 5, 1, 0, 0), (1236, 72, 2, 5, 7, 2, 3)]:
        nf = 200
        x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, nl, max_batch=F)
        imgs = np.stack([synth_image(h, w, 4100 + i, "natural" if i % 2 else "uniform") for i in range(F)])
        rs = ox + w + pad                                       # odd row stride
        fs = (oy + h) * rs + 11                                 # odd frame stride
        flat = np.full(F * fs, 0xEE, np.uint8)
        for i in range(F):
            for r in range(h):
                o = i * fs + (oy + r) * rs + ox
                flat[o:o + w] = imgs[i, r]
        last = (F - 1) * fs + (oy + h - 1) * rs + ox + w        # one past the last ROI byte: the allocation ends here
        d_img = torch_to_device(flat[:last], dev)
        kps = torch.zeros((F, nf, 4), dtype=torch.float32, device=dev)
        desc = torch.zeros((F, nf, 256), dtype=torch.float32, device=dev)
        glob = torch.zeros((F, engine.global_dim), dtype=torch.float32, device=dev)
        n_rows = torch.zeros((F,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        x.extract_batch_device(F, d_img.data_ptr() + oy * rs + ox, rs, fs, kps.data_ptr(), desc.data_ptr(), glob.data_ptr(), n_rows.data_ptr())
        engine.synchronize(); torch.cuda.synchronize()
        n_d = torch_to_host(n_rows); k_d = torch_to_host(kps); d_d = torch_to_host(desc); g_d = torch_to_host(glob)
        for i in range(F):
            rn, rk, rd, rg, _ = oracle_model.extract(imgs[i], nf, 0.01, nl, 1.2)
            assert n_d[i] == rn, (w, h, i, n_d[i], rn)
            for j, f in enumerate(("x", "y", "response")):
                _eq(f"{w}x{h} frame {i} kps.{f}", k_d[i, :rn, j], rk[f])
            _eq(f"{w}x{h} frame {i} desc", d_d[i, :rn], rd)
            _eq(f"{w}x{h} frame {i} global", g_d[i], rg)
        assert x.device_faults() == 0
        x.close()


@pytest.mark.parametrize("streams", [0, 1, 2, 3])
def test_device_resident_pipeline_matches_host_path(engine, streams, engine_options):
    """bench.py's path: on_device extract_batch + batched SearchByBoW over consecutive steps, the global branch on its own
    stream with the join deferred into the next step (engine option two_streams = 3) -- every step must equal the host-pointer path."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU (its HIP runtime must initialise before libhfnet_hip.so: run with -m gpu)")
    from hfnet_slam_amd import capi
    import ctypes as C
    engine_options({"two_streams": streams})
    w, h, nf, B, steps = 192, 144, 300, 3, 4
    dev = torch.device("cuda", 0)
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 3, max_batch=B)
    xh = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 3, max_batch=B)          # reference: host-pointer calls
    imgs = np.stack([synth_image(h, w, 900 + i, "natural" if i % 2 else "uniform") for i in range(B * steps)])
    d_img = torch_to_device(imgs, dev)
    kps = torch.zeros((steps * B, nf, 4), dtype=torch.float32, device=dev)
    desc = torch.zeros((steps * B, nf, 256), dtype=torch.float32, device=dev)
    glob = torch.zeros((steps * B, engine.global_dim), dtype=torch.float32, device=dev)
    n_rows = torch.zeros((steps * B,), dtype=torch.int32, device=dev)
    match = torch.zeros((steps * B, nf), dtype=torch.int32, device=dev)
    mdist = torch.zeros((steps * B, nf), dtype=torch.float32, device=dev)
    mcnt = torch.zeros((steps * B,), dtype=torch.int32, device=dev)
    tset = torch.arange(0, steps * B, dtype=torch.int32, device=dev)
    qset = torch.clamp(tset - 1, min=0).to(torch.int32)
    torch.cuda.synchronize()
    L = capi.lib()
    for s_ in range(steps):                                                    # nothing waits on the host inside the loop
        o = s_ * B
        x.extract_batch_device(B, d_img[o].data_ptr(), w, w * h, kps[o].data_ptr(), desc[o].data_ptr(), glob[o].data_ptr(), n_rows[o:].data_ptr())
        engine.fence()
        st = L.hfnet_match_search_by_bow_batch(engine.h, B, C.c_void_p(desc.data_ptr()), C.c_size_t(nf * 256), C.c_void_p(n_rows.data_ptr()), steps * B,
                                               C.c_void_p(qset[o:].data_ptr()), C.c_void_p(tset[o:].data_ptr()), nf, 256, C.c_float(0.6),
                                               C.c_void_p(match[o].data_ptr()), C.c_void_p(mdist[o].data_ptr()), C.c_void_p(mcnt[o:].data_ptr()), 1)
        assert st == capi.OK, capi.last_error()
    engine.synchronize(); torch.cuda.synchronize()
    n_d = torch_to_host(n_rows); k_d = torch_to_host(kps); d_d = torch_to_host(desc); g_d = torch_to_host(glob)
    m_d = torch_to_host(match); c_d = torch_to_host(mcnt)
    prev = None
    for i in range(steps * B):
        n, k, d, g, _ = xh.extract(imgs[i])
        assert n_d[i] == n, (i, n_d[i], n)
        kk = k_d[i, :n]
        for j, f in enumerate(("x", "y", "response")):
            _eq(f"frame {i} kps.{f}", kk[:, j], k[f])
        _eq(f"frame {i} desc", d_d[i, :n], d)
        _eq(f"frame {i} global", g_d[i], g)
        if prev is not None:
            cn, cm, _ = engine.search_by_bow(prev, d, 0.6)
            assert c_d[i] == cn
            _eq(f"frame {i} matches", m_d[i, :len(prev)], cm)
        prev = d
    x.close(); xh.close()


def test_concurrent_callers(engine, oracle_model):
    """SURVEY.md 8(b) threading contract: the per-level model instances are called concurrently from worker threads
    (HFextractor.cc:265), the global-only model from two SLAM threads at once (Tracking.cc:2026, LocalMapping.cc:367), the
    matcher from anywhere -- different contexts run concurrently, one context serialises internally; results stay exact."""
    import threading
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    sizes = [(120, 160), (100, 133), (83, 111), (69, 92)]
    imgs = [synth_image(h, w, 700 + i) for i, (h, w) in enumerate(sizes)]
    models = [capi.Model(engine, capi.MODE_LOCAL_AND_INTERMEDIATE if i == 0 else capi.MODE_LOCAL, h, w, 300) for i, (h, w) in enumerate(sizes)]
    refs = [oracle_model.detect(img, O.MODE_LOCAL_AND_INTERMEDIATE if i == 0 else O.MODE_LOCAL, 200, 0.01) for i, img in enumerate(imgs)]
    gm = capi.Model(engine, capi.MODE_INTERMEDIATE_TO_GLOBAL, sizes[0][0] // 8, sizes[0][1] // 8, 1)    # the intermediate map's size
    inter = refs[0][3]
    ok_g, ref_g = oracle_model.detect_global(inter)
    assert ok_g
    rng = np.random.default_rng(5)
    a = _unit_rows(rng, 200); b = _unit_rows(rng, 180)
    ref_m = O.search_by_bow(a, b, 0.6)
    errors = []

    def level_worker(i):
        try:
            for _ in range(8):
                st, kps, desc, aux = models[i].detect(imgs[i], 200, 0.01, with_aux=(i == 0))
                assert st == capi.OK, capi.last_error()
                ok, rk, rd, ri = refs[i]
                assert len(kps) == len(rk)
                for f in ("x", "y", "response"):
                    assert np.array_equal(kps[f], rk[f]), (i, f)
                assert np.array_equal(desc, rd), i
                if i == 0:
                    assert np.array_equal(aux, ri)
        except Exception as e:                                   # noqa: BLE001 -- reported below
            errors.append(("level", i, repr(e)))

    def global_worker(k):
        try:
            for _ in range(8):
                st, g = gm.detect_global(inter)
                assert st == capi.OK and np.array_equal(g, ref_g), k
        except Exception as e:                                   # noqa: BLE001
            errors.append(("global", k, repr(e)))

    def match_worker(k):
        try:
            for _ in range(8):
                n, m, d = engine.search_by_bow(a, b, 0.6)
                assert n == ref_m[0] and np.array_equal(m, ref_m[1]) and np.array_equal(d, ref_m[2]), k
        except Exception as e:                                   # noqa: BLE001
            errors.append(("match", k, repr(e)))

    threads = [threading.Thread(target=level_worker, args=(i,)) for i in range(4)]
    threads += [threading.Thread(target=global_worker, args=(k,)) for k in range(2)]
    threads += [threading.Thread(target=match_worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for m in models:
        m.close()
    gm.close()
    assert not errors, errors


def test_descriptor_store_matches_host_calls(engine):
    """device-resident keyframe descriptor sets: put once, match by slot pairs == the host-pointer entry points == the oracle"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    rng = np.random.default_rng(41)
    mr, S = 140, 6
    base = _unit_rows(rng, mr)
    store = capi.Store(engine, S, mr)
    sets = []
    for s_ in range(S):
        n = [140, 99, 1, 0, 140, 77][s_]
        v = base[rng.permutation(mr)][:n] + 0.03 * (s_ + 1) * rng.standard_normal((n, 256)).astype(np.float32)
        v = (v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)).astype(np.float32).reshape(n, 256)
        sets.append(v)
        store.put(s_, v)
        assert store.rows(s_) == n
    assert store.rows(S) == -1
    store.put(1, sets[5]); store.put(1, sets[1])                 # overwrite a slot
    pairs = [(0, 1), (1, 0), (4, 0), (2, 5), (3, 4), (5, 3), (4, 4)]
    cnt, match, dist = store.search_by_bow(pairs, 0.6)
    tcnt, tmatch = store.search_for_triangulation(pairs, 0.75)
    for p, (a, b) in enumerate(pairs):
        rn, rm, rd = O.search_by_bow(sets[a], sets[b], 0.6)
        assert cnt[p] == rn
        _eq(f"bow {p}", match[p, :len(sets[a])], rm); _eq(f"bow dist {p}", dist[p, :len(sets[a])], rd)
        tn, tm = O.search_for_triangulation(sets[a], sets[b], 0.75)
        assert tcnt[p] == tn
        _eq(f"tri {p}", tmatch[p, :len(sets[a])], tm)
    with pytest.raises(capi.HfnetError):
        store.search_by_bow([(0, S)], 0.6)
    store.close()


def test_descriptor_store_row_filters(engine):
    """flag-filtered sides (rows with / without a MapPoint, Matcher.cc:231-246, 808-834) == gather on the host + oracle + scatter"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    rng = np.random.default_rng(43)
    mr, S = 300, 5
    base = _unit_rows(rng, mr)
    store = capi.Store(engine, S, mr)
    sizes = [300, 257, 64, 300, 5]
    sets, flags = [], []
    for s_ in range(S):
        n = sizes[s_]
        v = base[rng.permutation(mr)][:n] + 0.03 * (s_ + 1) * rng.standard_normal((n, 256)).astype(np.float32)
        v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
        f = (rng.random(n) < [0.5, 0.3, 1.0, 0.0, 0.6][s_]).astype(np.uint8)
        sets.append(v); flags.append(f)
        store.put(s_, v); store.set_flags(s_, f)
    pairs = [(0, 1), (1, 0), (0, 2), (2, 3), (3, 0), (4, 1), (0, 0), (1, 4)]

    def rows(s_, filt):
        if filt == capi.ROWS_ALL:
            return np.arange(sizes[s_])
        return np.nonzero(flags[s_] == (1 if filt == capi.ROWS_FLAGGED else 0))[0]

    for f1, f2 in [(capi.ROWS_UNFLAGGED, capi.ROWS_UNFLAGGED), (capi.ROWS_FLAGGED, capi.ROWS_ALL), (capi.ROWS_ALL, capi.ROWS_FLAGGED)]:
        cnt, match, dist = store.search_by_bow(pairs, 0.6, f1, f2)
        tcnt, tmatch = store.search_for_triangulation(pairs, 0.75, f1, f2)
        for p, (a, b) in enumerate(pairs):
            ia, ib = rows(a, f1), rows(b, f2)
            rn, rm, rd = O.search_by_bow(sets[a][ia], sets[b][ib], 0.6)
            exp_m = np.full(sizes[a], -1, np.int32); exp_d = np.full(sizes[a], np.finfo(np.float32).max, np.float32)
            exp_m[ia] = np.where(rm >= 0, ib[np.maximum(rm, 0)] if len(ib) else -1, -1); exp_d[ia] = rd
            assert cnt[p] == rn, (f1, f2, p)
            _eq(f"bow {f1}{f2} {p}", match[p, :sizes[a]], exp_m); _eq(f"bow dist {f1}{f2} {p}", dist[p, :sizes[a]], exp_d)
            tn, tm = O.search_for_triangulation(sets[a][ia], sets[b][ib], 0.75)
            exp_t = np.full(sizes[a], -1, np.int32)
            exp_t[ia] = np.where(tm >= 0, ib[np.maximum(tm, 0)] if len(ib) else -1, -1)
            assert tcnt[p] == tn, (f1, f2, p)
            _eq(f"tri {f1}{f2} {p}", tmatch[p, :sizes[a]], exp_t)
    store.put(0, sets[0])                                            # put clears the flags of the slot
    cnt, _, _ = store.search_by_bow([(0, 1)], 0.6, capi.ROWS_FLAGGED, capi.ROWS_ALL)
    assert cnt[0] == 0
    with pytest.raises(capi.HfnetError):
        store.search_by_bow([(0, 1)], 0.6, 3, 0)
    store.close()


def test_random_geometries_match_oracle(engine, oracle_model):
    """sizes around the tile edges of the fused kernels (cell grids of 15..17, 31..33 columns, odd remainders, one to five
    levels), two frames per call: keypoints, descriptors and global descriptors are the oracle's, bit for bit"""
    from hfnet_slam_amd import capi
    rng = np.random.default_rng(20260928)
    cases = [(8 * 16 + 3, 8 * 15 + 1, 2), (8 * 17, 8 * 33 + 7, 3), (8 * 31 + 5, 8 * 9, 1), (8 * 32, 8 * 32, 4), (8 * 33 + 1, 8 * 7 + 6, 2)]
    for _ in range(5):
        cases.append((int(rng.integers(72, 400)), int(rng.integers(72, 300)), int(rng.integers(1, 6))))
    for ci, (w, h, nl) in enumerate(cases):
        nf = int(rng.integers(40, 400))
        x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, nl, max_batch=2)
        imgs = np.stack([synth_image(h, w, 4000 + 2 * ci, "natural"), synth_image(h, w, 4001 + 2 * ci)])
        nb, kb, db, gb = x.extract_batch(imgs)
        for f in range(2):
            rn, rk, rd, rg, _ = oracle_model.extract(imgs[f], nf, 0.01, nl, 1.2)
            assert nb[f] == rn, (w, h, nl, f)
            assert np.array_equal(kb[f, :rn], rk), (w, h, nl, f)
            _eq(f"desc {w}x{h}x{nl} frame {f}", db[f, :rn], rd)
            _eq(f"global {w}x{h}x{nl} frame {f}", gb[f], rg)
        x.close()


@pytest.mark.parametrize("mult,n_clusters,global_dim", [(1.0, 32, 4096), (0.5, 16, 256), (0.35, 8, 64), (0.75, 64, 1024), (1.4, 32, 512)])
def test_other_network_widths(engine, tmp_path, mult, n_clusters, global_dim):
    """the kernels are specialised for the depth multiplier 0.75 of the published model (hf_net.py:13-52); every other width,
    cluster count and global dimension goes through the generic kernels (unfused blocks, run-time loops) with the same bits"""
    from hfnet_slam_amd import capi, spec, weights
    from oracle import oracle as O
    p = str(tmp_path / "w.hfw")
    weights.save(p, weights.synthetic_weights(11, spec.net_spec(mult, n_clusters, global_dim)))
    m = O.Model(p)
    e = capi.Engine(p, 0)
    try:
        for (w, h, nl, nf) in [(248, 168, 3, 300), (131, 121, 2, 150)]:
            x = capi.Extractor(e, w, h, nf, 0.01, 1.2, nl, max_batch=2)
            imgs = np.stack([synth_image(h, w, 51, "natural"), synth_image(h, w, 52)])
            nb, kb, db, gb = x.extract_batch(imgs)
            for f in range(2):
                rn, rk, rd, rg, _ = m.extract(imgs[f], nf, 0.01, nl, 1.2)
                assert nb[f] == rn, (mult, w, h, f)
                assert np.array_equal(kb[f, :rn], rk), (mult, w, h, f)
                _eq(f"desc x{mult} {w}x{h} frame {f}", db[f, :rn], rd)
                _eq(f"global x{mult} {w}x{h} frame {f}", gb[f], rg)
            x.close()
    finally:
        e.close()


@pytest.mark.parametrize("fuse_min_wgs", [0, None], ids=["fused_forms", "default_dispatch"])
@pytest.mark.parametrize("mult", [0.75, 0.5, 1.4])
def test_split_bf16_options_on_ragged_geometries_and_other_widths(engine, oracle_model, weights_path, tmp_path, mult, fuse_min_wgs):
    """desc_bf16x3 / global_bf16x3 away from the two full-size configurations: level sizes that leave partial tiles on every border
    (the split-bf16 forms of the fused blocks have their own border handling), single-frame calls and calls of three frames, the
    fused forms forced onto launches this small (fuse_min_wgs = 0) and the default dispatch, and network widths whose layers only
    the generic kernels cover.  Keypoints == oracle bit for bit; descriptors / global descriptor within the header's tolerances."""
    from hfnet_slam_amd import capi, spec, weights
    from oracle import oracle as O
    if mult == 0.75:
        e, m, own = engine, oracle_model, False
    else:
        p = str(tmp_path / "w.hfw")
        weights.save(p, weights.synthetic_weights(13, spec.net_spec(mult, 32, 1024)))
        e, m, own = capi.Engine(p, 0), O.Model(p), True
    gdim = 4096 if mult == 0.75 else 1024
    tol_g = 2e-5 * np.sqrt(4096 / gdim)         # (the header's tolerance is on the elements of a unit vector: they scale with 1 / sqrt(dimension))
    opts = {"desc_bf16x3": 1, "global_bf16x3": 1}
    if fuse_min_wgs is not None:
        opts["fuse_min_wgs"] = fuse_min_wgs
    saved = {o: e.get_option(o) for o in opts}
    try:
        for o, v in opts.items():
            e.set_option(o, v)
        for (w, h, nl, nf) in [(200, 152, 4, 500), (131, 121, 2, 150), (248, 168, 3, 300), (376, 240, 2, 400)]:
            x = capi.Extractor(e, w, h, nf, 0.01, 1.2, nl, max_batch=3)
            imgs = np.stack([synth_image(h, w, 71, "natural"), synth_image(h, w, 72), synth_image(h, w, 73, "natural")])
            nb, kb, db, gb = x.extract_batch(imgs)
            n1, k1, d1, g1, _ = x.extract(imgs[1])
            for f in range(3):
                rn, rk, rd, rg, _ = m.extract(imgs[f], nf, 0.01, nl, 1.2)
                assert nb[f] == rn and np.array_equal(kb[f, :rn], rk), (mult, w, h, f)
                assert np.abs(db[f, :rn].astype(np.float64) - rd).max() <= 1e-5, (mult, w, h, f)
                assert np.abs(gb[f].astype(np.float64) - rg).max() <= tol_g, (mult, w, h, f)
                if f == 1:
                    assert n1 == rn and np.array_equal(k1, rk)
                    assert np.abs(d1.astype(np.float64) - rd).max() <= 1e-5 and np.abs(g1.astype(np.float64) - rg).max() <= tol_g
            x.close()
    finally:
        for o, v in saved.items():
            e.set_option(o, v)
        if own:
            e.close()


@pytest.mark.parametrize("B", [1, 4])
def test_split_bf16_fused_forms_on_calls_of_few_frames(engine, oracle_model, engine_options, B):
    """NOTEBOOK.md R4.8: with fuse_min_wgs lowered the split-bf16 forms of the fused blocks also run in calls of <= 4 frames, whose sampler
    overlaps the global branch.  Built with clang's default SLP packing, k_sample's v_pk_mul_f32 / v_pk_add_f32 then returned wrong values in
    lanes 48-63 (wrong descriptors in a few rows of nearly every call, or a GPU memory fault); the library is compiled without packed f32
    instructions since (hfnet_slam_amd/build.py, tests/test_abi.py).  One extractor, repeated calls on two alternating sets of frames, every
    frame of every call against the oracle."""
    from hfnet_slam_amd import capi
    engine_options({"global_bf16x3": 1, "fuse_min_wgs": 0, "join_fused_branch": 0})
    w, h, nf, nl = 752, 480, 1000, 4
    sets = [np.stack([synth_image(h, w, 8100 + 10 * s + i, "natural") for i in range(B)]) for s in range(2)]
    refs = [[oracle_model.extract(im[i], nf, 0.01, nl, 1.2) for i in range(B)] for im in sets]
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, nl, max_batch=B)
    try:
        for call in range(24):
            s = call & 1
            nb, kb, db, gb = x.extract_batch(sets[s])
            for f in range(B):
                rn, rk, rd, rg, _ = refs[s][f]
                assert nb[f] == rn and np.array_equal(kb[f, :rn], rk), (call, f)
                assert np.array_equal(db[f, :rn], rd), (call, f, int((np.abs(db[f, :rn] - rd).max(axis=1) > 0).sum()))      # (desc_bf16x3 is off: the oracle's bits)
                assert np.abs(gb[f].astype(np.float64) - rg).max() <= 2e-5, (call, f)
    finally:
        x.close()


def test_store_put_extracted_matches_host_round_trip(engine, oracle_model):
    """frame-to-frame tracking without the descriptors leaving the GPU: extract, keep the block in a store slot, match by slot"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    w, h, nf = 200, 152, 260
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 3, max_batch=2)
    store = capi.Store(engine, 3, nf)
    descs = []
    for i in range(3):
        img = synth_image(h, w, 8100 + i, "natural")
        if i == 2:                                               # two frames per call: staging frame 1
            nb, _, db, _ = x.extract_batch(np.stack([synth_image(h, w, 1, "uniform"), img]))
            n, d = int(nb[1]), db[1, :nb[1]]
            store.put_extracted(i, x, 1)
        else:
            n, _, d, _, _ = x.extract(img)
            store.put_extracted(i, x, 0)
        assert store.rows(i) == n
        descs.append(np.array(d))
    pairs = [(0, 1), (1, 2), (2, 0)]
    cnt, match, dist = store.search_by_bow(pairs, 0.6)
    for p, (a, b) in enumerate(pairs):
        rn, rm, rd = O.search_by_bow(descs[a], descs[b], 0.6)
        assert cnt[p] == rn
        _eq(f"match {p}", match[p, :len(descs[a])], rm); _eq(f"dist {p}", dist[p, :len(descs[a])], rd)
    with pytest.raises(capi.HfnetError):
        store.put_extracted(0, x, 2)
    small = capi.Store(engine, 1, 8)
    with pytest.raises(capi.HfnetError):
        small.put_extracted(0, x, 0)
    small.close(); store.close(); x.close()


@pytest.mark.parametrize("host_global", [1, 0])
def test_put_extracted_then_immediate_extract_then_match(engine, oracle_model, engine_options, host_global):
    """the invariant hfnet_store_put_extracted relies on since the latency path stopped draining its stream (engine option
    host_global = 1: the call returns while the global branch may still run): the block's local section is complete once the host
    has seen the local-results flag, and the NEXT extraction -- which overwrites the staging block the store's copies read -- waits
    for those copies on the device.  put_extracted, an immediate extract and a match, back to back, eight times."""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    engine_options({"host_global": host_global})
    w, h, nf = 320, 240, 400
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 4, max_batch=1)
    store = capi.Store(engine, 2, nf)
    prev = None
    for i in range(8):
        img = synth_image(h, w, 8200 + i, "natural" if i % 2 else "uniform")
        n, kps, desc, g, _ = x.extract(img)
        store.put_extracted(i & 1, x, 0)
        x.extract(synth_image(h, w, 8300 + i))                    # overwrites the staging block right behind the copies
        rn, rk, rd, rg, _ = oracle_model.extract(img, nf, 0.01, 4, 1.2)
        assert n == rn and np.array_equal(desc, rd) and np.array_equal(g, rg), i
        if prev is not None:
            cnt, match, dist = store.search_by_bow([(1 - (i & 1), i & 1)], 0.6)
            rc, rm, rdist = O.search_by_bow(prev, rd, 0.6)
            assert cnt[0] == rc, i
            _eq(f"match {i}", match[0, :len(prev)], rm); _eq(f"dist {i}", dist[0, :len(prev)], rdist)
        prev = rd
    store.close(); x.close()


def test_container_without_memberships_gamma(oracle_model, weights_path, tmp_path):
    """a real HF-Net checkpoint has no BatchNorm gamma for the NetVLAD memberships conv (slim.batch_norm scale=False): the
    library reads the missing tensor as 1 -- same bits as a container that stores ones, and as the oracle on either"""
    from hfnet_slam_amd import capi, weights
    from oracle import oracle as O
    w = weights.load(weights_path)
    gname = "global_head/vlad/memberships/BatchNorm/gamma"
    without = {k: v for k, v in w.items() if k != gname}
    ones = dict(w); ones[gname] = np.ones_like(w[gname])
    pa, pb = str(tmp_path / "without.hfw"), str(tmp_path / "ones.hfw")
    weights.save(pa, without); weights.save(pb, ones)
    img = synth_image(96, 128, 21)
    res = []
    for p in (pa, pb):
        e = capi.Engine(p, 0)
        m = capi.Model(e, capi.MODE_LOCAL_AND_GLOBAL, 96, 128, 200)
        st, kps, desc, g = m.detect(img, 150, 0.01)
        assert st == capi.OK, capi.last_error()
        res.append(g.copy())
        m.close(); e.close()
    _eq("global: missing gamma == ones", res[0], res[1])
    ok, _, _, rg = O.Model(pa).detect(img, O.MODE_LOCAL_AND_GLOBAL, 150, 0.01)
    assert ok
    _eq("global vs oracle", res[0], rg)


def test_host_pipeline_and_attached_store(engine, oracle_model):
    """host-pointer batch call over several chunks (the double-buffered pipeline: 7 frames through a max_batch=2 extractor,
    an odd tail, strided input) == per-frame oracle; the attached store receives every frame's block device to device, and
    matching by slot equals the oracle's SearchByBoW on the downloaded descriptors"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    w, h, nf, nl, F = 160, 120, 250, 3, 7
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, nl, max_batch=2)
    store = capi.Store(engine, 8, nf)
    x.attach_store(store, 3)                                   # frame f -> slot (3 + f) % 8
    big = np.stack([synth_image(h + 6, w + 10, 700 + i, "natural" if i % 2 else "uniform") for i in range(F)])
    imgs = big[:, 2:2 + h, 4:4 + w]                            # non-contiguous rows: row stride w + 10
    kps = np.zeros((F, nf), capi.KP_DTYPE); desc = np.zeros((F, nf, 256), np.float32)
    g = np.zeros((F, engine.global_dim), np.float32); n = np.zeros((F,), np.int32)
    import ctypes as C
    st = capi.lib().hfnet_extractor_extract_batch(x.h, F, C.c_void_p(imgs.ctypes.data), imgs.strides[1], C.c_size_t(imgs.strides[0]),
                                                  C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data), C.c_void_p(g.ctypes.data),
                                                  C.c_void_p(n.ctypes.data), 0)
    assert st == capi.OK, capi.last_error()
    ref = [oracle_model.extract(np.ascontiguousarray(imgs[i]), nf, 0.01, nl, 1.2) for i in range(F)]
    for i, (rn, rk, rd, rg, _) in enumerate(ref):
        assert n[i] == rn
        _eq(f"kps {i}", kps[i, :rn], rk); _eq(f"desc {i}", desc[i, :rn], rd); _eq(f"global {i}", g[i], rg)
        assert store.rows((3 + i) % 8) == rn
    pairs = [((3 + i - 1) % 8, (3 + i) % 8) for i in range(1, F)]
    cnt, match, dist = store.search_by_bow(pairs, 0.6)
    for p, i in enumerate(range(1, F)):
        rc, rm, rdist = O.search_by_bow(ref[i - 1][2], ref[i][2], 0.6)
        assert cnt[p] == rc
        _eq(f"match {i}", match[p, :ref[i - 1][0]], rm); _eq(f"dist {i}", dist[p, :ref[i - 1][0]], rdist)
    # the last chunk is what put_extracted sees (one frame: F is odd)
    store.put_extracted(2, x, 0)                               # (slot 2 is not one of the attached frames' slots)
    assert store.rows(2) == ref[F - 1][0]
    cnt2, m2, _ = store.search_by_bow([((3 + F - 2) % 8, 2)], 0.6)
    assert cnt2[0] == cnt[-1] and np.array_equal(m2[0], match[-1])
    x.attach_store(None)
    store.close(); x.close()


def test_registered_host_buffers_take_the_direct_path(engine, oracle_model):
    """hfnet_host_register: a host-pointer batch call whose image block and result buffers are all registered DMAs straight from /
    into the caller's memory (no pinned staging, no staging memcpy); same bits as the staged pipeline and the oracle, with an
    attached store, over several chunks incl. a ragged last one; a partly registered call falls back to the staged pipeline."""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    w, h, nf, B, F = 176, 136, 220, 3, 8
    x = capi.Extractor(engine, w, h, nf, 0.01, 1.2, 3, max_batch=B)
    store = capi.Store(engine, F, nf)
    x.attach_store(store, 0)
    imgs = np.stack([synth_image(h, w, 9900 + i, "natural" if i % 2 else "uniform") for i in range(F)])
    ref = x.extract_batch(imgs)                                                  # pageable: staged pipeline
    out = tuple(np.zeros_like(a) for a in ref)
    bufs = [imgs, out[1], out[2], out[3], out[0]]
    for b in bufs:
        capi.host_register(b)
    try:
        with pytest.raises(capi.HfnetError):
            capi.host_register(imgs)                                             # twice
        got = x.extract_batch(imgs, out)
        for f in range(F):
            n = int(ref[0][f])
            assert got[0][f] == n and np.array_equal(got[1][f, :n], ref[1][f, :n]) and np.array_equal(got[2][f, :n], ref[2][f, :n])
            assert np.array_equal(got[3][f], ref[3][f])
            rn, rk, rd, rg, _ = oracle_model.extract(imgs[f], nf, 0.01, 3, 1.2)
            assert n == rn and np.array_equal(got[1][f, :n], rk) and np.array_equal(got[2][f, :n], rd) and np.array_equal(got[3][f], rg)
            assert store.rows(f) == n
        cnt, match, dist = store.search_by_bow([(f - 1, f) for f in range(1, F)], 0.6)
        for p_, f in enumerate(range(1, F)):
            rc, rm, rdist = O.search_by_bow(ref[2][f - 1, :ref[0][f - 1]], ref[2][f, :ref[0][f]], 0.6)
            assert cnt[p_] == rc and np.array_equal(match[p_, :len(rm)], rm) and np.array_equal(dist[p_, :len(rm)], rdist)
        # strided frames (a ROI per frame) cannot be DMAed as one block: staged path, same results
        big = np.zeros((F, h + 4, w + 8), np.uint8); big[:, 2:2 + h, 4:4 + w] = imgs
        capi.host_register(big)
        try:
            L = capi.lib()
            import ctypes as C
            kps2 = np.zeros_like(out[1]); desc2 = np.zeros_like(out[2]); g2 = np.zeros_like(out[3]); n2 = np.zeros_like(out[0])
            roi = big[:, 2:2 + h, 4:4 + w]
            st = L.hfnet_extractor_extract_batch(x.h, F, C.c_void_p(roi.ctypes.data), roi.strides[1], C.c_size_t(roi.strides[0]), C.c_void_p(kps2.ctypes.data),
                                                 C.c_void_p(desc2.ctypes.data), C.c_void_p(g2.ctypes.data), C.c_void_p(n2.ctypes.data), 0)
            assert st == capi.OK, capi.last_error()
            for f in range(F):
                n = int(ref[0][f])
                assert n2[f] == n and np.array_equal(kps2[f, :n], ref[1][f, :n]) and np.array_equal(desc2[f, :n], ref[2][f, :n]) and np.array_equal(g2[f], ref[3][f])
        finally:
            capi.host_unregister(big)
    finally:
        for b in bufs:
            capi.host_unregister(b)
    with pytest.raises(capi.HfnetError):
        capi.host_unregister(imgs)                                               # not registered any more
    x.attach_store(None)
    store.close(); x.close()


def test_windowed_matcher_loop_and_distinctive_descriptors(engine):
    """the candidate loop of the windowed matchers (Matcher.cc:74-110 and siblings) and
    MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:366-400) on the device == oracle, bit for bit"""
    from hfnet_slam_amd import capi
    from oracle import oracle as O
    rng = np.random.default_rng(41)
    nq, nt = 700, 1000
    t = _unit_rows(rng, nt)
    q = (t[rng.integers(0, nt, nq)] + 0.05 * rng.standard_normal((nq, 256)).astype(np.float32))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    t[17] = t[3]                                                    # exact ties inside candidate lists
    lv = rng.integers(0, 4, nt).astype(np.int32)
    lens = rng.integers(0, 40, nq); lens[0] = 0; lens[1] = 300
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([rng.choice(nt, l, replace=False) for l in lens]).astype(np.int32)
    idx[off[1]:off[1] + 2] = [3, 17]
    for levels in (lv, None):
        got = engine.match_candidates(q, t, levels, off, idx)
        ref = O.match_candidates(q, t, levels, off, idx)
        for name, a, b in zip(("best_idx", "best_dist", "best_level", "second_dist", "second_level"), got, ref):
            _eq(name, a, b)
    assert got[0][0] == -1
    # error behaviour: a candidate outside the train matrix is rejected, not read
    bad = idx.copy(); bad[5] = nt
    with pytest.raises(capi.HfnetError) as ei:
        engine.match_candidates(q, t, lv, off, bad)
    assert ei.value.status == capi.ERR_INVALID_ARG
    # distinctive descriptors: 300 map points with 0..96 observations
    sizes = rng.integers(0, 40, 300); sizes[0] = 0; sizes[1] = 1; sizes[2] = 96
    soff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = _unit_rows(rng, 300)
    desc = np.concatenate([base[s][None] + 0.2 * rng.standard_normal((n, 256)).astype(np.float32) for s, n in enumerate(sizes)])
    desc = (desc / np.linalg.norm(desc, axis=1, keepdims=True)).astype(np.float32)
    desc[soff[5] + 1] = desc[soff[5]]                               # duplicate observations
    _eq("distinctive", engine.distinctive_descriptors(desc, soff), O.distinctive_descriptors(desc, soff))
    with pytest.raises(capi.HfnetError) as ei:
        engine.distinctive_descriptors(_unit_rows(rng, 97), np.array([0, 97], np.int32))
    assert ei.value.status == capi.ERR_CAPACITY


def test_checkpoint_importer_end_to_end(tmp_path):
    """SURVEY 8f rank 3 on the GPU: a full-width HF-Net checkpoint in TensorFlow's tensor-bundle format (as a real one:
    no gamma on the memberships BatchNorm, clusters as [1, 1, 1, D, K], an optimiser slot and global_step next to the
    weights) -> `python -m hfnet_slam_amd.tf_checkpoint` -> container -> engine; keypoints, descriptors and global
    descriptor equal the oracle's on the same container."""
    from hfnet_slam_amd import capi, tf_checkpoint as T, weights as W
    from oracle import oracle as O
    ref = W.synthetic_weights(11)
    gname = "global_head/vlad/memberships/BatchNorm/gamma"
    ck = {n: (a.reshape(1, 1, 1, *a.shape) if n == "global_head/vlad/clusters" else a) for n, a in ref.items() if n != gname}
    ck["global_step"] = np.array(83096, np.int64)
    ck["MobilenetV2/Conv/weights/RMSProp"] = np.zeros_like(ref["MobilenetV2/Conv/weights"])
    prefix = str(tmp_path / "model.ckpt-83096")
    T.write_bundle(prefix, ck)
    out = str(tmp_path / "hfnet.hfw")
    assert T.main([prefix, out]) == 0
    eng = capi.Engine(out, 0)
    model = O.Model(out)
    x = capi.Extractor(eng, 200, 136, 150, 0.01, 1.2, 3, max_batch=1)
    img = synth_image(136, 200, 77, "natural")
    n, kps, desc, g, npl = x.extract(img)
    rn, rk, rd, rg, rnpl = model.extract(img, 150, 0.01, 3, 1.2)
    assert n == rn and np.array_equal(npl, rnpl)
    _eq("importer kps", kps, rk); _eq("importer desc", desc, rd); _eq("importer global", g, rg)
    x.close(); eng.close()
