"""Randomised soak of the HIP path against the oracle (bit-exact): extractor geometries / budgets / batch sizes / engine
options, both matchers, the database scan.    python tools/dev/soak.py [seconds=600] [seed=1]
Prints one line per failure and a summary; exit code 1 if anything differed."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hfnet_slam_amd import capi, weights
from oracle import oracle as O
from conftest import synth_image



class CaseLog:
    """One line per case, written (flushed + fsync'd) BEFORE the case's first library call, so that a process abort
    (a device fault takes the host process down through the HIP runtime's queue-error callback) names its case.
    File: $HFNET_SOAK_LOG, else gpurun_out/soak_cases_<seed>.log under the repo root if that directory exists, else the
    temporary directory.  The last line of the file after an abort is the case that died."""

    def __init__(self, seed):
        path = os.environ.get("HFNET_SOAK_LOG")
        if not path:
            d = os.path.join(ROOT, "gpurun_out")
            path = os.path.join(d if os.path.isdir(d) else tempfile.gettempdir(), f"soak_cases_{seed}.log")
        self.path = path
        self.fh = open(path, "a")
        self.note(f"# soak seed {seed} pid {os.getpid()} build {capi.build_id() if hasattr(capi, 'build_id') else '?'}")

    def note(self, text):
        self.fh.write(text + "\n")
        self.fh.flush()
        try:
            os.fsync(self.fh.fileno())
        except OSError:                      # (/dev/null and friends)
            pass


def run(budget_s: float, seed: int, big_share: float = 0.02, max_cases: int = 0):
    """-> (cases, failures); big_share: fraction of full-size multi-frame cases; max_cases > 0: stop after that many cases
    whatever the clock says (the case sequence is a function of the seed alone: a replay of the first N cases is exact)"""
    rng = np.random.default_rng(seed)
    log = CaseLog(seed)
    wpath = os.path.join(tempfile.gettempdir(), f"hfnet_soak_{seed}.hfw")
    weights.save(wpath, weights.synthetic_weights(100 + seed))
    model = O.Model(wpath)
    fails, cases = [], 0
    t_end = time.time() + budget_s


    def unit(n, d=256):
        a = rng.standard_normal((n, d)).astype(np.float32)
        return (a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)).astype(np.float32)


    while time.time() < t_end and not (max_cases and cases >= max_cases):
        eng = capi.Engine(wpath, 0)
        opts = {"fuse_blocks": int(rng.integers(0, 2)), "fused_variant": int(rng.choice([2, 4, 6, 8])), "fuse_min_wgs": int(rng.choice([0, 256])), "fuse_stem": int(rng.integers(0, 2)),
                "dense_desc": int(rng.integers(0, 2)), "two_streams": int(rng.integers(0, 4)), "conv_wlds": int(rng.integers(0, 2)),
                "match_screen_bf16": int(rng.integers(0, 2)), "tri_screen_bf16": int(rng.integers(0, 2)),
                "fc_tile": int(rng.choice([0, 1, 2, 4])), "resize_band": int(rng.integers(0, 2)), "pyramid_fuse": int(rng.choice([0, 4]))}
        if rng.random() < 0.5:
            opts = {}
        for k, v in opts.items():
            eng.set_option(k, v)
        for _ in range(6):
            if time.time() >= t_end or (max_cases and cases >= max_cases):
                break
            kind = rng.random()
            cases += 1
            note = lambda what, *a: log.note(f"{cases} {what} {a} opts={opts}")      # noqa: E731
            try:
                if kind < big_share:
                    # full-size frames at a random call size: the kernel choices (column tiles per wave, LDS-weight kernels,
                    # slot skipping, low-latency 1x1) switch with the number of tiles of a call
                    w, h = (752, 480) if rng.random() < 0.6 else (512, 512)
                    nf = int(rng.choice([1000, 850, 300])); B = int(rng.integers(1, 41))
                    mb = int(rng.choice([B, 8, 32]))
                    note("extract_full", w, h, nf, B, mb)
                    x = capi.Extractor(eng, w, h, nf, 0.01, 1.2, 4, max_batch=mb)
                    imgs = np.stack([synth_image(h, w, int(rng.integers(1 << 30)), "natural" if rng.random() < 0.5 else "uniform") for _ in range(B)])
                    nb, kb, db_, gb = x.extract_batch(imgs)
                    for i in sorted(set(int(v) for v in rng.integers(0, B, 2))):
                        rn, rk, rd, rg, _ = model.extract(imgs[i], nf, 0.01, 4, 1.2)
                        if not (nb[i] == rn and np.array_equal(kb[i, :rn], rk) and np.array_equal(db_[i, :rn], rd) and np.array_equal(gb[i], rg)):
                            fails.append(("extract_full", w, h, nf, B, i, opts))
                            break
                    if x.device_faults():
                        fails.append(("device_fault", x.device_faults(), cases, opts))
                    x.close()
                elif kind < 0.5:
                    w, h = int(rng.integers(40, 420)), int(rng.integers(40, 340))
                    nl = int(rng.integers(1, 6)); nf = int(rng.integers(8, 1500)); thr = float(rng.choice([0.0, 0.002, 0.01, 0.02]))
                    sf = float(rng.choice([1.2, 1.1, 1.5]))
                    while nl > 1 and min(w, h) / sf ** (nl - 1) < 24:
                        nl -= 1
                    B = int(rng.choice([1, 1, 2, 3, 5, 12]))
                    mb = int(rng.choice([1, 2, 4, 16]))
                    note("extract", w, h, nl, nf, thr, sf, B, mb)
                    x = capi.Extractor(eng, w, h, nf, thr, sf, nl, max_batch=mb)
                    imgs = np.stack([synth_image(h, w, int(rng.integers(1 << 30)), "natural" if rng.random() < 0.5 else "uniform") for _ in range(B)])
                    nb, kb, db, gb = x.extract_batch(imgs)
                    for i in range(B):
                        rn, rk, rd, rg, _ = model.extract(imgs[i], nf, thr, nl, sf)
                        ok = nb[i] == rn and np.array_equal(kb[i, :rn], rk) and np.array_equal(db[i, :rn], rd) and np.array_equal(gb[i], rg)
                        if not ok:
                            fails.append(("extract", w, h, nl, nf, thr, sf, B, i, opts))
                            break
                    if x.device_faults():
                        fails.append(("device_fault", x.device_faults(), cases, opts))
                    x.close()
                elif kind < 0.62:
                    # one frame per call (graph path), twice with different images + the descriptor store fed from the extractor
                    w, h = int(rng.integers(64, 360)), int(rng.integers(64, 300))
                    nl = int(rng.integers(1, 5)); nf = int(rng.integers(16, 1200))
                    while nl > 1 and min(w, h) / 1.2 ** (nl - 1) < 24:
                        nl -= 1
                    mb = int(rng.choice([1, 3]))
                    note("extract1", w, h, nl, nf, mb)
                    x = capi.Extractor(eng, w, h, nf, 0.01, 1.2, nl, max_batch=mb)
                    for rep in range(3):
                        img = synth_image(h, w, int(rng.integers(1 << 30)), "natural" if rep % 2 else "uniform")
                        n, k, d, g, npl = x.extract(img)
                        rn, rk, rd, rg, rnpl = model.extract(img, nf, 0.01, nl, 1.2)
                        if n != rn or not np.array_equal(k, rk) or not np.array_equal(d, rd) or not np.array_equal(g, rg) or not np.array_equal(npl, rnpl):
                            fails.append(("extract1", w, h, nl, nf, rep, opts))
                            break
                    if x.device_faults():
                        fails.append(("device_fault", x.device_faults(), cases, opts))
                    x.close()
                elif kind < 0.72:
                    # device-resident store with row filters
                    mr, S = int(rng.integers(1, 400)), int(rng.integers(2, 7))
                    note("store", mr, S)
                    store = capi.Store(eng, S, mr)
                    base = unit(mr)
                    sets, flags = [], []
                    for s_ in range(S):
                        n = int(rng.integers(0, mr + 1))
                        v = base[rng.permutation(mr)][:n] + float(rng.choice([1e-4, 0.03, 0.08])) * rng.standard_normal((n, 256)).astype(np.float32)
                        v = (v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)).astype(np.float32).reshape(n, 256)
                        f = (rng.random(n) < rng.random()).astype(np.uint8)
                        sets.append(v); flags.append(f)
                        store.put(s_, v); store.set_flags(s_, f)
                    pairs = [(int(rng.integers(0, S)), int(rng.integers(0, S))) for _ in range(int(rng.integers(1, 9)))]
                    f1, f2 = int(rng.integers(0, 3)), int(rng.integers(0, 3))

                    def rows_of(s_, filt):
                        if filt == capi.ROWS_ALL:
                            return np.arange(len(sets[s_]))
                        return np.nonzero(flags[s_] == (1 if filt == capi.ROWS_FLAGGED else 0))[0]

                    cnt, match, dist = store.search_by_bow(pairs, 0.6, f1, f2)
                    tcnt, tmatch = store.search_for_triangulation(pairs, 0.75, f1, f2)
                    for pi, (a, b) in enumerate(pairs):
                        ia, ib = rows_of(a, f1), rows_of(b, f2)
                        rn, rm, rd = O.search_by_bow(sets[a][ia], sets[b][ib], 0.6)
                        na = len(sets[a])
                        exp_m = np.full(na, -1, np.int32); exp_d = np.full(na, np.finfo(np.float32).max, np.float32)
                        if len(ia):
                            exp_m[ia] = np.where(rm >= 0, ib[np.maximum(rm, 0)] if len(ib) else -1, -1); exp_d[ia] = rd
                        tn, tm = O.search_for_triangulation(sets[a][ia], sets[b][ib], 0.75)
                        exp_t = np.full(na, -1, np.int32)
                        if len(ia):
                            exp_t[ia] = np.where(tm >= 0, ib[np.maximum(tm, 0)] if len(ib) else -1, -1)
                        if cnt[pi] != rn or not np.array_equal(match[pi, :na], exp_m) or not np.array_equal(dist[pi, :na][exp_m >= 0], exp_d[exp_m >= 0]) \
                                or tcnt[pi] != tn or not np.array_equal(tmatch[pi, :na], exp_t):
                            fails.append(("store", mr, S, pairs[pi], f1, f2, na, len(sets[b])))
                            break
                    store.close()
                elif kind < 0.77:
                    # BaseModel mirror: the three Detect overloads
                    h, w = 8 * int(rng.integers(5, 40)), 8 * int(rng.integers(5, 50))
                    nk = int(rng.integers(1, 800)); thr = float(rng.choice([0.0, 0.01]))
                    mode = int(rng.choice([capi.MODE_LOCAL_AND_GLOBAL, capi.MODE_LOCAL, capi.MODE_LOCAL_AND_INTERMEDIATE]))
                    note("model", mode, h, w, nk, thr)
                    m = capi.Model(eng, mode, h, w, max_keypoints=nk)
                    img = synth_image(h, w, int(rng.integers(1 << 30)), "natural" if rng.random() < 0.5 else "uniform")
                    st, k, d, aux = m.detect(img, nk, thr)
                    ok, rk, rd, raux = model.detect(img, mode, nk, thr)
                    bad = (st == 0) != ok or not np.array_equal(k, rk) or not np.array_equal(d, rd) or (raux is not None and not np.array_equal(aux, raux))
                    if not bad and mode == capi.MODE_LOCAL_AND_INTERMEDIATE:
                        m2 = capi.Model(eng, capi.MODE_INTERMEDIATE_TO_GLOBAL, h // 8, w // 8)
                        st2, g = m2.detect_global(aux)
                        ok2, rg = model.detect_global(raux)
                        bad = (st2 == 0) != ok2 or not np.array_equal(g, rg)
                        m2.close()
                    if m.device_faults():
                        fails.append(("device_fault", m.device_faults(), cases, opts))
                    m.close()
                    if bad:
                        fails.append(("model", mode, h, w, nk, thr, opts))
                elif kind < 0.81:
                    # candidate loop of the windowed matchers + distinctive descriptors
                    nq, nt = int(rng.integers(1, 600)), int(rng.integers(1, 900))
                    note("candidates", nq, nt)
                    t = unit(nt)
                    q = t[rng.integers(0, nt, nq)] + float(rng.choice([0.0, 0.05])) * rng.standard_normal((nq, 256)).astype(np.float32)
                    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
                    lv = rng.integers(0, 4, nt).astype(np.int32)
                    lens = rng.integers(0, min(nt, 50) + 1, nq)
                    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
                    idx = (np.concatenate([rng.choice(nt, l, replace=False) for l in lens]) if lens.sum() else np.zeros(0)).astype(np.int32)
                    lvl = lv if rng.random() < 0.7 else None
                    got = eng.match_candidates(q, t, lvl, off, idx)
                    ref = O.match_candidates(q, t, lvl, off, idx)
                    if any(not np.array_equal(a, b) for a, b in zip(got, ref)):
                        fails.append(("candidates", nq, nt))
                    sizes = rng.integers(0, 97, int(rng.integers(1, 120)))
                    soff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
                    desc = unit(max(int(sizes.sum()), 1))[:int(sizes.sum())]
                    if len(desc) and not np.array_equal(eng.distinctive_descriptors(desc, soff), O.distinctive_descriptors(desc, soff)):
                        fails.append(("distinctive", len(sizes)))
                elif kind < 0.88:
                    n1, n2 = int(rng.integers(0, 1300)), int(rng.integers(0, 1300))
                    note("bow_tri", n1, n2)
                    a = unit(max(n1, 1))[:n1]
                    if n1 and n2 and rng.random() < 0.7:
                        b = a[rng.integers(0, n1, n2)] + float(rng.choice([1e-5, 1e-3, 0.02, 0.05])) * rng.standard_normal((n2, 256)).astype(np.float32)
                        b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
                    else:
                        b = unit(max(n2, 1))[:n2]
                    sc = float(rng.choice([1.0, 1.0, 0.2, 3.0]))
                    a, b = (a * sc).astype(np.float32), (b * sc).astype(np.float32)
                    n, m, d = eng.search_by_bow(a, b, 0.6 * sc)
                    rn, rm, rd = O.search_by_bow(a, b, 0.6 * sc)
                    if n != rn or not np.array_equal(m, rm) or not np.array_equal(d, rd):
                        fails.append(("bow", n1, n2, sc))
                    if sc == 1.0:
                        n, m = eng.search_for_triangulation(a, b, 0.75)
                        rn, rm = O.search_for_triangulation(a, b, 0.75)
                        if n != rn or not np.array_equal(m, rm):
                            fails.append(("tri", n1, n2))
                else:
                    n, dim = int(rng.integers(1, 1500)), int(rng.choice([4096, 4096, 4096, 1024, 256]))
                    note("db", n)
                    eng.set_option("db_screen_min_rows", int(rng.choice([0, 1, 6144])))      # (one query: exact scan / screened form)
                    rows = unit(n, dim)
                    db = capi.Database(eng, n + 5, dim)
                    for i in range(n):
                        db.add(i, rows[i])
                    holes = [int(i) for i in rng.choice(n, int(rng.integers(0, max(n // 8, 1))), replace=False)] if n > 4 and rng.random() < 0.5 else []
                    for i in holes:
                        db.erase(i)
                    keep = np.ones(n, bool); keep[holes] = False
                    nq = int(rng.choice([1, 3, 8, 20, 33, 64, 70, 130]))
                    qs = rows[rng.integers(0, n, nq)] + float(rng.choice([0.0, 0.003, 0.02])) * rng.standard_normal((nq, dim)).astype(np.float32)
                    qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
                    mode = int(rng.integers(0, 2))
                    if nq == 1:
                        slots, sc, best, allsc = db.query(qs[0], mode, want_scores=True)
                        ref = O.db_scores(qs[0], rows); ref[~keep] = -1         # (documented: -1 for empty slots)
                        ridx, rbest = O.db_candidates(ref, mode); ridx = ridx[keep[ridx]]
                        if not np.array_equal(allsc[:n], ref) or best != rbest or not np.array_equal(np.sort(slots), np.sort(ridx)):
                            fails.append(("db_q1", n, mode))
                    else:
                        res, best, allsc = db.query_batch(qs, mode, want_scores=True)
                        ref = np.stack([O.db_scores(q, rows) for q in qs]); ref[:, ~keep] = -1
                        # < 8 queries: the exact scan; >= 8: the integer screen + the exact chain for what it cannot rule out -- every score of
                        # every slot, best, candidates: the oracle's bits either way
                        ok = np.array_equal(allsc[:, :n], ref)
                        for i in range(nq):
                            ridx, rbest = O.db_candidates(ref[i], mode); ridx = ridx[keep[ridx]]
                            ok = ok and best[i] == rbest and np.array_equal(res[i][0], ridx) and np.array_equal(res[i][1], ref[i][ridx])
                        if not ok:
                            fails.append(("db_batch", n, nq, mode))
                    db.close()
            except Exception as e:                                        # noqa: BLE001
                fails.append(("exception", repr(e)[:200]))
        eng.close() if hasattr(eng, "close") else None
    return cases, fails


if __name__ == "__main__":
    cases, fails = run(float(sys.argv[1]) if len(sys.argv) > 1 else 600.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                       float(sys.argv[3]) if len(sys.argv) > 3 else 0.02)
    print(f"soak: {cases} cases, {len(fails)} failures")
    for f in fails[:20]:
        print("FAIL", f)
    sys.exit(1 if fails else 0)
