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

budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
wpath = os.path.join(tempfile.gettempdir(), f"hfnet_soak_{seed}.hfw")
weights.save(wpath, weights.synthetic_weights(100 + seed))
model = O.Model(wpath)
fails, cases = [], 0
t_end = time.time() + budget_s


def unit(n, d=256):
    a = rng.standard_normal((n, d)).astype(np.float32)
    return (a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)).astype(np.float32)


while time.time() < t_end:
    eng = capi.Engine(wpath, 0)
    opts = {"fuse_blocks": int(rng.integers(0, 2)), "fused_variant": int(rng.choice([2, 4])), "fuse_stem": int(rng.integers(0, 2)),
            "dense_desc": int(rng.integers(0, 2)), "two_streams": int(rng.integers(0, 4)), "conv_wlds": int(rng.integers(0, 2))}
    if rng.random() < 0.5:
        opts = {}
    for k, v in opts.items():
        eng.set_option(k, v)
    for _ in range(6):
        if time.time() >= t_end:
            break
        kind = rng.random()
        cases += 1
        try:
            if kind < 0.55:
                w, h = int(rng.integers(40, 420)), int(rng.integers(40, 340))
                nl = int(rng.integers(1, 6)); nf = int(rng.integers(8, 1500)); thr = float(rng.choice([0.0, 0.002, 0.01, 0.02]))
                sf = float(rng.choice([1.2, 1.1, 1.5]))
                while nl > 1 and min(w, h) / sf ** (nl - 1) < 24:
                    nl -= 1
                B = int(rng.choice([1, 1, 2, 3, 5, 12]))
                x = capi.Extractor(eng, w, h, nf, thr, sf, nl, max_batch=int(rng.choice([1, 2, 4, 16])))
                imgs = np.stack([synth_image(h, w, int(rng.integers(1 << 30)), "natural" if rng.random() < 0.5 else "uniform") for _ in range(B)])
                nb, kb, db, gb = x.extract_batch(imgs)
                for i in range(B):
                    rn, rk, rd, rg, _ = model.extract(imgs[i], nf, thr, nl, sf)
                    ok = nb[i] == rn and np.array_equal(kb[i, :rn], rk) and np.array_equal(db[i, :rn], rd) and np.array_equal(gb[i], rg)
                    if not ok:
                        fails.append(("extract", w, h, nl, nf, thr, sf, B, i, opts))
                        break
                x.close()
            elif kind < 0.8:
                n1, n2 = int(rng.integers(0, 1300)), int(rng.integers(0, 1300))
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
                n, dim = int(rng.integers(1, 1500)), 4096
                rows = unit(n, dim)
                db = capi.Database(eng, n + 5, dim)
                for i in range(n):
                    db.add(i, rows[i])
                nq = int(rng.choice([1, 3, 8, 20, 64]))
                qs = rows[rng.integers(0, n, nq)] + float(rng.choice([0.0, 0.003, 0.02])) * rng.standard_normal((nq, dim)).astype(np.float32)
                qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
                mode = int(rng.integers(0, 2))
                if nq == 1:
                    slots, sc, best, allsc = db.query(qs[0], mode, want_scores=True)
                    ref = O.db_scores(qs[0], rows)
                    ridx, rbest = O.db_candidates(ref, mode)
                    if not np.array_equal(allsc[:n], ref) or best != rbest or not np.array_equal(np.sort(slots), np.sort(ridx)):
                        fails.append(("db_q1", n, mode))
                else:
                    res, best, allsc = db.query_batch(qs, mode, want_scores=True)
                    ref = O.db_scores_gemm(qs, rows) if nq >= 8 else np.stack([O.db_scores(q, rows) for q in qs])
                    if not np.array_equal(allsc[:, :n], ref):
                        fails.append(("db_batch", n, nq, mode))
                db.close()
        except Exception as e:                                        # noqa: BLE001
            fails.append(("exception", repr(e)[:200]))
    eng.close() if hasattr(eng, "close") else None
print(f"soak: {cases} cases, {len(fails)} failures (seed {seed})")
for f in fails[:20]:
    print("FAIL", f)
sys.exit(1 if fails else 0)
