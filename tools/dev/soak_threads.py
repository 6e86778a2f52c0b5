"""Concurrent use of ONE engine from several host threads, the way the SLAM system's threads would (Tracking: extractor +
frame-to-frame match, LocalMapping: triangulation matches on the keyframe store, LoopClosing: database add / query):
every result is compared with the oracle.   python tools/dev/soak_threads.py [seconds] [seed]"""
import os, sys, time, tempfile, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hfnet_slam_amd import capi, weights
from oracle import oracle as O
from conftest import synth_image


def run(budget_s, seed):
    wpath = os.path.join(tempfile.gettempdir(), f"hfnet_soakt_{seed}.hfw")
    weights.save(wpath, weights.synthetic_weights(300 + seed))
    model = O.Model(wpath)
    eng = capi.Engine(wpath, 0)
    olock = threading.Lock()                                  # the oracle runs one call at a time (it is the checker, not the subject)
    fails, counts = [], {"tracking": 0, "mapping": 0, "loop": 0}
    t_end = time.time() + budget_s

    def unit(rng, n, d=256):
        a = rng.standard_normal((n, d)).astype(np.float32)
        return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)

    def tracking():
        rng = np.random.default_rng(seed * 10 + 1)
        w, h, nf, nl = 200, 152, 300, 3
        x = capi.Extractor(eng, w, h, nf, 0.01, 1.2, nl, max_batch=1)
        prev = None
        while time.time() < t_end:
            img = synth_image(h, w, int(rng.integers(1 << 30)), "natural")
            n, k, d, g, npl = x.extract(img)
            m = eng.search_by_bow(prev, d, 0.6) if prev is not None and len(prev) else None
            with olock:
                rn, rk, rd, rg, _ = model.extract(img, nf, 0.01, nl, 1.2)
                ok = n == rn and np.array_equal(k, rk) and np.array_equal(d, rd) and np.array_equal(g, rg)
                if ok and m is not None:
                    cn, cm, cd = O.search_by_bow(prev, rd, 0.6)
                    ok = m[0] == cn and np.array_equal(m[1], cm) and np.array_equal(m[2], cd)
            if not ok:
                fails.append(("tracking", counts["tracking"]))
            prev = d
            counts["tracking"] += 1
        x.close()

    def mapping():
        rng = np.random.default_rng(seed * 10 + 2)
        mr, S = 250, 8
        store = capi.Store(eng, S, mr)
        sets = [None] * S
        base = unit(rng, mr)
        it = 0
        while time.time() < t_end:
            s_ = it % S
            n = int(rng.integers(1, mr + 1))
            v = base[rng.permutation(mr)][:n] + 0.04 * rng.standard_normal((n, 256)).astype(np.float32)
            v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
            store.put(s_, v); sets[s_] = v
            others = [j for j in range(S) if sets[j] is not None and j != s_]
            if others:
                pairs = [(s_, j) for j in others]
                cnt, mt = store.search_for_triangulation(pairs, 0.75)
                with olock:
                    for p, (a, b) in enumerate(pairs):
                        tn, tm = O.search_for_triangulation(sets[a], sets[b], 0.75)
                        if cnt[p] != tn or not np.array_equal(mt[p, :len(sets[a])], tm):
                            fails.append(("mapping", it, p))
                            break
            it += 1
            counts["mapping"] += 1
        store.close()

    def loop_closing():
        rng = np.random.default_rng(seed * 10 + 3)
        cap, dim = 600, 4096
        db = capi.Database(eng, cap, dim)
        rows = np.zeros((cap, dim), np.float32); occ = np.zeros(cap, bool)
        it = 0
        while time.time() < t_end:
            slot = int(rng.integers(0, cap))
            if occ[slot] and rng.random() < 0.3:
                db.erase(slot); occ[slot] = False
            else:
                r = unit(rng, 1, dim)[0]
                db.add(slot, r); rows[slot] = r; occ[slot] = True
            if it % 4 == 3 and occ.any():
                q = rows[rng.choice(np.flatnonzero(occ))] + 0.01 * rng.standard_normal(dim).astype(np.float32)
                q = (q / np.linalg.norm(q)).astype(np.float32)
                slots, sc, best, allsc = db.query(q, int(rng.integers(0, 2)), want_scores=True)
                with olock:
                    ref = O.db_scores(q, rows); ref[~occ] = -1
                if not np.array_equal(allsc, ref):
                    fails.append(("loop", it, int((allsc != ref).sum())))
            it += 1
            counts["loop"] += 1
        db.close()

    th = [threading.Thread(target=f) for f in (tracking, mapping, loop_closing)]
    for t in th:
        t.start()
    for t in th:
        t.join(budget_s + 120)
    hung = [t.name for t in th if t.is_alive()]
    if hung:
        fails.append(("hung", hung))
    else:
        eng.close()
    return counts, fails


if __name__ == "__main__":
    counts, fails = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"thread soak: {counts}, {len(fails)} failures")
    for f in fails[:20]:
        print("FAIL", f)
    sys.exit(1 if fails else 0)
