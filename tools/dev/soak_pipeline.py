"""Randomised soak of the device-resident pipeline (the bench's path): several extract_batch(on_device) + fence +
search_by_bow_batch(on_device) calls are enqueued back to back WITHOUT host synchronisation (random call sizes, ring of
result buffers, pairs across call boundaries), then everything is compared with the oracle.  Exercises the cross-stream
ordering: deferred global branch, matcher stream vs extraction stream, fences.   python tools/dev/soak_pipeline.py [seconds] [seed]"""
import ctypes as C, os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from hfnet_slam_amd import capi, weights
from oracle import oracle as O
from conftest import synth_image, torch_to_host as H, torch_to_device


def run(budget_s, seed):
    rng = np.random.default_rng(seed)
    wpath = os.path.join(tempfile.gettempdir(), f"hfnet_soakp_{seed}.hfw")
    weights.save(wpath, weights.synthetic_weights(200 + seed))
    model = O.Model(wpath)
    dev = torch.device("cuda:0")
    L = capi.lib()
    fails, rounds = [], 0
    t_end = time.time() + budget_s
    while time.time() < t_end:
        eng = capi.Engine(wpath, 0)
        if rng.random() < 0.5:
            eng.set_option("two_streams", int(rng.integers(0, 4)))
        w, h = int(rng.integers(64, 260)), int(rng.integers(64, 200))
        nl = int(rng.integers(1, 5)); nf = int(rng.integers(16, 500))
        while nl > 1 and min(w, h) / 1.2 ** (nl - 1) < 24:
            nl -= 1
        MB = int(rng.choice([2, 4, 8]))
        ext = capi.Extractor(eng, w, h, nf, 0.01, 1.2, nl, max_batch=MB)
        n_calls = int(rng.integers(2, 6))
        sizes = [int(rng.integers(1, 2 * MB + 1)) for _ in range(n_calls)]          # calls may span several chunks
        total = sum(sizes)
        imgs = np.stack([synth_image(h, w, int(rng.integers(1 << 30)), "natural" if rng.random() < 0.5 else "uniform") for _ in range(total)])
        d_imgs = torch_to_device(imgs, dev)
        kps = torch.zeros((total, nf, 4), dtype=torch.float32, device=dev)
        desc = torch.zeros((total, nf, 256), dtype=torch.float32, device=dev)
        glob = torch.zeros((total, eng.global_dim), dtype=torch.float32, device=dev)
        nrow = torch.zeros((total,), dtype=torch.int32, device=dev)
        match = torch.full((total, nf), -9, dtype=torch.int32, device=dev)
        mdist = torch.zeros((total, nf), dtype=torch.float32, device=dev)
        mcnt = torch.full((total,), -9, dtype=torch.int32, device=dev)
        # pair lists made (and finished: torch's stream is not ordered with the library's) before anything is enqueued
        pair_t, pair_q, f0 = [], [], 0
        for n in sizes:
            first = max(f0, 1)
            t = torch.arange(first, f0 + n, dtype=torch.int32, device=dev) if f0 + n > first else None
            pair_t.append(t); pair_q.append(None if t is None else t - 1)
            f0 += n
        torch.cuda.synchronize()
        f0 = 0
        for ci, n in enumerate(sizes):
            ext.extract_batch_device(n, d_imgs[f0].data_ptr(), w, w * h, kps[f0].data_ptr(), desc[f0].data_ptr(), glob[f0].data_ptr(), nrow[f0:].data_ptr())
            eng.fence()
            if os.environ.get("SOAK_SYNC"):
                eng.synchronize()
            first = max(f0, 1)
            if pair_t[ci] is not None:                                              # frame f against frame f - 1 (across call boundaries too)
                t, q = pair_t[ci], pair_q[ci]
                st = L.hfnet_match_search_by_bow_batch(eng.h, int(t.numel()), C.c_void_p(desc.data_ptr()), C.c_size_t(nf * 256), C.c_void_p(nrow.data_ptr()), total,
                                                       C.c_void_p(q.data_ptr()), C.c_void_p(t.data_ptr()), nf, 256, C.c_float(0.6),
                                                       C.c_void_p(match[first].data_ptr()), C.c_void_p(mdist[first].data_ptr()), C.c_void_p(mcnt[first:].data_ptr()), 1)
                if st != 0:
                    fails.append(("status", st, capi.last_error()))
            f0 += n
        eng.synchronize(); torch.cuda.synchronize()
        nr = H(nrow); K = H(kps); D = H(desc); G = H(glob)
        M = H(match); MD = H(mdist); MC = H(mcnt)
        refs = [model.extract(imgs[i], nf, 0.01, nl, 1.2) for i in range(total)]
        for i, (rn, rk, rd, rg, _) in enumerate(refs):
            kk = K[i, :rn].view(capi.KP_DTYPE).reshape(-1) if hasattr(capi, "KP_DTYPE") else None
            ok = nr[i] == rn and np.array_equal(D[i, :rn], rd) and np.array_equal(G[i], rg) and (kk is None or np.array_equal(kk, rk))
            if ok and i >= 1:
                qn, qd = refs[i - 1][0], refs[i - 1][2]
                cn, cm, cd = O.search_by_bow(qd, rd, 0.6)
                ok = MC[i] == cn and np.array_equal(M[i, :qn], cm) and np.array_equal(MD[i, :qn][cm >= 0], cd[cm >= 0])
            if not ok:
                why = []
                if nr[i] != rn: why.append(("n", int(nr[i]), rn))
                elif not np.array_equal(D[i, :rn], rd): why.append("desc")
                elif not np.array_equal(G[i], rg): why.append("global")
                elif kk is not None and not np.array_equal(kk, rk): why.append("kps")
                elif i >= 1:
                    why.append(("match", int(MC[i]), cn, int((M[i, :qn] != cm).sum()), qn))
                fails.append(("pipeline", w, h, nl, nf, MB, sizes, i, why))
                break
        rounds += 1
        ext.close(); eng.close()
    return rounds, fails


if __name__ == "__main__":
    rounds, fails = run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"pipeline soak: {rounds} rounds, {len(fails)} failures")
    for f in fails[:20]:
        print("FAIL", f)
    sys.exit(1 if fails else 0)
