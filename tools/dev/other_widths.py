"""engine vs oracle for networks of other widths (depth multipliers 1.0 / 0.5 / 0.35, other cluster counts / global
dims): the shapes the specialised kernels do not cover must go through the generic ones with the same bits.
    python tools/dev/other_widths.py"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
try:
    import torch
    if torch.cuda.is_available(): torch.cuda.init()
except Exception:
    pass
from hfnet_slam_amd import capi, spec, weights
from oracle import oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from conftest import synth_image

O.build()
bad = 0
for mult, ncl, gdim in [(1.0, 32, 4096), (0.5, 16, 256), (0.35, 8, 64), (0.75, 64, 1024), (1.4, 32, 512)]:
    sp = spec.net_spec(mult, ncl, gdim)
    d = tempfile.mkdtemp()
    p = os.path.join(d, "w.hfw")
    weights.save(p, weights.synthetic_weights(11, sp))
    m = O.Model(p)
    e = capi.Engine(p, 0)
    for (w, h, nl, nf) in [(248, 168, 3, 300), (131, 121, 2, 150)]:
        x = capi.Extractor(e, w, h, nf, 0.01, 1.2, nl, max_batch=2)
        imgs = np.stack([synth_image(h, w, 51, "natural"), synth_image(h, w, 52)])
        nb, kb, db, gb = x.extract_batch(imgs)
        for f in range(2):
            rn, rk, rd, rg, _ = m.extract(imgs[f], nf, 0.01, nl, 1.2)
            ok = nb[f] == rn and np.array_equal(kb[f, :rn], rk) and np.array_equal(db[f, :rn].view(np.uint32), rd.view(np.uint32)) \
                and np.array_equal(gb[f].view(np.uint32), rg.view(np.uint32))
            if not ok:
                bad += 1
                print("MISMATCH", mult, ncl, gdim, w, h, nl, f, nb[f], rn,
                      "kp", nb[f] == rn and np.array_equal(kb[f, :rn], rk),
                      "desc", nb[f] == rn and np.array_equal(db[f, :rn].view(np.uint32), rd.view(np.uint32)),
                      "glob", np.array_equal(gb[f].view(np.uint32), rg.view(np.uint32)), flush=True)
        x.close()
    e.close()
    print("width", mult, "clusters", ncl, "global", gdim, "done", flush=True)
print("other widths:", "0 failures" if bad == 0 else f"{bad} FAILURES")
sys.exit(1 if bad else 0)
