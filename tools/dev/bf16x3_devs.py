# deviation of the split-bf16 options from the oracle over network widths, global dimensions and image sizes (needs the GPU):
#   OPT_D=1 OPT_G=1 python tools/dev/bf16x3_devs.py
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import synth_image
from hfnet_slam_amd import capi, spec, weights
from oracle import oracle as O
import tempfile, os
OD, OG = int(os.environ.get("OPT_D", 1)), int(os.environ.get("OPT_G", 1))
d = tempfile.mkdtemp()
for mult, gd in [(0.75, 4096), (0.75, 1024), (0.75, 4096)]:
    p = os.path.join(d, "w.hfw")
    weights.save(p, weights.synthetic_weights(13, spec.net_spec(mult, 32, gd)))
    e, m = capi.Engine(p, 0), O.Model(p)
    for fm in (0, None):
        e.set_option("desc_bf16x3", OD); e.set_option("global_bf16x3", OG)
        if fm is not None: e.set_option("fuse_min_wgs", fm)
        for (w, h, nl, nf) in [(200, 152, 4, 500), (131, 121, 2, 150), (248, 168, 3, 300), (376, 240, 2, 400), (752, 480, 4, 1000)]:
            x = capi.Extractor(e, w, h, nf, 0.01, 1.2, nl, max_batch=3)
            imgs = np.stack([synth_image(h, w, 71, "natural"), synth_image(h, w, 72), synth_image(h, w, 73, "natural")])
            nb, kb, db, gb = x.extract_batch(imgs)
            wd = wg = 0; ok = True; l2 = 0
            for f in range(3):
                rn, rk, rd, rg, _ = m.extract(imgs[f], nf, 0.01, nl, 1.2)
                ok &= nb[f] == rn and np.array_equal(kb[f, :rn], rk)
                wd = max(wd, np.abs(db[f, :rn].astype(np.float64) - rd).max()); wg = max(wg, np.abs(gb[f].astype(np.float64) - rg).max())
                l2 = max(l2, np.linalg.norm(gb[f].astype(np.float64) - rg))
            print(f"mult {mult} gd {gd} fm {fm} {w}x{h}: kp_ok {ok} desc {wd:.2e} global {wg:.2e} (x sqrt(D/4096) {wg*np.sqrt(gd/4096):.2e}) l2 {l2:.2e}", flush=True)
            x.close()
    e.close()
