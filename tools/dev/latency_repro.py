# one extractor, many calls on two alternating sets of frames, every frame of every call against the oracle (needs the GPU):
#   python tools/dev/latency_repro.py B option=value ...      e.g.  4 global_bf16x3=1 fuse_min_wgs=0 join_fused_branch=0   (NOTEBOOK.md R4.8)
import sys, numpy as np, os, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import synth_image
from hfnet_slam_amd import capi, spec, weights
from oracle import oracle as O
d = tempfile.mkdtemp(); p = os.path.join(d, "w.hfw")
weights.save(p, weights.synthetic_weights(13, spec.net_spec(0.75, 32, 4096)))
e, m = capi.Engine(p, 0), O.Model(p)
B = int(sys.argv[1])
for o in sys.argv[2:]:
    k, v = o.split("="); e.set_option(k, int(v))
W, H = 752, 480
sets = [np.stack([synth_image(H, W, 71 + 10 * s + i, "natural") for i in range(B)]) for s in range(2)]
refs = [[m.extract(im[i], 1000, 0.01, 4, 1.2) for i in range(B)] for im in sets]
x = capi.Extractor(e, W, H, 1000, 0.01, 1.2, 4, max_batch=B)
bad_calls = []
for call in range(int(os.environ.get("CALLS", 40))):
    s = call & 1
    nb, kb, db, gb = x.extract_batch(sets[s])
    wrong = sum(int((np.abs(db[f, :nb[f]] - refs[s][f][2]).max(axis=1) > 1e-4).sum()) for f in range(B))
    kok = all(nb[f] == refs[s][f][0] and np.array_equal(kb[f, :nb[f]], refs[s][f][1]) for f in range(B))
    if wrong or not kok: bad_calls.append((call, wrong, kok))
print("calls with wrong descriptor rows (call, rows, keypoints ok):", bad_calls)
x.close()
