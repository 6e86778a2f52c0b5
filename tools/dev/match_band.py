"""How wide must SearchByBoW's rounding band be for the split-bf16 screening (kernels_match.hip, launch_bow_pairs)?
A numpy emulation of the arithmetic -- operands split into hi + lo bf16 (round to nearest even), the three products that are
kept, fp32 accumulation (numpy's order here, the matrix unit's on the GPU: the bound in the kernel comment holds for any order)
-- against the distances in float64, relative to |q|^2 + |t|^2.  CPU only:  python tools/dev/match_band.py [n=1000] [dim=256]"""
import sys
import numpy as np


def bf16_rne(x):
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    b = ((b + 0x7fff + ((b >> 16) & 1)) >> 16) << 16
    return b.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_rne(x)
    lo = bf16_rne((x - hi).astype(np.float32))
    return hi, lo


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(3)
worst = 0.0
for scale, spread in ((1.0, None), (1.0, 1e-3), (7.5, 1e-4), (0.05, 1e-4), (30.0, 1e-3)):
    if spread is None:
        q = rng.standard_normal((n, dim)); t = rng.standard_normal((n, dim))
    else:
        c = rng.standard_normal((12, dim)); c /= np.linalg.norm(c, axis=1, keepdims=True)
        q = np.repeat(c, n // 12 + 1, axis=0)[:n] + spread * rng.standard_normal((n, dim))
        t = np.repeat(c, n // 12 + 1, axis=0)[:n] + spread * rng.standard_normal((n, dim))
    q = (scale * q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    t = (scale * t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float32)
    qh, ql = split(q); th, tl = split(t)
    s = (qh @ th.T + qh @ tl.T + ql @ th.T).astype(np.float32)
    qn = np.einsum("ij,ij->i", q, q).astype(np.float32); tn = np.einsum("ij,ij->i", t, t).astype(np.float32)
    g = (qn[:, None] + tn[None, :]).astype(np.float32) - 2.0 * s
    d2 = ((q.astype(np.float64)[:, None, :] - t.astype(np.float64)[None, :, :]) ** 2).sum(-1) if n <= 400 else \
        (q.astype(np.float64) ** 2).sum(1)[:, None] + (t.astype(np.float64) ** 2).sum(1)[None, :] - 2.0 * q.astype(np.float64) @ t.astype(np.float64).T
    rel = np.abs(g - d2) / (qn[:, None].astype(np.float64) + tn[None, :])
    worst = max(worst, rel.max())
    print(f"scale {scale:5.2f} spread {spread}: max |G - d^2| / (|q|^2 + |t|^2) = {rel.max():.3e}   (band {1.25e-6 * dim:.3e})")
print(f"worst {worst:.3e} = band / {1.25e-6 * dim / worst:.0f}")
