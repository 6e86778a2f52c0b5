"""The split-bf16 scheme of the tolerance options (include/hfnet_hip.h; kernels_conv.hip `split2`): x = hi + lo + e with hi = bf16(x), lo = bf16(x - hi),
both round-to-nearest-even, and a . w ~ ah.wh + ah.wl + al.wh.  A numpy emulation of the two roundings checks the bounds the header states:
|e| <= 2^-16 |x| (in fact 2^-17), the three-product dot product within 3 * 2^-16 sum |a_k||w_k| of the exact one, and that (x - 128) / 128 of a u8 pixel and
the constant one are exact in ONE piece (the bias row / stem input arguments of NOTEBOOK.md R6)."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split(x):
    x = np.asarray(x, np.float32)
    hi = bf16_rne(x)
    lo = bf16_rne((x - hi).astype(np.float32))             # (x - hi is exact in fp32)
    return hi, lo


def test_two_pieces_leave_2_to_the_minus_17():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 6, 200000).astype(np.float32),
                        np.float32([0.0, 1.0, -1.0, 6.0, 1.0 / 65, 0.0078125, 255.0 / 128 - 1, 1e-30, -3.0e38])])
    hi, lo = split(x)
    e = np.abs(x.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
    nz = np.abs(x) > 1e-30
    assert np.all(e[nz] <= 2.0 ** -17 * np.abs(x[nz]).astype(np.float64) * (1 + 1e-12))
    assert np.all(np.abs(lo[nz]) <= 2.0 ** -8 * np.abs(x[nz]) * (1 + 1e-6))
    assert np.all(split(np.float32([0.0]))[0] == 0) and np.all(split(np.float32([0.0]))[1] == 0)


def test_three_products_within_the_stated_bound():
    rng = np.random.default_rng(2)
    for K in (16, 96, 864, 4096):
        a = rng.standard_normal((64, K)).astype(np.float32) * 3
        w = (rng.standard_normal((K, 32)) * np.sqrt(2.0 / K)).astype(np.float32)
        ah, al = split(a); wh, wl = split(w)
        s3 = ah.astype(np.float64) @ wh.astype(np.float64) + ah.astype(np.float64) @ wl.astype(np.float64) + al.astype(np.float64) @ wh.astype(np.float64)
        exact = a.astype(np.float64) @ w.astype(np.float64)
        bound = 3 * 2.0 ** -16 * (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64))
        assert np.all(np.abs(s3 - exact) <= bound), K
        # ... and the measured size of it: rounding residues of random sign, an order of magnitude (long sums: two) below the bound
        assert np.abs(s3 - exact).max() <= (0.25 if K <= 96 else 0.05) * bound.max(), K


def test_values_that_are_exact_in_one_piece():
    px = np.arange(256, dtype=np.float32)
    v = (px * np.float32(0.0078125) - np.float32(1.0)).astype(np.float32)      # (x - 128) / 128 as the stem computes it
    assert np.array_equal(bf16_rne(v), v)
    assert bf16_rne(np.float32([1.0]))[0] == 1.0
