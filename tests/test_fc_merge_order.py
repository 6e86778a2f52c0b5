"""The blocked FC kernel (k_fc_mfma_tile, kernels_global.hip) merges the 16 range partials as a binary counter in registers instead of
storing them and adding them as fc_combine_one's balanced tree (oracle/hfnet_oracle.c global_head: p[q] = p[2q] + p[2q+1], four
times, then the bias).  fp32 addition is not associative, so "same bits" rests on the counter adding exactly the tree's pairs in the
tree's operand order: checked here on the CPU, in fp32, for random and badly scaled partials."""
import numpy as np


def tree16(p):
    p = [np.float32(v) for v in p]
    m = 16
    while m > 1:
        p = [np.float32(p[2 * q] + p[2 * q + 1]) for q in range(m // 2)]
        m //= 2
    return p[0]


def counter16(p):
    lvl = [None] * 4
    total = None
    for q in range(16):
        v = np.float32(p[q])
        placed = False
        for l in range(4):
            if placed:
                break
            if (q >> l) & 1:
                v = np.float32(lvl[l] + v)             # the earlier block first, as the tree writes it
            else:
                lvl[l] = v
                placed = True
        if not placed:
            total = v
    return total


def test_binary_counter_merge_adds_the_balanced_trees_pairs():
    rng = np.random.default_rng(5)
    for trial in range(2000):
        scale = np.float32(10.0) ** rng.integers(-6, 7, 16)
        p = (rng.standard_normal(16) * scale).astype(np.float32)
        if trial % 5 == 0:
            p[rng.integers(0, 16, 5)] = 0.0                     # empty ranges add zeros
        a, b = tree16(p), counter16(p)
        assert a.tobytes() == b.tobytes(), (trial, p, a, b)
