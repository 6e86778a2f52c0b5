"""k_block_fused8 keeps the expanded halo tile in registers; which halo position a lane half holds in which accumulator slot, and what the
v_permlane32_swap exchange hands over, is index arithmetic that the GPU parity tests only check end to end.  This restates that arithmetic
(kernels_block.hip: the comment above k_block_fused8 / F8Geo) in Python and checks it position by position: every halo position is
computed exactly once, and every tap of every depthwise output reads the register that holds the right position."""
import itertools

import pytest


def slot_position(stride, h, s):
    """halo (hy, hx) that lane half h holds in accumulator slot s = 16 m + i, or None for a padding slot"""
    if stride == 2:                                   # 9 x 17 halo, five M tiles: 80 slots per half
        if s < 72:
            return (s >> 3, 8 * h + (s & 7))
        return ((s - 72) if h else 8, 16) if (h or s == 72) else None
    if s >= 30:                                       # 6 x 10 halo, two M tiles: 32 slots per half, 30 used
        return None
    hy, k = divmod(s, 5)
    return (hy, (9 if k == 0 else 4 + k) if h else k)


def swp(x, y, which):
    """__builtin_amdgcn_permlane32_swap(x, y)[which] as (lower half, upper half): [0] = (x.lower, y.lower), [1] = (x.upper, y.upper)"""
    return (x[0], y[0]) if which == 0 else (x[1], y[1])


@pytest.mark.parametrize("stride", [1, 2])
def test_every_halo_position_is_computed_exactly_once(stride):
    ih, iw, slots = (9, 17, 80) if stride == 2 else (6, 10, 32)
    seen = {}
    for h, s in itertools.product((0, 1), range(slots)):
        p = slot_position(stride, h, s)
        if p is not None:
            assert p not in seen, (p, seen[p], (h, s))
            seen[p] = (h, s)
    assert set(seen) == set(itertools.product(range(ih), range(iw)))
    # the A row that feeds slot (m, i) of half h is rho = 8 (i >> 2) + 4 h + (i & 3): a bijection onto the 32 rows of the M tile
    for m in range(slots // 16):
        rows = sorted(8 * (i >> 2) + 4 * h + (i & 3) for h in (0, 1) for i in range(16))
        assert rows == list(range(32))
        for r in range(32):                           # ... and the kernel's inverse (hr, ir) of a row
            hr, ir = (r >> 2) & 1, ((r >> 3) << 2) | (r & 3)
            assert 8 * (ir >> 2) + 4 * hr + (ir & 3) == r


@pytest.mark.parametrize("stride", [1, 2])
def test_every_depthwise_tap_reads_the_right_position(stride):
    ih, nc = (9, 8) if stride == 2 else (6, 5)
    # a register = (value in the lower half, value in the upper half); values are the halo positions themselves
    acc = lambda s: (slot_position(stride, 0, s), slot_position(stride, 1, s))
    E, E0 = {}, {}
    if stride == 2:
        lo72 = swp(acc(72), acc(72), 0)
        E[8] = swp(acc(64), lo72, 1)
        for hy in range(8):
            E[hy] = swp(acc(hy * 8), acc(72 + hy), 1)
    else:
        for hy in range(ih):
            a0, a1, a4 = acc(hy * nc), acc(hy * nc + 1), acc(hy * nc + 4)
            E[hy] = swp(a1, a0, 1)
            E0[hy] = swp(a0, a4, 0)
    for h, oy, ox, ky, kx in itertools.product((0, 1), range(4), range(4), range(3), range(3)):
        hy, c = stride * oy + ky, stride * ox + kx
        if stride == 2:
            reg = acc(hy * nc + c) if c < nc else E[hy]
        else:
            reg = E0[hy] if c == 0 else acc(hy * nc + c) if c < nc else E[hy]
        want = (stride * oy + ky, stride * (4 * h + ox) + kx)         # output column 4 h + ox of the 4 x 8 tile
        assert reg[h] == want, (stride, h, oy, ox, ky, kx, reg[h], want)
