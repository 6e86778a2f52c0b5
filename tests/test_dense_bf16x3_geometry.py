"""k_conv3x3_dense_bf16x3 (hfnet_slam_amd/csrc/kernels_conv.hip) stages the halo of a tile of 256 consecutive pixels through LDS in PADDED LINEAR
coordinates q = (y + 1) (W + 2) + (x + 1): a tap is then one uniform offset for every lane and the zero border is written by the staging pass.  This
restates the kernel's index arithmetic on the CPU and checks, for the shipped pyramids and for random map sizes, that
  * the staged range [qbase, qbase + ncell) holds every cell any tap of any pixel of the tile reads, and never more cells than the LDS block has;
  * a lane's window centre + tap offset lands on the staged cell of exactly the neighbour (y + ky - 1, x + kx - 1), or on a border cell (zeros);
  * the staging pass's (piece -> cell -> image pixel | border) map covers every staged cell once with the right source pixel;
  * the launcher's bound (conv3x3_dense_bf16x3_supported) is the worst tile's cell count."""
import numpy as np
import pytest

CELLS = 512      # k_conv3x3_dense_bf16x3<512>
TILE = 256


def tile_geometry(H, W, tile):
    Wp, nrows = W + 2, H * W
    p0 = tile * TILE
    y0, x0 = divmod(p0, W)
    plast = min(p0 + TILE - 1, nrows - 1)
    yl, xl = divmod(plast, W)
    qbase = y0 * Wp + x0
    ncell = (yl + 2) * Wp + xl + 2 - qbase + 1
    return Wp, nrows, p0, plast, qbase, ncell


def supported_bound(W):
    span = (TILE - 1 + W - 1) // W
    return TILE - 1 + 2 * span + 2 * (W + 2) + 3


def check_map(H, W):
    Wp, nrows, *_ = tile_geometry(H, W, 0)
    worst = 0
    for tile in range((nrows + TILE - 1) // TILE):
        Wp, nrows, p0, plast, qbase, ncell = tile_geometry(H, W, tile)
        worst = max(worst, ncell)
        assert 0 < ncell <= supported_bound(W), (H, W, tile, ncell)
        # staging: cell j <-> padded index q = qbase + j <-> image pixel (yq - 1, xq - 1) or border
        j = np.arange(ncell)
        q = qbase + j
        yq, xq = q // Wp, q % Wp
        sy, sx = yq - 1, xq - 1
        inside = (sy >= 0) & (sy < H) & (sx >= 0) & (sx < W)
        # pixels of the tile: window centre in staged coordinates
        p = np.arange(p0, plast + 1)
        y, x = p // W, p % W
        jc = (y + 1) * Wp + (x + 1) - qbase
        for ky in range(3):
            for kx in range(3):
                jt = jc + (ky - 1) * Wp + (kx - 1)
                assert jt.min() >= 0 and jt.max() < ncell, (H, W, tile, ky, kx)
                ny, nx = y + ky - 1, x + kx - 1
                nb_in = (ny >= 0) & (ny < H) & (nx >= 0) & (nx < W)
                # the staged cell is the neighbour where it exists, a border cell (zeros) where it does not
                assert np.array_equal(inside[jt], nb_in), (H, W, tile, ky, kx)
                assert np.array_equal(sy[jt][nb_in], ny[nb_in]) and np.array_equal(sx[jt][nb_in], nx[nb_in]), (H, W, tile, ky, kx)
    return worst


@pytest.mark.parametrize("size", [(752, 480), (512, 512)])
def test_shipped_pyramids(size):
    from hfnet_slam_amd import spec
    for (w, h) in spec.level_sizes(size[0], size[1], 4, 1.2):
        H, W = h // 8, w // 8                                   # the detector head's cell grid of a level
        worst = check_map(H, W)
        assert worst <= CELLS, (w, h, worst)
        assert supported_bound(W) <= CELLS


def test_random_maps_and_the_launcher_bound():
    rng = np.random.default_rng(5)
    for _ in range(60):
        H, W = int(rng.integers(1, 90)), int(rng.integers(1, 123))
        worst = check_map(H, W)
        assert worst <= supported_bound(W)
    # the bound is tight somewhere: a map whose tiles start at the last column of a row
    assert any(check_map(40, W) == supported_bound(W) for W in range(20, 120))
    # maps too wide for the LDS block are refused by the launcher's test (the engine then takes the exact kernel)
    assert supported_bound(122) <= CELLS < supported_bound(124)
