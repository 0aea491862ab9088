"""Host logic of the chained greedy PCA -> GLR call (pipeline.glr_bands_for): which row bands of
the GLR may start while some areas still iterate (CPU only)."""
import numpy as np
import pytest

from origin_amd import pipeline


def check_partition(early, late, Ny):
    bands = sorted(early + late)
    assert bands[0][0] == 0 and bands[-1][1] == Ny
    for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
        assert a1 == b0                                   # no gap, no overlap
    for y0, y1 in bands:
        assert y0 % 64 == 0 and (y1 % 64 == 0 or y1 == Ny) and y0 < y1


def test_bands_of_the_bench_field():
    # one straggler area on rows 200..299 of a 600-row field, 25 x 25 PSF (halo 12)
    early, late = pipeline.glr_bands_for([(200, 299)], 600, 12)
    assert early == [(0, 128), (320, 600)] and late == [(128, 320)]
    check_partition(early, late, 600)
    # a second one on rows 400..499
    early, late = pipeline.glr_bands_for([(200, 299), (400, 499)], 600, 12)
    assert early == [(0, 128), (320, 384), (512, 600)] and late == [(128, 320), (384, 512)]


def test_budget_takes_the_largest_bands_first_and_merges_the_rest():
    early, late = pipeline.glr_bands_for([(200, 299)], 900, 12, max_early_rows=256)
    assert early == [(320, 576)] and late == [(0, 320), (576, 900)]
    check_partition(early, late, 900)
    # a budget below one block still gives one block; a huge one changes nothing
    early, late = pipeline.glr_bands_for([(200, 299)], 900, 12, max_early_rows=10)
    assert early == [(320, 384)]
    check_partition(early, late, 900)
    assert (pipeline.glr_bands_for([(200, 299)], 900, 12, max_early_rows=10 ** 9) ==
            pipeline.glr_bands_for([(200, 299)], 900, 12))


@pytest.mark.parametrize("seed", range(20))
def test_early_bands_never_read_a_row_of_an_active_area(seed):
    rng = np.random.default_rng(seed)
    Ny = int(rng.integers(1, 1200))
    halo = int(rng.integers(0, 13))
    active = []
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, Ny))
        active.append((a, min(Ny - 1, a + int(rng.integers(0, 300)))))
    budget = None if seed % 3 == 0 else int(rng.integers(1, 1000))
    early, late = pipeline.glr_bands_for(active, Ny, halo, budget)
    check_partition(early, late, Ny)
    for y0, y1 in early:
        for ymin, ymax in active:                         # the band's spatial stage reads
            assert y1 + halo <= ymin or y0 - halo > ymax  # rows [y0 - halo, y1 + halo)
    if budget is not None:
        assert sum(y1 - y0 for y0, y1 in early) <= max(64, budget // 64 * 64)
    if not active:
        assert late == [] or budget is not None


# ---- tiles: which 64 x 64 regions of a halo-extended tile need no halo data (multigpu.py) ----
def test_interior_regions_of_the_bench_tilings():
    from origin_amd import multigpu as m
    # two ranks at 600^2: tiles 600 x 300, one halo column strip of 12
    ok = m.interior_regions(600, 312, (0, 0, 0, 12), 12)
    assert m.region_rects(ok, 600, 312) == [(0, 600, 0, 256)]
    assert m.region_rects(~ok, 600, 312) == [(0, 600, 256, 312)]
    # four ranks: 300 x 300 tiles, halos on two sides
    ok = m.interior_regions(312, 312, (12, 0, 12, 0), 12)
    assert m.region_rects(ok, 312, 312) == [(64, 312, 64, 312)]
    assert m.region_rects(~ok, 312, 312) == [(0, 64, 0, 312), (64, 312, 0, 64)]
    # eight ranks: 100-wide tiles have no halo-free region
    ok = m.interior_regions(524, 124, (12, 12, 12, 12), 12)
    assert not ok.any() and m.region_rects(~ok, 524, 124) == [(0, 524, 0, 124)]
    # no neighbours at all: everything is interior
    ok = m.interior_regions(600, 600, (0, 0, 0, 0), 12)
    assert ok.all()


@pytest.mark.parametrize("seed", range(12))
def test_region_rects_cover_the_mask_exactly(seed):
    from origin_amd import multigpu as m
    rng = np.random.default_rng(seed)
    Ny, Nx = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    nry, nrx = (Ny + 63) // 64, (Nx + 63) // 64
    ok = rng.random((nry, nrx)) < 0.6
    cover = np.zeros((Ny, Nx), int)
    for y0, y1, x0, x1 in m.region_rects(ok, Ny, Nx):
        assert y0 % 64 == 0 and x0 % 64 == 0 and (y1 % 64 == 0 or y1 == Ny)
        assert (x1 % 64 == 0 or x1 == Nx) and y0 < y1 and x0 < x1
        cover[y0:y1, x0:x1] += 1
    want = np.kron(ok.astype(int), np.ones((64, 64), int))[:Ny, :Nx]
    assert np.array_equal(cover, want)                   # every True cell once, no False cell


@pytest.mark.parametrize("seed", range(12))
def test_interior_regions_read_no_halo_row_or_column(seed):
    from origin_amd import multigpu as m
    rng = np.random.default_rng(100 + seed)
    Ny, Nx = int(rng.integers(30, 700)), int(rng.integers(30, 700))
    reach = int(rng.integers(0, 13))
    halos = tuple(int(h) for h in rng.choice([0, 12, 13], 4))
    ok = m.interior_regions(Ny, Nx, halos, reach)
    top, bot, left, right = halos
    for ry in range(ok.shape[0]):
        for rx in range(ok.shape[1]):
            a, b = 64 * ry - reach, min(Ny, 64 * ry + 64) + reach
            c, d = 64 * rx - reach, min(Nx, 64 * rx + 64) + reach
            touches = ((top and a < top) or (bot and b > Ny - bot) or
                       (left and c < left) or (right and d > Nx - right))
            assert ok[ry, rx] == (not touches)
