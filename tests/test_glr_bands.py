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
