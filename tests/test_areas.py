"""Area construction (reference steps.py:492-569, lib_origin.py:367-765).

Host-side geometry: the tests run without a GPU.  Golden G10 holds the outputs of the
reference's own functions (oracle/gen_golden.py g10), stage by stage, on two fields: one with
sources in most squares, one with few sources (areas without a source are dropped and the
others grow over them); plus the two merge criteria and one filled hull in isolation."""
import os

import numpy as np
import pytest

from oracle import golden_cases as gc
from origin_amd import areas

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_areas.npz"))
INP = gc.g10_inputs()


def planes(stack):
    return (np.arange(1, len(stack) + 1)[:, None, None] * (stack > 0)).sum(axis=0)


@pytest.mark.parametrize("name", ["many", "few"])
def test_stages_match_the_reference(name):
    c = INP[name]
    mask, segmap, minsize = c["mask"], c["segmap"], c["minsize"]
    maxsize = c["maxsize"] if c["maxsize"] is not None else 2 * minsize
    Ny, Nx = segmap.shape
    nexpmap = ((~mask).sum(axis=0) > 0).astype(int)
    nsub = int(G[f"{name}_nsub"])
    assert nsub == max(1, int(np.sqrt(nexpmap.sum() / minsize ** 2)))
    sq = areas.area_segmentation_square_fusion(nexpmap, minsize ** 2, maxsize ** 2, nsub, Ny, Nx)
    assert np.array_equal(planes(sq), G[f"{name}_squares"])
    assert np.array_equal(sq.sum(axis=(1, 2)), G[f"{name}_square_sizes"])
    ws, src = areas.area_segmentation_sources_fusion(segmap, sq.copy(), c["pfa"], Ny, Nx)
    assert np.array_equal(planes(ws), G[f"{name}_with_src"]) and np.array_equal(src, G[f"{name}_src"])
    hull = areas.area_segmentation_convex_fusion(ws, src)
    assert np.array_equal(planes(hull), G[f"{name}_hulls"])
    grown = areas.area_growing(hull, nexpmap)
    assert np.array_equal(planes(grown), G[f"{name}_grown"])
    amap = areas.area_segmentation_final(grown, minsize ** 2, maxsize ** 2)
    assert np.array_equal(amap.astype(int), G[f"{name}_areamap"])


@pytest.mark.parametrize("name", ["many", "few"])
def test_create_areamap_matches_the_reference(name):
    c = INP[name]
    amap, nb = areas.create_areamap(c["mask"], c["segmap"], c["pfa"], c["minsize"], c["maxsize"])
    assert amap.dtype.kind == "i" and np.array_equal(amap, G[f"{name}_areamap"])
    assert nb == int(G[f"{name}_nbareas"])
    exposed = (~c["mask"]).any(axis=0)
    assert np.array_equal(amap > 0, exposed)                    # the areas tile the exposed field
    for s in range(1, c["segmap"].max() + 1):                   # no source is cut
        assert len(np.unique(amap[c["segmap"] == s])) == 1


def test_fusion_criteria_and_hull_in_isolation():
    lab = np.zeros((5, 40, 44))
    lab[0, :12, :20] = 1
    lab[1, :12, 20:] = 1
    lab[2, 12:, :9] = 1
    lab[3, 12:30, 9:] = 1
    lab[4, 30:, 9:] = 1
    assert np.array_equal(areas.fusion_areas(lab.copy(), 300, 900).sum(axis=(1, 2)), G["fusion_min"])
    assert np.array_equal(areas.fusion_areas(lab.copy(), 300, 900, option='var').sum(axis=(1, 2)),
                          G["fusion_var"])
    with pytest.raises(ValueError):
        areas.fusion_areas(lab.copy(), 300, 900, option='nope')
    filled = areas.Convexline(G["hull_points"].copy(), 0, 0)
    assert np.array_equal(np.asarray(filled).astype(np.uint8), G["hull_filled"])


def test_iterated_morphology_equals_distance_transform():
    """The two shortcuts of area_growing against SciPy's iterated operators."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(4)
    for shape, p in (((60, 70), 0.004), ((33, 31), 0.02), ((50, 50), 0.0)):
        a = rng.random(shape) < p
        assert np.array_equal(areas._dilate_taxicab(a, 21), ndi.binary_dilation(a, iterations=21))
        d = ndi.binary_dilation(a, iterations=21)
        assert np.array_equal(areas._erode_taxicab(d, 20),
                              ndi.binary_erosion(d, border_value=1, iterations=20))
    full = np.ones((9, 9), bool)
    assert areas._erode_taxicab(full, 20).all()


def test_single_area_field_and_step():
    """NbSubcube == 1: the area map is the exposure map (steps.py:552-553); the step stores the
    map and the number of areas like the reference (:556-566)."""
    from origin_amd.steps import CreateAreas, Status
    mask = np.zeros((4, 30, 32), bool)
    mask[:, :2, :] = True
    amap, nb = areas.create_areamap(mask, np.zeros((30, 32), int), minsize=100)
    assert nb == 1 and np.array_equal(amap, (~mask).any(axis=0).astype(int))

    class Orig:
        pass
    o = Orig()
    c = INP["many"]
    o.mask, o.param, o.Ny, o.Nx = c["mask"], {}, *c["segmap"].shape
    o.segmap_merged = c["segmap"]
    o.steps = {}
    step = CreateAreas(o, 2, o.param)
    step(pfa=0.2, minsize=c["minsize"])
    assert step.status is Status.RUN and o.param["nbareas"] == int(G["many_nbareas"])
    assert np.array_equal(step.areamap, G["many_areamap"])
    assert o.param["areas"]["params"] == dict(pfa=0.2, minsize=c["minsize"], maxsize=None)


@pytest.mark.gpu
def test_areas_step_inside_the_gpu_chain():
    """The real step between preprocessing (which makes segmap_merged on the GPU path) and the
    threshold / PCA / GLR steps, which then work on its non-rectangular areas."""
    from origin_amd import synth
    from origin_amd.device import default_context
    from origin_amd.steps import SimpleOrig, Status
    f, raw, var, mask = synth.small_case(Nz=120, Ny=64, Nx=72, seed=9, psf_size=9, nprof=3)
    orig = SimpleOrig(raw, var, mask, f.PSF, f.profiles, ctx=default_context(0))
    orig.step01_preprocessing()
    seg = np.asarray(orig.segmap_merged)
    try:
        orig.step02_areas(minsize=30)
    except Exception as exc:  # QhullError: a square whose only sources are collinear pixels
        if "Qhull" not in type(exc).__name__:
            raise
        pytest.skip("degenerate sources for ConvexHull in this synthetic field (reference raises too)")
    amap = np.asarray(orig.areamap)
    assert orig.nbAreas == len(np.unique(amap[amap > 0])) >= 2
    assert np.array_equal(amap > 0, (~mask).any(axis=0))
    for s in range(1, seg.max() + 1):
        assert len(np.unique(amap[seg == s])) == 1
    orig.step03_compute_PCA_threshold()
    orig.step04_compute_greedy_PCA()
    orig.step05_compute_TGLR()
    assert all(s.status is Status.RUN for s in list(orig.steps.values())[:5])
    assert np.isfinite(orig.cube_faint._data).all() and np.isfinite(orig.maxmap).all()
    assert len(orig.thresO2) == orig.nbAreas
