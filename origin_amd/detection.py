"""Thresholding of step 7 on the device (reference muse_origin/steps.py:935-994).

``Detection.run`` opens with three ``np.where(cube > threshold)`` scans of full host cubes
(:958, :968; ``det_correl_min`` :938) and fancy-indexed gathers at the hits (:962-963, :971).
With the cubes resident in HBM those are ordered stream compactions
(``origin_where_above``): only the detections leave the GPU, in the order ``np.where``
returns them, so everything downstream (``spatiospectral_merging``, the segmentation labels,
the WCS columns -- host code on 10^3..10^5 rows) sees the same table.

Tables are plain dicts of NumPy columns (astropy is not a dependency of this package);
``astropy.table.Table(cat0)`` gives the reference's ``Cat0`` before ``_format_cat``.
"""
import numpy as np

from . import kernels

CAT0_COLUMNS = ("x0", "y0", "z0", "comp", "STD", "T_GLR", "profile")


def det_correl_min(ctx, cube_local_min, thresh):
    """``Detection.det_correl_min`` (steps.py:935-939): positions above ``thresh`` in
    cube_local_min -> (zm, ym, xm)."""
    w = _where_above(ctx, cube_local_min, thresh)
    return w["z"], w["y"], w["x"]


def _where_above(ctx, cube, threshold, aux=None):
    """``kernels.where_above`` for a DeviceArray; cubes that live in pieces on several devices
    (session.TiledCube) bring their own."""
    if hasattr(cube, "where_above"):
        return cube.where_above(threshold, aux=aux)
    return kernels.where_above(ctx, cube, threshold, aux=aux)


def threshold_detections(ctx, cube_local_max, cube_profile, cube_std_local_max,
                         threshold_correl, threshold_std, maxdist_lines=2.5):
    """The first half of ``Detection.run`` (steps.py:956-994) on device cubes.

    Returns ``(cat0, cat_correl, cat_std_kept)``: ``cat0`` the raw detection table (correl
    rows, then std rows: the ``vstack`` of :981), ``cat_correl`` its correl rows and
    ``cat_std_kept`` the std rows farther than ``maxdist_lines`` from every correl detection
    (:983-994; in ascending row order) -- the two tables the reference stacks again at :1010.
    """
    c = _where_above(ctx, cube_local_max, threshold_correl, aux=cube_profile)
    s = _where_above(ctx, cube_std_local_max, threshold_std)
    n, m = c["z"].size, s["z"].size
    cat = dict(x0=c["x"], y0=c["y"], z0=c["z"], comp=np.zeros(n, int), STD=np.full(n, np.nan),
               T_GLR=c["value"], profile=c["aux"])
    cat_std = dict(x0=s["x"], y0=s["y"], z0=s["z"], comp=np.ones(m, int), STD=s["value"],
                   T_GLR=np.full(m, np.nan), profile=np.zeros(m, np.uint8))
    cat0 = {k: np.concatenate([cat[k], cat_std[k]]) for k in CAT0_COLUMNS}
    keep = unmatched_std(cat, cat_std, maxdist_lines)
    return cat0, cat, {k: v[keep] for k, v in cat_std.items()}


def from_session(orig, threshold=None, threshold_std=None, maxdist_lines=2.5):
    """``threshold_detections`` on the cubes a session holds (device copies left by steps 1 and
    5 when there are any, host cubes uploaded otherwise) with the thresholds of step 6
    (``param['threshold']``, ``param['threshold_std']``) unless given, like steps.py:951-954."""
    from .steps import _HipStepMixin, _ctx_of
    ctx = _ctx_of(orig)
    get = _HipStepMixin()._get_cube
    thr = orig.param['threshold'] if threshold is None else threshold
    thr_std = orig.param['threshold_std'] if threshold_std is None else threshold_std
    prof = get(orig, ctx, 'cube_profile')
    if prof.dtype != np.uint8 and not hasattr(prof, "where_above"):
        prof = ctx.to_device(prof.to_host().astype(np.uint8))
    return threshold_detections(ctx, get(orig, ctx, 'cube_local_max'), prof,
                                get(orig, ctx, 'cube_std_local_max'), thr, thr_std,
                                maxdist_lines)


def unmatched_std(cat, cat_std, maxdist_lines=2.5):
    """Rows of ``cat_std`` with no correl detection within ``maxdist_lines`` voxels
    (steps.py:983-992: two cKDTrees and ``query_ball_tree``), ascending."""
    from scipy.spatial import cKDTree

    n, m = len(cat["z0"]), len(cat_std["z0"])
    if n == 0 or m == 0:
        return np.arange(m)
    kdt_cor = cKDTree(np.array([cat["x0"], cat["y0"], cat["z0"]]).T)
    kdt_std = cKDTree(np.array([cat_std["x0"], cat_std["y0"], cat_std["z0"]]).T)
    hit = np.zeros(m, dtype=bool)
    for lst in kdt_cor.query_ball_tree(kdt_std, maxdist_lines):
        hit[lst] = True
    return np.flatnonzero(~hit)
