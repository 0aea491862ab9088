"""Function seam (tier B1): the hot functions of the reference's
``muse_origin/lib_origin.py`` with identical names, signatures and return tuples, running
on the MI355X through liborigin_hip.so.

``muse_origin.steps`` resolves these names through its module globals (steps.py:19-41), so

    import muse_origin.steps, origin_amd.lib_origin as hip
    for name in hip.__all__:
        setattr(muse_origin.steps, name, getattr(hip, name))

swaps the implementation without touching the reference.  Inputs and
outputs are host ``ndarray`` s (float64 out, as the reference produces); internally the
cubes are float32 in HBM.  There is no CPU fallback: without the library or a GPU every
function raises.
"""
import numpy as np

from . import kernels
from .device import default_context
from .pca import GreedyPCA
from .thresholds import compute_thresh_gaussfit

__all__ = (
    'dct_residual',
    'O2test',
    'Compute_PCA_threshold',
    'Compute_GreedyPCA',
    'Compute_GreedyPCA_area',
    'Correlation_GLR_test',
    'compute_local_max',
    'compute_thresh_gaussfit',
    'Compute_threshold_purity',
)


def _ctx():
    return default_context(0)


def _mask_u8(mask, shape):
    if mask is None or mask is np.ma.nomask:
        return np.zeros(shape, dtype=np.uint8)
    return np.ascontiguousarray(np.broadcast_to(mask, shape), dtype=np.uint8)


def dct_residual(w_raw, order, var, approx, mask):
    """Continuum estimated from the DCT decomposition (reference lib_origin.py:150-240;
    despite its name the reference returns the continuum, steps.py:431-434)."""
    ctx = _ctx()
    shape = w_raw.shape
    raw = ctx.to_device(w_raw, np.float32)
    dvar = ctx.to_device(np.broadcast_to(var, shape), np.float32)
    dmask = ctx.to_device(_mask_u8(mask, shape))
    coef = kernels.dct_fit(ctx, raw, dvar, dmask, order, approx)
    cont = kernels.dct_continuum(ctx, coef, shape[0])
    return cont.to_host_f64()


def O2test(arr):
    """mean(arr**2, axis=0) (reference lib_origin.py:957-974)."""
    ctx = _ctx()
    arr = np.asarray(arr)
    d = ctx.to_device(arr, np.float32)
    return kernels.o2test(ctx, d).to_host().reshape(arr.shape[1:])


def Compute_PCA_threshold(faint, pfa):
    """test, histO2, frecO2, thresO2, mea, std (reference lib_origin.py:824-845)."""
    test = O2test(faint)
    histO2, frecO2, thresO2, mea, std = compute_thresh_gaussfit(test, pfa)
    return test, histO2, frecO2, thresO2, mea, std


def Compute_GreedyPCA(cube_in, test, thresO2, Noise_population, itermax):
    """faint, mapO2, nstop for one area given as (Nz, S) (reference lib_origin.py:848-954)."""
    ctx = _ctx()
    cube_in = np.asarray(cube_in)
    Nz, S = cube_in.shape
    F = ctx.to_device(cube_in.reshape(Nz, 1, S), np.float32)
    spx = np.arange(S, dtype=np.int32)
    mapO2, nstop = GreedyPCA(ctx).run(F, [spx], [np.asarray(test, dtype=np.float64)],
                                      [float(thresO2)], Noise_population, itermax)
    return F.to_host_f64().reshape(Nz, S), mapO2[0], nstop


def Compute_GreedyPCA_area(NbArea, cube_std, areamap, Noise_population, threshold_test,
                           itermax, testO2):
    """cube_faint, mapO2, nstop over all areas (reference lib_origin.py:769-821)."""
    ctx = _ctx()
    cube_std = np.asarray(cube_std)
    Nz, Ny, Nx = cube_std.shape
    F = ctx.to_device(cube_std, np.float32)
    flat = np.asarray(areamap).reshape(-1)
    area_spx = [np.nonzero(flat == i)[0].astype(np.int32) for i in range(1, NbArea + 1)]
    thr = [float(threshold_test[i]) for i in range(NbArea)]
    maps, nstop = GreedyPCA(ctx).run(F, area_spx, [np.asarray(t) for t in testO2], thr,
                                     Noise_population, itermax)
    mapO2 = np.zeros(Ny * Nx)
    for spx, m in zip(area_spx, maps):
        mapO2[spx] = m
    return F.to_host_f64(), mapO2.reshape(Ny, Nx), nstop


def Correlation_GLR_test(cube, fsf, weights, profiles, nthreads=1, pcut=None, pmeansub=True):
    """correl, profile (uint8), correl_min (reference lib_origin.py:1070-1217).
    ``nthreads`` is accepted for signature compatibility and ignored."""
    ctx = _ctx()
    cube = np.asarray(cube)
    plan = kernels.GLRPlan(ctx, cube.shape, fsf, weights, profiles, pcut, pmeansub)
    try:
        d = ctx.to_device(cube, np.float32)
        out = plan.run(d, mask=None, want_maps=False)
        correl = out["correl"].to_host_f64()
        profile = out["profile"].to_host()
        correl_min = out["correl_min"].to_host_f64()
    finally:
        plan.close()
    return correl, profile, correl_min


def compute_local_max(correl, correl_min, mask, size=3):
    """local maxima of correl and of -correl_min (reference lib_origin.py:1220-1256)."""
    ctx = _ctx()
    if not np.isscalar(size):
        if len(set(size)) != 1:
            raise ValueError("only cubic windows are supported")
        size = size[0]
    correl = np.asarray(correl)
    dc = ctx.to_device(correl, np.float32)
    dm = dc if correl_min is correl else ctx.to_device(correl_min, np.float32)
    dmask = ctx.to_device(_mask_u8(mask, correl.shape))
    from . import sparse
    # (sparse pass where it exists: only the maxima cross PCIe, the zeros are made on the host)
    lmax, lmin = sparse.local_max(ctx, dc, dm, dmask, size)
    return lmax.to_host_f64(), lmin.to_host_f64()


class PurityTable(dict):
    """Columns Tval_r, Pval_r, Det_m, Det_M of the purity curve, sorted by Tval_r (what the
    reference returns as an astropy Table, lib_origin.py:1455-1461).  ``as_table()`` builds the
    astropy Table when astropy is installed."""

    colnames = ('Tval_r', 'Pval_r', 'Det_m', 'Det_M')

    def as_table(self):
        from astropy.table import Table
        res = Table([self[c] for c in self.colnames], names=self.colnames)
        res['Tval_r'].format = '.2f'
        res['Pval_r'].format = '.2f'
        return res


def Compute_threshold_purity(purity, cube_local_max, cube_local_min, segmap=None,
                             threshlist=None):
    """Threshold for a given purity (reference lib_origin.py:1391-1479).  The cubes may be host
    arrays or float32 ``DeviceArray`` s (no PCIe traffic then); only per-spaxel maxima and the
    counts per threshold leave the GPU.  Returns (threshold, PurityTable)."""
    import logging

    from .device import DeviceArray
    ctx = _ctx()
    logger = logging.getLogger(__name__)

    class OnDevice:
        """The two reductions on one DeviceArray; cubes that live in pieces on several devices
        (session.TiledCube) bring their own ``zmax_map`` / ``count_above``."""

        def __init__(self, a):
            self.a, self.shape = a, a.shape

        def _keep(self, keep):
            return None if keep is None else ctx.to_device(
                np.ascontiguousarray(keep, dtype=np.uint8).reshape(-1))

        def zmax_map(self, keep=None):
            return kernels.zmax_map(ctx, self.a, self._keep(keep))

        def count_above(self, thresholds, keep=None):
            return kernels.count_above(ctx, self.a, thresholds, self._keep(keep))

    def dev(c):
        if hasattr(c, "count_above"):
            return c
        return OnDevice(c if isinstance(c, DeviceArray) else ctx.to_device(np.asarray(c), np.float32))

    lmax, lmin = dev(cube_local_max), dev(cube_local_min)
    L1 = int(np.prod(lmin.shape[1:]))                                          # :1425
    keep = None
    if segmap is not None:                                                     # :1428-1435
        keep = np.asarray(segmap) == 0
        L0 = int(np.count_nonzero(keep))
        logger.info('using only background pixels (%.1f%%)', L0 / L1 * 100)
    else:
        L0 = L1
    if threshlist is None:                                                     # :1437-1442
        map_max = lmax.zmax_map()
        threshmax = min(lmin.zmax_map(keep).max(), map_max.max())
        threshmin = np.median(map_max) * 1.1
        threshlist = np.linspace(threshmin, threshmax, 50)
    threshlist = np.asarray(threshlist, dtype=np.float64)
    n1 = lmax.count_above(threshlist)                                          # :1444-1450
    n0 = lmin.count_above(threshlist, keep) * (L1 / L0)                        # :1452
    with np.errstate(divide='ignore', invalid='ignore'):
        est_purity = 1 - n0 / n1                                               # :1454
    order = np.argsort(threshlist, kind='stable')                              # res.sort('Tval_r')
    res = PurityTable(Tval_r=threshlist[order], Pval_r=est_purity[order],
                      Det_m=n0.astype(int)[order], Det_M=n1[order])
    if est_purity[-1] < purity:                                                # :1464-1468
        logger.warning('Maximum computed purity %.2f is below %.2f', est_purity[-1], purity)
        threshold = np.inf
    else:
        threshold = np.interp(purity, res['Pval_r'], res['Tval_r'])            # :1470
        detect = np.interp(threshold, res['Tval_r'], res['Det_M'])
        logger.info('Interpolated Threshold %.2f Detection %d for Purity %.2f', threshold,
                    detect, purity)
    try:  # the reference returns an astropy Table (what Step.dump writes, steps.py:321-322)
        return float(threshold), res.as_table()
    except ImportError:
        return float(threshold), res
