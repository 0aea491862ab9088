"""Sparse local maxima (origin_local_max_sparse and its consumers, origin_amd/sparse.py) against
the dense pass, the oracle and golden G6 / G8: bit exact -- index work and copies of float32 values.
Reference: compute_local_max lib_origin.py:1220-1256, its consumers :1391-1479 and
steps.py:935-974."""
import os

import numpy as np
import pytest

from oracle import cpu_ref
from oracle import golden_cases as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from origin_amd.device import default_context
    return default_context(0)


def cubes(shape, seed, smooth=True):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape).astype(np.float32)
    if smooth:   # like a GLR output: neighbouring voxels correlated
        from scipy import ndimage as ndi
        a = ndi.uniform_filter(a, 3).astype(np.float32)
        b = ndi.uniform_filter(b, 3).astype(np.float32)
    mask = rng.random(shape) < 0.02
    a[mask] = 0                      # correl[mask] = 0 (steps.py:781)
    return a, b, mask


@pytest.mark.parametrize("shape", [(5, 4, 4), (17, 9, 12), (40, 33, 64), (70, 31, 100),
                                   (131, 64, 252), (300, 100, 128)])
def test_sparse_pass_is_the_dense_pass(ctx, shape):
    """Lists -> dense (device and host) equal the dense kernel's cubes bit for bit; so do the
    reductions of steps 6 and 7 (counts with and without a keep map, negative thresholds, per-spaxel
    maxima, np.where order with the gathered profile cube)."""
    from origin_amd import kernels, sparse
    a, b, mask = cubes(shape, sum(shape), smooth=shape[0] % 2 == 1)
    Nz, Ny, Nx = shape
    da, db, dm = ctx.to_device(a), ctx.to_device(b), ctx.to_device(mask.astype(np.uint8))
    dmax, dmin = kernels.local_max(ctx, da, db, dm, 3)
    hmax, hmin = dmax.to_host(), dmin.to_host()
    rmax, rmin = cpu_ref.compute_local_max(a.astype(float), b.astype(float), mask, 3)
    assert np.array_equal(hmax, rmax) and np.array_equal(hmin, rmin)       # the dense pass itself
    smax, smin = sparse.local_max_sparse(ctx, da, db, dm)
    for sp, dense_h, dense_d in ((smax, hmax, dmax), (smin, hmin, dmin)):
        assert sp.nnz == np.count_nonzero(dense_h)
        assert np.array_equal(sp.dense().to_host(), dense_h)
        assert np.array_equal(sp.to_host(), dense_h)
        got64 = sp.to_host_f64()
        assert got64.dtype == np.float64 and np.array_equal(got64, dense_h.astype(np.float64))
        idx, val = sp.entries()
        assert np.all(np.diff(idx) > 0) and np.array_equal(val, dense_h.reshape(-1)[idx])
        keep = (np.arange(Ny * Nx) % 3 != 0).astype(np.uint8)
        thr = [0.05, 0.0, 0.3, -0.2, 1e9, 0.11]
        for kp in (None, keep):
            kd = None if kp is None else ctx.to_device(kp)
            assert np.array_equal(sp.count_above(thr, kp), kernels.count_above(ctx, dense_d, thr, kd))
            assert np.array_equal(sp.zmax_map(kp), kernels.zmax_map(ctx, dense_d, kd))
        aux = ctx.to_device((np.arange(a.size) % 251).astype(np.uint8).reshape(shape))
        for t in (0.2, 0.0, 5.0, -1.0):
            w1 = sp.where_above(t, aux=aux, cap=7)
            w2 = kernels.where_above(ctx, dense_d, t, aux=aux)
            for k in ("z", "y", "x", "value", "aux"):
                assert np.array_equal(w1[k], w2[k]), (t, k)


def test_sparse_local_max_golden_and_fallbacks(ctx):
    """G6 (the reference's compute_local_max on G5's correl) through the B1 seam, which now takes
    the sparse pass; a constant cube (every voxel a non-zero maximum: a segment overflows) and a
    shape without a sparse form (Nx % 4) come back through the dense kernels, same values."""
    import origin_amd.lib_origin as hip
    from origin_amd import kernels, sparse
    from origin_amd.device import DeviceArray
    g5 = np.load(os.path.join(gc.GOLDEN_DIR, "g5_glr.npz"))
    g = np.load(os.path.join(gc.GOLDEN_DIR, "g6_localmax.npz"))
    mask = gc.g5_mask(g5["a_correl"].shape)
    cm = g5["a_correl"].astype(np.float32)
    cm[mask] = 0
    cmin = g5["a_correl_min"].astype(np.float32)
    lmax, lmin = hip.compute_local_max(cm, cmin, mask, 3)
    rmax, rmin = cpu_ref.compute_local_max(cm.astype(float), cmin.astype(float), mask, 3)
    assert np.array_equal(lmax, rmax) and np.array_equal(lmin, rmin)
    assert np.mean((lmax != 0) != (g["local_max"] != 0)) < 1e-3
    d = ctx.to_device(cm)
    assert (cm.shape[2] % 4 == 0) == isinstance(sparse.local_max(ctx, d, d, None)[0],
                                                sparse.SparseCube)
    const = ctx.to_device(np.full((40, 16, 64), 2.0, np.float32))
    a, b = sparse.local_max(ctx, const, const, None)
    assert isinstance(a, DeviceArray)                       # overflow -> dense
    assert np.all(a.to_host() == 2.0) and np.all(b.to_host() == -2.0)  # (plateaus: v == max)
    sp = sparse.local_max_sparse(ctx, const, const, None)[0]
    with pytest.raises(sparse.SparseOverflow):
        sp.counts()
    odd = ctx.to_device(np.random.default_rng(3).standard_normal((9, 7, 10)).astype(np.float32))
    assert sparse.plan(ctx, odd.shape)[0] == 0
    a, b = sparse.local_max(ctx, odd, odd, None)
    ka, kb = kernels.local_max(ctx, odd, odd, None, 3)
    assert np.array_equal(a.to_host(), ka.to_host()) and np.array_equal(b.to_host(), kb.to_host())


@pytest.mark.parametrize("name", ["seg", "noseg", "lst"])
def test_purity_golden_on_sparse_cubes(ctx, name):
    """G8 (Compute_threshold_purity of the reference) with the local-maximum cubes handed over as
    SparseCubes (lists made from the golden's own cubes): the table is the golden's, as with
    dense device cubes."""
    import origin_amd.lib_origin as hip
    g = np.load(os.path.join(gc.GOLDEN_DIR, "g8_purity.npz"))
    inp = gc.g8_inputs()
    smax = _as_sparse(ctx, inp["lmax"].astype(np.float32))
    smin = _as_sparse(ctx, inp["lmin"].astype(np.float32))
    segmap = None if name == "noseg" else inp["segmap"]
    tl = list(inp["threshlist"]) if name == "lst" else None
    with np.errstate(all="ignore"):
        thr, res = hip.Compute_threshold_purity(float(g[name + "_purity"]), smax, smin, segmap,
                                                threshlist=tl)
    assert thr == float(g[name + "_threshold"])
    for c in ("Tval_r", "Pval_r", "Det_m", "Det_M"):
        assert np.array_equal(np.asarray(res[c], float), np.asarray(g[f"{name}_{c}"], float),
                              equal_nan=True), c


def _as_sparse(ctx, dense):
    """A SparseCube holding exactly the non-zero voxels of a host cube (one segment per 64
    entries): the consumers' side of the format, independent of the pass that fills it."""
    from origin_amd import sparse
    idx = np.flatnonzero(dense.reshape(-1)).astype(np.int64)
    val = dense.reshape(-1)[idx]
    cap = 64
    nseg = max(1, -(-len(idx) // cap))

    class B:
        pass
    b = B()
    b.ctx, b.shape, b.nseg, b.seg_cap = ctx, dense.shape, nseg, cap
    pi = np.zeros(nseg * cap, np.int64)
    pv = np.zeros(nseg * cap, np.float32)
    pi[:len(idx)], pv[:len(idx)] = idx, val
    cnt = np.zeros(2 * nseg, np.int32)
    full, rest = divmod(len(idx), cap)
    cnt[:full] = cap
    if rest:
        cnt[full] = rest
    b.idx, b.val = [ctx.to_device(pi), None], [ctx.to_device(pv), None]
    b.counts = ctx.to_device(cnt)
    return sparse.SparseCube(b, 0)


def test_step_chain_keeps_sparse_local_maxima(ctx):
    """The Step chain leaves SparseCubes in the session's device cache for cube_local_max / _min
    and the std pair; the DataObjs read dense (what the reference's interface promises) and equal
    the oracle's compute_local_max of the device's own correl, bit for bit."""
    from origin_amd import sparse, synth
    from origin_amd.steps import SimpleOrig
    f, raw, var, mask = synth.small_case(Nz=160, Ny=48, Nx=52, seed=3, psf_size=9, nprof=3,
                                         area_size=24)
    orig = SimpleOrig(raw, var, mask, f.PSF.astype(float), f.profiles, ctx=ctx)
    orig.step01_preprocessing()
    orig.step02_areas.set_areamap(f.areamap)
    orig.step03_compute_PCA_threshold()
    orig.step04_compute_greedy_PCA()
    orig.step05_compute_TGLR()
    for name in ("cube_local_max", "cube_local_min", "cube_std_local_max", "cube_std_local_min"):
        assert isinstance(orig._hip_cache[name], sparse.SparseCube), name
    correl = orig._hip_cache["cube_correl"].to_host().astype(float)
    cmin = orig._hip_cache["cube_correl_min"].to_host().astype(float)
    rmax, rmin = cpu_ref.compute_local_max(correl, cmin, mask, 3)
    assert np.array_equal(orig.cube_local_max._data, rmax)
    assert np.array_equal(orig.cube_local_min._data, rmin)
    std = orig._hip_cache["cube_std"].to_host().astype(float)
    smax, smin = cpu_ref.compute_local_max(std, std, mask, 3)
    assert np.array_equal(orig.cube_std_local_max._data, smax)
    assert np.array_equal(orig.cube_std_local_min._data, smin)
