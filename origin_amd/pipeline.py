"""Device-resident stages of the hot path, in Step order:

    preprocess  (steps.py:431-450)  ->  pca_threshold (steps.py:610-631)
    -> greedy_pca (steps.py:681-704) -> tglr (steps.py:770-802)

Cubes stay in HBM between stages (``DeviceArray``); only per-spaxel maps, thresholds and
what a caller asks for cross PCIe.  ``origin_amd.steps`` wraps these behind the reference's
Step API, ``bench.py`` times them directly.
"""
import numpy as np

from . import kernels
from .pca import GreedyPCA
from .thresholds import areas_fit


def preprocess(ctx, raw, var, mask, dct_order=10, dct_approx=False, allreduce=None,
               want_cont=True, allreduce_dev=None):
    """DCT continuum + standardisation.  raw/var: float32 (Nz,Ny,Nx) DeviceArrays, mask
    uint8.  ``allreduce``: callable summing a float64 host vector over all ranks (the
    per-channel mean of steps.py:442 is over the *whole* field when the cube is tiled);
    ``allreduce_dev``: the same on the device, ``f(ctx, [DeviceArray, ...])`` in place
    (``TileComm.allreduce_sum_device``: RCCL on the context's stream, no host round trip)."""
    coef, zsum, zcnt = kernels.dct_fit_sums(ctx, raw, var, mask, dct_order, dct_approx)
    if allreduce_dev is not None:
        allreduce_dev(ctx, [zsum, zcnt])
    elif allreduce is not None:
        both = np.concatenate([zsum.to_host(), zcnt.to_host()])
        both = allreduce(both)
        n = zsum.size
        zsum.upload(both[:n])
        zcnt.upload(both[n:])
    # cube_std, cont_dct (var is read once for both) and the O2 map in one pass.  (A continuum
    # pass of its own after the O2 map has left -- kernels.dct_cont_std, optionally on the
    # auxiliary stream -- overlaps the host's threshold fit but costs the PCA's first iterations
    # its 10.6 GB of traffic: measured equal within noise, bench.py ORIGIN_BENCH_CONT.)
    out = kernels.dct_standardize(ctx, raw, var, mask, coef, zsum, zcnt, want_cont=want_cont)
    out["o2_host"] = out["o2"].to_host()
    out["coef"] = coef  # (freeing it here would wait for the pass just enqueued)
    return out


_CAT = {}  # concatenated area lists of the last pca_threshold call (keyed by the list object)


def area_lists(areamap, nbAreas):
    """Flat spaxel indices of each area, in the column order of ``cube[:, areamap == i]``."""
    flat = np.asarray(areamap).reshape(-1)
    order = np.argsort(flat, kind="stable")
    sorted_lab = flat[order]
    lo = np.searchsorted(sorted_lab, np.arange(1, nbAreas + 1), side="left")
    hi = np.searchsorted(sorted_lab, np.arange(1, nbAreas + 1), side="right")
    return [order[a:b].astype(np.int32) for a, b in zip(lo, hi)]


def pca_threshold(o2_map, areamap, nbAreas, pfa_test=0.01, spx=None):
    """``ComputePCAThreshold.run`` (steps.py:610-631) on the O2 map of cube_std."""
    spx = area_lists(areamap, nbAreas) if spx is None else spx
    flat = np.asarray(o2_map, dtype=np.float64).reshape(-1)

    # the concatenated lists are kept per list object; gather, clip, histogram, Gaussian fit and
    # threshold of all areas: one native call spread over the host worker pool (csrc/thresh.hip,
    # csrc/lmfit.hip)
    cat = _CAT.get(id(spx))
    if cat is None or cat[0] is not spx:
        off = np.zeros(len(spx) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in spx])
        idx = (np.concatenate(spx) if len(spx) else np.zeros(0)).astype(np.int32)
        _CAT.clear()
        cat = _CAT[id(spx)] = (spx, idx, off)
    _, idx, off = cat
    tests, fits = areas_fit(flat, idx, off, pfa_test) if len(spx) else ([], [])
    results = [(t,) + f for t, f in zip(tests, fits)]
    testO2, histO2, binO2, thresO2, meaO2, stdO2 = zip(*results)
    return dict(testO2=testO2, histO2=histO2, binO2=binO2, thresO2=thresO2, meaO2=meaO2,
                stdO2=stdO2)


def greedy_pca(ctx, cube_std, areamap, nbAreas, thresholds, testO2, Noise_population=50,
               itermax=100, spx=None, inplace=False, driver=None, o2_dev=None, out=None,
               into=None):
    """``Compute_GreedyPCA_area`` (lib_origin.py:769-821) on a device cube.  Returns
    (cube_faint DeviceArray, mapO2 (Ny,Nx) float64, nstop, driver).  ``o2_dev``: the O2 map of
    cube_std still on the device (float64 [Ny,Nx]) -- used instead of ``testO2`` when given;
    ``driver``: a GreedyPCA to reuse (keeps the area lists on the device between calls);
    ``into = (ext, top, left)``: cube_faint is written into that box of the larger cube ``ext``
    (the halo-extended tile of multigpu.TiledGLR) and the first return value is None."""
    Nz, Ny, Nx = cube_std.shape
    spx = area_lists(areamap, nbAreas) if spx is None else spx
    if into is not None:
        drv = driver or GreedyPCA(ctx)
        if sum(len(s) for s in spx) < Ny * Nx:
            # spaxels outside every area keep their cube_std values (:799): the box first, the
            # areas over it
            from .multigpu import _copy_box
            ext, top, left = into
            _copy_box(ctx, ext, ext.shape, (0, top, left), cube_std, cube_std.shape, (0, 0, 0),
                      (Nz, Ny, Nx))
        hmap, nstop = drv.run(None, spx, testO2, [float(t) for t in thresholds],
                              Noise_population, itermax, test_map=o2_dev, src=cube_std,
                              want_map="full", into=into)
        return None, hmap.astype(np.float64).reshape(Ny, Nx), nstop, drv
    # cube_faint = cube_std.copy() (:799): the copy is folded into the final F = X - U C pass
    F = cube_std if inplace else (out if out is not None else ctx.empty(cube_std.shape,
                                                                        np.float32))
    if not inplace and sum(len(s) for s in spx) < Ny * Nx:
        F.copy_from(cube_std)  # spaxels outside every area keep their cube_std values (:799)
    drv = driver or GreedyPCA(ctx)
    hmap, nstop = drv.run(F, spx, testO2, [float(t) for t in thresholds], Noise_population,
                          itermax, test_map=o2_dev, src=None if inplace else cube_std,
                          want_map="full")
    # (the device map is zero outside the areas: the same array as scattering the per-area maps)
    return F, hmap.astype(np.float64).reshape(Ny, Nx), nstop, drv


def glr_bands_for(active_rows, Ny, halo, max_early_rows=None):
    """(early, late): row bands [y0, y1) in units the row-band GLR accepts (multiples of 64, the
    last one ends at Ny) that together cover [0, Ny).  ``active_rows``: (ymin, ymax) inclusive of
    every area that still iterates; a band is late when its spatial stage would read a row of one
    of them (+- halo).  ``max_early_rows``: at most that many rows early (rounded down to a
    multiple of 64, at least 64) -- the largest ready bands first, cut at multiples of 64, the rest
    joins the late bands; adjacent late bands are merged."""
    nb = (Ny + 63) // 64
    blocked = np.zeros(nb, bool)
    for ymin, ymax in active_rows:
        b0 = max(0, (ymin - halo) // 64)
        b1 = min(nb - 1, (ymax + halo) // 64)
        blocked[b0:b1 + 1] = True

    def runs(flag):
        out, b = [], 0
        while b < nb:
            if blocked[b] == flag:
                e = b
                while e < nb and blocked[e] == flag:
                    e += 1
                out.append((64 * b, min(Ny, 64 * e)))
                b = e
            else:
                b += 1
        return out
    early, late = runs(False), runs(True)
    if max_early_rows is not None:
        rows_left = max(64, int(max_early_rows) // 64 * 64)
        kept, moved = [], []
        for y0, y1 in sorted(early, key=lambda band: band[0] - band[1]):
            take = min(y1 - y0, rows_left)
            take = take if y0 + take == y1 else take // 64 * 64
            if take > 0:
                kept.append((y0, y0 + take))
                rows_left -= take
            if y0 + take < y1:
                moved.append((y0 + take, y1))
        early, merged = sorted(kept), []
        for y0, y1 in sorted(late + moved):
            if merged and merged[-1][1] == y0:
                merged[-1] = (merged[-1][0], y1)
            else:
                merged.append((y0, y1))
        late = merged
    return early, late


def greedy_pca_then_glr(ctx, plan, cube_std, areamap, nbAreas, thresholds, testO2, mask,
                        correl, profile, correl_min, cube_faint, Noise_population=50, itermax=100,
                        spx=None, driver=None, o2_dev=None, max_active=2, area_rows=None,
                        local_max=None, early_budget=8.5e8):
    """greedy PCA and GLR of one field with the GLR of the finished part of the field started in
    the shadow of the PCA's tail: when at most ``max_active`` areas still iterate, the library
    writes the others to ``cube_faint`` and calls back; the row bands whose spatial stage reads
    none of the rows of the areas that go on run at once on the context's side stream (all CUs
    but a reserve the PCA's small kernels keep), the remaining bands behind the PCA on the main
    stream.  Same results as ``greedy_pca`` followed by ``plan.run`` (the bands run the same
    kernels on the same waves and regions).  ``local_max = (out_max, out_min)``: the 3x3x3 local
    maxima of correl / correl_min behind the last band (steps.py:796), dense; a
    ``sparse.SparseBuffers`` instead of the pair: the same in sparse form (``out["local_max"]`` /
    ``["local_min"]`` are ``SparseCube`` s then).  ``early_budget``: voxels
    of GLR given to the side stream at most (None: every band that is ready).  The PCA's tail
    is a few milliseconds of mostly idle device; GLR work beyond what fits beside it only keeps
    the PCA's small kernels on the reserved CUs for longer (3681x900x900: 139.6 ms per step with
    every ready band early, against 135.1 in sequence; 8.5e8 voxels ~ 12 ms of GLR on an MI355X).  (Their row bands on the
    side stream behind the early GLR bands were built and measured in round 3: 51.98 against
    51.9-52.1 ms per step -- the pass is HBM-bound and the GLR is not, but their workgroups do not
    share a CU, so they take turns; dropped.)  Returns (cube_faint, mapO2, nstop, driver, out) with
    ``out`` as ``plan.run`` returns it, plus ``out["bands"] = (early, late)`` (and ``local_max``
    / ``local_min``)."""
    Nz, Ny, Nx = cube_std.shape
    spx = area_lists(areamap, nbAreas) if spx is None else spx
    if area_rows is None:
        area_rows = [(int(s.min()) // Nx, int(s.max()) // Nx) if len(s) else None for s in spx]
    halo = plan.P // 2
    state = {"early": None, "late": None}

    def hook(areas):
        early, late = glr_bands_for(
            [area_rows[a] for a in areas if area_rows[a]], Ny, halo,
            None if early_budget is None else early_budget / (Nz * Nx))
        for i, (y0, y1) in enumerate(early):
            plan.run_rows(cube_faint, mask, correl, profile, correl_min, y0, y1, first=(i == 0),
                          side=True)
        state["early"], state["late"] = early, late

    use = plan.rows_supported() and max_active > 0
    if use:
        ctx.set_pca_tail_hook(hook, max_active)
    try:
        F, mapO2, nstop, drv = greedy_pca(ctx, cube_std, areamap, nbAreas, thresholds, testO2,
                                          Noise_population, itermax, spx=spx, driver=driver,
                                          o2_dev=o2_dev, out=cube_faint)
    finally:
        if use:
            ctx.set_pca_tail_hook(None)
    err = ctx.pop_tail_hook_error() if use else None
    if err is not None:
        raise err
    def finish_local_max(out):
        if local_max is None:
            return out
        if isinstance(local_max, tuple):
            kernels.local_max(ctx, correl, correl_min, mask, 3, out_max=local_max[0],
                              out_min=local_max[1])
            out["local_max"], out["local_min"] = local_max
        else:    # a sparse.SparseBuffers: (index, value) lists instead of two dense cubes
            from . import sparse
            out["local_max"], out["local_min"] = sparse.local_max_sparse(ctx, correl, correl_min,
                                                                         mask, local_max)
        return out

    if state["early"] is None:     # the hook did not fire (every area finished together)
        out = plan.run(F, mask=mask, correl=correl, profile=profile, correl_min=correl_min,
                       want_maps=True)
        out["bands"] = ([], [(0, Ny)])
        return F, mapO2, nstop, drv, finish_local_max(out)
    first = len(state["early"]) == 0
    for y0, y1 in state["late"]:
        plan.run_rows(F, mask, correl, profile, correl_min, y0, y1, first=first, side=False)
        first = False
    maxmap, minmap = plan.run_finish(want_maps=True)
    out = dict(correl=correl, profile=profile, correl_min=correl_min, maxmap=maxmap,
               minmap=minmap, bands=(state["early"], state["late"]))
    return F, mapO2, nstop, drv, finish_local_max(out)


def tglr(ctx, plan, cube_faint, mask, size=3, want_local=True):
    """``ComputeTGLR.run`` dense part (steps.py:770-802): GLR + mask glue + maps + local
    maxima, all on the device."""
    out = plan.run(cube_faint, mask=mask, want_maps=True)
    if want_local:
        # (sparse where the pass has a sparse form: the cubes are > 98 % zeros and their consumers
        # -- steps 6 and 7 -- count and pick; a DataObj turns dense when somebody reads it)
        from . import sparse
        lmax, lmin = sparse.local_max(ctx, out["correl"], out["correl_min"], mask, size)
        out["local_max"], out["local_min"] = lmax, lmin
    return out
