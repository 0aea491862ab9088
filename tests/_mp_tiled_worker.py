"""Worker of the multi-process tiling tests (one process per rank).  The host group is the
package's socket rendezvous, or torch.distributed/gloo behind the same methods
(TILED_GROUP=gloo, tests/_gloo_group.py).

mode 'cpu' : host arrays + the CPU oracle per tile -- checks the tiling arithmetic
             (area-aligned tiles, global per-channel mean, two-phase halo exchange, border
             classes of the extended tile) against the untiled oracle, bit-for-bit level.
mode 'gpu' : the same decomposition through the HIP path (all ranks share GPU 0, strips are
             host-staged through the host group) against the single-context HIP result.
mode 'rccl': one GPU per rank, native RCCL communicator on the library's stream: device
             all-reduce of the per-channel sums and GPU-to-GPU halo strips (what bench.py --gpus N
             runs); needs as many devices as ranks.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from origin_amd import multigpu, synth  # noqa: E402


def areas_field():
    """A field cut into IRREGULAR areas: the area map the reference's own CreateAreas functions
    made for golden G10 ("many": 132 x 150 spaxels, five areas out of squares, sources, convex
    hulls and growing, label 0 where no exposure covers the field; tests/golden/g10_areas.npz)."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "g10_areas.npz"))
    amap = G["many_areamap"].astype(int)
    Ny, Nx = amap.shape
    f = synth.SyntheticField(64, Ny, Nx, seed=8, psf_size=9, nprof=3, blob_density=1 / 300,
                             emitter_density=1 / 900, area_size=50)
    raw, var, mask = f.arrays()
    mask[:, amap == 0] = True          # unexposed spaxels: masked in every channel
    mask[5:9, 60, 70] = True
    # a masked spaxel right next to a cut between two areas (the 3x3x3 local maxima of the
    # neighbouring rank's edge spaxels look at it)
    ys, xs = np.nonzero((amap[:, :-1] != amap[:, 1:]) & (amap[:, :-1] > 0) & (amap[:, 1:] > 0))
    k = len(ys) // 2
    mask[:, ys[k], xs[k]] = True
    mask[20:40, ys[k // 2], xs[k // 2] + 1] = True
    raw[mask] = 0
    var[mask] = np.inf
    f.areamap, f.nbAreas = amap, int(amap.max())
    return f, raw, var, mask


def make_tiling(f, world, Ny, Nx):
    # (+1: the 3x3x3 local maxima of the tile look one spaxel beyond it)
    halo = f.PSF.shape[1] // 2 + 1
    kind = os.environ.get("TILED_FIELD")
    if kind == "areas":
        return multigpu.OwnerTiling.from_areamap(f.areamap, world, halo)
    return multigpu.Tiling(Ny, Nx, world, halo=halo, area_size=50 if kind == "big" else 20)


def field():
    if os.environ.get("TILED_FIELD") == "areas":
        return areas_field()
    if os.environ.get("TILED_FIELD") == "big":
        # tiles wide enough for 64 x 64 regions that need no halo data (TiledGLR runs those on
        # the side stream while the strips travel)
        f = synth.SyntheticField(64, 150, 320, seed=6, psf_size=9, nprof=3, blob_density=1 / 200,
                                 emitter_density=1 / 900, area_size=50)
        raw, var, mask = f.arrays()
        # one area (rows 50-99, columns 100-149) with many spectra of comparable strength: its
        # PCA goes on long after the others (the tail hook of TiledGLR has something to shadow)
        rng = np.random.default_rng(9)
        for j in range(40):
            y, x = 52 + (7 * j) % 46, 101 + (11 * j) % 47
            raw[:, y, x] += ((6.5 + 0.1 * j) * np.sqrt(var[:, y, x]) *
                             rng.standard_normal(raw.shape[0])).astype(raw.dtype)
        mask[10:14, 3, 7] = True
        mask[:, 125, 241] = True
        raw[mask] = 0
        var[mask] = np.inf
        return f, raw, var, mask
    f = synth.SyntheticField(96, 40, 60, seed=5, psf_size=9, nprof=3, blob_density=1 / 80,
                             emitter_density=1 / 300, area_size=20)
    raw, var, mask = f.arrays()
    mask[10:14, 3, 7] = True
    mask[:, 25, 41] = True
    # masked voxels in the one-spaxel ring just outside a tile (the two-rank tiling cuts the 40
    # rows at 20, the four-rank one also the 60 columns at 40 / 20): correl[mask] = 0 must hold in
    # the halo too, or the local maxima of the tile's edge row see the wrong neighbours
    mask[:, 20, 10:30] = True
    mask[30:60, 19, 45] = True
    mask[:, 5:9, 40] = True
    raw[mask] = 0
    var[mask] = np.inf
    return f, raw, var, mask


def mosaic(f):
    """Two fields over the test field: PSFs and weight maps (TILED_WEIGHTS=1)."""
    Nz, Ny, Nx = 96, 40, 60
    x = np.linspace(0, 1, Nx)[None, :] * np.ones((Ny, 1))
    w0 = (0.15 + 0.7 * x).astype(np.float32).astype(np.float64)
    psfs = [f.PSF.astype(float), synth.moffat_psf(Nz, f.PSF.shape[1], fwhm0=3.0, fwhm1=3.4).astype(float)]
    return psfs, [w0, 1.0 - w0]


def main():
    mode, out = sys.argv[1], sys.argv[2]
    weighted = os.environ.get("TILED_WEIGHTS") == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    group = None
    if os.environ.get("TILED_GROUP") == "gloo":
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _gloo_group import GlooGroup
        group = GlooGroup(rank, world)
    comm = multigpu.init_comm(rank, world, rank if mode == "rccl" else 0,
                              backend="rccl" if mode == "rccl" else "host", group=group)
    f, raw, var, mask = field()
    Nz, Ny, Nx = raw.shape
    tiling = make_tiling(f, world, Ny, Nx)
    t = tiling.tile(rank)
    sl = (slice(None), slice(t.y0, t.y1), slice(t.x0, t.x1))
    traw, tvar, tmask = raw[sl], var[sl], mask[sl]
    amap = f.areamap[t.y0:t.y1, t.x0:t.x1]
    owned = None
    if isinstance(tiling, multigpu.OwnerTiling):
        # the rank works on the bounding box of its areas; the spaxels of the box that belong to
        # other ranks count as masked in the DCT stage (they stay out of the per-channel sums
        # that are reduced over the ranks, and what is computed there is discarded) and carry
        # label 0 (in no area of this rank)
        owned = tiling.owned_tile(rank)
        amap = np.where(owned, amap, 0)
        dmask = tmask | ~owned[None]
    else:
        dmask = tmask
    labels = np.unique(amap[amap > 0]) if owned is not None else np.unique(amap)
    lmap = np.where(amap > 0, np.searchsorted(labels, amap) + 1, 0) if owned is not None \
        else np.searchsorted(labels, amap) + 1
    res = {}
    if mode == "cpu":
        from oracle import cpu_ref
        r64, v64 = traw.astype(float), tvar.astype(float)
        cont = cpu_ref.dct_residual(r64, 10, v64, False, dmask)
        data = r64 - cont
        data[dmask] = np.nan
        both = comm.allreduce_sum(np.concatenate([np.nansum(data, axis=(1, 2)),
                                                  np.sum(~dmask, axis=(1, 2)).astype(float)]))
        mean = both[:Nz] / both[Nz:]
        with np.errstate(invalid="ignore"):
            data = (data - mean[:, None, None]) / np.sqrt(v64)
        data[dmask] = 0
        thr = cpu_ref.pca_threshold_areas(data, lmap, len(labels), 0.01)
        faint, mapO2, nstop = cpu_ref.Compute_GreedyPCA_area(len(labels), data, lmap, 50, thr[3],
                                                             100, thr[0])
        ext = multigpu.exchange_halo_host(comm, tiling, rank, faint)
        (_, _, _, _), (top, bot, left, right) = tiling.extended(rank)
        # the TRUE mask over the extended tile: this rank's part, the halo from its owners
        emask = multigpu.exchange_halo_host(comm, tiling, rank, tmask.astype(np.uint8)) > 0
        psf_, wts_ = (f.PSF.astype(float), None)
        if weighted:
            psf_, wfull = mosaic(f)
            (ey0, ey1, ex0, ex1), _ = tiling.extended(rank)
            wts_ = [w_[ey0:ey1, ex0:ex1] for w_ in wfull]
        g = cpu_ref.compute_TGLR(ext, psf_, wts_, f.profiles, emask, pcut=1e-8)
        crop = (slice(None), slice(top, top + tmask.shape[1]), slice(left, left + tmask.shape[2]))
        lmax, lmin = cpu_ref.compute_local_max(g["cube_correl"], g["cube_correl_min"], emask, 3)
        res = dict(cube_std=data, cube_faint=faint, correl=g["cube_correl"][crop],
                   correl_min=g["cube_correl_min"][crop], mapO2=mapO2,
                   maxmap=g["maxmap"][crop[1:]], thr=np.array(thr[3]),
                   local_max=lmax[crop], local_min=lmin[crop])
    else:
        from origin_amd import kernels, pipeline
        from origin_amd.device import Context
        ctx = Context(rank if mode == "rccl" else 0)
        d_raw, d_var = ctx.to_device(traw, np.float32), ctx.to_device(tvar, np.float32)
        d_mask = ctx.to_device(tmask.astype(np.uint8))
        d_dmask = d_mask if owned is None else ctx.to_device(dmask.astype(np.uint8))
        if mode == "rccl":
            comm.attach(ctx)  # RCCL communicator on this context (collective)
            assert comm.backend == "rccl", (comm.backend, comm.note)
            pre = pipeline.preprocess(ctx, d_raw, d_var, d_dmask, 10, False,
                                      allreduce_dev=comm.allreduce_sum_device)
        else:
            pre = pipeline.preprocess(ctx, d_raw, d_var, d_dmask, 10, False,
                                      allreduce=comm.allreduce_sum)
        spx = pipeline.area_lists(lmap, len(labels))
        thr = pipeline.pca_threshold(pre["o2"].to_host(), lmap, len(labels), 0.01, spx=spx)
        psf_, wts_ = (f.PSF.astype(float), None)
        if weighted:
            psf_, wts_ = mosaic(f)
        glr = multigpu.TiledGLR(ctx, comm, tiling, rank, Nz, psf_, f.profiles, pcut=1e-8,
                                weights=wts_)
        shape = d_raw.shape
        # TILED_INTO=1: the PCA writes cube_faint straight into the interior of the GLR's
        # halo-extended tile (origin_pca_run_into), no copy in between -- what bench.py does
        into = os.environ.get("TILED_INTO", "1") == "1"
        # TILED_HOOK=1: the regions that depend neither on the halo nor on the areas still
        # iterating start their GLR inside the PCA's tail (TiledGLR.make_tail_hook)
        hook = None
        if into and os.environ.get("TILED_HOOK") == "1":
            nx_t = shape[2]
            boxes = [(int(s_.min()) // nx_t, int(s_.max()) // nx_t, int((s_ % nx_t).min()),
                      int((s_ % nx_t).max())) if len(s_) else None for s_ in spx]
            hook = glr.make_tail_hook(boxes, d_mask)
        if hook is not None:
            ctx.set_pca_tail_hook(hook, 2)
        faint, mapO2, nstop, _ = pipeline.greedy_pca(ctx, pre["cube_std"], lmap, len(labels),
                                                     thr["thresO2"], thr["testO2"], spx=spx,
                                                     into=glr.faint_target() if into else None)
        if hook is not None:
            ctx.set_pca_tail_hook(None)
            err = ctx.pop_tail_hook_error()
            if err is not None:
                raise err
        correl = ctx.empty(shape, np.float32)
        cmin = ctx.empty(shape, np.float32)
        prof = ctx.empty(shape, np.uint8)
        lm = (ctx.empty(shape, np.float32), ctx.empty(shape, np.float32))
        o = glr.run(faint, d_mask, correl, prof, cmin, local_max=lm)
        n_hook = glr.last_rects[2]
        n_early = len(glr.last_rects[0])
        if os.environ.get("TILED_NOCROP", "1") == "1":
            # the same step without the crop (what bench.py times on tiles): the results stay in
            # the halo-extended arrays, res["box"] is the tile inside them -- bit for bit the
            # cropped cubes
            o2 = glr.run(faint, d_mask, None, None, None, local_max=True)
            by, bx, bny, bnx = o2["box"]
            assert (bny, bnx) == shape[1:]
            for name, ref_arr in (("correl", correl), ("correl_min", cmin), ("profile", prof),
                                  ("local_max", lm[0]), ("local_min", lm[1])):
                got = o2[name].window(by, by + bny, bx, bx + bnx)
                assert np.array_equal(got, ref_arr.to_host()), name
            assert np.array_equal(o2["maxmap"].to_host(), o["maxmap"].to_host())
            # and with the local maxima as lists of the extended tile's non-zero voxels
            from origin_amd import sparse
            # (run() exchanges the halo: a collective -- every rank takes this branch or none)
            has = float(sparse.plan(ctx, glr.eshape)[0] > 0)
            if comm.group.allreduce(np.array([has]), "min")[0] == 1.0:
                dense = [o2["local_max"].to_host(), o2["local_min"].to_host()]
                o3 = glr.run(faint, d_mask, None, None, None, local_max="sparse")
                assert isinstance(o3["local_max"], sparse.SparseCube)
                assert np.array_equal(o3["local_max"].to_host(), dense[0])
                assert np.array_equal(o3["local_min"].to_host(), dense[1])
                res["sparse_tile"] = 1
        if into:   # the tile lives inside the extended buffer
            top, _, left, _ = glr.halos
            faint_host = glr.ext.window(top, top + shape[1], left, left + shape[2])
        else:
            faint_host = faint.to_host()
        res = dict(cube_std=pre["cube_std"].to_host(), cube_faint=faint_host,
                   correl=correl.to_host(), correl_min=cmin.to_host(), mapO2=mapO2,
                   maxmap=o["maxmap"].to_host(), thr=np.array(thr["thresO2"]),
                   local_max=lm[0].to_host(), local_min=lm[1].to_host(),
                   n_early=n_early,   # rectangles run ahead of the halo exchange
                   n_hook=n_hook)     # regions the tail hook started
    if owned is not None:
        res["owned"] = owned
    np.savez(f"{out}.rank{rank}.npz", y0=t.y0, y1=t.y1, x0=t.x0, x1=t.x1, **res)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
