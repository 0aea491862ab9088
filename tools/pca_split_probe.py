"""Would two groups of areas, each with its own lock-step loop on its own stream, overlap?  The
greedy PCA of the 3681 x 600 x 600 bench field as one run (36 areas) against the two halves of
the field (18 areas each, as cubes of their own) run at the same time from two host threads on two
contexts of the same device.  Prints the wall times; round 4's decision about splitting
origin_pca_run into concurrent groups rests on this."""
import multiprocessing as mp
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from origin_amd import synth  # noqa: E402


def _gen(args):
    fa, ic, window = args
    f = _gen.cache.get(fa)
    if f is None:
        f = _gen.cache[fa] = synth.SyntheticField(*fa)
    return ic, f.chunk(ic, window)


_gen.cache = {}


def main():
    N = int(os.environ.get("PROBE_SIZE", "600"))
    Nz = 3681
    fa = (Nz, N, N, None, 25, 20, 0, 1.0 / 400, 1.0 / 900, 100)
    pool = mp.get_context("fork").Pool(12)
    from origin_amd import kernels, pipeline
    from origin_amd.device import Context
    from origin_amd.pca import GreedyPCA
    ctx = Context(0)
    ctx2 = Context(0)
    field = synth.SyntheticField(*fa)
    raw, var = ctx.empty((Nz, N, N), np.float32), ctx.empty((Nz, N, N), np.float32)
    mask = ctx.empty((Nz, N, N), np.uint8)
    plane = N * N
    for ic, (r, v, m) in pool.imap_unordered(_gen, [(fa, ic, (0, N, 0, N)) for ic in range(field.nchunks)]):
        z0 = ic * synth.ZCHUNK
        raw.view(z0 * plane, r.shape).upload(r)
        var.view(z0 * plane, v.shape).upload(v)
        mask.view(z0 * plane, m.shape).upload(m)
    pool.close()
    pre = pipeline.preprocess(ctx, raw, var, mask, want_cont=False)
    amap = field.areamap
    thr = pipeline.pca_threshold(pre["o2_host"], amap, field.nbAreas, 0.01)
    std = pre["cube_std"]
    out = ctx.empty((Nz, N, N), np.float32)
    spx = pipeline.area_lists(amap, field.nbAreas)
    drv = GreedyPCA(ctx)

    def full():
        pipeline.greedy_pca(ctx, std, amap, field.nbAreas, thr["thresO2"], thr["testO2"], spx=spx,
                            driver=drv, o2_dev=pre["o2"], out=out)
        ctx.sync()
    for _ in range(2):
        full()
    t = time.perf_counter()
    for _ in range(5):
        full()
    t_full = (time.perf_counter() - t) / 5
    print(f"one run, {field.nbAreas} areas: {1e3 * t_full:.2f} ms, {drv.iterations} iterations")

    # halves: rows [0, N/2) and [N/2, N) as cubes of their own (contiguous copies)
    h = N // 2 // 100 * 100
    halves = []
    for (y0, y1), c in (((0, h), ctx), ((h, N), ctx2)):
        ny = y1 - y0
        sub = ctx.empty((Nz, ny, N), np.float32)
        from origin_amd.multigpu import _copy_box
        _copy_box(ctx, sub, sub.shape, (0, 0, 0), std, std.shape, (0, y0, 0), (Nz, ny, N))
        am = amap[y0:y1]
        labels = np.unique(am)
        lm = np.searchsorted(labels, am) + 1
        sp = pipeline.area_lists(lm, len(labels))
        th = [thr["thresO2"][l - 1] for l in labels]
        te = [thr["testO2"][l - 1] for l in labels]
        halves.append(dict(ctx=c, sub=sub, lm=lm, n=len(labels), sp=sp, th=th, te=te,
                           out=ctx.empty((Nz, ny, N), np.float32), drv=GreedyPCA(c)))
    ctx.sync()

    def half(hh):
        pipeline.greedy_pca(hh["ctx"], hh["sub"], hh["lm"], hh["n"], hh["th"], hh["te"], spx=hh["sp"],
                            driver=hh["drv"], out=hh["out"])
        hh["ctx"].sync()

    def both():
        ts = [threading.Thread(target=half, args=(hh,)) for hh in halves]
        [t_.start() for t_ in ts]
        [t_.join() for t_ in ts]
    for _ in range(2):
        both()
    t = time.perf_counter()
    for _ in range(5):
        both()
    t_both = (time.perf_counter() - t) / 5
    t = time.perf_counter()
    for _ in range(5):
        half(halves[0])
        half(halves[1])
    t_seq = (time.perf_counter() - t) / 5
    print(f"two halves at the same time (two contexts, two threads): {1e3 * t_both:.2f} ms "
          f"({halves[0]['drv'].iterations} / {halves[1]['drv'].iterations} iterations)")
    print(f"two halves one after the other: {1e3 * t_seq:.2f} ms")


if __name__ == "__main__":
    main()
