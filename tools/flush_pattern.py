"""Is the final F = X - U C pass of the greedy PCA slowed by area boundaries that cut cache
lines?  With thresholds at +inf the PCA does nothing but that pass (a copy through the area
lists); timed for 100-wide areas (rows of 400 B: boundaries inside 128-B lines) and 128-wide
areas (line-aligned) on fields of the same size.   python tools/flush_pattern.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import pipeline, synth  # noqa: E402
from origin_amd.device import default_context  # noqa: E402
from origin_amd.pca import GreedyPCA  # noqa: E402

ctx = default_context(0)
Nz, N = 3681, 640
rng = np.random.default_rng(0)
plane = rng.standard_normal((32, N, N)).astype(np.float32)
cube = ctx.to_device(np.concatenate([plane] * (Nz // 32 + 1))[:Nz])
out = ctx.empty((Nz, N, N), np.float32)
for size in (100, 128, 160):
    amap, nb = synth.grid_areamap(N, N, size)
    spx = pipeline.area_lists(amap, nb)
    o2 = ctx.to_device(np.ones(N * N))
    drv = GreedyPCA(ctx)
    thr = [1e30] * nb
    for rep in range(3):
        ctx.sync()
        t = time.perf_counter()
        drv.run(out, spx, None, thr, 50, 100, test_map=o2, want_map=False, src=cube)
        ctx.sync()
        dt = time.perf_counter() - t
    gb = 2 * 4.0 * Nz * N * N / 1e9
    print(f"areas {size:3d} wide ({nb:2d}): copy pass {dt * 1e3:6.2f} ms  {gb / dt / 1e3:5.2f} TB/s")
