# Step-7 thresholding (origin_where_above) on 3681x600x600 device cubes: time per call for a
# sparse local-maximum cube at several thresholds, next to one HBM pass over the cube.
import sys, time, numpy as np
sys.path.insert(0, '.')
from origin_amd.device import default_context
from origin_amd import kernels
ctx = default_context(0)
Nz, N = 3681, 600
rng = np.random.default_rng(0)
a = ctx.empty((Nz, N, N), np.float32); p = ctx.empty((Nz, N, N), np.uint8)
for z0 in range(0, Nz, 64):
    n = min(64, Nz - z0)
    v = np.abs(rng.standard_normal((n, N, N), dtype=np.float32)) * 3 + 2
    v[rng.random((n, N, N), dtype=np.float32) > 0.03] = 0      # ~1/27 of the voxels are maxima
    a.view(z0 * N * N, (n, N, N)).upload(v)
    p.view(z0 * N * N, (n, N, N)).upload(rng.integers(0, 20, (n, N, N)).astype(np.uint8))
gb = a.nbytes / 1e9
for thr in (16.0, 12.0, 8.0, 0.0):
    cap = max(1 << 20, kernels.where_above(ctx, a, thr, aux=p)["z"].size)
    ctx.sync(); t = time.perf_counter()
    for _ in range(5):
        w = kernels.where_above(ctx, a, thr, aux=p, cap=cap)
    ctx.sync(); ms = (time.perf_counter() - t) / 5 * 1e3
    print("where_above thr %5.1f: %9d detections  %7.2f ms per call (host arrays out; cube %.2f GB -> %.2f TB/s if one pass)"
          % (thr, w["z"].size, ms, gb, gb / ms))
