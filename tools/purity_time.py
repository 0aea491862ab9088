import sys, time, numpy as np
sys.path.insert(0, '.')
from origin_amd.device import default_context
from origin_amd import kernels, lib_origin
ctx = default_context(0)
Nz, N = 3681, 600
rng = np.random.default_rng(0)
a = ctx.empty((Nz, N, N), np.float32); b = ctx.empty((Nz, N, N), np.float32)
for z0 in range(0, Nz, 64):
    n = min(64, Nz - z0)
    for arr in (a, b):
        v = np.abs(rng.standard_normal((n, N, N), dtype=np.float32)) * 3 + 2
        v[rng.random((n, N, N), dtype=np.float32) > 0.02] = 0
        arr.view(z0 * N * N, (n, N, N)).upload(v)
seg = np.zeros((N, N), int); seg[100:200, 300:420] = 1
lib_origin.Compute_threshold_purity(0.9, a, b, seg)
ctx.sync(); t = time.perf_counter()
for _ in range(3):
    thr, res = lib_origin.Compute_threshold_purity(0.9, a, b, seg)
ctx.sync(); print("purity threshold on 3681x600x600 device cubes: %.2f ms" % ((time.perf_counter() - t) / 3 * 1e3), thr, res["Det_M"][:3])
