"""The sparse local-maximum pass (origin_local_max_sparse) on device cubes of the bench size: time
per call, algorithmic bandwidth (9 B per voxel read) and the density of maxima.  The kernel's
variants are chosen by environment variables read once per process (ORIGIN_LOCALMAX_SPARSE_V1,
ORIGIN_LOCALMAX_XCD, ORIGIN_LOCALMAX_PREFETCH): run once per variant.
    python tools/localmax_sparse_time.py [N] [Nz]"""
import os, sys, time
import numpy as np
from scipy import ndimage as ndi
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels, sparse
from origin_amd.device import Context

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 3681
ctx = Context(0)
rng = np.random.default_rng(0)
blk = 64
a = ctx.empty((Nz, N, N), np.float32)
b = ctx.empty((Nz, N, N), np.float32)
m = ctx.zeros((Nz, N, N), np.uint8)
# smoothed noise: one voxel in ~100 is a 3x3x3 maximum, like a GLR output
ha = ndi.gaussian_filter(rng.standard_normal((blk, N, N)), 1.2, mode="wrap").astype(np.float32)
hb = -np.abs(ndi.gaussian_filter(rng.standard_normal((blk, N, N)), 1.2, mode="wrap")).astype(np.float32)
for z0 in range(0, Nz, blk):
    n = min(blk, Nz - z0)
    a.view(z0 * N * N, (n, N, N)).upload(ha[:n])
    b.view(z0 * N * N, (n, N, N)).upload(hb[:n])
bufs = sparse.SparseBuffers(ctx, a.shape)
for _ in range(2):
    sm, sn = sparse.local_max_sparse(ctx, a, b, m, bufs)
ctx.sync()
t = time.perf_counter()
reps = 10
for _ in range(reps):
    sm, sn = sparse.local_max_sparse(ctx, a, b, m, bufs)
ctx.sync()
dt = (time.perf_counter() - t) / reps
vox = float(Nz) * N * N
tag = " ".join(f"{k[16:]}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("ORIGIN_LOCALMAX_"))
print(f"[{tag or 'defaults'}] {1e3 * dt:.3f} ms per call, {9 * vox / dt / 1e12:.2f} TB/s of algorithmic "
      f"bytes (9 B/voxel); maxima {sm.nnz / vox:.4f} / minima {sn.nnz / vox:.4f} of the voxels")
oa, ob = kernels.local_max(ctx, a, b, m, 3)
same = np.array_equal(sm.dense().window(0, 64, 0, 64), oa.window(0, 64, 0, 64)) and \
    np.array_equal(sn.dense().window(N - 64, N, N - 64, N), ob.window(N - 64, N, N - 64, N))
print("two 64 x 64 columns against the dense pass:", "identical" if same else "MISMATCH")
