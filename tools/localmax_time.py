"""compute_local_max on device cubes of the bench size: time per call and algorithmic bandwidth
(17 B per voxel) of the kernel forms (ORIGIN_LOCALMAX_FORM / ORIGIN_LOCALMAX_SCALAR are read once
per process: run once per form).      python tools/localmax_time.py [N] [Nz]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels
from origin_amd.device import Context

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 3681
ctx = Context(0)
rng = np.random.default_rng(0)
blk = 64
a = ctx.empty((Nz, N, N), np.float32)
b = ctx.empty((Nz, N, N), np.float32)
m = ctx.zeros((Nz, N, N), np.uint8)
ha = rng.standard_normal((blk, N, N)).astype(np.float32)
hb = -np.abs(rng.standard_normal((blk, N, N))).astype(np.float32)
for z0 in range(0, Nz, blk):
    n = min(blk, Nz - z0)
    a.view(z0 * N * N, (n, N, N)).upload(ha[:n])
    b.view(z0 * N * N, (n, N, N)).upload(hb[:n])
oa, ob = ctx.empty(a.shape, np.float32), ctx.empty(a.shape, np.float32)
for _ in range(2):
    kernels.local_max(ctx, a, b, m, 3, out_max=oa, out_min=ob)
ctx.sync()
t = time.perf_counter()
reps = 10
for _ in range(reps):
    kernels.local_max(ctx, a, b, m, 3, out_max=oa, out_min=ob)
ctx.sync()
dt = (time.perf_counter() - t) / reps
vox = float(Nz) * N * N
print(f"form {os.environ.get('ORIGIN_LOCALMAX_FORM', '0')} scalar {os.environ.get('ORIGIN_LOCALMAX_SCALAR', '-')}: "
      f"{1e3 * dt:.3f} ms per call, {17 * vox / dt / 1e12:.2f} TB/s of algorithmic bytes (17 B/voxel)")
# spot check against NumPy on a slab
from oracle import cpu_ref
sl = (slice(100, 108), slice(0, 40), slice(N - 44, N))
ha_ = a.window(0, 42, N - 48, N)[98:110].astype(np.float64)
hb_ = b.window(0, 42, N - 48, N)[98:110].astype(np.float64)
r0, r1 = cpu_ref.compute_local_max(ha_, hb_, np.zeros(ha_.shape, bool), 3)
g0 = oa.window(0, 42, N - 48, N)[98:110]
g1 = ob.window(0, 42, N - 48, N)[98:110]
# (the slab is cut in z, at its bottom row and at its left column: compare inside those cuts)
ok = np.array_equal(g0[1:-1, :41, 1:], r0[1:-1, :41, 1:]) and \
    np.array_equal(g1[1:-1, :41, 1:], r1[1:-1, :41, 1:])
print("spot check against the oracle (field corner slab):", "OK" if ok else "MISMATCH")
