import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from origin_amd import kernels
from origin_amd.device import default_context
ctx = default_context(0)
Nz, N = 3681, 600
rng = np.random.default_rng(0)
raw = ctx.empty((Nz, N, N), np.float32); var = ctx.empty((Nz, N, N), np.float32); mask = ctx.zeros((Nz, N, N), np.uint8)
pl = rng.standard_normal((64, N, N)).astype(np.float32) + 50
pv = (1 + rng.random((64, N, N))).astype(np.float32)
for z0 in range(0, Nz, 64):
    m = min(64, Nz - z0)
    raw.view(z0 * N * N, (m, N, N)).upload(pl[:m]); var.view(z0 * N * N, (m, N, N)).upload(pv[:m])
coef = ctx.empty((11, N, N), np.float64); zs = ctx.empty((Nz,), np.float64); zc = ctx.empty((Nz,), np.float64)
def t(f, n=5):
    f(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    ctx.sync()
    return 1e3 * (time.perf_counter() - t0) / n
print("dct_fit (no fold)      %.3f ms" % t(lambda: kernels.dct_fit(ctx, raw, var, mask, 10, False, coef=coef)))
print("dct_fit_sums (fold)    %.3f ms" % t(lambda: kernels.dct_fit_sums(ctx, raw, var, mask, 10, False, coef=coef, zsum=zs, zcnt=zc)))
print("dct_resid_sums         %.3f ms" % t(lambda: kernels.dct_resid_sums(ctx, raw, mask, coef, zsum=zs, zcnt=zc)))
# the whole preprocessing phase with the per-kernel times of the context's profiler
from origin_amd import pipeline
for want_cont in (False, True):
    pipeline.preprocess(ctx, raw, var, mask, want_cont=want_cont); ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(5):
        pre = pipeline.preprocess(ctx, raw, var, mask, want_cont=want_cont)
    ctx.sync()
    dt = 1e3 * (time.perf_counter() - t0) / 5
    ctx.prof_enable(False)
    print("preprocess(want_cont=%s) %.3f ms" % (want_cont, dt),
          {k: round(v[0] / 5, 3) for k, v in ctx.prof_report().items()})
    del pre
