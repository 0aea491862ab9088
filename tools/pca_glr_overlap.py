"""Would a GLR started in the shadow of the greedy PCA's tail pay?  (measurement for DESIGN 8)

Two contexts = two streams of one process.  Stream A runs the greedy PCA of the 3681x600x600
bench field (host-driven loop, ~25 ms, its last ~6 ms one straggler area's one-block kernels).
Stream B runs a GLR of ANOTHER cube of the same size (~20 ms), either right away or behind a
delay of `delay_ms` made of one-block eigen-solves (so that it starts when the PCA enters its
tail).  Printed: PCA alone, GLR alone, both, and the time the PCA thread itself took in the
concurrent run."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels, pipeline, synth, _capi
from origin_amd.device import Context, DeviceArray

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
delay_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 17.0
Nz = 3681
# ORIGIN_OVERLAP_PRIO=1: the PCA's context on a high-priority queue, the GLR's on a low-priority one
if os.environ.get("ORIGIN_OVERLAP_PRIO") == "1":
    os.environ["ORIGIN_CTX_PRIORITY"] = "high"
    a = Context(0)
    os.environ["ORIGIN_CTX_PRIORITY"] = "low"
    b = Context(0)
    del os.environ["ORIGIN_CTX_PRIORITY"]
elif os.environ.get("ORIGIN_OVERLAP_GLR_CUS"):
    # the GLR's stream restricted to the first n compute units: the rest stays free for the PCA
    os.environ["ORIGIN_CTX_PRIORITY"] = "high"
    a = Context(0)
    del os.environ["ORIGIN_CTX_PRIORITY"]
    os.environ["ORIGIN_CTX_CUS"] = os.environ["ORIGIN_OVERLAP_GLR_CUS"]
    b = Context(0)
    del os.environ["ORIGIN_CTX_CUS"]
else:
    a, b = Context(0), Context(0)
f = synth.SyntheticField(Nz, N, N)
raw, var, mask = f.arrays()
d_raw, d_var, d_mask = a.to_device(raw), a.to_device(var), a.to_device(mask.astype(np.uint8))
del raw, var
pre = pipeline.preprocess(a, d_raw, d_var, d_mask, want_cont=False)
thr = pipeline.pca_threshold(pre["o2_host"], f.areamap, f.nbAreas, 0.01)
spx = pipeline.area_lists(f.areamap, f.nbAreas)
faint = a.empty(pre["cube_std"].shape, np.float32)
drv = None
def pca():
    global drv
    _, _, _, drv = pipeline.greedy_pca(a, pre["cube_std"], f.areamap, f.nbAreas, thr["thresO2"],
                                       thr["testO2"], 50, 100, spx=spx, driver=drv,
                                       o2_dev=pre["o2"], out=faint)
    a.sync()
# stream B: GLR of another cube (the raw one: any data)
plan = kernels.GLRPlan(b, d_raw.shape, f.PSF.astype(np.float64), None, f.profiles, pcut=1e-8,
                       precision="f16x2")
out = plan.run(d_raw, mask=d_mask)
b.sync()
# delay on stream B: Lanczos solves of one 700-column matrix, one block each
rng = np.random.default_rng(0)
nn, ld = 700, 704
X = rng.standard_normal((900, nn)); G = np.zeros((ld, ld)); G[:nn, :nn] = X.T @ X
dG = b.to_device(G); z64 = b.to_device(np.zeros(1, np.int64))
dld = b.to_device(np.full(1, ld, np.int64)); dn = b.to_device(np.full(1, nn, np.int64))
qrows = _capi.load().origin_pca_eig_qrows()
dv = DeviceArray(b, (ld,), np.float64)
def eig(k):
    for _ in range(k):
        _capi.call("origin_pca_eig", b.handle, dG.p, z64.p, dld.p, dn.p, 1, qrows * ld, z64.p, dv.p, z64.p, None)
eig(2); b.sync()
t = time.perf_counter(); eig(10); b.sync(); one = (time.perf_counter() - t) / 10 * 1e3
k = max(0, int(round(delay_ms / one)))
def glr(delayed):
    if delayed:
        eig(k)
    plan.run(d_raw, mask=d_mask, correl=out["correl"], profile=out["profile"], correl_min=out["correl_min"])
    b.sync()
def timed(fn, reps=5):
    fn(); t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e3
t_pca = timed(pca)
t_glr = timed(lambda: glr(False))
t_delay = timed(lambda: (eig(k), b.sync()))
res = {}
for delayed in (False, True):
    tot, tp = [], []
    for _ in range(6):
        box = {}
        def run_pca():
            t0 = time.perf_counter(); pca(); box["pca"] = (time.perf_counter() - t0) * 1e3
        th = threading.Thread(target=run_pca)
        t0 = time.perf_counter(); th.start(); glr(delayed); th.join()
        tot.append((time.perf_counter() - t0) * 1e3); tp.append(box["pca"])
    res[delayed] = (np.median(tot[1:]), np.median(tp[1:]))
print(f"PCA alone {t_pca:.1f} ms, GLR alone {t_glr:.1f} ms (sum {t_pca + t_glr:.1f}); one-block delay {k} x {one:.2f} = {t_delay:.1f} ms")
print(f"both from t=0      : total {res[False][0]:.1f} ms, the PCA thread {res[False][1]:.1f} ms")
print(f"GLR behind the delay: total {res[True][0]:.1f} ms, the PCA thread {res[True][1]:.1f} ms  "
      f"(serial order would be {t_pca + t_glr:.1f}; perfect shadow {max(t_pca, t_delay + t_glr):.1f})")
