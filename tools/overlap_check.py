import ctypes as C, sys, os, threading, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from origin_amd import _capi
from origin_amd.device import Context, DeviceArray
a = Context(0); b = Context(0)
n = 3681*600*600
rng = np.random.default_rng(0)
src = a.to_device(rng.standard_normal(n//4).astype(np.float32))
raw = DeviceArray(a, (n//4*8,), np.uint8)
# latency-bound work on ctx b: Lanczos on 18 matrices n=300
nm, nn = 18, 300
ld = 304
G = np.zeros((nm, ld, ld))
for i in range(nm):
    X = rng.standard_normal((400, nn)); G[i,:nn,:nn] = X.T@X
dG = b.to_device(G); goff = b.to_device(np.arange(nm, dtype=np.int64)*ld*ld)
dld = b.to_device(np.full(nm, ld, np.int64)); dn = b.to_device(np.full(nm, nn, np.int64))
qrows = _capi.load().origin_pca_eig_qrows()
qoff = b.to_device(np.arange(nm, dtype=np.int64)*qrows*ld); voff = b.to_device(np.arange(nm, dtype=np.int64)*ld)
dv = DeviceArray(b, (nm*ld,), np.float64)
def enc(reps):
    for _ in range(reps):
        _capi.call("origin_fits_encode", a.handle, src.p, 0, n//4, -64, raw.p)
    a.sync()
def eig(reps):
    for _ in range(reps):
        _capi.call("origin_pca_eig", b.handle, dG.p, goff.p, dld.p, dn.p, nm, nm*qrows*ld, qoff.p, dv.p, voff.p, None)
    b.sync()
enc(2); eig(2)
t=time.perf_counter(); enc(40); te=time.perf_counter()-t
t=time.perf_counter(); eig(10); tg=time.perf_counter()-t
t=time.perf_counter()
th=threading.Thread(target=eig, args=(10,)); th.start(); enc(40); th.join()
tb=time.perf_counter()-t
print(f"encode x40 alone {te*1e3:.1f} ms; eig x10 alone {tg*1e3:.1f} ms; both concurrently {tb*1e3:.1f} ms (sum {1e3*(te+tg):.1f})")
