"""Phase timing of spatial2_kernel (variant library built with -DS2_TIMING): clock64 stamps of one
block (1,1,0), phases 10..25, per wave: phase start, MFMA start, MFMA end, work end."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels, synth, _capi
from origin_amd.device import Context
ctx = Context(0)
Nz, size = 256, 600
rng = np.random.default_rng(0)
cube = ctx.to_device(rng.standard_normal((Nz, size, size), dtype=np.float32))
plan = kernels.GLRPlan(ctx, cube.shape, synth.moffat_psf(Nz).astype(np.float64), None, synth.dico_fwhm(), pcut=1e-8, precision="f16x2")
out = plan.run(cube, None)
ctx.sync()
buf = (C.c_longlong * (16 * 8 * 8))()
lib = _capi.load()
lib.origin_debug_s2_timing.argtypes = [C.c_void_p]
print("rc", lib.origin_debug_s2_timing(buf))
t = np.array(buf[:], dtype=np.int64).reshape(16, 8, 8)
t0 = t[0, :, 0].min()
for p in range(15):
    row = []
    for w in range(8):
        s, m0, m1, e, c4, c5 = t[p, w, :6]
        nxt = t[p + 1, w, 0]
        if m1 > 0 and m0 >= s and m1 <= e:
            row.append(f"w{w} MFMA: pre {m0 - s:5d} mfma {m1 - m0:6d} post {e - m1:5d} bar {nxt - e:5d}")
        else:
            row.append(f"w{w} conv: scale {c4 - s:5d} image {c5 - c4:5d} table {e - c5:5d} bar {nxt - e:5d}")
    print(f"phase {p + 10} (start +{t[p, :, 0].min() - t0:7d}):", " | ".join(row[:1] + row[4:5]))
    print("      all waves work:", [int(t[p, w, 3] - t[p, w, 0]) for w in range(8)])
