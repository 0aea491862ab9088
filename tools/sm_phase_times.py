"""Phase timing of spectral_mfma2_kernel's FOLD tile loop (variant library built with -DSM_TIMING:
ORIGIN_HIPCC_FLAGS=-DSM_TIMING python -m origin_amd.build --force): clock64 stamps of block (3, 1),
tiles 2..9, per wave: tile start, window ready, [pairs done, drained, stores issued] x 2 halves."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels, synth, _capi
from origin_amd.device import Context
ctx = Context(0)
Nz, size = 3681, 600
rng = np.random.default_rng(0)
cube = ctx.empty((Nz, size, size), np.float32)
blk = rng.standard_normal((64, size, size), dtype=np.float32)
for z0 in range(0, Nz, 64):
    n = min(64, Nz - z0)
    cube.view(z0 * size * size, (n, size, size)).upload(blk[:n])
plan = kernels.GLRPlan(ctx, cube.shape, synth.moffat_psf(Nz).astype(np.float64), None, synth.dico_fwhm(), pcut=1e-8, precision="f16x2")
mask = ctx.zeros(cube.shape, np.uint8)
for _ in range(3):
    out = plan.run(cube, mask)
ctx.sync()
NW = 12
buf = (C.c_longlong * (8 * NW * 8))()
lib = _capi.load()
lib.origin_debug_sm_timing.argtypes = [C.c_void_p]
print("rc", lib.origin_debug_sm_timing(buf))
t = np.array(buf[:], dtype=np.int64).reshape(8, NW, 8)
names = ["window", "pairs0", "drain0", "store0", "pairs1", "drain1", "store1"]
print("per wave and tile, cycles: " + ", ".join(names) + " | tile total")
for w in range(NW):
    rows = []
    for ti in range(7):
        d = np.diff(t[ti, w, :8])
        tot = t[ti + 1, w, 0] - t[ti, w, 0]
        rows.append(list(d[:7]) + [tot])
    r = np.median(np.array(rows), axis=0).astype(int)
    print(f"wave {w:2d} (SIMD {w % 4}): " + " ".join(f"{v:6d}" for v in r[:7]) + f" | {r[7]:6d}")
allr = []
for w in range(NW):
    for ti in range(7):
        allr.append(list(np.diff(t[ti, w, :8])[:7]) + [t[ti + 1, w, 0] - t[ti, w, 0]])
m = np.median(np.array(allr), axis=0)
print("median over waves and tiles:", " ".join(f"{n}={int(v)}" for n, v in zip(names + ["tile"], m)))
print("a tile is 240 MFMAs = 7 680 matrix-pipe cycles per wave, three waves per SIMD")
wb = (C.c_longlong * (4 * NW * 4))()
lib.origin_debug_sm_waves.argtypes = [C.c_void_p]
print("rc", lib.origin_debug_sm_waves(wb))
wt = np.array(wb[:], dtype=np.int64).reshape(4, NW, 4)
for b in range(4):
    t0 = wt[b, :, 0].min()
    print(f"block {b + 3}: wave (start, end, columns) relative to the block's first wave:")
    print("   " + "  ".join(f"w{w}:{int(wt[b, w, 0] - t0)}-{int(wt[b, w, 1] - t0)}/{int(wt[b, w, 2])}" for w in range(NW)))
    print(f"   block lifetime {int(wt[b, :, 1].max() - t0)}, sum of wave lifetimes / (12 x lifetime) = "
          f"{float((wt[b, :, 1] - wt[b, :, 0]).sum()) / (NW * float(wt[b, :, 1].max() - t0)):.2f}")
    dt_ticks = (wt[b, :, 1] - wt[b, :, 0]).astype(float)
    dt_wall = wt[b, :, 3].astype(float)          # 100 MHz ticks
    ok = dt_wall > 0
    if ok.any():
        mhz = dt_ticks[ok] / dt_wall[ok] * 100.0
        print(f"   clock64 ticks per microsecond over the waves' lifetimes: {mhz.min():.0f} .. {mhz.max():.0f} "
              f"(clock64 against the 100 MHz wall clock: the shader clock in MHz if clock64 counts shader cycles); "
              f"block lifetime {float(wt[b, :, 3].max()) / 100.0:.1f} us")
