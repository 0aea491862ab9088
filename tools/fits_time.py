"""Timing of the FITS dump path: device encode kernel alone, whole
``fitsio.write_image`` / ``read_image`` to a RAM-backed file, and the host conversion the
reference pays (NumPy ``astype('>f8')``, what astropy.io.fits does when mpdaf writes a cube).

    python tools/fits_time.py [N] [dir]      # cube 3681 x N x N float32, default N=600, /dev/shm
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import _capi, fitsio  # noqa: E402
from origin_amd.device import DeviceArray, default_context  # noqa: E402
from origin_amd.steps import LazyCube  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
out = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
ctx = default_context(0)
shape = (3681, N, N)
n = int(np.prod(shape))
rng = np.random.default_rng(0)
plane = rng.standard_normal((64, N, N)).astype(np.float32)
host = np.concatenate([plane] * (shape[0] // 64 + 1))[:shape[0]]
dev = ctx.to_device(host)
raw = DeviceArray(ctx, (n * 8,), np.uint8)
for bitpix in (-64, -32):
    _capi.call("origin_fits_encode", ctx.handle, dev.p, 0, n, bitpix, raw.p)
    ctx.sync()
    ctx.timer_start(0)
    for _ in range(5):
        _capi.call("origin_fits_encode", ctx.handle, dev.p, 0, n, bitpix, raw.p)
    ctx.timer_stop(0)
    ms = ctx.timer_ms(0) / 5
    by = n * (4 + abs(bitpix) // 8)
    print(f"encode float32 -> BITPIX {bitpix}: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s "
          f"({n / ms / 1e6:.2f} Gvoxel/s)")
    ctx.timer_start(0)
    for _ in range(5):
        _capi.call("origin_fits_decode", ctx.handle, raw.p, bitpix, n, 0, dev.p)
    ctx.timer_stop(0)
    ms = ctx.timer_ms(0) / 5
    print(f"decode BITPIX {bitpix} -> float32: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s")
raw.free()
path = os.path.join(out, "origin_fits_time.fits")
t = time.perf_counter()
fitsio.write_image(path, LazyCube(dev, dtype=np.float64), ctx=ctx)
tw = time.perf_counter() - t
size = os.path.getsize(path)
print(f"write_image float64 file ({size / 1e9:.2f} GB) to {out}: {tw:.2f} s  {size / tw / 1e9:.2f} GB/s")
t = time.perf_counter()
back, _ = fitsio.read_image(path, dtype=np.float32, ctx=ctx)
tr = time.perf_counter() - t
print(f"read_image -> float32 device array: {tr:.2f} s  {size / tr / 1e9:.2f} GB/s")
os.remove(path)
# host conversion of the reference's path on a bounded sample (one thread, like astropy)
sample = host[:256].astype(np.float64)
t = time.perf_counter()
be = sample.astype(">f8")
tc = time.perf_counter() - t
print(f"host astype('>f8') of {sample.size / 1e6:.0f} Mvoxel float64: {tc:.2f} s "
      f"{sample.size / tc / 1e6:.0f} Mvoxel/s  (x{n / sample.size:.1f} for the cube: "
      f"{tc * n / sample.size:.1f} s)")
