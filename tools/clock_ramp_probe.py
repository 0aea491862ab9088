"""Does a kernel that follows a quiet stretch run slower than the same kernel in a busy stream?

    python tools/clock_ramp_probe.py [size=600]

Times the GLR (two long matrix-core kernels) back to back, then after host-side pauses of
0.2 .. 50 ms with the device idle, then behind a stretch of tiny kernels (the shape of the greedy
PCA's tail: ~10 us kernels, device mostly idle).  The step's GLR kernels run ~1 ms slower each than
in tools/glr_only.py; this separates "clock ramps down when the device is quiet" from the rest.
(A resident wave that sleeps and polls a flag does not keep the clock up -- same numbers with it --
and waves that spin on the VALU take the CUs away from the GLR's workgroups: 22 ms with 32 of them,
10 x slower with one per CU.  Measured in round 3, not kept.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels, synth  # noqa: E402
from origin_amd.device import Context  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    Nz = 3681
    ctx = Context(0)
    rng = np.random.default_rng(0)
    cube = ctx.empty((Nz, size, size), np.float32)
    plane = size * size
    for z0 in range(0, Nz, 64):
        n = min(64, Nz - z0)
        cube.view(z0 * plane, (n, size, size)).upload(
            rng.standard_normal((n, size, size), dtype=np.float32))
    psf = synth.moffat_psf(Nz).astype(np.float64)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, synth.dico_fwhm(), pcut=1e-8,
                           pmeansub=True, precision="f16x2")
    out = plan.run(cube)
    ctx.sync()

    def one():
        ctx.prof_reset()
        ctx.prof_enable(True)
        t = time.perf_counter()
        plan.run(cube, correl=out["correl"], profile=out["profile"], correl_min=out["correl_min"])
        ctx.sync()
        dt = time.perf_counter() - t
        ctx.prof_enable(False)
        r = ctx.prof_report()
        return 1e3 * dt, r["glr_spatial"][0], r["glr_spectral"][0]

    for _ in range(3):
        one()
    print("back to back      :", ["%.2f / %.2f / %.2f" % one() for _ in range(3)])
    for pause in (0.0002, 0.001, 0.005, 0.02, 0.05, 0.2):
        res = []
        for _ in range(3):
            time.sleep(pause)
            res.append("%.2f / %.2f / %.2f" % one())
        print(f"after {1e3 * pause:6.1f} ms idle:", res)
    # a stretch of tiny dependent kernels (small device-to-device copies), ~6 ms
    small = ctx.empty((1024,), np.float32)
    small2 = ctx.empty((1024,), np.float32)
    res = []
    for _ in range(3):
        for _ in range(600):
            small2.copy_from(small)
        res.append("%.2f / %.2f / %.2f" % one())
    print("after 600 tiny copies:", res)
    print("(wall ms / spatial ms / spectral ms)")


if __name__ == "__main__":
    main()
