"""GLR stage alone on a random cube (profiling driver).

    python tools/glr_only.py [size=600] [precision=f16x2|f32] [reps=2] [nz=3681]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels, synth  # noqa: E402
from origin_amd.device import Context  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    prec = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    Nz = int(sys.argv[4]) if len(sys.argv) > 4 else 3681
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
                           pmeansub=True, precision=prec)
    out = plan.run(cube)
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t = time.perf_counter()
    for _ in range(reps):
        plan.run(cube, correl=out["correl"], profile=out["profile"], correl_min=out["correl_min"])
    ctx.sync()
    dt = (time.perf_counter() - t) / reps
    ctx.prof_enable(False)
    print("fold (eps, active):", plan.fold_eps())
    print(plan.precision, f"{1e3 * dt:.2f} ms per GLR",
          {k: round(v[0] / reps, 3) for k, v in ctx.prof_report().items()})


if __name__ == "__main__":
    main()
