"""Two weighted fields at 3681 x N x N: the spectral stage with the FOLD form on the norm cube
(NORMW) and with the two-product kernel everywhere (ORIGIN_GLR_NO_FOLD=1).

    python tools/glr_weighted_fold_time.py [size=600]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels, synth  # noqa: E402
from origin_amd.device import Context  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    Nz = 3681
    ctx = Context(0)
    rng = np.random.default_rng(0)
    cube = ctx.empty((Nz, N, N), np.float32)
    for z0 in range(0, Nz, 64):
        n = min(64, Nz - z0)
        cube.view(z0 * N * N, (n, N, N)).upload(rng.standard_normal((n, N, N), dtype=np.float32))
    x = np.linspace(0, 1, N)[None, :] * np.ones((N, 1))
    ws = [(0.3 + x) / 1.4, (1.1 - x) / 1.4]
    psfs = [synth.moffat_psf(Nz, 25, fwhm0=3.6 - 0.3 * f, fwhm1=3.0 + 0.2 * f).astype(np.float64)
            for f in range(2)]
    plan = kernels.GLRPlan(ctx, cube.shape, psfs, ws, synth.dico_fwhm(), pcut=1e-8, pmeansub=True,
                           precision="f16x2")
    t = time.perf_counter()
    out = plan.run(cube)
    ctx.sync()
    print(f"first run (norm cube + eps measurement): {1e3 * (time.perf_counter() - t):.1f} ms; "
          f"fold (eps, active): {plan.fold_eps()}")
    for env in (None, "1"):
        if env:
            os.environ["ORIGIN_GLR_NO_FOLD"] = env
        else:
            os.environ.pop("ORIGIN_GLR_NO_FOLD", None)
        plan.run(cube, correl=out["correl"], profile=out["profile"], correl_min=out["correl_min"])
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        t = time.perf_counter()
        for _ in range(3):
            plan.run(cube, correl=out["correl"], profile=out["profile"],
                     correl_min=out["correl_min"])
        ctx.sync()
        dt = (time.perf_counter() - t) / 3
        ctx.prof_enable(False)
        print("two-product kernel" if env else "FOLD on the norm cube", f"{1e3 * dt:.2f} ms per run",
              {k: round(v[0] / 3, 3) for k, v in ctx.prof_report().items()})
    plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
