"""GLR of a two-field weighted mosaic on a random cube: wall time per run (spatial stage on the
matrix cores per field, spectral stage = the fp32 kernel with the explicit norm cube).
python tools/glr_weighted_time.py [size] [nz] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import kernels, synth  # noqa: E402
from origin_amd.device import default_context  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    nz = int(sys.argv[2]) if len(sys.argv) > 2 else 3681
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = default_context(0)
    rng = np.random.default_rng(0)
    x = np.linspace(0, 1, n)[None, :] * np.ones((n, 1))
    ws = [(0.2 + 0.6 * x), 1.0 - (0.2 + 0.6 * x)]
    psfs = [synth.moffat_psf(nz, 25, fwhm0=3.6, fwhm1=3.0).astype(np.float64),
            synth.moffat_psf(nz, 25, fwhm0=3.2, fwhm1=2.8).astype(np.float64)]
    cube = ctx.empty((nz, n, n), np.float32)
    plane = rng.standard_normal((64, n, n)).astype(np.float32)
    for z0 in range(0, nz, 64):
        m = min(64, nz - z0)
        cube.view(z0 * n * n, (m, n, n)).upload(plane[:m])
    for label, w, p in (("one field, no weights", None, psfs[0]), ("two weighted fields", ws, psfs)):
        plan = kernels.GLRPlan(ctx, (nz, n, n), p, w, synth.dico_fwhm(20), 1e-8, True)
        out = plan.run(cube, mask=None, want_maps=True)   # (first run: norm cube of a weighted plan)
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            out = plan.run(cube, mask=None, want_maps=True, correl=out["correl"],
                           profile=out["profile"], correl_min=out["correl_min"])
        ctx.sync()
        dt = (time.perf_counter() - t) / reps
        print(f"{label}: {1e3 * dt:.1f} ms per run ({nz * n * n / dt / 1e9:.1f} Gvoxel/s), precision "
              f"{plan.precision}, spatial on matrix cores {plan.spatial_on_matrix_cores}, spectral "
              f"{plan.spectral_on_matrix_cores}")
        plan.close()


if __name__ == "__main__":
    main()
