import sys, numpy as np, time
sys.path.insert(0, '.')
from origin_amd.device import default_context
from origin_amd import kernels, synth
from oracle import cpu_ref
ctx = default_context(0)
rng = np.random.default_rng(3)
Nz, Ny, Nx = 300, 40, 52
cube = rng.normal(size=(Nz, Ny, Nx)).astype(np.float32)
cube[100:110, 10:14, 20:25] += 30.0
cube[:, 30, 30] *= 1e-3
prof = synth.dico_fwhm()
psf = synth.moffat_psf(Nz).astype(np.float64)
ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=4, pcut=1e-8, pmeansub=True)
d = ctx.to_device(cube)
mask = np.zeros(cube.shape, np.uint8); mask[5:9, 3, :] = 1
dm = ctx.to_device(mask)
for prec in ("f32", "f16x2"):
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, pcut=1e-8, pmeansub=True, precision=prec)
    out = plan.run(d, mask=None)
    c, p, cm = out["correl"].to_host(), out["profile"].to_host(), out["correl_min"].to_host()
    print(prec, plan.precision, "correl err", np.abs(c - ref[0]).max(), "min err", np.abs(cm - ref[2]).max(),
          "profile mismatch", np.mean(p != ref[1]), "maxmap err", np.abs(out["maxmap"].to_host() - ref[0].max(axis=0)).max(),
          "scale", np.abs(ref[0]).max())
