"""GPU parity tests: the HIP path (through the ctypes C ABI) against the golden vectors
recorded from the reference and against the CPU oracle on the same seeded inputs.

Tolerances (fp32 device storage vs the float64 reference), stated per test:
  cube_std / cont : |d| <= 1e-5 * max(1, |x|)                       
  cube_faint      : rel-Frobenius <= 2e-6, max-abs <= 1e-4, mapO2 and nstop identical
  GLR             : |dT| <= 1e-4, argmax-profile mismatches <= 0.01 % of voxels
  local maxima    : bit exact on identical float32 inputs
"""
import os

import numpy as np
import pytest

from oracle import cpu_ref
from origin_amd import synth
from oracle import golden_cases as gc

pytestmark = pytest.mark.gpu


def load(name):
    return np.load(os.path.join(gc.GOLDEN_DIR, name + ".npz"))


@pytest.fixture(scope="module")
def hip():
    import origin_amd.lib_origin as lib
    return lib


@pytest.fixture(scope="module")
def ctx():
    from origin_amd.device import default_context
    return default_context(0)


def assert_close_scaled(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert err.max() <= tol, f"max scaled error {err.max():.3e} > {tol:.1e}"


def test_to_host_f64_is_the_exact_widening(ctx):
    """origin_d2h_f32_as_f64 (pinned staging, widened by the host pool while the next chunk is
    in flight) against to_host().astype(float64): one element, a non-multiple of the task piece,
    more than two 64-MiB chunks."""
    rng = np.random.default_rng(1)
    for n in (1, 1000, (1 << 18) + 17, 2 * (16 << 20) + 12345):
        x = rng.standard_normal(n).astype(np.float32)
        x[::7] *= 1e-30
        x[::11] *= 1e30
        d = ctx.to_device(x, np.float32)
        got = d.to_host_f64()
        assert got.dtype == np.float64 and np.array_equal(got, x.astype(np.float64))
    d64 = ctx.to_device(np.arange(5.0), np.float64)
    assert np.array_equal(d64.to_host_f64(), np.arange(5.0))
    # and the way in: float64 host array -> float32 device array = astype(float32)
    for n in (1, (1 << 18) + 17, 2 * (16 << 20) + 12345):
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-20, 20, n)
        assert np.array_equal(ctx.to_device(x, np.float32).to_host(), x.astype(np.float32))
    m = rng.random((3, 4, 5)) < 0.5
    assert np.array_equal(ctx.to_device(m, np.uint8).to_host(), m.astype(np.uint8))


# ------------------------------------------------------------------------------- DCT
@pytest.mark.parametrize("approx", [False, True])
def test_dct_residual_golden(hip, approx):
    g = load("g1_dct")
    inp = gc.g1_inputs()
    cont = hip.dct_residual(inp["raw"], 10, inp["var"], approx, inp["mask"])
    ref = g["cont_approx" if approx else "cont"]
    assert cont.shape == ref.shape and cont.dtype == np.float64
    assert_close_scaled(cont, ref, 1e-5)


@pytest.mark.parametrize("approx", [False, True])
def test_preprocessing_golden(ctx, approx):
    from origin_amd import pipeline
    g = load("g2_preproc_approx" if approx else "g2_preproc")
    inp = gc.g1_inputs()
    raw = ctx.to_device(inp["raw"], np.float32)
    var = ctx.to_device(inp["var"], np.float32)
    mask = ctx.to_device(inp["mask"].astype(np.uint8))
    out = pipeline.preprocess(ctx, raw, var, mask, 10, approx)
    cube_std = out["cube_std"].to_host()
    assert_close_scaled(cube_std, g["cube_std"], 1e-5)
    assert np.all(cube_std[inp["mask"]] == 0)                  # data[mask] = 0
    assert_close_scaled(out["cont_dct"].to_host(), g["cont_dct"], 1e-5)
    assert_close_scaled(out["ima_std"].to_host(), g["ima_std"], 1e-5)
    assert_close_scaled(out["ima_dct"].to_host(), g["ima_dct"], 1e-5)
    assert_close_scaled(out["o2"].to_host(), g["o2"], 1e-5)


def test_o2test_matches_numpy(hip):
    rng = np.random.default_rng(3)
    cube = rng.standard_normal((301, 7, 13)).astype(np.float32)
    got = hip.O2test(cube)
    assert got.shape == (7, 13)
    np.testing.assert_allclose(got, np.mean(cube.astype(float) ** 2, axis=0), rtol=1e-12)
    # (Nz, S) form used by Compute_PCA_threshold
    got2 = hip.O2test(cube.reshape(301, -1))
    np.testing.assert_allclose(got2, got.ravel(), rtol=0, atol=0)


def test_dct_exact_on_dct_spectra(hip):
    """Property: a spectrum lying in the span of the atoms is its own continuum."""
    Nz, Ny, Nx = 3681, 5, 70
    rng = np.random.default_rng(5)
    D = cpu_ref.DCTMAT(Nz, 10)
    coef = rng.standard_normal((11, Ny * Nx)) * 10
    raw = (D @ coef).reshape(Nz, Ny, Nx).astype(np.float32)
    var = (1 + rng.random((Nz, Ny, Nx))).astype(np.float32)
    mask = np.zeros((Nz, Ny, Nx), bool)
    for approx in (False, True):
        cont = hip.dct_residual(raw, 10, var, approx, mask)
        assert np.max(np.abs(cont - raw)) <= 2e-5 * np.max(np.abs(raw))


@pytest.mark.parametrize("shape,order,approx", [
    ((3681, 9, 70), 10, False), ((200, 24, 28), 10, False), ((37, 5, 13), 10, False),
    ((13, 3, 5), 10, False), ((611, 64, 64), 10, False), ((96, 40, 60), 10, False),
    ((300, 20, 33), 3, False), ((300, 20, 33), 12, False), ((300, 20, 33), 10, True)])
def test_dct_fit_sums_equals_the_two_calls(ctx, shape, order, approx):
    """origin_dct_fit_sums (per-channel sums folded into the moments pass, rows brought in by
    register batches) against origin_dct_fit + origin_dct_resid_sums and against NumPy: clean and masked
    rows, fully masked spaxels, S % 64 != 0, Nz below / not a multiple of
    the 16-row trip, fields small enough for several waves per spaxel group."""
    from origin_amd import kernels
    Nz, Ny, Nx = shape
    rng = np.random.default_rng(Nz + Nx)
    raw = (rng.standard_normal(shape) * 3 + 50 + 10 * np.linspace(0, 1, Nz)[:, None, None])
    var = 1.0 + rng.random(shape)
    mask = rng.random(shape) < 0.002
    mask[:, 0, :] = True                       # a fully masked image row
    mask[Nz // 3: Nz // 3 + 5, Ny - 1, Nx - 2] = True
    mask[Nz - 1, 1, 1] = True
    raw = raw.astype(np.float32)
    raw[mask] = 0                               # ORIGIN.init: data.filled(0), var.filled(inf)
    var = var.astype(np.float32)
    var[mask] = np.inf
    d_raw, d_var = ctx.to_device(raw, np.float32), ctx.to_device(var, np.float32)
    d_mask = ctx.to_device(mask.astype(np.uint8))
    c1 = kernels.dct_fit(ctx, d_raw, d_var, d_mask, order, approx)
    s1, n1 = kernels.dct_resid_sums(ctx, d_raw, d_mask, c1)
    c2, s2, n2 = kernels.dct_fit_sums(ctx, d_raw, d_var, d_mask, order, approx)
    coef2 = c2.to_host()
    assert np.array_equal(c1.to_host(), coef2)
    scale = np.abs(raw).sum(axis=(1, 2), dtype=float) + 1.0
    assert np.max(np.abs(s1.to_host() - s2.to_host()) / scale) <= 1e-14
    assert np.array_equal(n1.to_host(), n2.to_host())
    # NumPy: sum over the unmasked spaxels of raw - cont
    cont = kernels.dct_continuum(ctx, c2, Nz).to_host().astype(float)   # (float32 cube)
    ref = np.where(mask, 0.0, raw.astype(float) - cont).sum(axis=(1, 2))
    assert np.max(np.abs(s2.to_host() - ref) / scale) <= 1e-6
    assert np.array_equal(n2.to_host(), (~mask).sum(axis=(1, 2)).astype(float))


# ------------------------------------------------------------------------------- PCA
def test_gram_mfma_matches_numpy(ctx):
    """G = X^T X from v_mfma_f64_16x16x4_f64 with a deliberately asymmetric X."""
    from origin_amd import _capi
    rng = np.random.default_rng(11)
    Nz, n = 517, 77
    ld = (n + 15) // 16 * 16
    X = np.zeros((Nz, ld))
    X[:, :n] = rng.standard_normal((Nz, n)) * (1 + np.arange(n))[None, :]
    dX = ctx.to_device(X)
    T = (ld + 31) // 32
    iu, ju = np.triu_indices(T)
    d_ti = ctx.to_device(iu.astype(np.int32))
    d_tj = ctx.to_device(ju.astype(np.int32))
    d_ta = ctx.to_device(np.zeros(len(iu), np.int32))
    d_ld = ctx.to_device(np.array([ld], np.int64))
    d_off = ctx.to_device(np.array([0, 0], np.int64))
    G = ctx.zeros((ld, ld), np.float64)
    _capi.call("origin_pca_gram", ctx.handle, dX.p, d_off.p, d_ld.p, Nz, len(iu), d_ti.p, d_tj.p,
               d_ta.p, ld * ld, G.p, d_off.p)
    ref = X.T @ X
    got = G.to_host()
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))


@pytest.mark.parametrize("n,spectrum", [(1, "flat"), (2, "gap"), (20, "flat"), (37, "gap"),
                                        (48, "close"), (49, "gap"), (64, "close"), (90, "flat"),
                                        (96, "gap"), (97, "close"), (130, "close"),
                                        (150, "gap"), (200, "close"), (208, "flat"),
                                        (209, "close"), (256, "flat"), (257, "gap"),
                                        (300, "flat"), (400, "close"), (512, "flat"),
                                        (513, "gap"), (700, "close"), (120, "lowrank"),
                                        (230, "lowrank"), (300, "lowrank")])
def test_lanczos_leading_eigenvector(ctx, n, spectrum):
    """Device eigen-solvers vs LAPACK on PSD matrices with wide, close and flat spectra:
    repeated squaring on the f64 matrix cores for n <= 48 and n <= 96; above, plain Lanczos
    with the matrix resident on the CU (all of it up to 208 columns, the rest streamed; four
    rows per lane up to 256 columns, eight up to 512), and the round-2 kernel beyond 512."""
    from origin_amd import _capi
    rng = np.random.default_rng(100 + n)
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    if spectrum == "gap":
        lam = np.concatenate([[1000.0], rng.uniform(0.1, 10, n - 1)])
    elif spectrum == "close":
        lam = np.concatenate([[1.0, 0.999], rng.uniform(0.0, 0.9, n - 2)])
    elif spectrum == "lowrank":   # rank 2: the Krylov space is exhausted after two steps
        lam = np.concatenate([[1000.0, 3.0], np.zeros(n - 2)])
    else:
        lam = 1.0 + 0.05 * rng.random(n)
    lam = np.sort(lam)[::-1]
    A = (Qm * lam) @ Qm.T
    A = 0.5 * (A + A.T)
    ld = (n + 15) // 16 * 16
    G = np.zeros((ld, ld))
    G[:n, :n] = A
    rows = _capi.load().origin_pca_eig_qrows()
    dG = ctx.to_device(G)
    d0 = ctx.to_device(np.array([0], np.int64))
    dld = ctx.to_device(np.array([ld], np.int64))
    dn = ctx.to_device(np.array([n], np.int64))
    v = ctx.zeros((ld,), np.float64)
    info = ctx.zeros((3,), np.float64)
    _capi.call("origin_pca_eig", ctx.handle, dG.p, d0.p, dld.p, dn.p, 1, rows * ld, d0.p, v.p,
               d0.p, info.p)
    got = v.to_host()[:n]
    theta, resid, restarts = info.to_host()
    w, V = np.linalg.eigh(A)
    assert abs(theta - w[-1]) <= 1e-12 * w[-1]
    assert abs(np.linalg.norm(got) - 1) < 1e-12
    # residual of the returned pair and alignment with the LAPACK vector (gap permitting)
    r = np.linalg.norm(A @ got - w[-1] * got)
    assert r <= 1e-11 * w[-1], (r, resid, restarts)
    gap = (w[-1] - w[-2]) / w[-1] if n > 1 else 1.0
    assert 1 - abs(got @ V[:, -1]) <= 1e-10 / max(gap, 1e-12) ** 2 + 1e-13


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_greedy_pca_golden(hip, name):
    g = load("g4_pca")
    cube = gc.g4_inputs()[name]
    faint, mapO2, nstop = hip.Compute_GreedyPCA(cube, g[name + "_test"],
                                                float(g[name + "_thr"][0]), 50, 100)
    ref = g[name + "_faint"]
    assert np.array_equal(mapO2, g[name + "_mapO2"])
    assert nstop == int(g[name + "_nstop"])
    assert np.linalg.norm(faint - ref) <= 2e-6 * np.linalg.norm(ref)
    assert np.max(np.abs(faint - ref)) <= 1e-4


def test_greedy_pca_itermax_guard(hip):
    g = load("g4_pca")
    cube = gc.g4_inputs()["a"]
    faint, mapO2, nstop = hip.Compute_GreedyPCA(cube, g["a_test"], float(g["a_thr"][0]), 50, 2)
    assert nstop == 1 and np.array_equal(mapO2, g["a_it2_mapO2"])
    assert np.max(np.abs(faint - g["a_it2_faint"])) <= 1e-4


def test_greedy_pca_area_golden(hip):
    g = load("g4_pca")
    inp = gc.g4_inputs()
    cube, areamap, nb = inp["area_cube"], inp["areamap"], inp["nbAreas"]
    res = [hip.Compute_PCA_threshold(cube[:, areamap == i], 0.01) for i in range(1, nb + 1)]
    thr = [r[3] for r in res]
    np.testing.assert_allclose(thr, g["area_thr"], rtol=1e-6)
    faint, mapO2, nstop = hip.Compute_GreedyPCA_area(nb, cube, areamap, 50, list(g["area_thr"]),
                                                     100, [r[0] for r in res])
    ref = g["area_faint"]
    assert np.array_equal(mapO2, g["area_mapO2"]) and nstop == int(g["area_nstop"])
    assert np.linalg.norm(faint - ref) <= 2e-6 * np.linalg.norm(ref)
    assert np.max(np.abs(faint - ref)) <= 1e-4


def test_greedy_pca_pipelined_host_loop_is_identical(hip, monkeypatch):
    """ORIGIN_PCA_PIPELINED=1: the host builds each work list from the previous selection's counts
    and the selection patches the real ones in on the device (csrc/pca.hip).  Areas that finish
    at different iterations (2 x 2 areas golden) and a long single-area run: same cube, same map,
    bit for bit, as the synchronous order."""
    g = load("g4_pca")
    inp = gc.g4_inputs()
    cube, areamap, nb = inp["area_cube"], inp["areamap"], inp["nbAreas"]
    res = [hip.Compute_PCA_threshold(cube[:, areamap == i], 0.01) for i in range(1, nb + 1)]
    args = (nb, cube, areamap, 50, list(g["area_thr"]), 100, [r[0] for r in res])
    a = hip.Compute_GreedyPCA_area(*args)
    one = (inp["a"], g["a_test"], float(g["a_thr"][0]), 50, 100)
    a1 = hip.Compute_GreedyPCA(*one)
    monkeypatch.setenv("ORIGIN_PCA_PIPELINED", "1")
    b = hip.Compute_GreedyPCA_area(*args)
    b1 = hip.Compute_GreedyPCA(*one)
    for x, y in ((a, b), (a1, b1)):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2]
    assert np.array_equal(b[1], g["area_mapO2"])


# ------------------------------------------------------------------------------- GLR
@pytest.mark.parametrize("name", list("abcde"))
def test_glr_golden(hip, name):
    g = load("g5_glr")
    c = gc.g5_inputs()[name]
    correl, profile, correl_min = hip.Correlation_GLR_test(
        c["cube"], c["fsf"], c["weights"], c["profiles"], nthreads=1, pcut=c["pcut"],
        pmeansub=c["pmeansub"])
    assert profile.dtype == np.uint8
    assert np.max(np.abs(correl - g[name + "_correl"])) <= 1e-4
    assert np.max(np.abs(correl_min - g[name + "_correl_min"])) <= 1e-4
    assert np.mean(profile != g[name + "_profile"]) <= 1e-4


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", list("ade"))
def test_glr_golden_both_arithmetics(ctx, name, precision):
    """The spectral stage exists twice for plans without weight maps: matrix cores on a
    two-term f16 split (default) and the fp32 FMA kernel.  Both against the reference's output
    (case d has asymmetric profiles: the Toeplitz operand orientation shows there)."""
    from origin_amd import kernels
    g = load("g5_glr")
    c = gc.g5_inputs()[name]
    plan = kernels.GLRPlan(ctx, c["cube"].shape, c["fsf"], None, c["profiles"], c["pcut"],
                           c["pmeansub"], precision=precision)
    assert plan.precision == precision
    out = plan.run(ctx.to_device(c["cube"], np.float32), mask=None, want_maps=True)
    correl, cmin = out["correl"].to_host(), out["correl_min"].to_host()
    assert np.max(np.abs(correl - g[name + "_correl"])) <= 1e-4
    assert np.max(np.abs(cmin - g[name + "_correl_min"])) <= 1e-4
    assert np.mean(out["profile"].to_host() != g[name + "_profile"]) <= 1e-4
    assert np.max(np.abs(out["maxmap"].to_host() - g[name + "_correl"].max(axis=0))) <= 1e-4
    plan.close()


@pytest.mark.parametrize("shape,P", [((70, 67, 132), 25), ((40, 30, 20), 25), ((50, 26, 140), 9),
                                      ((33, 130, 260), 17), ((45, 70, 131), 25),
                                      ((37, 65, 66), 9), ((41, 66, 70), 13), ((36, 70, 133), 21),
                                      ((30, 66, 69), 5), ((32, 40, 72), 7), ((31, 67, 64), 11),
                                      ((33, 64, 130), 15), ((30, 70, 66), 19), ((34, 66, 71), 23)])
def test_glr_matrix_core_spatial_stage(ctx, shape, P):
    """Shapes that take the matrix-core spatial kernel (odd P from 5 to 25): several 64x64 regions,
    partial regions, fields narrower than a region, row lengths that are not multiples of four
    (element-wise tile loads and stores), against the float64 oracle and against the fp32
    kernels."""
    from origin_amd import kernels
    rng = np.random.default_rng(P + shape[2])
    Nz, Ny, Nx = shape
    cube = rng.standard_normal(shape).astype(np.float32)
    cube[Nz // 2, Ny // 2, Nx // 3] += 40.0
    psf = synth.moffat_psf(Nz, P).astype(np.float64)
    psf *= 1.0 + 0.3 * rng.random(psf.shape)          # asymmetric in x and y
    psf /= psf.sum(axis=(1, 2), keepdims=True)
    prof = synth.dico_fwhm(3)
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    d = ctx.to_device(cube)
    got = {}
    for prec in ("f16x2", "f32"):
        plan = kernels.GLRPlan(ctx, shape, psf, None, prof, 1e-8, True, precision=prec)
        # (a field smaller than the PSF has no border-class table: such plans stay on fp32)
        assert plan.spatial_on_matrix_cores == (plan.precision == "f16x2")
        assert plan.precision == (prec if min(Ny, Nx) >= P else "f32")
        out = plan.run(d, mask=None, want_maps=False)
        got[prec] = out["correl"].to_host()
        assert np.max(np.abs(got[prec] - ref[0])) <= 1e-4
        assert np.max(np.abs(out["correl_min"].to_host() - ref[2])) <= 1e-4
        assert np.mean(out["profile"].to_host() != ref[1]) <= 1e-4
        plan.close()
    assert np.max(np.abs(got["f16x2"] - got["f32"])) <= 1e-4


@pytest.mark.parametrize("shape,P,nf,nprof", [((60, 70, 132), 9, 2, 3), ((40, 66, 67), 25, 3, 3),
                                               ((48, 30, 140), 17, 2, 3), ((150, 34, 45), 9, 2, 20),
                                               ((130, 33, 40), 7, 2, 15)])
def test_glr_weighted_fields_on_matrix_cores(ctx, shape, P, nf, nprof):
    """A mosaic: several fields, each with its PSF and weight map (origin.py:600-609,
    lib_origin.py:1029-1031, :1134-1147).  The spatial stage runs per field on the matrix cores
    (weight map multiplied in while the tile is staged, fields accumulated), the norm cube is a
    constant of the plan (computed by the first run), the spectral stage convolves it next to
    the data -- a second Toeplitz product on the matrix cores, two launches over the halves of
    the dictionary when it has more than 13 profiles.  Against the float64 oracle, against the fp32 plan, and a second run of
    the same plan (cached norm cube) against the first."""
    from origin_amd import kernels
    rng = np.random.default_rng(P + shape[2] + nf)
    Nz, Ny, Nx = shape
    cube = rng.standard_normal(shape).astype(np.float32)
    cube[Nz // 2, Ny // 2, Nx // 3] += 40.0
    psfs, ws = [], []
    x = np.linspace(0, 1, Nx)[None, :] * np.ones((Ny, 1))
    y = np.linspace(0, 1, Ny)[:, None] * np.ones((1, Nx))
    raww = [0.2 + x, 1.2 - x, 0.1 + y * y][:nf]
    tot = sum(raww)
    for f in range(nf):
        p = synth.moffat_psf(Nz, P, fwhm0=3.6 - 0.4 * f, fwhm1=3.0 + 0.2 * f).astype(np.float64)
        p *= 1.0 + 0.2 * rng.random(p.shape)
        p /= p.sum(axis=(1, 2), keepdims=True)
        psfs.append(p)
        ws.append((raww[f] / tot).astype(np.float32).astype(np.float64))
    ws[0][:5, :7] = 0.0                     # a corner one field does not cover
    if nf == 2:
        ws[1][-4:, -6:] = 0.0               # and a corner NO field covers: den = 0 -> T = 0
    prof = synth.dico_fwhm(nprof)           # (more than 13 profiles: two launches, merged)
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psfs, ws, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    d = ctx.to_device(cube)
    got = {}
    for prec in ("f16x2", "f32"):
        plan = kernels.GLRPlan(ctx, shape, psfs, ws, prof, 1e-8, True, precision=prec)
        assert plan.precision == prec
        assert plan.spatial_on_matrix_cores == (prec == "f16x2")
        assert plan.spectral_on_matrix_cores == (prec == "f16x2")
        for rep in range(2):
            out = plan.run(d, mask=None, want_maps=True)
            c = out["correl"].to_host()
            if rep == 1:
                assert np.array_equal(c, got[prec])
            got[prec] = c
            assert np.max(np.abs(c - ref[0])) <= 1e-4
            assert np.max(np.abs(out["correl_min"].to_host() - ref[2])) <= 1e-4
            assert np.mean(out["profile"].to_host() != ref[1]) <= 1e-4
            assert np.max(np.abs(out["maxmap"].to_host() - ref[0].max(axis=0))) <= 1e-4
        plan.close()
    assert np.max(np.abs(got["f16x2"] - got["f32"])) <= 1e-4


@pytest.mark.parametrize("precision", ["f16x2", "bf16"])
def test_glr_weighted_fields_fold_form_on_the_norm_cube(ctx, monkeypatch, precision):
    """NORMW: a mosaic whose PSFs vary smoothly with the channel.  den_k[z, s] is the plan's norm
    cube smoothed by p_k^2, = norm[z, s] sum p_k^2 (1 + eps)^2 away from the cube's ends; the first
    run measures eps over every voxel and profile, and where it is <= 2e-6 the spectral stage is
    the FOLD form of the table kernel with rsq(norm) of each voxel behind the profile loop -- one
    Toeplitz product instead of two, one launch instead of two -- and the two-product kernel only
    for the 32 channels at either end.  Checked: eps and `active` as the plan reports them, the
    form against the two-product form of the same plan (ORIGIN_GLR_NO_FOLD=1) within eps, both
    against the float64 oracle, maps, a corner no field covers (T = 0), a second run bit for bit
    the first.  bf16 plans take the same road (one bf16 MFMA per product between the ends, the
    ends on the f16 split): bf16 bounds."""
    from scipy.ndimage import maximum_filter1d
    from origin_amd import kernels
    rng = np.random.default_rng(41)
    Nz, Ny, Nx, P = 330, 40, 70, 9
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    cube[150:190] *= 23.0
    x = np.linspace(0, 1, Nx)[None, :] * np.ones((Ny, 1))
    raww = [0.3 + x, 1.1 - x]
    tot = sum(raww)
    psfs = [synth.moffat_psf(3681, P, fwhm0=3.6 - 0.3 * f, fwhm1=3.0 + 0.2 * f)[:Nz].astype(np.float64)
            for f in range(2)]
    ws = [(raww[f] / tot).astype(np.float32).astype(np.float64) for f in range(2)]
    ws[0][:5, :7] = 0.0
    ws[1][-9:, -10:] = 0.0
    ws[0][-9:, -10:] = 0.0                    # a corner NO field covers: norm = 0 -> T = 0 where
    prof = synth.dico_fwhm(20)                # the whole 9 x 9 window lies in it
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psfs, ws, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    monkeypatch.delenv("ORIGIN_GLR_NO_FOLD", raising=False)
    plan = kernels.GLRPlan(ctx, cube.shape, psfs, ws, prof, 1e-8, True, precision=precision)
    assert plan.precision == precision
    assert plan.fold_eps()[1] is False        # (measured by the first run)
    d = ctx.to_device(cube)
    out = plan.run(d, mask=None, want_maps=True)
    eps, active = plan.fold_eps()
    assert active and 0.0 < eps <= 2e-6
    fold = {k: out[k].to_host() for k in ("correl", "correl_min", "profile", "maxmap", "minmap")}
    out = plan.run(d, mask=None, want_maps=True)
    assert np.array_equal(out["correl"].to_host(), fold["correl"])
    monkeypatch.setenv("ORIGIN_GLR_NO_FOLD", "1")
    out = plan.run(d, mask=None, want_maps=True)
    exact = {k: out[k].to_host() for k in ("correl", "correl_min", "profile", "maxmap", "minmap")}
    monkeypatch.delenv("ORIGIN_GLR_NO_FOLD")
    plan.close()
    local = maximum_filter1d(np.abs(ref[0]).max(axis=(1, 2)), size=193, mode="nearest")
    rnd = 1.2e-2 if precision == "bf16" else 3e-6   # (as in the test of the table kernel's FOLD)
    tol = ((eps + rnd) * local)[:, None, None]
    for k, r in (("correl", ref[0]), ("correl_min", ref[2])):
        assert np.all(np.abs(fold[k] - exact[k]) <= 2 * tol), k
        assert np.all(np.abs(fold[k] - r) <= tol), k
    assert np.all(fold["correl"][:, -5:, -6:] == 0.0)        # the uncovered corner's interior
    assert np.all(fold["correl_min"][:, -5:, -6:] == 0.0)
    assert np.max(np.abs(fold["maxmap"] - fold["correl"].max(axis=0))) == 0.0
    assert np.max(np.abs(fold["minmap"] - fold["correl_min"].min(axis=0))) == 0.0
    # (where no field reaches, every T is 0 and the index is whatever the oracle's FFT noise makes
    # it: compare the rest)
    sel = np.ones((Ny, Nx), bool)
    sel[-9:, -10:] = False
    if precision == "f16x2":
        assert np.mean(fold["profile"][:, sel] != ref[1][:, sel]) <= 1e-4
        assert np.mean(exact["profile"][:, sel] != ref[1][:, sel]) <= 1e-4


def test_glr_bf16_precision_meets_the_bf16_tolerance(ctx):
    """precision="bf16" (BASELINE config 4): one bf16 MFMA per product in the spectral stage.
    SURVEY 8c tolerance for bf16 operands with wide accumulation: |dT| <= 5e-2, rms <= 5e-3,
    arg-max mismatches <= 2 % and only where the two best profiles are closer than 5e-2."""
    from origin_amd import kernels
    rng = np.random.default_rng(12)
    Nz, Ny, Nx = 300, 40, 44
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    cube[150, 20, 22] += 30.0
    psf = synth.moffat_psf(Nz, 25).astype(np.float64)
    prof = synth.dico_fwhm(20)
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, 1e-8, True, precision="bf16")
    assert plan.precision == "bf16"
    out = plan.run(ctx.to_device(cube), mask=None, want_maps=True)
    T, Tmin, arg = out["correl"].to_host(), out["correl_min"].to_host(), out["profile"].to_host()
    d = T - ref[0]
    assert np.max(np.abs(d)) <= 5e-2 and np.sqrt(np.mean(d * d)) <= 5e-3
    assert np.max(np.abs(Tmin - ref[2])) <= 5e-2
    wrong = arg != ref[1]
    assert wrong.mean() <= 0.02
    # a wrong index only where the runner-up is within 5e-2: T at the device's index is that close
    # to the true maximum by the bound above
    assert np.max(np.abs(out["maxmap"].to_host() - ref[0].max(axis=0))) <= 5e-2
    plan.close()
    # and it is really a different arithmetic from the fp32-class path
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, 1e-8, True, precision="f16x2")
    T2 = plan.run(ctx.to_device(cube), mask=None, want_maps=False)["correl"].to_host()
    plan.close()
    assert np.max(np.abs(T2 - ref[0])) <= 1e-4 < np.max(np.abs(d))


@pytest.mark.parametrize("nprof,expect", [(20, "even"), (3, "odd tail"), (5, "mixed pair"),
                                          (7, "mixed pair + odd tail"), (2, "two"), (1, "one")])
def test_glr_matrix_core_profile_list_shapes(ctx, nprof, expect):
    """The spectral kernel's stage sequence is compiled per shape of the profile list (narrow
    profiles first, in pairs; one mixed pair; an odd one out): every variant against the
    oracle, with the first-maximum rule checked exactly by duplicating a profile."""
    from origin_amd import kernels
    rng = np.random.default_rng(nprof)
    Nz, Ny, Nx = 130, 26, 36
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    cube[:, 3, 4:9] = 0.0                 # spaxels of zeros: every T equal -> index 0
    psf = synth.moffat_psf(Nz, 25).astype(np.float64)
    prof = synth.dico_fwhm(nprof) if nprof > 1 else synth.dico_fwhm(3)[1:2]
    if nprof == 5:
        prof = list(prof) + [prof[1]]     # a duplicate: ties must go to the FIRST index
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, 1e-8, True, precision="f16x2")
    assert plan.precision == "f16x2"
    out = plan.run(ctx.to_device(cube), mask=None, want_maps=False)
    assert np.max(np.abs(out["correl"].to_host() - ref[0])) <= 1e-4
    assert np.max(np.abs(out["correl_min"].to_host() - ref[2])) <= 1e-4
    arg = out["profile"].to_host()
    sel = ref[0] > 0 if nprof == 5 else np.ones(ref[0].shape, bool)
    assert np.mean((arg != ref[1]) & sel) <= 1e-4
    if nprof == 5:
        # the duplicate (index 5) never beats index 1 where the maximum is positive; the kernel's
        # arg-max key breaks exact ties among NEGATIVE maxima towards the larger index (only
        # profiles whose T agree to 2^-18 are affected: glr_spectral_mfma.hip)
        assert not np.any((arg == 5) & (ref[0] > 0))
    plan.close()


@pytest.mark.parametrize("case", ["dictionary", "odd count", "wide first", "bf16", "P25 borders"])
def test_glr_fold_form_against_the_exact_form_and_the_oracle(ctx, case, monkeypatch):
    """FOLD of the matrix-core spectral stage (glr_spectral_mfma.hip): away from the cube's first
    and last 32 channels the pair loop compares accumulators that carry a_k = 1/sqrt(sum p_k^2)
    and the class factor s(z) is applied once, behind the loop -- T carries a relative error
    <= the plan's measured eps (<= 2e-6, here a few 1e-7).  The PSF is the first channels of the
    benchmark's 3681-channel FWHM ramp (the smoothness the plan's eps test asks for).  Checked:
    the plan reports FOLD active with eps under the limit; FOLD against the exact form of the
    same plan (ORIGIN_GLR_NO_FOLD=1) within eps of the window's largest |T| (+ the f16 split's
    rounding); both against the float64 oracle at the usual bounds; the arg-max rate.  Cases:
    the 20-profile dictionary (identity processing order), an odd number of profiles (the last
    one sits in LDS twice), wide profiles first (processing order != caller's order: look-up per
    pair), the bf16 arithmetic, and a 25 x 25 PSF on a field where most spaxels are border
    classes (each lane its own s)."""
    from scipy.ndimage import maximum_filter1d
    from origin_amd import kernels
    rng = np.random.default_rng(31)
    P = 25 if case == "P25 borders" else 9
    Nz, Ny, Nx = (330, 30, 44) if P == 25 else (330, 40, 70)
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    cube[100:140] *= 37.0                    # a scale step inside the FOLD range
    psf = synth.moffat_psf(3681, P)[:Nz].astype(np.float64)
    prof = synth.dico_fwhm(20)
    if case == "odd count":
        prof = synth.dico_fwhm(7)
    elif case == "wide first":
        prof = list(prof[::-1])
    precision = "bf16" if case == "bf16" else "f16x2"
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, 1e-8, True, precision=precision)
    assert plan.precision == precision
    monkeypatch.delenv("ORIGIN_GLR_NO_FOLD", raising=False)
    eps, active = plan.fold_eps()
    assert active and 0.0 < eps <= 2e-6
    d_cube = ctx.to_device(cube)
    out = plan.run(d_cube, mask=None, want_maps=True)
    fold = {k: out[k].to_host() for k in ("correl", "correl_min", "profile", "maxmap")}
    monkeypatch.setenv("ORIGIN_GLR_NO_FOLD", "1")
    assert plan.fold_eps()[1] is False
    out = plan.run(d_cube, mask=None, want_maps=True)
    exact = {k: out[k].to_host() for k in ("correl", "correl_min", "profile", "maxmap")}
    monkeypatch.delenv("ORIGIN_GLR_NO_FOLD")
    plan.close()
    # the ends of the cube run the exact pair loop in both forms (on folded / plain taps)
    local = maximum_filter1d(np.abs(ref[0]).max(axis=(1, 2)), size=193, mode="nearest")
    # the arithmetic's own bound per unit of the window's largest |T| (bf16: the 5e-2 of
    # test_glr_bf16_precision_meets_the_bf16_tolerance at unit noise, where that maximum is ~4)
    rnd = 1.2e-2 if precision == "bf16" else 3e-6
    tol = ((eps + rnd) * local)[:, None, None]
    for k, r in (("correl", ref[0]), ("correl_min", ref[2])):
        assert np.all(np.abs(fold[k] - exact[k]) <= 2 * tol), k
        assert np.all(np.abs(fold[k] - r) <= tol), k
        assert np.all(np.abs(exact[k] - r) <= tol), k
    assert np.max(np.abs(fold["maxmap"] - fold["correl"].max(axis=0))) == 0.0
    if precision == "f16x2":
        assert np.mean(fold["profile"] != ref[1]) <= 1e-4
        assert np.mean(fold["profile"] != exact["profile"]) <= 1e-4


def test_glr_fold_is_refused_for_a_steep_psf_ramp(ctx):
    """A PSF whose width changes quickly with the channel fails the plan's eps test: the plan
    reports FOLD inactive and the stage runs the exact form (same bounds as ever)."""
    from origin_amd import kernels
    rng = np.random.default_rng(32)
    Nz, Ny, Nx = 200, 26, 36
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    psf = synth.moffat_psf(Nz, 9, fwhm0=6.0, fwhm1=2.0).astype(np.float64)
    prof = synth.dico_fwhm(20)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, 1e-8, True, precision="f16x2")
    eps, active = plan.fold_eps()
    assert eps > 2e-6 and not active
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    out = plan.run(ctx.to_device(cube), mask=None, want_maps=False)
    assert np.max(np.abs(out["correl"].to_host() - ref[0])) <= 1e-4
    assert np.max(np.abs(out["correl_min"].to_host() - ref[2])) <= 1e-4
    plan.close()


def test_glr_f16_split_survives_huge_dynamic_range(ctx):
    """Per-tile power-of-two scaling: slabs of channels at 1e-6, 1 and 1e+7 times unit noise must
    neither overflow the f16 halves nor lose the faint slabs.  Error bound relative to the
    largest |T| within the 96-channel window a tile shares (float64 oracle)."""
    from scipy.ndimage import maximum_filter1d
    from origin_amd import kernels
    rng = np.random.default_rng(77)
    Nz, Ny, Nx = 448, 26, 30
    amp = np.repeat([1e-6, 1.0, 1e7, 1e-3, 1e3, 1.0, 1e-6], 64)[:, None, None]
    cube = (rng.standard_normal((Nz, Ny, Nx)) * amp).astype(np.float32)
    psf = synth.moffat_psf(Nz, 25).astype(np.float64)
    prof = synth.dico_fwhm(20)
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, 1e-8, True, precision="f16x2")
    assert plan.precision == "f16x2"
    out = plan.run(ctx.to_device(cube), mask=None, want_maps=False)
    got, gmin = out["correl"].to_host(), out["correl_min"].to_host()
    assert np.all(np.isfinite(got)) and np.all(np.isfinite(gmin))
    local = maximum_filter1d(np.abs(ref[0]).max(axis=(1, 2)), size=193, mode="nearest")
    tol = 3e-6 * local[:, None, None]
    assert np.all(np.abs(got - ref[0]) <= tol)
    assert np.all(np.abs(gmin - ref[2]) <= tol)
    plan.close()


@pytest.mark.parametrize("pattern", ["ramps", "random"])
def test_glr_kept_fragments_follow_a_moving_scale(ctx, pattern):
    """The spectral kernel keeps four of a window's six 16-channel blocks as f16 fragments from
    tile to tile and multiplies them by 2^d when the tile scale moves by |d| <= 4 (larger steps
    re-convert the whole window).  Amplitude ramps of x1.3 per 16 channels (d = +-1 every other
    tile), steps of x8 and x1/16 (d = 3, -4) and x64 (re-conversion), long enough for several
    tiles per z chunk; same error bound as above."""
    from scipy.ndimage import maximum_filter1d
    from origin_amd import kernels
    rng = np.random.default_rng(78)
    Ny, Nx = 26, 30
    seg = [1.3 ** (np.arange(96) // 16), 1.3 ** 5 * 1.3 ** (-(np.arange(160) // 16)),
           np.full(64, 8.0), np.full(64, 0.5), np.full(96, 32.0), np.full(64, 1.0),
           1.3 ** (np.arange(352) // 16 % 7)]
    amp = np.concatenate(seg)
    if pattern == "random":
        # a random walk of the block amplitude in steps of 2^U(-6, 6): kept and re-converted
        # windows in random order, several z chunks
        amp = np.repeat(2.0 ** np.clip(np.cumsum(rng.uniform(-6, 6, 80)), -30, 30), 16)
    Nz = len(amp)
    cube = (rng.standard_normal((Nz, Ny, Nx)) * amp[:, None, None]).astype(np.float32)
    psf = synth.moffat_psf(Nz, 25).astype(np.float64)
    prof = synth.dico_fwhm(20)
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=1,
                                       pcut=1e-8, pmeansub=True)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, 1e-8, True, precision="f16x2")
    out = plan.run(ctx.to_device(cube), mask=None, want_maps=False)
    got, gmin = out["correl"].to_host(), out["correl_min"].to_host()
    local = maximum_filter1d(np.abs(ref[0]).max(axis=(1, 2)), size=193, mode="nearest")
    tol = 3e-6 * local[:, None, None]
    assert np.all(np.abs(got - ref[0]) <= tol)
    assert np.all(np.abs(gmin - ref[2]) <= tol)
    if pattern == "ramps":
        assert np.mean(out["profile"].to_host() != ref[1]) <= 1e-4
    plan.close()


def test_glr_wide_profiles_fallback(hip):
    """pcut=None with the full 201-tap dictionary takes the generic (wide-window) kernel."""
    c = gc.g5_inputs()["a"]
    args = (c["cube"], c["fsf"], None, c["profiles"])
    correl, profile, correl_min = hip.Correlation_GLR_test(*args, pcut=None, pmeansub=True)
    r = cpu_ref.Correlation_GLR_test(*args, nthreads=1, pcut=None, pmeansub=True)
    assert np.max(np.abs(correl - r[0])) <= 1e-4
    assert np.max(np.abs(correl_min - r[2])) <= 1e-4


@pytest.mark.parametrize("lengths", [(8, 12, 30, 11), (64, 9), (140, 20)])
def test_glr_even_length_profiles_keep_the_reference_centre(ctx, lengths):
    """pcut=None hands the dictionary entries over untrimmed; an even-length entry is centred
    on startind = (L-1)//2 = L/2 - 1 (reference lib_origin.py:1179-1181).  Asymmetric profiles
    (a shift by one channel cannot hide), every spectral kernel family: matrix-core / packed
    fp32 (lw <= 32) and the generic wide-window kernel (140 taps)."""
    from origin_amd import kernels
    rng = np.random.default_rng(sum(lengths))
    Nz, Ny, Nx = 200, 28, 32
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    cube[100, 14, 16] += 60.0
    psf = synth.moffat_psf(Nz, 25).astype(np.float64)
    prof = [np.exp(-0.5 * ((np.arange(L) - 0.37 * L) / (0.18 * L)) ** 2) * (1 + 0.2 * rng.random(L))
            for L in lengths]
    ref = cpu_ref.Correlation_GLR_test(cube.astype(np.float64), psf, None, prof, nthreads=1,
                                       pcut=None, pmeansub=True)
    for prec in ("f16x2", "f32"):
        plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, None, True,
                               precision=prec if max(lengths) <= 65 else None)
        out = plan.run(ctx.to_device(cube), mask=None, want_maps=False)
        assert np.max(np.abs(out["correl"].to_host() - ref[0])) <= 1e-4
        assert np.max(np.abs(out["correl_min"].to_host() - ref[2])) <= 1e-4
        assert np.mean(out["profile"].to_host() != ref[1]) <= 1e-4
        plan.close()


def test_glr_linearity_property(hip):
    """T(a * cube) = a * T(cube) for a > 0 (size-independent property)."""
    c = gc.g5_inputs()["e"]
    r1 = hip.Correlation_GLR_test(c["cube"], c["fsf"], None, c["profiles"], pcut=1e-8)
    r2 = hip.Correlation_GLR_test(4.0 * c["cube"], c["fsf"], None, c["profiles"], pcut=1e-8)
    np.testing.assert_allclose(r2[0], 4.0 * r1[0], rtol=0, atol=1e-4)
    assert np.array_equal(r1[1], r2[1])


def test_local_max_golden(hip):
    g5, g = load("g5_glr"), load("g6_localmax")
    mask = gc.g5_mask(g5["a_correl"].shape)
    cm = g5["a_correl"].astype(np.float32)
    cm[mask] = 0
    cmin = g5["a_correl_min"].astype(np.float32)
    lmax, lmin = hip.compute_local_max(cm, cmin, mask, 3)
    rmax, rmin = cpu_ref.compute_local_max(cm.astype(float), cmin.astype(float), mask, 3)
    assert np.array_equal(lmax, rmax) and np.array_equal(lmin, rmin)
    # and the float64 golden agrees where rounding to float32 does not create ties
    assert np.mean((lmax != 0) != (g["local_max"] != 0)) < 1e-3


# ------------------------------------------------------------------------------- chain
def test_step_chain_golden(ctx):
    """G7: minicube-shaped chain through the Step API (SimpleOrig) vs the reference."""
    from origin_amd.steps import SimpleOrig, Status
    g = load("g7_chain")
    inp = gc.g7_inputs()
    orig = SimpleOrig(inp["raw"], inp["var"], inp["mask"], inp["PSF"], inp["profiles"], ctx=ctx)
    with pytest.raises(RuntimeError):
        orig.step03_compute_PCA_threshold()          # require: preprocessing, areas
    orig.step01_preprocessing()
    orig.step02_areas.set_areamap(inp["areamap"])
    orig.step03_compute_PCA_threshold()
    orig.step04_compute_greedy_PCA()
    orig.step05_compute_TGLR()
    assert all(s.status is Status.RUN for s in list(orig.steps.values())[:5])
    assert orig.param["compute_TGLR"]["params"]["pcut"] == 1e-8
    zs = g["zs"]
    np.testing.assert_allclose(orig.thresO2, g["thresO2"], rtol=1e-5)
    assert_close_scaled(orig.cube_std._data[zs], g["cube_std_z"], 1e-5)
    assert np.array_equal(orig.mapO2, g["mapO2"])
    assert np.max(np.abs(orig.cube_faint._data[zs] - g["cube_faint_z"])) <= 1e-4
    assert np.max(np.abs(orig.cube_correl._data[zs] - g["correl_z"])) <= 2e-4
    assert np.max(np.abs(orig.cube_correl_min._data[zs] - g["correl_min_z"])) <= 2e-4
    assert np.max(np.abs(orig.maxmap - g["maxmap"])) <= 2e-4
    assert np.max(np.abs(orig.minmap - g["minmap"])) <= 2e-4
    assert np.mean(orig.cube_profile._data[zs] != g["profile_z"]) <= 1e-3
    assert orig.cube_local_max._data.shape == inp["raw"].shape


def test_errors_are_raised_not_aborted(ctx):
    from origin_amd import _capi, kernels
    raw = ctx.zeros((8, 4, 4), np.float32)
    mask = ctx.zeros((8, 4, 4), np.uint8)
    with pytest.raises(_capi.OriginHipError) as e:
        kernels.dct_fit(ctx, raw, raw, mask, order=40)
    assert e.value.code == -1 and "order" in str(e.value)
    with pytest.raises(_capi.OriginHipError):
        kernels.GLRPlan(ctx, (8, 4, 4), np.ones((8, 4, 4)), None, [np.ones(5)])  # even PSF


@pytest.mark.parametrize("name", ["seg", "noseg", "lst"])
def test_purity_threshold_golden(hip, name):
    """Compute_threshold_purity on the device (per-spaxel maxima + counts per threshold are the
    only things that leave the GPU) against the reference's output, bit for bit: counts are
    integers, thresholds are float64 functions of float32-representable cube values."""
    g = load("g8_purity")
    inp = gc.g8_inputs()
    segmap = None if name == "noseg" else inp["segmap"]
    tl = list(inp["threshlist"]) if name == "lst" else None
    with np.errstate(all="ignore"):
        thr, res = hip.Compute_threshold_purity(float(g[name + "_purity"]), inp["lmax"],
                                                inp["lmin"], segmap, threshlist=tl)
    assert thr == float(g[name + "_threshold"])
    for c in ("Tval_r", "Pval_r", "Det_m", "Det_M"):
        assert np.array_equal(np.asarray(res[c], float), np.asarray(g[f"{name}_{c}"], float),
                              equal_nan=True), c


def test_purity_step_on_device_cubes(ctx):
    """Step 6 after the GPU chain: the cubes come from the device cache, nothing is uploaded."""
    from origin_amd.steps import SimpleOrig
    f, raw, var, mask = synth.small_case(Nz=160, Ny=48, Nx=52, seed=3, psf_size=9, nprof=3,
                                         area_size=24)
    orig = SimpleOrig(raw, var, mask, f.PSF.astype(float), f.profiles)
    orig.step01_preprocessing()
    orig.step02_areas.set_areamap(f.areamap)
    orig.step03_compute_PCA_threshold()
    orig.step04_compute_greedy_PCA()
    orig.step05_compute_TGLR()
    orig.step06_compute_purity_threshold(purity=0.8)
    with np.errstate(all="ignore"):
        thr, cols = cpu_ref.Compute_threshold_purity(
            0.8, orig.cube_local_max._data.astype(float), orig.cube_local_min._data.astype(float),
            np.asarray(orig.segmap_purity))
    assert orig.param["threshold"] == thr
    pv = orig.Pval
    assert np.array_equal(np.asarray(pv["Det_M"]), cols["Det_M"])
    assert np.array_equal(np.asarray(pv["Det_m"]), cols["Det_m"])


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 5, 7), (16, 16, 16), (17, 31, 29),
                                   (64, 64, 65), (130, 47, 53)])
def test_where_above_is_numpy_where(ctx, shape):
    """origin_where_above against np.where on the same float32 cube: positions in C order, the
    values and the gathered uint8 cube there -- bit exact (index work).  Shapes below one
    4096-voxel chunk, an exact multiple of it and ragged tails; sparse hits, no hit, every voxel a
    hit, NaN / inf entries, a threshold equal to values of the cube (strict >), and more hits
    than the caller's capacity (second call with room for all)."""
    from origin_amd import kernels
    rng = np.random.default_rng(sum(shape))
    cube = rng.standard_normal(shape).astype(np.float32)
    cube[rng.random(shape) < 0.6] = 0.0                       # mostly zeros, like a local-max cube
    if cube.size > 8:
        flat = cube.reshape(-1)
        flat[rng.integers(0, flat.size, 3)] = np.nan
        flat[rng.integers(0, flat.size, 2)] = np.inf
        flat[rng.integers(0, flat.size, 2)] = -np.inf
        flat[-1] = 5.0                                        # a hit in the last (ragged) voxel
        flat[0] = 4.0
    aux = rng.integers(0, 255, shape).astype(np.uint8)
    d_cube, d_aux = ctx.to_device(cube), ctx.to_device(aux)
    some = float(np.float32(cube[np.isfinite(cube)].max())) if np.isfinite(cube).any() else 0.0
    for thr in (1.5, 0.0, -1e30, 1e30, some, float(np.float32(0.7)) + 1e-12):
        for cap in (1 << 20, 3):
            got = kernels.where_above(ctx, d_cube, thr, aux=d_aux, cap=cap)
            with np.errstate(invalid="ignore"):
                z, y, x = np.where(cube.astype(np.float64) > thr)
            assert np.array_equal(got["z"], z) and np.array_equal(got["y"], y)
            assert np.array_equal(got["x"], x)
            assert np.array_equal(got["value"], cube[z, y, x].astype(np.float64))
            assert np.array_equal(got["aux"], aux[z, y, x])
    only = kernels.where_above(ctx, d_cube, 1.5)              # without the gathered cube
    assert "aux" not in only and np.array_equal(only["z"], np.where(cube > 1.5)[0])
    # a NaN threshold (step 6 found no purity crossing): np.where(cube > nan) is empty
    none = kernels.where_above(ctx, d_cube, float("nan"), aux=d_aux)
    assert none["z"].size == 0 and none["value"].size == 0 and none["aux"].size == 0


def test_where_above_large_cube_properties(ctx):
    """3681 x 200 x 200 (BASELINE config 1's shape; 36 000 chunks, several rounds of the prefix
    scan): planted detections come back exactly, in order; the count agrees with
    origin_count_above; every returned voxel is above the threshold and the flat indices are
    strictly increasing (C order, no duplicates)."""
    from origin_amd import kernels
    Nz, Ny, Nx = 3681, 200, 200
    rng = np.random.default_rng(7)
    cube = np.zeros((Nz, Ny, Nx), np.float32)
    n = 50000
    flat = np.unique(rng.integers(0, cube.size, n))
    cube.reshape(-1)[flat] = rng.uniform(0.5, 30.0, flat.size).astype(np.float32)
    d = ctx.to_device(cube)
    thr = 8.25
    got = kernels.where_above(ctx, d, thr, cap=1000)          # forces the second call
    want = flat[cube.reshape(-1)[flat].astype(np.float64) > thr]
    gi = (got["z"] * Ny + got["y"]) * Nx + got["x"]
    assert np.array_equal(gi, want)
    assert np.all(np.diff(gi) > 0) and np.all(got["value"] > thr)
    assert np.array_equal(got["value"], cube.reshape(-1)[want].astype(np.float64))
    assert kernels.count_above(ctx, d, [thr])[0] == want.size


def test_step7_thresholding_after_the_chain(ctx):
    """Detection.run's first half (steps.py:956-994) on the device cubes the chain left in HBM,
    against the oracle's restatement on their host copies: Cat0 identical column by column, the
    same std detections survive the merging."""
    from origin_amd import detection
    from origin_amd.steps import SimpleOrig
    f, raw, var, mask = synth.small_case(Nz=160, Ny=48, Nx=52, seed=3, psf_size=9, nprof=3,
                                         area_size=24)
    orig = SimpleOrig(raw, var, mask, f.PSF.astype(float), f.profiles)
    orig.step01_preprocessing()
    orig.step02_areas.set_areamap(f.areamap)
    orig.step03_compute_PCA_threshold()
    orig.step04_compute_greedy_PCA()
    orig.step05_compute_TGLR()
    lmax = orig.cube_local_max._data
    prof = orig.cube_profile._data
    smax = orig.cube_std_local_max._data
    d_lmax = orig._hip_cache["cube_local_max"]                 # what step 5 left in HBM
    # thresholds that give a few hundred detections of each kind on this small field
    t_cor = float(np.sort(lmax[lmax > 0])[-300])
    t_std = float(np.sort(smax[smax > 0])[-400])
    cat0, cat, cat_std = detection.from_session(orig, threshold=t_cor, threshold_std=t_std)
    ref0, keep = cpu_ref.detection_threshold(lmax.astype(np.float32).astype(float), prof,
                                             smax.astype(np.float32).astype(float), t_cor, t_std)
    assert 250 <= len(cat["z0"]) <= 350 and len(cat0["z0"]) > len(cat["z0"])
    for k in detection.CAT0_COLUMNS:
        assert np.array_equal(np.asarray(cat0[k], float), np.asarray(ref0[k], float),
                              equal_nan=True), k
    n = len(cat["z0"])
    for k in ("x0", "y0", "z0", "STD"):
        assert np.array_equal(cat_std[k], ref0[k][n:][keep]), k
    assert 0 < len(keep) <= len(ref0["z0"]) - n
    zm, ym, xm = detection.det_correl_min(ctx, d_lmax, t_cor)          # steps.py:935-939
    assert np.array_equal(zm, cat["z0"]) and np.array_equal(xm, cat["x0"])


def test_graft_entry_smoke():
    """The driver's smoke(): small Step chain on the GPU against the oracle."""
    import __graft_entry__
    __graft_entry__.smoke()


def test_greedy_pca_full_depth_properties(ctx):
    """BASELINE config 1 size (3681 x 200 x 200, four 100 x 100 areas), too big for the oracle:
    size-independent properties of the greedy PCA.  (a) At the end at most ONE spaxel per area
    has an O2 test above the area's threshold (a single remaining nuisance spaxel ends the loop
    untouched, lib_origin.py:927-928) -- recomputed from the float32 cube_faint, not from the
    running values the loop keeps.  (b) Idempotence: a second run on cube_faint with the same
    thresholds only counts those leftovers once more in mapO2 and leaves the cube bit for bit."""
    from origin_amd import kernels, pipeline, synth
    f = synth.SyntheticField(3681, 200, 200)
    raw, var, mask = f.arrays()
    d_raw, d_var, d_mask = ctx.to_device(raw), ctx.to_device(var), ctx.to_device(mask.astype(np.uint8))
    del raw, var
    pre = pipeline.preprocess(ctx, d_raw, d_var, d_mask, want_cont=False)
    thr = pipeline.pca_threshold(pre["o2_host"], f.areamap, f.nbAreas, 0.01)
    F, mapO2, nstop, drv = pipeline.greedy_pca(ctx, pre["cube_std"], f.areamap, f.nbAreas,
                                               thr["thresO2"], thr["testO2"], 50, 100,
                                               o2_dev=pre["o2"])
    assert nstop == 0 and mapO2.max() >= 3
    o2 = kernels.o2test(ctx, F).to_host()
    leftovers = 0
    for a in range(f.nbAreas):
        above = o2[f.areamap == a + 1] > thr["thresO2"][a] * (1 + 2e-6)
        assert above.sum() <= 1, a
        leftovers += int(above.sum())
    before = F.to_host()
    test2 = [o2.reshape(-1)[s] for s in pipeline.area_lists(f.areamap, f.nbAreas)]
    F2, map2, nstop2, _ = pipeline.greedy_pca(ctx, F, f.areamap, f.nbAreas, thr["thresO2"], test2,
                                              50, 100, inplace=True)
    assert nstop2 == 0 and map2.sum() == leftovers and map2.max() <= 1
    assert np.array_equal(F2.to_host(), before)
