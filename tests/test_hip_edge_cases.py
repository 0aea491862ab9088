"""GPU edge cases of the HIP path: ragged / odd sizes (which select the non-vectorised and
unpacked kernel variants), PSF sizes without a specialised kernel, one profile, weights with
the production PSF, degenerate PCA inputs, fully masked channels, and cross-checks between the
table-normalised and the explicit-norm GLR paths.  Oracle: oracle/cpu_ref.py."""
import numpy as np
import pytest

from oracle import cpu_ref
from origin_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import origin_amd.lib_origin as lib
    return lib


@pytest.fixture(scope="module")
def ctx():
    from origin_amd.device import default_context
    return default_context(0)


def noise(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32).astype(float)


@pytest.mark.parametrize("Ny,Nx,P,nprof", [(27, 31, 9, 3), (26, 30, 7, 3), (9, 11, 3, 1),
                                           (33, 34, 5, 20), (40, 44, 25, 3), (25, 25, 25, 3)])
def test_glr_ragged_shapes_vs_oracle(hip, Ny, Nx, P, nprof):
    Nz = 90
    cube = noise((Nz, Ny, Nx), 1000 + Ny)
    psf = synth.moffat_psf(Nz, P).astype(float)
    profs = synth.dico_fwhm(nprof)
    got = hip.Correlation_GLR_test(cube, psf, None, profs, pcut=1e-8)
    ref = cpu_ref.Correlation_GLR_test_direct(cube, psf, None, profs, pcut=1e-8)
    assert np.max(np.abs(got[0] - ref[0])) <= 1e-4
    assert np.max(np.abs(got[2] - ref[2])) <= 1e-4
    assert np.mean(got[1] != ref[1]) <= 2e-4


def test_glr_weights_production_psf(hip):
    Nz, Ny, Nx = 60, 30, 32
    cube = noise((Nz, Ny, Nx), 7)
    p0 = synth.moffat_psf(Nz, 25).astype(float)
    p1 = synth.moffat_psf(Nz, 25, fwhm0=2.7, fwhm1=3.4).astype(float)
    w0 = (0.2 + 0.6 * np.linspace(0, 1, Ny)[:, None] * np.ones((1, Nx))).astype(np.float32)
    w = [w0.astype(float), 1.0 - w0.astype(float)]
    got = hip.Correlation_GLR_test(cube, [p0, p1], w, synth.dico_fwhm(3), pcut=1e-8)
    ref = cpu_ref.Correlation_GLR_test_direct(cube, [p0, p1], w, synth.dico_fwhm(3), pcut=1e-8)
    assert np.max(np.abs(got[0] - ref[0])) <= 1e-4 and np.max(np.abs(got[2] - ref[2])) <= 1e-4


def test_glr_table_path_equals_explicit_norm_path(hip):
    """weights=None (border-class tables + packed kernels) vs one field with weights == 1
    (explicit norm cube + generic kernels): same algebra, two code paths."""
    Nz, Ny, Nx = 400, 64, 72
    cube = noise((Nz, Ny, Nx), 11)
    psf = synth.moffat_psf(Nz, 25).astype(float)
    profs = synth.dico_fwhm(20)
    a = hip.Correlation_GLR_test(cube, psf, None, profs, pcut=1e-8)
    b = hip.Correlation_GLR_test(cube, [psf], [np.ones((Ny, Nx))], profs, pcut=1e-8)
    assert np.max(np.abs(a[0] - b[0])) <= 5e-5 and np.max(np.abs(a[2] - b[2])) <= 5e-5
    assert np.mean(a[1] != b[1]) <= 2e-4


def test_glr_mask_glue_and_maps(ctx):
    from origin_amd import kernels
    Nz, Ny, Nx = 80, 28, 30
    cube = noise((Nz, Ny, Nx), 13)
    mask = np.random.default_rng(5).random((Nz, Ny, Nx)) < 0.05
    psf = synth.moffat_psf(Nz, 9).astype(float)
    profs = synth.dico_fwhm(3)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, profs, 1e-8, True)
    out = plan.run(ctx.to_device(cube, np.float32), mask=ctx.to_device(mask.astype(np.uint8)))
    ref = cpu_ref.compute_TGLR(cube, psf, None, profs, mask, pcut=1e-8)
    correl = out["correl"].to_host()
    assert np.all(correl[mask] == 0) and np.all(out["profile"].to_host()[mask] == 0)
    assert np.max(np.abs(correl - ref["cube_correl"])) <= 1e-4
    assert np.max(np.abs(out["maxmap"].to_host() - ref["maxmap"])) <= 1e-4
    assert np.max(np.abs(out["minmap"].to_host() - ref["minmap"])) <= 1e-4
    plan.close()


def test_pca_degenerate_inputs(hip):
    rng = np.random.default_rng(3)
    cube = rng.standard_normal((120, 150)).astype(np.float32).astype(float)
    test = cpu_ref.O2test(cube)
    # threshold above every O2: nothing to do
    faint, mapO2, nstop = hip.Compute_GreedyPCA(cube, test, 1e9, 50, 100)
    assert np.array_equal(faint, cube) and not mapO2.any() and nstop == 0
    # itermax = 0: every area with a nuisance stops at once
    faint, mapO2, nstop = hip.Compute_GreedyPCA(cube, test, float(np.median(test)), 50, 0)
    r = cpu_ref.Compute_GreedyPCA(cube, test, float(np.median(test)), 50, 0)
    assert np.array_equal(faint, cube) and nstop == r[2] == 1 and np.array_equal(mapO2, r[1])
    # exactly one nuisance spaxel: the reference breaks without touching the data
    thr = float(np.sort(test)[-2])
    faint, mapO2, nstop = hip.Compute_GreedyPCA(cube, test, thr, 50, 100)
    r = cpu_ref.Compute_GreedyPCA(cube, test, thr, 50, 100)
    assert np.array_equal(faint, cube) and np.array_equal(mapO2, r[1]) and nstop == r[2]
    # fractional Noise_population and a spaxel with O2 == 0 (filtered-index quirk, lib :908-917)
    cube2 = cube.copy()
    cube2[:, 4] = 0.0
    cube2[:, 7] *= 6
    cube2[:, 90] *= 4
    test2 = cpu_ref.O2test(cube2)
    thr2 = 2.0
    got = hip.Compute_GreedyPCA(cube2, test2, thr2, 33.3, 100)
    ref = cpu_ref.Compute_GreedyPCA(cube2, test2, thr2, 33.3, 100, svd="dense")
    assert np.array_equal(got[1], ref[1]) and got[2] == ref[2]
    assert np.max(np.abs(got[0] - ref[0])) <= 1e-4


def test_pca_area_with_unassigned_spaxels(hip):
    """areamap label 0 (outside every area, reference steps.py:559-565) is left untouched."""
    rng = np.random.default_rng(8)
    cube = rng.standard_normal((100, 12, 14)).astype(np.float32).astype(float)
    cube[:, 3, 3] *= 5
    cube[:, 8, 9] *= 7
    areamap = np.ones((12, 14), int)
    areamap[:, 7:] = 2
    areamap[0, :] = 0
    res = [cpu_ref.Compute_PCA_threshold(cube[:, areamap == i], 0.01) for i in (1, 2)]
    thr = [1.5, 1.5]
    got = hip.Compute_GreedyPCA_area(2, cube, areamap, 50, thr, 100, [r[0] for r in res])
    ref = cpu_ref.Compute_GreedyPCA_area(2, cube, areamap, 50, thr, 100, [r[0] for r in res],
                                         svd="dense")
    assert np.array_equal(got[1], ref[1]) and got[2] == ref[2]
    assert np.max(np.abs(got[0] - ref[0])) <= 1e-4
    assert np.array_equal(got[0][:, 0, :], cube[:, 0, :])


def test_preprocess_fully_masked_channel_and_spaxel(ctx):
    from origin_amd import pipeline
    f, raw, var, mask = synth.small_case(Nz=96, Ny=10, Nx=13, seed=21, psf_size=7)
    mask[17] = True          # a whole channel
    mask[:, 2, 5] = True     # a whole spaxel
    raw[mask] = 0
    var[mask] = np.inf
    out = pipeline.preprocess(ctx, ctx.to_device(raw), ctx.to_device(var),
                              ctx.to_device(mask.astype(np.uint8)))
    with np.errstate(all="ignore"):
        ref = cpu_ref.preprocessing(raw.astype(float), var.astype(float), mask)
    got = out["cube_std"].to_host()
    assert np.all(np.isfinite(got)) and np.all(got[mask] == 0)
    err = np.abs(got - ref["cube_std"]) / np.maximum(1, np.abs(ref["cube_std"]))
    assert err.max() <= 1e-5
    assert out["o2"].to_host()[2, 5] == 0


@pytest.mark.parametrize("order", [1, 4, 12])
def test_dct_other_orders(hip, order):
    f, raw, var, mask = synth.small_case(Nz=150, Ny=6, Nx=9, seed=order, psf_size=7)
    got = hip.dct_residual(raw, order, var, False, mask)
    ref = cpu_ref.dct_residual(raw.astype(float), order, var.astype(float), False, mask)
    assert np.max(np.abs(got - ref) / np.maximum(1, np.abs(ref))) <= 1e-5


def test_local_max_sizes(hip):
    a = noise((20, 9, 11), 2).astype(np.float32)
    b = noise((20, 9, 11), 3).astype(np.float32)
    mask = np.zeros(a.shape, bool)
    mask[3, 4, 5] = True
    for size in (1, 2, 3, 5):
        got = hip.compute_local_max(a, b, mask, size)
        ref = cpu_ref.compute_local_max(a.astype(float), b.astype(float), mask, size)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_bad_arguments_raise(ctx, hip):
    from origin_amd import _capi, kernels
    with pytest.raises(_capi.OriginHipError):
        kernels.GLRPlan(ctx, (10, 8, 8), np.ones((10, 5, 5)), None, [np.ones(7)] * 300)  # K > 255
    with pytest.raises(ValueError):
        kernels.GLRPlan(ctx, (10, 8, 8), np.ones((9, 5, 5)), None, [np.ones(7)])  # Nz mismatch
    with pytest.raises(ValueError):
        hip.Compute_GreedyPCA_area(1, np.zeros((5, 2, 2)), np.ones((2, 2), int), 50, [1.0], 10,
                                   [np.zeros(3)])  # testO2 length != area size


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 3, 5), (7, 1, 70), (5, 17, 1), (40, 19, 130),
                                   (3, 1, 4), (9, 5, 8), (33, 7, 260), (70, 9, 600), (12, 66, 132),
                                   (6, 3, 1028)])
@pytest.mark.parametrize("size", [3, 5])
def test_local_max_small_and_ragged_shapes(hip, shape, size):
    """3x3x3 z-marching kernels (size 3: the four-samples-per-lane form when Nx % 4 == 0 -- rows
    that end inside a wave, waves that end inside a row, rows shorter than a wave, a row count
    that is not a multiple of the lane's row group, more than one z chunk --, the one-sample form
    otherwise) and the generic kernel on shapes smaller than a tile, single planes / rows /
    columns; plateaus (equal neighbours) and masked voxels included."""
    rng = np.random.default_rng(sum(shape) + size)
    correl = np.round(rng.standard_normal(shape) * 2).astype(np.float32).astype(float)  # ties
    cmin = -np.abs(np.round(rng.standard_normal(shape) * 2)).astype(np.float32).astype(float)
    mask = rng.random(shape) < 0.1
    got = hip.compute_local_max(correl, cmin, mask, size)
    ref = cpu_ref.compute_local_max(correl, cmin, mask, size)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_out_of_place_pca_keeps_unassigned_spaxels(ctx):
    """cube_faint starts as a copy of cube_std (reference lib_origin.py:799): spaxels with
    areamap == 0 must come out unchanged when the device path writes into a fresh buffer."""
    from origin_amd import pipeline
    rng = np.random.default_rng(12)
    cube = rng.standard_normal((90, 10, 12)).astype(np.float32)
    cube[:, 4, 4] *= 6
    areamap = np.ones((10, 12), int)
    areamap[:2, :] = 0
    test = cpu_ref.O2test(cube.astype(float)[:, areamap == 1])
    d = ctx.to_device(cube)
    F, mapO2, nstop, _ = pipeline.greedy_pca(ctx, d, areamap, 1, [1.5], [test], 50, 100)
    got = F.to_host()
    assert np.array_equal(got[:, :2, :], cube[:, :2, :])
    ref = cpu_ref.Compute_GreedyPCA_area(1, cube.astype(float), areamap, 50, [1.5], 100, [test],
                                         svd="dense")
    assert np.max(np.abs(got - ref[0])) <= 1e-4 and np.array_equal(mapO2, ref[1])


def test_pca_more_iterations_than_vector_slots(hip):
    """An area that needs more iterations than the 64 removed vectors the device keeps per area:
    the cube is flushed (F = X - U C) in the middle of the run and the loop goes on from the
    flushed cube -- nuisance-block reuse, running background sums and the memory-order dot
    products all restart from it.  ~80 independent strong nuisance spaxels, one removed per
    iteration."""
    rng = np.random.default_rng(31)
    Nz, S = 240, 500
    cube = rng.standard_normal((Nz, S)).astype(np.float32).astype(float)
    for j in range(80):
        cube[:, 5 * j] += (6.0 + 0.05 * j) * rng.standard_normal(Nz).astype(np.float32)
    cube = cube.astype(np.float32).astype(float)
    test = cpu_ref.O2test(cube)
    thr = float(np.percentile(test, 83.5))
    ref = cpu_ref.Compute_GreedyPCA(cube, test, thr, 50, 300, svd="dense")
    got = hip.Compute_GreedyPCA(cube, test, thr, 50, 300)
    assert ref[1].max() > 64                      # the case does cross the flush
    assert got[2] == ref[2] == 0
    assert np.array_equal(got[1], ref[1])
    assert np.max(np.abs(got[0] - ref[0])) <= 1e-4


def test_pca_area_larger_than_the_register_resident_selection(hip):
    """An area of more than 1024 x 12 spaxels takes the general selection kernel (O2 values
    streamed instead of held in registers / LDS), the re-gathered background mean and the
    per-area deflation kernels."""
    rng = np.random.default_rng(47)
    Nz, S = 64, 13000
    cube = rng.standard_normal((Nz, S)).astype(np.float32).astype(float)
    for j in range(12):
        cube[:, 1000 * j + 7] += (5.0 + 0.3 * j) * rng.standard_normal(Nz).astype(np.float32)
    cube[:, 4321] = 0.0      # a spaxel with O2 == 0: the filtered-index quirk (lib :908-917)
    cube = cube.astype(np.float32).astype(float)
    test = cpu_ref.O2test(cube)
    thr = float(np.sort(test)[-13])
    ref = cpu_ref.Compute_GreedyPCA(cube, test, thr, 50, 100, svd="dense")
    got = hip.Compute_GreedyPCA(cube, test, thr, 50, 100)
    assert ref[1].max() >= 5
    assert got[2] == ref[2] and np.array_equal(got[1], ref[1])
    assert np.max(np.abs(got[0] - ref[0])) <= 1e-4


@pytest.mark.parametrize("deep", [False, True])
def test_pca_into_a_box_of_a_larger_cube(ctx, deep):
    """origin_pca_run_into (the tiled path: cube_faint goes straight into the interior of the
    halo-extended tile): bit for bit what the contiguous run writes, the rest of the larger cube
    untouched -- with spaxels outside every area (copied through), and (deep) with an area that
    needs more iterations than the 64 vector slots, whose flush in the middle of the run goes to a
    contiguous work cube while only the final pass writes the strided box."""
    from origin_amd import pipeline
    rng = np.random.default_rng(77 + deep)
    Nz, Ny, Nx = (240, 20, 25) if deep else (96, 12, 16)
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    flat = cube.reshape(Nz, -1)
    n_src = 80 if deep else 9
    for j in range(n_src):
        flat[:, (5 if deep else 11) * j + 3] += (6.0 + 0.05 * j) * rng.standard_normal(Nz).astype(np.float32)
    areamap = np.ones((Ny, Nx), int)
    areamap[0, :] = 0                      # a row outside every area
    if not deep:
        areamap[:, Nx // 2:] = 2           # two areas
        areamap[0, :] = 0
    nb = int(areamap.max())
    X = cube.astype(float)
    tests = [cpu_ref.O2test(X[:, areamap == a + 1]) for a in range(nb)]
    thr = [float(np.percentile(t, 83.5 if deep else 90.0)) for t in tests]
    d = ctx.to_device(cube)
    F, map_a, nstop_a, _ = pipeline.greedy_pca(ctx, d, areamap, nb, thr, tests, 50, 300)
    want = F.to_host()
    if deep:
        assert map_a.max() > 64            # the run does flush in the middle
    top, left = 3, 5
    ext_h = np.full((Nz, Ny + 7, Nx + 9), -7.0, np.float32)
    ext = ctx.to_device(ext_h)
    none, map_b, nstop_b, _ = pipeline.greedy_pca(ctx, d, areamap, nb, thr, tests, 50, 300,
                                                  into=(ext, top, left))
    got = ext.to_host()
    assert none is None and nstop_a == nstop_b and np.array_equal(map_a, map_b)
    assert np.array_equal(got[:, top:top + Ny, left:left + Nx], want)
    assert np.array_equal(want[:, 0, :], cube[:, 0, :])       # outside every area: copied
    got[:, top:top + Ny, left:left + Nx] = -7.0
    assert np.all(got == -7.0)                                 # nothing else was written


@pytest.mark.parametrize("precision", ["f16x2", "bf16"])
def test_glr_row_bands_write_what_the_whole_run_writes(ctx, precision):
    """origin_glr_run_rows: the run split into row bands (three on the main stream in shuffled
    order, one on the CU-masked side stream) -- bit for bit the cubes and maps of origin_glr_run,
    with a mask, a field height that is no multiple of 64 and a width that is no multiple of 32."""
    from origin_amd import kernels
    rng = np.random.default_rng(5)
    Nz, Ny, Nx = 200, 230, 77
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    cube[40:90] *= 21.0
    mask = (rng.random((Nz, Ny, Nx)) < 0.01).astype(np.uint8)
    psf = synth.moffat_psf(3681, 9)[:Nz].astype(np.float64)
    prof = synth.dico_fwhm(20)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, prof, 1e-8, True, precision=precision)
    assert plan.rows_supported()
    d_cube, d_mask = ctx.to_device(cube), ctx.to_device(mask)
    whole = plan.run(d_cube, mask=d_mask, want_maps=True)
    want = {k: whole[k].to_host() for k in ("correl", "correl_min", "profile", "maxmap", "minmap")}
    correl = ctx.empty(cube.shape, np.float32)
    cmin = ctx.empty(cube.shape, np.float32)
    prof_i = ctx.empty(cube.shape, np.uint8)
    for a in (correl, cmin, prof_i):
        a.fill_bytes(0x7f)
    plan.run_rows(d_cube, d_mask, correl, prof_i, cmin, 128, 192, first=True)
    plan.run_rows(d_cube, d_mask, correl, prof_i, cmin, 0, 64, side=True)
    plan.run_rows(d_cube, d_mask, correl, prof_i, cmin, 192, Ny)
    plan.run_rows(d_cube, d_mask, correl, prof_i, cmin, 64, 128)
    maxmap, minmap = plan.run_finish()
    ctx.sync()
    got = dict(correl=correl.to_host(), correl_min=cmin.to_host(), profile=prof_i.to_host(),
               maxmap=maxmap.to_host(), minmap=minmap.to_host())
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    with pytest.raises(Exception):
        plan.run_rows(d_cube, d_mask, correl, prof_i, cmin, 32, 128)   # not a multiple of 64
    plan.close()


@pytest.mark.parametrize("max_active", [1, 2])
def test_pca_tail_hook_and_the_glr_started_in_its_shadow(ctx, max_active):
    """pipeline.greedy_pca_then_glr: one area of six keeps iterating long after the others; when
    at most max_active areas are left the library writes the finished ones out and calls back, the
    row bands that do not touch the stragglers run on the side stream while the PCA goes on.
    cube_faint, mapO2, nstop and every GLR output are bit for bit those of the sequential calls;
    the bands are the ones the areas' rows imply."""
    from origin_amd import kernels, pipeline
    rng = np.random.default_rng(11)
    Nz, Ny, Nx = 150, 300, 128
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    areamap = np.zeros((Ny, Nx), int)
    for i in range(3):
        for j in range(2):
            areamap[100 * i:100 * (i + 1), 64 * j:64 * (j + 1)] = 2 * i + j + 1
    nb = 6
    flat = cube.reshape(Nz, -1)
    # a few bright spectra per area; many more, of comparable strength, in area 4 (rows 100-199)
    for a in range(nb):
        idx = np.flatnonzero(areamap.reshape(-1) == a + 1)
        for j in range(40 if a == 3 else 3):
            flat[:, idx[17 * j + 5]] += (7.0 + 0.11 * j) * rng.standard_normal(Nz).astype(np.float32)
    if max_active == 2:       # a second straggler, in the top band
        idx = np.flatnonzero(areamap.reshape(-1) == 2)
        for j in range(25):
            flat[:, idx[13 * j + 7]] += (6.5 + 0.13 * j) * rng.standard_normal(Nz).astype(np.float32)
    X = cube.astype(float)
    tests = [cpu_ref.O2test(X[:, areamap == a + 1]) for a in range(nb)]
    thr = [float(np.percentile(t, 99.0)) for t in tests]
    mask = (rng.random((Nz, Ny, Nx)) < 0.005).astype(np.uint8)
    psf = synth.moffat_psf(3681, 9)[:Nz].astype(np.float64)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, synth.dico_fwhm(20), 1e-8, True,
                           precision="f16x2")
    d, d_mask = ctx.to_device(cube), ctx.to_device(mask)
    F0, map0, nstop0, _ = pipeline.greedy_pca(ctx, d, areamap, nb, thr, tests, 50, 100)
    out0 = plan.run(F0, mask=d_mask, want_maps=True)
    want = {k: out0[k].to_host() for k in ("correl", "correl_min", "profile", "maxmap", "minmap")}
    it = [map0[areamap == a + 1].max() for a in range(nb)]
    assert it[3] > 2 * max(it[i] for i in (0, 2, 4, 5))          # area 4 is a straggler
    correl, cmin = ctx.empty(cube.shape, np.float32), ctx.empty(cube.shape, np.float32)
    prof_i, faint = ctx.empty(cube.shape, np.uint8), ctx.empty(cube.shape, np.float32)
    for a in (correl, cmin, prof_i, faint):
        a.fill_bytes(0x7f)
    lm0 = kernels.local_max(ctx, out0["correl"], out0["correl_min"], d_mask, 3)
    want_lm = [a.to_host() for a in lm0]
    lmax, lmin = ctx.empty(cube.shape, np.float32), ctx.empty(cube.shape, np.float32)
    F1, map1, nstop1, _, out1 = pipeline.greedy_pca_then_glr(
        ctx, plan, d, areamap, nb, thr, tests, d_mask, correl, prof_i, cmin, faint,
        max_active=max_active, local_max=(lmax, lmin))
    ctx.sync()
    assert np.array_equal(out1["local_max"].to_host(), want_lm[0])
    assert np.array_equal(out1["local_min"].to_host(), want_lm[1])
    assert nstop0 == nstop1 and np.array_equal(map0, map1)
    assert np.array_equal(F1.to_host(), F0.to_host())
    for k in want:
        assert np.array_equal(out1[k].to_host(), want[k]), k
    early, late = out1["bands"]
    assert early and late, (early, late)                         # the hook did fire
    # area 4 = rows 100..199, halo 4: regions 64..255 wait; with the second straggler (area 2,
    # rows 0..99) everything above row 256 does
    assert late == ([(64, 256)] if max_active == 1 and it[1] < it[3] / 2 else late)
    assert all(y0 % 64 == 0 for y0, _ in early + late)
    assert sorted(early + late)[0][0] == 0 and sorted(early + late)[-1][1] == Ny
    plan.close()


def test_glr_rectangles_agree_with_the_whole_run(ctx):
    """origin_glr_run_rect: the field in four rectangles of 64 x 64 regions (one on the side
    stream) -- what a tile of a tiled field does around its halo exchange.  A rectangle that spans
    all columns is bit for bit the whole run's; one that does not holds 32 columns of one row per
    wave of the spectral stage (other tile scales): equal to rounding, arg-max but for near ties."""
    from origin_amd import kernels
    rng = np.random.default_rng(8)
    Nz, Ny, Nx = 160, 150, 170
    cube = rng.standard_normal((Nz, Ny, Nx)).astype(np.float32)
    cube[30:70] *= 19.0
    mask = (rng.random((Nz, Ny, Nx)) < 0.01).astype(np.uint8)
    psf = synth.moffat_psf(3681, 9)[:Nz].astype(np.float64)
    plan = kernels.GLRPlan(ctx, cube.shape, psf, None, synth.dico_fwhm(20), 1e-8, True,
                           precision="f16x2")
    d_cube, d_mask = ctx.to_device(cube), ctx.to_device(mask)
    whole = plan.run(d_cube, mask=d_mask, want_maps=True)
    want = {k: whole[k].to_host() for k in ("correl", "correl_min", "profile", "maxmap", "minmap")}
    correl, cmin = ctx.empty(cube.shape, np.float32), ctx.empty(cube.shape, np.float32)
    prof_i = ctx.empty(cube.shape, np.uint8)
    for a in (correl, cmin, prof_i):
        a.fill_bytes(0x7f)
    plan.run_rect(d_cube, d_mask, correl, prof_i, cmin, 64, Ny, 64, Nx, first=True, side=True)
    plan.run_rect(d_cube, d_mask, correl, prof_i, cmin, 0, 64, 0, Nx)          # all columns
    plan.run_rect(d_cube, d_mask, correl, prof_i, cmin, 64, 128, 0, 64)
    plan.run_rect(d_cube, d_mask, correl, prof_i, cmin, 128, Ny, 0, 64)
    maxmap, minmap = plan.run_finish()
    ctx.sync()
    got = dict(correl=correl.to_host(), correl_min=cmin.to_host(), profile=prof_i.to_host(),
               maxmap=maxmap.to_host(), minmap=minmap.to_host())
    for k in ("correl", "correl_min", "profile"):
        assert np.array_equal(got[k][:, :64], want[k][:, :64]), k     # the band of whole rows
    scale = np.abs(want["correl"]).max()
    for k in ("correl", "correl_min", "maxmap", "minmap"):
        assert np.max(np.abs(got[k] - want[k])) <= 3e-6 * scale, k
    assert np.mean(got["profile"] != want["profile"]) <= 1e-4
    assert np.max(np.abs(got["maxmap"] - got["correl"].max(axis=0))) == 0.0
    with pytest.raises(Exception):
        plan.run_rect(d_cube, d_mask, correl, prof_i, cmin, 0, 64, 32, Nx)     # x0 not a multiple of 64
    plan.close()


def test_allocation_cache_reuses_released_blocks():
    """origin_free keeps blocks of >= 1 MiB for the next origin_malloc of their size (or up to an
    eighth less); smaller requests get a block of their own; a reused block is ordered behind the
    work that used it before (same stream): what is written next is what is read back.
    origin_mem_info counts the kept blocks as free."""
    from origin_amd.device import Context
    ctx = Context(0)      # (a context of its own: its cache holds nothing yet)
    n = 1 << 20
    a = ctx.empty((n,), np.float32)
    a.fill_bytes(0x3f)
    p = a.ptr
    free0, total = ctx.mem_info()
    a.free()
    free1, _ = ctx.mem_info()
    assert free1 >= free0 + 4 * n - (1 << 16)           # the kept block counts as free
    b = ctx.empty((n,), np.float32)
    assert b.ptr == p
    b.upload(np.arange(n, dtype=np.float32))
    assert np.array_equal(b.to_host(), np.arange(n, dtype=np.float32))
    b.free()
    c = ctx.empty((n - 1000,), np.float32)              # within an eighth: the same block
    assert c.ptr == p
    c.free()
    d = ctx.empty((n // 2,), np.float32)                # much smaller: a block of its own
    assert d.ptr != p
    small = ctx.empty((1000,), np.float32)              # below 1 MiB: never cached
    sp = small.ptr
    small.free()
    e = ctx.empty((n,), np.float32)
    assert e.ptr == p and sp != p
    ctx.close()
