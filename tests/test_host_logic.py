"""CPU: host-side pieces of the product (no GPU needed): the threshold fit that stays on the
host by design, profile preparation, area lists, synthetic inputs."""
import os

import numpy as np

from oracle import cpu_ref
from oracle import golden_cases as gc
from origin_amd import kernels, pipeline, synth, thresholds


def test_threshold_fit_matches_reference_goldens():
    g = np.load(os.path.join(gc.GOLDEN_DIR, "g3_thresh.npz"))
    for name in "abc":
        t = g["in_" + name]
        for pfa in (0.01, 0.2):
            key = f"{name}_{str(pfa).replace('.', 'p')}"
            h, e, thr, mea, std = thresholds.compute_thresh_gaussfit(t, pfa)
            np.testing.assert_allclose([thr, mea, std], g[key + "_res"], rtol=1e-10)
            np.testing.assert_allclose(h, g[key + "_hist"], rtol=1e-12)
            np.testing.assert_allclose(e, g[key + "_edges"], rtol=1e-12)
            assert isinstance(thr, float)


def test_native_lm_fit_equals_scipy_leastsq():
    """csrc/lmfit.hip (MINPACK's lmder algorithm, native) against scipy.optimize.leastsq, the
    reference's own solver behind astropy's LevMarLSQFitter (lib_origin.py:1014-1018): same
    stopping point, parameters equal to a few ulp (libm exp vs NumPy's vector exp)."""
    g = np.load(os.path.join(gc.GOLDEN_DIR, "g3_thresh.npz"))
    cases = [g["in_" + n] for n in "abc"]
    rng = np.random.default_rng(5)
    cases += [(rng.standard_normal((200, n)) ** 2).mean(0) * s for n, s in
              ((900, 1.0), (5000, 3.0), (20000, 0.2))]
    for t in cases:
        for pfa in (0.01, 0.2):
            a = thresholds.compute_thresh_gaussfit(t, pfa)
            b = thresholds.compute_thresh_gaussfit(t, pfa, _fit=thresholds.fit_gauss1d_scipy)
            np.testing.assert_allclose(a[2:], b[2:], rtol=1e-13)
    # the batch call (histogram + fit + threshold natively, all areas at once) is the same
    res = thresholds.thresholds_batch(cases, 0.01)
    for t, r in zip(cases, res):
        one = thresholds.compute_thresh_gaussfit(t, 0.01)
        assert r[2] == one[2] and r[3] == one[3] and r[4] == one[4]
        np.testing.assert_array_equal(r[0], one[0])
        np.testing.assert_array_equal(r[1], one[1])
    # and so is the single-dispatch form that starts from the O2 map and the areas' spaxel lists
    # (what ComputePCAThreshold.run uses): scattered, interleaved index lists
    lens = [len(t) for t in cases]
    perm = rng.permutation(sum(lens))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    o2map = np.empty(sum(lens))
    o2map[perm] = np.concatenate(cases)
    tests_, fits = thresholds.areas_fit(o2map, perm.astype(np.int32), off, 0.01)
    for t, t2, r, f in zip(cases, tests_, res, fits):
        np.testing.assert_array_equal(t, t2)
        assert f[2:] == r[2:]
        np.testing.assert_array_equal(f[0], r[0])
        np.testing.assert_array_equal(f[1], r[1])
    with __import__("pytest").raises(ValueError):
        bad = np.concatenate([np.full(500, 1.0), np.linspace(1.0, 3.0, 40)])
        thresholds.areas_fit(bad, np.arange(len(bad), dtype=np.int32),
                             np.array([0, len(bad)], np.int64), 0.01)


def test_threshold_batch_raises_where_the_reference_does():
    import pytest
    # histogram maximum in the first bin: np.argmin of an empty slice (lib_origin.py:1006)
    t = np.concatenate([np.full(500, 1.0), np.linspace(1.0, 3.0, 40)])
    with pytest.raises(ValueError):
        cpu_ref.compute_thresh_gaussfit(t, 0.01)
    with pytest.raises(ValueError):
        thresholds.thresholds_batch([t], 0.01)


def test_prepare_profiles_matches_oracle():
    for pcut, sub in ((1e-8, True), (None, True), (1e-3, False)):
        a = kernels.prepare_profiles(synth.dico_fwhm(20), pcut, sub)
        b = cpu_ref.prepare_profiles(synth.dico_fwhm(20), pcut, sub)
        assert [len(x) for x in a] == [len(x) for x in b]
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    taps = [len(p) for p in kernels.prepare_profiles(synth.dico_fwhm(20), 1e-8)]
    assert sum(taps) == 704 and taps[0] == 11 and taps[-1] == 59


def test_area_lists_follow_boolean_mask_order():
    rng = np.random.default_rng(0)
    areamap = rng.integers(0, 5, (13, 17))
    lists = pipeline.area_lists(areamap, 4)
    cube = rng.standard_normal((3, 13, 17))
    for i, s in enumerate(lists, start=1):
        np.testing.assert_array_equal(cube.reshape(3, -1)[:, s], cube[:, areamap == i])


def test_pca_threshold_per_area_equals_oracle():
    rng = np.random.default_rng(1)
    o2 = (rng.standard_normal((200, 60 * 60)) ** 2).mean(0).reshape(60, 60)
    areamap, nb = synth.grid_areamap(60, 60, 20)
    res = pipeline.pca_threshold(o2, areamap, nb, 0.01)
    for i in range(nb):
        ref = cpu_ref.compute_thresh_gaussfit(o2[areamap == i + 1], 0.01)
        assert res["thresO2"][i] == ref[2]
        np.testing.assert_array_equal(res["testO2"][i], o2[areamap == i + 1])


def test_synthetic_field_is_deterministic_and_windowed():
    f = synth.SyntheticField(70, 20, 24, seed=9, psf_size=7, nprof=3)
    a = f.chunk(1)
    b = synth.SyntheticField(70, 20, 24, seed=9, psf_size=7, nprof=3).chunk(1)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    w = f.chunk(1, window=(4, 15, 6, 20))
    np.testing.assert_array_equal(w[0], a[0][:, 4:15, 6:20])
    np.testing.assert_array_equal(w[1], a[1][:, 4:15, 6:20])
    assert a[0].dtype == np.float32 and a[2].dtype == np.uint8
    psf = synth.moffat_psf(10, 25)
    np.testing.assert_allclose(psf.sum(axis=(1, 2)), 1, rtol=1e-6)


def test_native_clipped_histogram_is_bit_identical_to_numpy():
    """origin_o2_histogram (native host code in liborigin_hip.so) vs
    sigma_clip + np.histogram(bins='fd', density=True)."""
    from origin_amd import build
    build.build()
    rng = np.random.default_rng(0)
    tests = []
    for trial in range(60):
        n = int(rng.integers(30, 9000))
        t = (rng.standard_normal((int(rng.integers(20, 300)), n)) ** 2).mean(0)
        k = max(1, n // 40)
        t[rng.integers(0, n, k)] *= np.exp(rng.uniform(0.1, 4, k))
        if trial % 7 == 0:
            t[rng.integers(0, n, 3)] = 0
        tests.append(t)
        a = thresholds.clipped_histogram(t)
        b = thresholds.clipped_histogram_numpy(t)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    for (h, e), t in zip(thresholds.clipped_histograms(tests), tests):
        b = thresholds.clipped_histogram_numpy(t)
        np.testing.assert_array_equal(h, b[0])
        np.testing.assert_array_equal(e, b[1])


def test_step7_merging_of_std_detections():
    """Host half of Detection.run's opening (steps.py:983-994): std detections within
    maxdist_lines of a correl detection are dropped.  detection.unmatched_std against a brute
    force distance matrix, and the oracle's restatement of :956-994 on a hand-made cube."""
    from origin_amd import detection
    rng = np.random.default_rng(5)
    for n, m in ((0, 7), (9, 0), (40, 60), (300, 500)):
        cor = {k: rng.integers(0, 30, n) for k in ("x0", "y0", "z0")}
        std = {k: rng.integers(0, 30, m) for k in ("x0", "y0", "z0")}
        got = detection.unmatched_std(cor, std, 2.5)
        if n and m:
            d2 = sum((std[k][:, None] - cor[k][None, :]) ** 2 for k in ("x0", "y0", "z0"))
            want = np.flatnonzero(~(d2 <= 2.5 ** 2).any(axis=1))
        else:
            want = np.arange(m)
        np.testing.assert_array_equal(got, want)
    lmax = np.zeros((4, 3, 5))
    prof = np.arange(60, dtype=np.uint8).reshape(4, 3, 5)
    smax = np.zeros((4, 3, 5))
    lmax[2, 1, 3], lmax[0, 2, 4], lmax[3, 0, 0] = 9.0, 7.5, 7.0       # 7.0 is not > 7.0
    smax[2, 1, 4], smax[0, 0, 0], smax[1, 1, 1] = 5.0, 6.0, 2.0       # first one is 1 px from a line
    cat0, keep = cpu_ref.detection_threshold(lmax, prof, smax, 7.0, 3.0)
    np.testing.assert_array_equal(cat0["z0"], [0, 2, 0, 2])           # np.where order, then std rows
    np.testing.assert_array_equal(cat0["x0"], [4, 3, 0, 4])
    np.testing.assert_array_equal(cat0["comp"], [0, 0, 1, 1])
    np.testing.assert_array_equal(cat0["T_GLR"], [7.5, 9.0, np.nan, np.nan])
    np.testing.assert_array_equal(cat0["STD"], [np.nan, np.nan, 6.0, 5.0])
    np.testing.assert_array_equal(cat0["profile"], [prof[0, 2, 4], prof[2, 1, 3], 0, 0])
    np.testing.assert_array_equal(keep, [0])                          # (0,0,0) survives
