"""CPU: the oracle (oracle/cpu_ref.py) reproduces the golden vectors that
oracle/gen_golden.py recorded from the reference's own lib_origin.py."""
import os

import numpy as np
import pytest

from oracle import cpu_ref
from oracle import golden_cases as gc


def load(name):
    return np.load(os.path.join(gc.GOLDEN_DIR, name + ".npz"))


def close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    scale = max(np.max(np.abs(b)), 1e-300)
    assert np.max(np.abs(a - b)) <= tol * scale


def test_g1_dct_residual():
    g = load("g1_dct")
    inp = gc.g1_inputs()
    assert gc.digest(inp["raw"], inp["var"], inp["mask"]) == str(g["sha"])
    raw, var = inp["raw"].astype(float), inp["var"].astype(float)
    close(cpu_ref.dct_residual(raw, 10, var, False, inp["mask"]), g["cont"], 1e-12)
    close(cpu_ref.dct_residual(raw, 10, var, True, inp["mask"]), g["cont_approx"], 1e-12)


@pytest.mark.parametrize("approx", [False, True])
def test_g2_preprocessing(approx):
    g = load("g2_preproc_approx" if approx else "g2_preproc")
    inp = gc.g1_inputs()
    assert gc.digest(inp["raw"], inp["var"], inp["mask"]) == str(g["sha"])
    out = cpu_ref.preprocessing(inp["raw"].astype(float), inp["var"].astype(float),
                                inp["mask"], 10, approx)
    close(out["cube_std"], g["cube_std"], 1e-12)
    close(out["ima_std"], g["ima_std"], 1e-12)
    close(out["cont_dct"], g["cont_dct"], 1e-6)
    close(out["ima_dct"], g["ima_dct"], 1e-6)
    close(cpu_ref.O2test(out["cube_std"]), g["o2"], 1e-12)
    # masked voxels are exactly zero, fully masked spaxel has O2 == 0
    assert np.all(out["cube_std"][inp["mask"]] == 0)


def test_g3_threshold_fit():
    g = load("g3_thresh")
    for name in "abc":
        t = g["in_" + name]
        for pfa in (0.01, 0.2):
            key = f"{name}_{str(pfa).replace('.', 'p')}"
            h, e, thr, mea, std = cpu_ref.compute_thresh_gaussfit(t, pfa)
            close([thr, mea, std], g[key + "_res"], 1e-10)
            close(h, g[key + "_hist"], 1e-12)
            close(e, g[key + "_edges"], 1e-12)
            assert isinstance(thr, float)


@pytest.mark.parametrize("svd", ["svds", "dense"])
def test_g4_greedy_pca(svd):
    g = load("g4_pca")
    inp = gc.g4_inputs()
    assert gc.digest(inp["a"], inp["b"], inp["c"], inp["area_cube"],
                     inp["areamap"]) == str(g["sha"])
    for name in ("a", "b", "c"):
        cube = inp[name]
        test, _, _, thr, mea, std = cpu_ref.Compute_PCA_threshold(cube, 0.01)
        close(test, g[name + "_test"], 1e-13)
        close([thr, mea, std], g[name + "_thr"], 1e-10)
        faint, mapO2, nstop = cpu_ref.Compute_GreedyPCA(cube, g[name + "_test"],
                                                        float(g[name + "_thr"][0]), 50, 100,
                                                        svd=svd)
        close(faint, g[name + "_faint"], 1e-10)
        assert np.array_equal(mapO2, g[name + "_mapO2"])
        assert nstop == int(g[name + "_nstop"])
    faint, mapO2, nstop = cpu_ref.Compute_GreedyPCA(inp["a"], g["a_test"],
                                                    float(g["a_thr"][0]), 50, 2, svd=svd)
    close(faint, g["a_it2_faint"], 1e-10)
    assert nstop == 1 and np.array_equal(mapO2, g["a_it2_mapO2"])


def test_g4_greedy_pca_area():
    g = load("g4_pca")
    inp = gc.g4_inputs()
    cube, areamap, nb = inp["area_cube"], inp["areamap"], inp["nbAreas"]
    testO2, _, _, thr, _, _ = cpu_ref.pca_threshold_areas(cube, areamap, nb, 0.01)
    close(thr, g["area_thr"], 1e-10)
    faint, mapO2, nstop = cpu_ref.Compute_GreedyPCA_area(nb, cube, areamap, 50, thr, 100, testO2)
    close(faint, g["area_faint"], 1e-10)
    assert np.array_equal(mapO2, g["area_mapO2"]) and nstop == int(g["area_nstop"])


@pytest.mark.parametrize("name", list("abcde"))
def test_g5_glr(name):
    g = load("g5_glr")
    c = gc.g5_inputs()[name]
    fs = c["fsf"] if isinstance(c["fsf"], list) else [c["fsf"]]
    assert gc.digest(c["cube"], *fs, *c["profiles"]) == str(g["sha"]["abcde".index(name)])
    for fn in (cpu_ref.Correlation_GLR_test, cpu_ref.Correlation_GLR_test_direct):
        kw = dict(pcut=c["pcut"], pmeansub=c["pmeansub"])
        if fn is cpu_ref.Correlation_GLR_test:
            kw["nthreads"] = 1
        correl, profile, correl_min = fn(c["cube"], c["fsf"], c["weights"], c["profiles"], **kw)
        close(correl, g[name + "_correl"], 1e-11)
        close(correl_min, g[name + "_correl_min"], 1e-11)
        assert np.mean(profile != g[name + "_profile"]) < 1e-4


def test_g5_glr_threads():
    """joblib path (nthreads > 1) gives the same numbers (lib_origin.py:1130,1204)."""
    g = load("g5_glr")
    c = gc.g5_inputs()["a"]
    correl, profile, correl_min = cpu_ref.Correlation_GLR_test(
        c["cube"], c["fsf"], c["weights"], c["profiles"], nthreads=2, pcut=c["pcut"])
    close(correl, g["a_correl"], 1e-11)


def test_g6_local_max_and_glue():
    g5 = load("g5_glr")
    g = load("g6_localmax")
    c = gc.g5_inputs()["a"]
    mask = gc.g5_mask(c["cube"].shape)
    out = cpu_ref.compute_TGLR(c["cube"], c["fsf"], c["weights"], c["profiles"], mask,
                               pcut=c["pcut"], pmeansub=c["pmeansub"])
    close(out["maxmap"], g["maxmap"], 1e-11)
    close(out["minmap"], g["minmap"], 1e-11)
    cm = g5["a_correl"].copy()
    cm[mask] = 0
    lmax, lmin = cpu_ref.compute_local_max(cm, g5["a_correl_min"], mask, 3)
    assert np.array_equal(lmax, g["local_max"]) and np.array_equal(lmin, g["local_min"])


def test_g7_chain():
    g = load("g7_chain")
    inp = gc.g7_inputs()
    assert gc.digest(inp["raw"], inp["var"], inp["mask"]) == str(g["sha"])
    out = cpu_ref.run_chain(inp["raw"].astype(float), inp["var"].astype(float), inp["mask"],
                            inp["PSF"], None, inp["profiles"], inp["areamap"], inp["nbAreas"])
    zs = g["zs"]
    close(out["thresO2"], g["thresO2"], 1e-10)
    assert np.array_equal(out["mapO2"], g["mapO2"]) and out["nstop"] == int(g["nstop"])
    close(out["cube_std"][zs], g["cube_std_z"], 1e-11)
    close(out["cube_faint"][zs], g["cube_faint_z"], 1e-10)
    close(out["cube_correl"][zs], g["correl_z"], 1e-10)
    close(out["cube_correl_min"][zs], g["correl_min_z"], 1e-10)
    close(out["maxmap"], g["maxmap"], 1e-10)
    close(out["minmap"], g["minmap"], 1e-10)
    close(out["ima_std"], g["ima_std"], 1e-10)
    for k, a in (("std", out["cube_std"]), ("faint", out["cube_faint"]),
                 ("correl", out["cube_correl"]), ("correl_min", out["cube_correl_min"])):
        close([a.mean(), a.std(), a.min(), a.max()], g["stats_" + k], 1e-9)


def test_next_fast_len():
    assert cpu_ref.next_fast_len(3681 + 59 - 1) == 3750
    for n in (7, 97, 1000, 1159, 4097):
        m = cpu_ref.next_fast_len(n)
        k = m
        for p in (2, 3, 5):
            while k % p == 0:
                k //= p
        assert k == 1 and m >= n


def test_g8_purity_threshold_oracle_matches_reference():
    """oracle.cpu_ref.Compute_threshold_purity against the reference's outputs (G8: default
    list with / without segmap, explicit unsorted list), lib_origin.py:1391-1479."""
    g = load("g8_purity")
    inp = gc.g8_inputs()
    assert str(g["sha"]) == gc.digest(inp["lmax"], inp["lmin"], inp["segmap"])
    cases = dict(seg=(inp["segmap"], None), noseg=(None, None),
                 lst=(inp["segmap"], list(inp["threshlist"])))
    for name, (segmap, tl) in cases.items():
        with np.errstate(all="ignore"):
            thr, cols = cpu_ref.Compute_threshold_purity(float(g[name + "_purity"]),
                                                         inp["lmax"].copy(), inp["lmin"].copy(),
                                                         segmap, threshlist=tl)
        assert thr == float(g[name + "_threshold"])
        for c in ("Tval_r", "Pval_r", "Det_m", "Det_M"):
            assert np.array_equal(np.asarray(cols[c], float), np.asarray(g[f"{name}_{c}"], float),
                                  equal_nan=True), (name, c)
