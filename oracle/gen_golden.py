"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz from the REAL reference.

Run in the build container only:

    /opt/conda/bin/python3.9 oracle/gen_golden.py

It imports the reference's ``lib_origin.py`` unmodified (oracle/ref_import.py), feeds
it the seeded inputs of oracle/golden_cases.py, stores the reference's outputs, and
at the same time checks the CPU restatement (oracle/cpu_ref.py) against them, writing
the comparison to oracle/PINNING_REPORT.txt.  Step ``run`` bodies cannot be imported
(they need mpdaf); their few glue lines (steps.py:434-446, :781-793) are applied here
to the reference functions' outputs in the same order.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import golden_cases as gc  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from oracle.ref_import import load_reference  # noqa: E402

ref = load_reference()
REPORT = []

# Inputs are produced by the SYSTEM python (the one that runs the tests) and loaded
# here, so their bits -- and the sha256 stored in each golden -- match in the tests.
import subprocess  # noqa: E402
import tempfile  # noqa: E402

_tmp = os.path.join(tempfile.mkdtemp(prefix="origin_golden_in_"), "inputs.npz")
subprocess.check_call([os.environ.get("SYSTEM_PYTHON", "/usr/bin/python3"),
                       os.path.join(HERE, "golden_cases.py"), _tmp],
                      env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
INPUTS = gc.load_inputs(_tmp)


def rel(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    d = np.max(np.abs(a - b)) if a.size else 0.0
    s = max(np.max(np.abs(b)) if b.size else 0.0, 1e-300)
    return d / s


def report(name, what, err, tol=1e-10):
    ok = err <= tol
    line = f"{name:10s} {what:28s} max-rel-err {err:9.2e}  (tol {tol:.0e})  {'OK' if ok else 'FAIL'}"
    print(line)
    REPORT.append(line)
    if not ok:
        raise SystemExit("oracle disagrees with the reference: " + line)


def save(name, **arrays):
    os.makedirs(gc.GOLDEN_DIR, exist_ok=True)
    path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"   wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def gen_g1_g2():
    inp = INPUTS["g1"]
    raw, var, mask = (inp["raw"].astype(float), inp["var"].astype(float), inp["mask"])
    out = {}
    for approx in (False, True):
        cont = ref.dct_residual(raw, int(inp["order"]), var, approx, mask)
        mine = cpu_ref.dct_residual(raw, int(inp["order"]), var, approx, mask)
        report("G1", f"dct_residual approx={approx}", rel(mine, cont))
        out["cont_approx" if approx else "cont"] = cont
    report("G1", "DCTMAT", rel(cpu_ref.DCTMAT(256, 10), ref.DCTMAT(256, 10)), 0.0)
    save("g1_dct", sha=gc.digest(inp["raw"], inp["var"], inp["mask"]), **out)

    # G2: glue of Preprocessing.run (steps.py:431-450, 463-465) on the reference function
    for approx in (False, True):
        cont_dct = ref.dct_residual(raw, int(inp["order"]), var, approx, mask)
        data = raw - cont_dct
        data[mask] = np.nan
        std = np.sqrt(var)
        cont_dct /= std
        mean = np.nanmean(data, axis=(1, 2))
        data -= mean[:, np.newaxis, np.newaxis]
        data /= std
        data[mask] = 0
        cont32 = cont_dct.astype(np.float32)
        g = dict(cube_std=data, ima_std=data.mean(axis=0), cont_dct=cont32,
                 ima_dct=cont32.mean(axis=0), o2=ref.O2test(data))
        mine = cpu_ref.preprocessing(raw, var, mask, int(inp["order"]), approx)
        for k in ("cube_std", "ima_std", "cont_dct", "ima_dct"):
            report("G2", f"{k} approx={approx}", rel(mine[k], g[k]), 1e-6 if "dct" in k else 1e-10)
        report("G2", "O2test", rel(cpu_ref.O2test(data), g["o2"]), 0.0)
        save("g2_preproc_approx" if approx else "g2_preproc",
             sha=gc.digest(inp["raw"], inp["var"], inp["mask"]), **g)


def gen_g3():
    inp = INPUTS["g3"]
    out = {}
    for name, t in inp.items():
        for pfa in (0.01, 0.2):
            h, e, thr, mea, std = ref.compute_thresh_gaussfit(t, pfa)
            h2, e2, thr2, mea2, std2 = cpu_ref.compute_thresh_gaussfit(t, pfa)
            report("G3", f"{name} pfa={pfa} thr/mea/std",
                   max(rel(thr2, thr), rel(mea2, mea), rel(std2, std)), 1e-9)
            report("G3", f"{name} pfa={pfa} hist/edges", max(rel(h2, h), rel(e2, e)), 1e-12)
            key = f"{name}_{str(pfa).replace('.', 'p')}"
            out[key + "_res"] = np.array([thr, mea, std])
            out[key + "_hist"] = h
            out[key + "_edges"] = e
    # inputs are float64 results of libm/numpy reductions (not bit-stable across numpy
    # versions), so G3 stores them next to the outputs
    for name, t in inp.items():
        out["in_" + name] = t
    save("g3_thresh", **out)


def gen_g4():
    inp = INPUTS["g4"]
    out = {}
    for name in ("a", "b", "c"):
        cube = inp[name]
        test, h, e, thr, mea, std = ref.Compute_PCA_threshold(cube, 0.01)
        np.random.seed(1234)
        faint, mapO2, nstop = ref.Compute_GreedyPCA(cube, test, thr, 50, 100)
        t2 = cpu_ref.Compute_PCA_threshold(cube, 0.01)
        report("G4", f"{name} threshold", rel(t2[3], thr), 1e-9)
        trace = []
        f2, m2, n2 = cpu_ref.Compute_GreedyPCA(cube, test, thr, 50, 100, trace=trace)
        report("G4", f"{name} faint (svds)", rel(f2, faint), 1e-9)
        f3, m3, n3 = cpu_ref.Compute_GreedyPCA(cube, test, thr, 50, 100, svd="dense")
        report("G4", f"{name} faint (dense svd)", rel(f3, faint), 1e-9)
        assert np.array_equal(m2, mapO2) and np.array_equal(m3, mapO2) and n2 == nstop
        print(f"   G4 {name}: {int(mapO2.max())} iterations, trace(n_nuis, nb)={trace[:6]}...")
        out[name + "_test"] = test
        out[name + "_thr"] = np.array([thr, mea, std])
        out[name + "_faint"] = faint
        out[name + "_mapO2"] = mapO2
        out[name + "_nstop"] = np.array(nstop)
    # itermax guard (lib_origin.py:902-905): stop after 2 iterations
    cube = inp["a"]
    test = ref.O2test(cube)
    thr = float(out["a_thr"][0])
    faint, mapO2, nstop = ref.Compute_GreedyPCA(cube, test, thr, 50, 2)
    f2, m2, n2 = cpu_ref.Compute_GreedyPCA(cube, test, thr, 50, 2)
    report("G4", "a itermax=2", rel(f2, faint), 1e-9)
    assert n2 == nstop == 1 and np.array_equal(m2, mapO2)
    out["a_it2_faint"] = faint
    out["a_it2_mapO2"] = mapO2
    out["a_it2_nstop"] = np.array(nstop)

    # area version
    cube = inp["area_cube"]
    areamap, nb = inp["areamap"], int(inp["nbAreas"])
    res = [ref.Compute_PCA_threshold(cube[:, areamap == i], 0.01) for i in range(1, nb + 1)]
    testO2 = [r[0] for r in res]
    thr = [r[3] for r in res]
    faint, mapO2, nstop = ref.Compute_GreedyPCA_area(nb, cube, areamap, 50, thr, 100, testO2)
    f2, m2, n2 = cpu_ref.Compute_GreedyPCA_area(nb, cube, areamap, 50, thr, 100, testO2)
    report("G4", "area faint", rel(f2, faint), 1e-9)
    assert np.array_equal(m2, mapO2) and n2 == nstop
    t3 = cpu_ref.pca_threshold_areas(cube, areamap, nb, 0.01)
    report("G4", "area thresholds", rel(np.array(t3[3]), np.array(thr)), 1e-9)
    out["area_thr"] = np.array(thr)
    out["area_faint"] = faint
    out["area_mapO2"] = mapO2
    out["area_nstop"] = np.array(nstop)
    print(f"   G4 area: iterations per area max {int(mapO2.max())}, thr={np.round(thr, 4)}")
    save("g4_pca", sha=gc.digest(inp["a"], inp["b"], inp["c"], inp["area_cube"],
                                  inp["areamap"]), **out)


def PC(c):
    return None if c["pcut"] is None else float(c["pcut"])


def gen_g5_g6():
    inp = INPUTS["g5"]
    out = {}
    shas = []
    for name, c in inp.items():
        correl, profile, correl_min = ref.Correlation_GLR_test(
            c["cube"], c["fsf"], c["weights"], c["profiles"], nthreads=1, pcut=PC(c),
            pmeansub=bool(c["pmeansub"]))
        profile = np.where(np.isfinite(correl), profile, 0)  # np.empty in the reference
        m = cpu_ref.Correlation_GLR_test(c["cube"], c["fsf"], c["weights"], c["profiles"],
                                         nthreads=1, pcut=PC(c), pmeansub=bool(c["pmeansub"]))
        report("G5", f"{name} correl (fft restatement)", rel(m[0], correl), 1e-11)
        report("G5", f"{name} correl_min", rel(m[2], correl_min), 1e-11)
        assert np.array_equal(m[1], profile), "profile index mismatch"
        d = cpu_ref.Correlation_GLR_test_direct(c["cube"], c["fsf"], c["weights"],
                                                c["profiles"], pcut=PC(c),
                                                pmeansub=bool(c["pmeansub"]))
        report("G5", f"{name} correl (direct algebra)", rel(d[0], correl), 1e-10)
        report("G5", f"{name} correl_min (direct)", rel(d[2], correl_min), 1e-10)
        nmis = int(np.sum(d[1] != profile))
        line = f"G5         {name} profile mismatches direct-vs-ref: {nmis} / {profile.size}"
        print(line)
        REPORT.append(line)
        out[name + "_correl"] = correl
        out[name + "_profile"] = profile.astype(np.uint8)
        out[name + "_correl_min"] = correl_min
        fs = c["fsf"] if isinstance(c["fsf"], list) else [c["fsf"]]
        shas.append(gc.digest(c["cube"], *fs, *c["profiles"]))
        if name == "a":
            # ComputeTGLR glue (steps.py:781-793) + compute_local_max (G6)
            mask = gc.g5_mask(correl.shape)
            cm = correl.copy()
            cm[mask] = 0
            lmax, lmin = ref.compute_local_max(cm, correl_min, mask, 3)
            l2 = cpu_ref.compute_local_max(cm.copy(), correl_min, mask, 3)
            report("G6", "local_max", rel(l2[0], lmax), 0.0)
            report("G6", "local_min", rel(l2[1], lmin), 0.0)
            g = cpu_ref.compute_TGLR(c["cube"], c["fsf"], c["weights"], c["profiles"], mask,
                                     pcut=PC(c), pmeansub=bool(c["pmeansub"]))
            report("G6", "glue maxmap", rel(g["maxmap"], np.amax(cm, axis=0)), 1e-11)
            np.savez_compressed(os.path.join(gc.GOLDEN_DIR, "g6_localmax.npz"),
                                sha=shas[-1], local_max=lmax, local_min=lmin,
                                maxmap=np.amax(cm, axis=0),
                                minmap=np.amin(correl_min, axis=0))
    save("g5_glr", sha=np.array(shas), **out)


def gen_g7():
    inp = INPUTS["g7"]
    raw, var, mask = inp["raw"].astype(float), inp["var"].astype(float), inp["mask"]
    t0 = time.time()
    # --- step 1 (steps.py:431-450)
    cont = ref.dct_residual(raw, 10, var, False, mask)
    data = raw - cont
    data[mask] = np.nan
    std = np.sqrt(var)
    cont /= std
    mean = np.nanmean(data, axis=(1, 2))
    data -= mean[:, None, None]
    data /= std
    data[mask] = 0
    t1 = time.time()
    # --- step 3 (steps.py:610-631)
    areamap, nb = inp["areamap"], int(inp["nbAreas"])
    res = [ref.Compute_PCA_threshold(data[:, areamap == i], 0.01) for i in range(1, nb + 1)]
    testO2 = [r[0] for r in res]
    thr = [r[3] for r in res]
    t2 = time.time()
    # --- step 4 (steps.py:681-704)
    faint, mapO2, nstop = ref.Compute_GreedyPCA_area(nb, data, areamap, 50, thr, 100, testO2)
    t3 = time.time()
    # --- step 5 (steps.py:770-793)
    correl, profile, correl_min = ref.Correlation_GLR_test(
        faint, inp["PSF"], None, inp["profiles"], nthreads=1, pcut=1e-8, pmeansub=True)
    correl[mask] = 0
    profile[mask] = 0
    maxmap = np.amax(correl, axis=0)
    minmap = np.amin(correl_min, axis=0)
    t4 = time.time()
    line = (f"G7 reference timings: dct+std {t1 - t0:.2f}s  thr {t2 - t1:.2f}s  "
            f"pca {t3 - t2:.2f}s  glr {t4 - t3:.2f}s")
    print(line)
    REPORT.append(line)

    tm = {}
    mine = cpu_ref.run_chain(raw, var, mask, inp["PSF"], None, inp["profiles"], areamap, nb,
                             timings=tm)
    line = "G7 oracle    timings: " + "  ".join(f"{k} {v:.2f}s" for k, v in tm.items())
    print(line)
    REPORT.append(line)
    report("G7", "cube_std", rel(mine["cube_std"], data))
    report("G7", "thresholds", rel(mine["thresO2"], np.array(thr)), 1e-9)
    report("G7", "cube_faint", rel(mine["cube_faint"], faint), 1e-9)
    assert np.array_equal(mine["mapO2"], mapO2) and mine["nstop"] == nstop
    report("G7", "cube_correl", rel(mine["cube_correl"], correl), 1e-9)
    report("G7", "maxmap", rel(mine["maxmap"], maxmap), 1e-9)
    report("G7", "minmap", rel(mine["minmap"], minmap), 1e-9)
    zs = [100, 550, 1000]

    def stats(a):
        return np.array([a.mean(), a.std(), a.min(), a.max()])

    save("g7_chain", sha=gc.digest(inp["raw"], inp["var"], inp["mask"]),
         thresO2=np.array(thr), mapO2=mapO2, nstop=np.array(nstop), maxmap=maxmap,
         minmap=minmap, zs=np.array(zs), cube_std_z=data[zs], cube_faint_z=faint[zs],
         correl_z=correl[zs], correl_min_z=correl_min[zs], profile_z=profile[zs],
         stats_std=stats(data), stats_faint=stats(faint), stats_correl=stats(correl),
         stats_correl_min=stats(correl_min), ima_std=data.mean(axis=0))


def check_dictionary():
    """The analytic dictionary equals the reference's FITS files."""
    from astropy.io import fits
    for fname, n in (("Dico_FWHM_2_12.fits", 20), ("Dico_3FWHM.fits", 3)):
        with fits.open(os.path.join("/root/reference/muse_origin", fname)) as hdul:
            profs = [h.data.astype(float) for h in hdul[1:]]
        mine = gc.synth.dico_fwhm(n)
        assert len(profs) == n
        err = max(np.max(np.abs(a - b)) for a, b in zip(mine, profs))
        report("DICO", fname, err, 1e-15)
    taps = [len(p) for p in cpu_ref.prepare_profiles(gc.synth.dico_fwhm(20), 1e-8)]
    line = f"DICO       trimmed taps (pcut=1e-8): {taps} sum={sum(taps)}"
    print(line)
    REPORT.append(line)


def gen_g8():
    """Compute_threshold_purity (lib_origin.py:1391-1479): default threshold list with and
    without segmap, explicit (unsorted) list."""
    inp = INPUTS["g8"]
    out = {}
    cases = dict(seg=(0.9, inp["segmap"], None), noseg=(0.8, None, None),
                 lst=(0.7, inp["segmap"], list(inp["threshlist"])))
    for name, (purity, segmap, tl) in cases.items():
        with np.errstate(all="ignore"):
            thr, res = ref.Compute_threshold_purity(purity, inp["lmax"].copy(), inp["lmin"].copy(),
                                                    segmap, threshlist=tl)
            thr2, cols = cpu_ref.Compute_threshold_purity(purity, inp["lmax"].copy(),
                                                          inp["lmin"].copy(), segmap, threshlist=tl)
        for c in ("Tval_r", "Pval_r", "Det_m", "Det_M"):
            a, b = np.asarray(res[c], dtype=float), np.asarray(cols[c], dtype=float)
            ok = np.array_equal(a, b, equal_nan=True)
            report("G8", f"{name} {c}", 0.0 if ok else 1.0, 0.0)
            out[f"{name}_{c}"] = np.asarray(res[c])
        report("G8", f"{name} threshold", 0.0 if (thr == thr2 or abs(thr - thr2) < 1e-12) else 1.0,
               0.0)
        out[f"{name}_threshold"] = np.array(thr)
        out[f"{name}_purity"] = np.array(purity)
    save("g8_purity", sha=np.array(gc.digest(inp["lmax"], inp["lmin"], inp["segmap"])), **out)


def gen_g10():
    """Area construction: the reference's functions chained exactly as CreateAreas.run does
    (steps.py:530-563), with the stages kept so that a mismatch can be located."""
    inp = INPUTS["g10"]
    out = {}
    for name, c in inp.items():
        mask, segmap = np.asarray(c["mask"], dtype=bool), np.asarray(c["segmap"])
        minsize = int(c["minsize"])
        maxsize = None if c["maxsize"] is None else int(c["maxsize"])
        Ny, Nx = segmap.shape
        nexpmap = (np.sum(~mask, axis=0) > 0).astype(int)
        NbSubcube = np.maximum(1, int(np.sqrt(np.sum(nexpmap) / (minsize ** 2))))
        assert NbSubcube > 1
        if maxsize is None:
            maxsize = minsize * 2
        MinSize, MaxSize = minsize ** 2, maxsize ** 2
        sq = ref.area_segmentation_square_fusion(nexpmap, MinSize, MaxSize, NbSubcube, Ny, Nx)
        sq_sizes = sq.sum(axis=(1, 2))
        ws, src = ref.area_segmentation_sources_fusion(segmap, sq.copy(), float(c["pfa"]), Ny, Nx)
        hull = ref.area_segmentation_convex_fusion(ws, src)
        grown = ref.area_growing(hull, nexpmap)
        # (area_segmentation_final merges in place: give it a copy, keep the grown planes)
        areamap = ref.area_segmentation_final(grown.copy(), MinSize, MaxSize).astype(int)
        labels = np.unique(areamap)
        nb = len(labels) - (1 if 0 in labels else 0)
        planes = lambda st: (np.arange(1, len(st) + 1)[:, None, None] * (st > 0)).sum(axis=0)
        out.update({f"{name}_squares": planes(sq).astype(np.int16), f"{name}_square_sizes": sq_sizes,
                    f"{name}_with_src": planes(ws).astype(np.int16), f"{name}_src": src,
                    f"{name}_hulls": planes(hull).astype(np.int16),
                    f"{name}_grown": planes(grown).astype(np.int16),
                    f"{name}_areamap": areamap.astype(np.int16), f"{name}_nbareas": np.array(nb),
                    f"{name}_nsub": np.array(NbSubcube)})
        line = (f"G10        {name}: {NbSubcube}^2 squares -> {len(sq)} -> hulls {len(hull)} -> "
                f"{nb} areas, sizes {sorted(int(x) for x in np.bincount(areamap.ravel())[1:])}")
        print(line)
        REPORT.append(line)
    # isolated pieces: both merge criteria of fusion_areas, one hull
    rng = np.random.default_rng(3)
    lab = np.zeros((5, 40, 44))
    lab[0, :12, :20] = 1
    lab[1, :12, 20:] = 1
    lab[2, 12:, :9] = 1
    lab[3, 12:30, 9:] = 1
    lab[4, 30:, 9:] = 1
    out["fusion_min"] = ref.fusion_areas(lab.copy(), 300, 900).sum(axis=(1, 2))
    out["fusion_var"] = ref.fusion_areas(lab.copy(), 300, 900, option='var').sum(axis=(1, 2))
    pts = np.unique(rng.integers(0, 30, size=(25, 2)), axis=0)
    pts = pts - pts.min(axis=0)
    out["hull_points"] = pts
    out["hull_filled"] = np.asarray(ref.Convexline(pts.copy(), 0, 0)).astype(np.uint8)
    save("g10_areas", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["dico", "g1", "g3", "g4", "g5", "g7", "g8", "g10"]
    if "dico" in which:
        check_dictionary()
    if "g1" in which:
        gen_g1_g2()
    if "g3" in which:
        gen_g3()
    if "g4" in which:
        gen_g4()
    if "g5" in which:
        gen_g5_g6()
    if "g7" in which:
        gen_g7()
    if "g8" in which:
        gen_g8()
    if "g10" in which:
        gen_g10()
    if not sys.argv[1:]:
        with open(os.path.join(HERE, "PINNING_REPORT.txt"), "w") as f:
            f.write("oracle/cpu_ref.py vs the reference's lib_origin.py (imported unmodified), "
                    "written by oracle/gen_golden.py\n")
            f.write(f"numpy {np.__version__}\n")
            f.write("\n".join(REPORT) + "\n")
