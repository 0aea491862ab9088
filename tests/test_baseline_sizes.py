"""GPU parity at the sizes BASELINE.json names (configs 1-3), against the float64 oracle.

config 1  3681 x 200 x 200, DCT + greedy PCA: the whole HIP run against the oracle run from the
          same raw inputs -- cube_std over the full field, cube_faint / mapO2 / iteration counts
          on two whole 100 x 100 areas (reference lib_origin.py:150-240, steps.py:431-450,
          lib_origin.py:824-954).
config 2  3681 x 300 x 300, 20 profiles: the HIP chain on the full field, the oracle GLR on
          haloed (48+24)^2 windows of the device's own cube_faint at a corner, an edge and in the
          interior (lib_origin.py:1070-1217, steps.py:781-793).
config 3  the 3681 x 600 x 600 headline workload: ``bench.py --check full`` (same windows + two
          PCA areas + a DCT window), so that the number the driver records carries the same
          correctness bit.

Tolerances: cube_std |d| <= 1e-5 max(1,|x|); cube_faint rel-Frobenius <= 2e-6 and
max-abs <= 1e-4, mapO2 / nstop identical; GLR |dT| <= 1e-4, argmax mismatches <= 0.01 %.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cpu_ref, window_check
from origin_amd import kernels, pipeline, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from origin_amd.device import default_context
    return default_context(0)


def _hip_chain(ctx, f, raw, var, mask, upto="pca"):
    d_raw, d_var = ctx.to_device(raw), ctx.to_device(var)
    d_mask = ctx.to_device(mask.astype(np.uint8))
    pre = pipeline.preprocess(ctx, d_raw, d_var, d_mask)
    thr = pipeline.pca_threshold(pre["o2_host"], f.areamap, f.nbAreas, 0.01)
    spx = pipeline.area_lists(f.areamap, f.nbAreas)
    F, mapO2, nstop, drv = pipeline.greedy_pca(ctx, pre["cube_std"], f.areamap, f.nbAreas,
                                               thr["thresO2"], thr["testO2"], 50, 100, spx=spx,
                                               o2_dev=pre["o2"])
    return dict(raw=d_raw, var=d_var, mask=d_mask, pre=pre, thr=thr, spx=spx, faint=F,
                mapO2=mapO2, nstop=nstop, iterations=drv.iterations)


def test_config1_dct_pca_3681x200x200_against_numpy(ctx):
    Nz, N = 3681, 200
    f = synth.SyntheticField(Nz, N, N)
    raw, var, mask = f.arrays()
    hip = _hip_chain(ctx, f, raw, var, mask)
    assert hip["nstop"] == 0

    # ---- oracle preprocessing of the whole field, one area (100 x 100 x 3681 float64) at a
    # time: dct_residual is per spaxel, the glue of steps.py:434-446 needs the per-channel mean
    # over ALL unmasked spaxels, accumulated here as sum / count
    spx = hip["spx"]
    zsum, zcnt = np.zeros(Nz), np.zeros(Nz)
    data_std = {}
    for a, s in enumerate(spx):
        ys, xs = np.unravel_index(s, (N, N))
        y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
        r = raw[:, y0:y1, x0:x1].astype(np.float64)
        v = var[:, y0:y1, x0:x1].astype(np.float64)
        m = mask[:, y0:y1, x0:x1]
        cont = cpu_ref.dct_residual(r, 10, v, False, m)
        data = r - cont                                            # steps.py:434
        data[m] = np.nan                                           # :435
        zsum += np.nansum(data, axis=(1, 2))
        zcnt += np.sum(~m, axis=(1, 2))
        data_std[a] = (data, np.sqrt(v), m, (y0, y1, x0, x1))      # :439
    mean = zsum / zcnt                                             # :442
    cube_std = np.empty((Nz, N, N))
    for a, (data, std, m, (y0, y1, x0, x1)) in data_std.items():
        data -= mean[:, None, None]                                # :444
        data /= std                                                # :445
        data[m] = 0                                                # :446
        cube_std[:, y0:y1, x0:x1] = data
    del data_std
    got = hip["pre"]["cube_std"].to_host()
    err = np.abs(got - cube_std) / np.maximum(1.0, np.abs(cube_std))
    assert err.max() <= 1e-5, err.max()

    # ---- thresholds and greedy PCA of two whole areas, oracle end to end from its own cube_std
    for a in (0, f.nbAreas - 1):
        X = cube_std.reshape(Nz, -1)[:, spx[a]]
        test = cpu_ref.O2test(X)
        _, _, thr, mea, sd = cpu_ref.compute_thresh_gaussfit(test, 0.01)
        assert abs(hip["thr"]["thresO2"][a] - thr) <= 1e-5 * thr
        trace = []
        faint, m2, nstop = cpu_ref.Compute_GreedyPCA(X, test, thr, 50, 100, trace=trace)
        assert nstop == 0 and len(trace) >= 3
        gF = hip["faint"].to_host().reshape(Nz, -1)[:, spx[a]]
        d = gF - faint
        assert np.linalg.norm(d) <= 2e-6 * np.linalg.norm(faint)
        assert np.max(np.abs(d)) <= 1e-4
        assert np.array_equal(hip["mapO2"].reshape(-1)[spx[a]], m2), a
        assert int(m2.max()) == len(trace)


def test_config2_glr_3681x300x300_oracle_windows(ctx):
    Nz, N = 3681, 300
    f = synth.SyntheticField(Nz, N, N)
    raw, var, mask = f.arrays()
    hip = _hip_chain(ctx, f, raw, var, mask)
    del raw, var
    psf = f.PSF.astype(np.float64)
    plan = kernels.GLRPlan(ctx, (Nz, N, N), psf, None, f.profiles, 1e-8, True)
    assert plan.precision == "f16x2" and plan.spatial_on_matrix_cores
    out = pipeline.tglr(ctx, plan, hip["faint"], hip["mask"], want_local=False)
    ctx.sync()
    ncpu = min(32, os.cpu_count() or 1)
    for w in window_check.glr_windows(N, N):
        res = window_check.check_glr_window(hip["faint"], out, hip["mask"], psf, f.profiles, w,
                                            nthreads=ncpu)
        assert res["ok"], res
        assert res["T_range"][1] > 5.0          # the window saw real signal, not zeros
    # the stage before it, on two areas of this field too (one has the brightest blob)
    for a in (0, 4):
        res = window_check.check_pca_area(hip["pre"]["cube_std"], hip["faint"], hip["mapO2"],
                                          hip["spx"][a], hip["thr"]["thresO2"][a], a)
        assert res["ok"], res
    plan.close()


def test_masked_border_3681x300x300(ctx):
    """SURVEY 8(d)'s masked-border variant at size: 5 spaxels along every edge masked in every
    channel (raw 0, var inf -- origin.py:262-274).  DCT window across the border (masked spaxels
    exactly 0, the plain-DCT fallback of lib_origin.py:226 never reached for them); the corner area
    holds O2 == 0 spaxels (the index quirk of lib_origin.py:908-917) and its mapO2 is identical;
    GLR corner / edge windows with correl[mask] = 0 (steps.py:781); local maxima bit exact, in the
    sparse form the chain keeps."""
    from origin_amd import sparse
    Nz, N = 3681, 300
    f = synth.SyntheticField(Nz, N, N, masked_border=5)
    raw, var, mask = f.arrays()
    assert mask[:, :5].all() and mask[:, :, -5:].all() and not mask[:, 5:-5, 5:-5].any()
    hip = _hip_chain(ctx, f, raw, var, mask)
    assert hip["nstop"] == 0
    o2 = hip["pre"]["o2_host"]
    assert np.all(o2[:5] == 0) and np.all(o2[:, -5:] == 0) and np.all(o2[5:-5, 5:-5] > 0)
    for w in (("dct_border", 0, 16, 140, 164), ("dct_corner", 284, 300, 280, 300)):
        res = window_check.check_dct_window(hip["raw"], hip["var"], hip["mask"],
                                            hip["pre"]["cube_std"], hip["pre"]["cont_dct"], w)
        res.pop("_zmean")
        assert res["ok"], res
    # the corner area (masked spaxels among its columns) and an interior one
    for a in (0, 4):
        res = window_check.check_pca_area(hip["pre"]["cube_std"], hip["faint"], hip["mapO2"],
                                          hip["spx"][a], hip["thr"]["thresO2"][a], a)
        assert res["ok"] and res["iterations"] >= 2, res
    psf = f.PSF.astype(np.float64)
    plan = kernels.GLRPlan(ctx, (Nz, N, N), psf, None, f.profiles, 1e-8, True)
    out = pipeline.tglr(ctx, plan, hip["faint"], hip["mask"])
    ctx.sync()
    assert isinstance(out["local_max"], sparse.SparseCube)
    dense = dict(out, local_max=out["local_max"].dense(), local_min=out["local_min"].dense())
    ncpu = min(32, os.cpu_count() or 1)
    for w in window_check.glr_windows(N, N, which=("corner", "edge", "far_corner")):
        res = window_check.check_glr_window(hip["faint"], dense, hip["mask"], psf, f.profiles, w,
                                            nthreads=ncpu)
        assert res["ok"] and res["local_max_mismatch"] == 0, res
    m = mask[0]
    assert np.all(out["maxmap"].to_host()[m] == 0)           # correl[mask] = 0 in every channel
    assert np.all(dense["local_max"].window(0, 5, 0, N) == 0)
    plan.close()


def test_masked_border_two_ranks_on_one_gpu():
    """The same variant tiled over two ranks that share the GPU (`bench.py --gpus 2
    --masked-border 5`, host-staged strips): runs end to end, and the single-rank run of the same
    field passes its oracle check (DCT border window, corner area, GLR windows)."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", "200", "--masked-border", "5",
            "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--e2e-size", "0"]
    env = dict(os.environ, ORIGIN_BENCH_SHARE_GPU="1")
    r = subprocess.run(base + ["--gpus", "2", "--check", "off"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    two = json.loads(r.stdout.decode().strip().splitlines()[-1])
    r = subprocess.run(base + ["--check", "full"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    one = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert one["check"]["ok"], one["check"]
    assert any(d_["window"] == "dct_border" for d_ in one["check"]["dct"])
    assert 0 in [p_["area"] for p_ in one["check"]["pca"]]
    assert two["config"]["masked_border"] == 5 and two["config"]["tiles"] == 2
    assert sum(two["per_rank"]["areas"]) == 4
    assert sum(two["per_rank"]["pca_iterations"]) > 0


def test_config3_headline_600_bench_check():
    """bench.py on the headline workload with --check full: three GLR windows, two PCA areas
    and a DCT window of the very arrays the timed steps produced."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1",
           "--check", "full", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    chk = line["check"]
    assert chk["level"] == "full" and len(chk["glr"]) == 3 and len(chk["pca"]) == 2
    assert chk["ok"], chk
    assert line["config"]["workload"].startswith("synthetic 3681x600x600")


def test_config5_900_bf16_bench_check():
    """BASELINE config 5's workload on one GPU (the largest single-GPU configuration): synthetic
    3681 x 900 x 900, fp32 PCA + bf16 GLR, `bench.py --check full` at the bf16 tolerances of
    SURVEY 8c (|dT| <= 5e-2, rms <= 5e-3, arg-max mismatch <= 2 %); the PCA / DCT checks keep the
    fp32 tolerances."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", "900", "--glr-precision",
           "bf16", "--steps", "1", "--warmup", "1", "--check", "full", "--no-cpu-baseline",
           "--e2e-size", "0"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    chk = line["check"]
    assert line["dtype"] == "f32+bf16" and line["config"]["glr_spectral_arithmetic"] == "bf16"
    assert line["config"]["workload"].startswith("synthetic 3681x900x900")
    assert chk["level"] == "full" and len(chk["glr"]) == 3 and len(chk["pca"]) == 2
    assert "bf16" in chk["tolerances"]
    print(json.dumps(chk, indent=1))   # (shown in full when the test fails)
    assert chk["ok"]
    assert all(p_["mapO2_mismatch"] == 0 for p_ in chk["pca"])


def test_config5_900_f16x2_glr_windows_only():
    """The same field with the fp32-class GLR (two-term f16 split): only the GLR windows are
    checked again (|dT| <= 1e-4) -- DCT and PCA do not depend on the GLR's arithmetic and were
    checked by the bf16 twin above."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", "900", "--steps", "1",
           "--warmup", "1", "--check", "glr", "--no-cpu-baseline", "--e2e-size", "0"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    chk = line["check"]
    assert line["dtype"] == "f32+f16x2" and len(chk["glr"]) == 3 and not chk["pca"]
    print(json.dumps(chk, indent=1))
    assert chk["ok"]


def test_bench_two_ranks_on_one_gpu_prints_its_line():
    """`python bench.py --gpus 2` end to end on ONE device (ORIGIN_BENCH_SHARE_GPU=1: the ranks
    share the GPU and the halo strips go through the host -- the transport is not what is tested):
    the parent spawns its ranks, every rank runs the tiled step (PCA into the extended tile, halo
    exchange, GLR and local maxima without a crop), rank 0's line comes back with the per-rank
    table, exit status 0.  A change to the step that breaks the multi-rank branch shows up here and
    not first on the multi-GPU node."""
    env = dict(os.environ, ORIGIN_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "200",
           "--steps", "2", "--warmup", "1", "--check", "off", "--no-cpu-baseline", "--e2e-size", "0"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT,
                       env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["config"]["tiles"] == 2 and line["n_gpus"] == 1   # (one device, honestly reported)
    assert line["value"] > 0 and line["sequential"] is None
    pr = line["per_rank"]
    assert len(pr["pca_iterations"]) == 2 and sum(pr["areas"]) == 4
    assert set(pr["phases_ms"]) >= {"dct_std", "greedy_pca"}
