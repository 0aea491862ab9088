"""bench.py's roofline block must follow from the counter passes committed under profiles/
(VERDICT r2 #2): `executed` from the MFMA instruction count the plan issues, `traffic` with the
gfx950 FETCH_SIZE correction.  CPU only: the plan-free count model of the C ABI + the profiles."""
import csv
import ctypes as C
import glob
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _count_model(terms=3, K=20, n_narrow=10, Nz=3681, N=600, P=25, num_cu=256):
    from origin_amd import _capi
    a, b = C.c_long(), C.c_long()
    _capi.call("origin_glr_mfma_count_model", num_cu, terms, K, n_narrow, Nz, N, N, P,
               C.byref(a), C.byref(b))
    return a.value, b.value


def _profile_pair():
    """Newest (counter json, kernel stats csv) pair of one round that holds both kernels."""
    for tag in ("r04", "r03", "r02"):
        pj = os.path.join(ROOT, "profiles", f"{tag}_glr_pmc.json")
        pc = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")
        if os.path.exists(pj) and os.path.exists(pc):
            return tag, json.load(open(pj)), list(csv.DictReader(open(pc)))
    pytest.skip("no committed counter profile")


def _avg_ns(rows, kernel):
    for r in rows:
        if kernel in r["Name"]:
            return float(r["AverageNs"])
    raise KeyError(kernel)


def test_mfma_count_model_matches_the_counter_pass():
    """Dico_FWHM_2_12 at 3681 x 600 x 600: 10 narrow + 10 wide profiles, P = 25."""
    tag, pmc, _ = _profile_pair()
    n_sp, n_sc = _count_model()
    assert abs(n_sc / pmc["spectral_mfma2_kernel"]["SQ_INSTS_MFMA"] - 1) <= 0.01
    assert abs(n_sp / pmc["spatial2_kernel"]["SQ_INSTS_MFMA"] - 1) <= 0.01
    # 240 per 32 channels x 32 spaxels, 168 per 1024 outputs -- plus the partial edge tiles
    vox = 3681 * 600 * 600
    assert 240 <= 1024 * n_sc / vox <= 244 and 168 <= 1024 * n_sp / vox <= 192


def test_executed_follows_from_counters_and_kernel_stats():
    """SQ_INSTS_MFMA x 32768 / AverageNs of the committed rocprofv3 passes against the formula
    the bench line uses, within 3 %."""
    b = _bench()
    tag, pmc, rows = _profile_pair()
    n_sp, n_sc = _count_model()
    for kernel, n_model in (("spectral_mfma2_kernel", n_sc), ("spatial2_kernel", n_sp)):
        avg_s = _avg_ns(rows, kernel) * 1e-9
        from_counters = pmc[kernel]["SQ_INSTS_MFMA"] * 32768.0 / avg_s / 1e12
        from_bench = b.executed_tflops(n_model, avg_s)
        assert abs(from_bench / from_counters - 1) <= 0.03, (kernel, from_bench, from_counters)
        assert from_bench / b.F16_MFMA_PEAK_TFLOPS < 0.5      # (round 2's line said 0.438 at 0.353)


def test_traffic_uses_the_fetch_correction():
    b = _bench()
    name, pmc = b.load_pmc_profile()
    assert pmc is not None
    vox = 3681 * 600 * 600
    for kernel, must_read, must_write in (("spectral_mfma2_kernel", 5.0 * vox, 9.0 * vox),
                                          ("dct_moments_kernel", 9.0 * vox, 0.0)):
        t = b.traffic_from_profile(pmc, kernel)
        assert t is not None
        assert t >= 0.97 * (must_read + must_write), (kernel, t)   # never below the compulsory bytes
        raw = (pmc[kernel]["FETCH_SIZE_GB_per_launch"] + pmc[kernel]["WRITE_SIZE_GB_per_launch"]) * 1e9
        assert t > raw


def test_gpus_n_without_a_launcher_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` starts its own ranks; here (no GPU) every rank fails at the
    first device call and the parent must exit non-zero, quickly, with no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "200",
                        "--steps", "1", "--warmup", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: covered by the rehearsal under profiles/")
    assert r.returncode != 0
    assert b"exited with code" in r.stderr
    assert not r.stdout.strip()
