"""Where the PCIe-inclusive pass (bench.py `e2e`) spends its time: cProfile of the Step seam on a
3681 x 200 x 200 sub-field, host arrays in, float64 host arrays out."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import synth  # noqa: E402
from origin_amd.device import default_context  # noqa: E402
from origin_amd.steps import SimpleOrig  # noqa: E402


def run(fe, eraw, evar, emask, ctx):
    o = SimpleOrig(eraw, evar, emask, fe.PSF.astype(np.float64), fe.profiles, ctx=ctx)
    o.step01_preprocessing()
    o.step02_areas.set_areamap(fe.areamap)
    o.step03_compute_PCA_threshold()
    o.step04_compute_greedy_PCA()
    o.step05_compute_TGLR()
    return [o.cube_std._data, o.cube_faint._data, o.cube_correl._data, o.maxmap]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ctx = default_context(0)
    fe = synth.SyntheticField(3681, n, n, None, 25, 20, 0, 1.0 / 400, 1.0 / 900, 100)
    eraw, evar, emask = fe.arrays()
    run(fe, eraw, evar, emask, ctx)
    t = time.perf_counter()
    run(fe, eraw, evar, emask, ctx)
    print(f"second pass: {time.perf_counter() - t:.3f} s")
    pr = cProfile.Profile()
    pr.enable()
    run(fe, eraw, evar, emask, ctx)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(28)


if __name__ == "__main__":
    main()
