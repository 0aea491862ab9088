"""Wall time of the reference-signature functions (B1 seam: float64 ndarrays in and out, every
call crosses PCIe) on a 3681 x N x N synthetic cube."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from origin_amd import synth  # noqa: E402
import origin_amd.lib_origin as lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    f = synth.SyntheticField(3681, n, n, None, 25, 20, 0, 1.0 / 400, 1.0 / 900, 100)
    raw, var, mask = f.arrays()
    raw, var = raw.astype(np.float64), var.astype(np.float64)
    vox = raw.size
    for rep in range(2):
        t0 = time.perf_counter()
        cont = lib.dct_residual(raw, 10, var, False, mask)
        t1 = time.perf_counter()
        data = (raw - cont)
        data -= np.mean(data, axis=(1, 2))[:, None, None]
        data /= np.sqrt(var)
        t2 = time.perf_counter()
        thr = [lib.Compute_PCA_threshold(data[:, f.areamap == i], 0.01) for i in
               range(1, f.nbAreas + 1)]
        t3 = time.perf_counter()
        faint, mapO2, nstop = lib.Compute_GreedyPCA_area(f.nbAreas, data, f.areamap, 50,
                                                         [t[3] for t in thr], 100,
                                                         [t[0] for t in thr])
        t4 = time.perf_counter()
        correl, profile, cmin = lib.Correlation_GLR_test(faint, f.PSF.astype(np.float64), None,
                                                         f.profiles, nthreads=1, pcut=1e-8)
        t5 = time.perf_counter()
        print(f"pass {rep}: dct_residual {t1 - t0:.3f} s ({vox / (t1 - t0) / 1e9:.2f} Gvox/s), "
              f"[numpy glue {t2 - t1:.3f}], PCA thresholds {t3 - t2:.3f}, "
              f"Compute_GreedyPCA_area {t4 - t3:.3f} ({vox / (t4 - t3) / 1e9:.2f}), "
              f"Correlation_GLR_test {t5 - t4:.3f} ({vox / (t5 - t4) / 1e9:.2f})")


if __name__ == "__main__":
    main()
