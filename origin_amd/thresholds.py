"""Host-side threshold estimation of the O2 test.

``compute_thresh_gaussfit`` works on one O2 value per spaxel of an area (~1e4 numbers):
sigma-clip, Freedman-Diaconis histogram, Levenberg-Marquardt Gaussian fit of the left
half.  It is scalar work on a tiny vector and stays on the host by design; the dense part
(the O2 values themselves) comes from the GPU.  Follows the reference
muse_origin/lib_origin.py:977-1024, with astropy's ``sigma_clip`` / ``LevMarLSQFitter`` /
``Gaussian1D`` restated on NumPy/SciPy (astropy is not needed at run time).
"""
import ctypes as C

import numpy as np
from scipy.special import ndtri

_SIGMA_TO_FWHM = 2.0 * np.sqrt(2.0 * np.log(2.0))


def sigma_clip_compressed(data, sigma, maxiters=5):
    """``astropy.stats.sigma_clip(data, sigma).compressed()`` with astropy's defaults
    (median centre, std spread, 5 iterations), lib_origin.py:1000-1001."""
    d = np.asarray(data, dtype=float).ravel()
    kept = d[np.isfinite(d)]
    lo, hi = -np.inf, np.inf
    changed, it = 1, 0
    while changed and it < maxiters:
        it += 1
        n = kept.size
        cen, sd = np.median(kept), np.std(kept)
        lo, hi = cen - sd * sigma, cen + sd * sigma
        kept = kept[(kept >= lo) & (kept <= hi)]
        changed = n - kept.size
    return d[(d >= lo) & (d <= hi)]


def fit_gauss1d(x, y, amplitude, mean, stddev):
    """Levenberg-Marquardt fit of amplitude*exp(-(x-mean)^2/(2 stddev^2)) with the analytic
    Jacobian (what LevMarLSQFitter does with Gaussian1D.fit_deriv, lib_origin.py:1014-1018):
    MINPACK's lmder algorithm as native host code (csrc/lmfit.hip), no Python callbacks."""
    from . import _capi
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    p = np.array([amplitude, mean, stddev], dtype=np.float64)
    info, nfev = C.c_int(0), C.c_int(0)
    _capi.call("origin_gauss_fit", x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
               x.size, p.ctypes.data_as(C.c_void_p), C.byref(info), C.byref(nfev))
    return p[0], p[1], max(p[2], float(np.finfo(np.float32).tiny))


def fit_gauss1d_scipy(x, y, amplitude, mean, stddev):
    """The same fit through scipy.optimize.leastsq (the reference's own route); kept as the
    cross-check of the native solver in tests/test_host_logic.py."""
    from scipy import optimize
    tiny = float(np.finfo(np.float32).tiny)

    def unpack(p):
        return p[0], p[1], max(p[2], tiny)

    def resid(p):
        a, m, s = unpack(p)
        return a * np.exp(-0.5 * (x - m) ** 2 / s ** 2) - y

    def jac(p):
        a, m, s = unpack(p)
        g = np.exp(-0.5 / s ** 2 * (x - m) ** 2)
        return [g, a * g * (x - m) / s ** 2, a * g * (x - m) ** 2 / s ** 3]

    p, _ = optimize.leastsq(resid, [amplitude, mean, stddev], Dfun=jac, col_deriv=True,
                            maxfev=100, epsfcn=np.sqrt(np.finfo(float).eps), xtol=1e-7)
    return unpack(p)


def clipped_histogram(data, bins='fd', sigclip=10):
    """data[data > 0] -> sigma clip -> density histogram (lib_origin.py:999-1002).  For the
    default 'fd' bins this runs as native host code in liborigin_hip.so (bit-identical to
    the NumPy path below, tests/test_host_logic.py); other estimators use NumPy."""
    data = np.ascontiguousarray(data, dtype=np.float64).ravel()
    if bins == 'fd' and data.size:
        from . import _capi
        cap = max(4096, data.size)
        hist = np.empty(cap)
        edges = np.empty(cap + 1)
        nb, nk = C.c_long(0), C.c_long(0)
        _capi.call("origin_o2_histogram", data.ctypes.data_as(C.c_void_p), data.size,
                   float(sigclip), 5, hist.ctypes.data_as(C.c_void_p),
                   edges.ctypes.data_as(C.c_void_p), cap, C.byref(nb), C.byref(nk))
        return hist[:nb.value].copy(), edges[:nb.value + 1].copy()
    return clipped_histogram_numpy(data, bins, sigclip)


def clipped_histograms(tests, sigclip=10, _cat=None):
    """``clipped_histogram`` for a list of per-area O2 vectors in one native call that spreads
    the areas over host threads.  ``_cat``: (data, offsets) when the vectors already sit one
    after the other in one float64 array (then ``tests`` is only counted)."""
    from . import _capi
    if _cat is not None:
        data, off = _cat
        lens = np.diff(off)
    else:
        lens = np.array([len(t) for t in tests], dtype=np.int64)
        off = np.zeros(len(tests) + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        data = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.float64).ravel()
                                                    for t in tests]))
    cap = int(max(4096, lens.max()))
    hist = np.empty((len(tests), cap + 1))
    edges = np.empty((len(tests), cap + 1))
    nb = np.zeros(len(tests), dtype=np.int64)
    _capi.call("origin_o2_histogram_batch", data.ctypes.data_as(C.c_void_p),
               off.ctypes.data_as(C.c_void_p), len(tests), float(sigclip), 5,
               hist.ctypes.data_as(C.c_void_p), edges.ctypes.data_as(C.c_void_p), cap,
               nb.ctypes.data_as(C.c_void_p))
    return [(hist[a, :nb[a]].copy(), edges[a, :nb[a] + 1].copy()) for a in range(len(tests))]


def thresholds_batch(tests, pfa, sigclip=10, _cat=None):
    """``compute_thresh_gaussfit`` for a list of per-area O2 vectors, everything (clip,
    histogram, fit, threshold) in two native calls on the host worker pool.  Returns one
    (histO2, frecO2, thresO2, mea, std) per area.  Raises ValueError where the reference would
    (histogram maximum in the first bin: ``argmin`` of an empty slice, lib_origin.py:1006)."""
    from . import _capi
    na = len(tests)
    if _cat is not None:
        data, off = _cat
        lens = np.diff(off)
    else:
        lens = np.array([len(t) for t in tests], dtype=np.int64)
        off = np.zeros(na + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        data = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.float64).ravel()
                                                    for t in tests]))
    cap = int(max(4096, lens.max()))
    hist = np.empty((na, cap + 1))
    edges = np.empty((na, cap + 1))
    nb = np.zeros(na, dtype=np.int64)
    _capi.call("origin_o2_histogram_batch", data.ctypes.data_as(C.c_void_p),
               off.ctypes.data_as(C.c_void_p), na, float(sigclip), 5,
               hist.ctypes.data_as(C.c_void_p), edges.ctypes.data_as(C.c_void_p), cap,
               nb.ctypes.data_as(C.c_void_p))
    res = np.empty((na, 3))
    status = np.zeros(na, dtype=np.int32)
    _capi.call("origin_o2_threshold_batch", hist.ctypes.data_as(C.c_void_p),
               edges.ctypes.data_as(C.c_void_p), nb.ctypes.data_as(C.c_void_p), na, cap,
               float(ndtri(pfa)), res.ctypes.data_as(C.c_void_p),
               status.ctypes.data_as(C.c_void_p))
    if np.any(status == 1):
        raise ValueError("attempt to get argmin of an empty sequence")   # np.argmin's message
    if np.any(status == 2):
        raise ValueError("fewer than three histogram bins left of the mode: no Gaussian fit")
    return [(hist[a, :nb[a]].copy(), edges[a, :nb[a] + 1].copy(), float(res[a, 0]),
             res[a, 1], res[a, 2]) for a in range(na)]


def areas_fit(o2_flat, idx, off, pfa, sigclip=10):
    """``thresholds_batch`` straight from the O2 map: ``idx`` (int32) are the concatenated flat
    spaxel indices of the areas, area a at ``idx[off[a]:off[a+1]]``.  Gather, clip, histogram,
    fit and threshold of every area run in ONE dispatch of the native worker pool
    (``origin_o2_areas_fit``).  Returns (tests, fits): the per-area O2 vectors (views into one
    buffer, the reference's ``testO2``) and one (histO2, frecO2, thresO2, mea, std) per area."""
    from . import _capi
    o2_flat = np.ascontiguousarray(o2_flat, dtype=np.float64).reshape(-1)
    na = len(off) - 1
    lens = np.diff(off)
    cap = int(max(4096, lens.max() if na else 0))
    data = np.empty(int(off[-1]))
    hist = np.empty((na, cap + 1))
    edges = np.empty((na, cap + 1))
    nb = np.zeros(na, dtype=np.int64)
    res = np.empty((na, 3))
    status = np.zeros(na, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _capi.call("origin_o2_areas_fit", p(o2_flat), p(idx), p(off), na, float(sigclip), 5,
               float(ndtri(pfa)), p(data), p(hist), p(edges), cap, p(nb), p(res), p(status))
    if np.any(status == 3):
        raise ValueError("origin_o2_histogram failed for at least one area (empty or too many bins)")
    if np.any(status == 1):
        raise ValueError("attempt to get argmin of an empty sequence")   # np.argmin's message
    if np.any(status == 2):
        raise ValueError("fewer than three histogram bins left of the mode: no Gaussian fit")
    tests = [data[off[a]:off[a + 1]] for a in range(na)]
    fits = [(hist[a, :nb[a]].copy(), edges[a, :nb[a] + 1].copy(), float(res[a, 0]),
             res[a, 1], res[a, 2]) for a in range(na)]
    return tests, fits


def clipped_histogram_numpy(data, bins='fd', sigclip=10):
    data = np.asarray(data, dtype=float)
    data = data[data > 0]
    data = sigma_clip_compressed(data, sigclip)
    return np.histogram(data, bins=bins, density=True)


def compute_thresh_gaussfit(data, pfa, bins='fd', sigclip=10, _hist=None, _fit=None):
    """Same signature and return tuple as the reference (lib_origin.py:977-1024):
    histO2, frecO2, thresO2 (python float), mea, std."""
    histO2, frecO2 = _hist if _hist is not None else clipped_histogram(data, bins, sigclip)
    ind = np.argmax(histO2)
    mod = frecO2[ind]
    ind2 = np.argmin((histO2[ind] / 2 - histO2[:ind]) ** 2)
    fwhm = mod - frecO2[ind2]
    sigma = fwhm / np.sqrt(2 * np.log(2))
    coef = ndtri(pfa)  # == scipy.stats.norm.ppf(pfa) without the distribution machinery
    x = (frecO2[1:] + frecO2[:-1]) / 2
    xcut = mod + _SIGMA_TO_FWHM * sigma / 2
    ksel = x < xcut
    _, mea, std = (_fit or fit_gauss1d)(x[ksel], histO2[ksel], histO2.max(), mod, sigma)
    thresO2 = float(mea - std * coef)
    return histO2, frecO2, thresO2, mea, std
