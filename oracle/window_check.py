"""TEST INFRASTRUCTURE (never imported by origin_amd): oracle checks of device results at sizes
the oracle cannot run whole -- BASELINE.json configs 1-3 (3681 x 200^2 / 300^2 / 600^2).

The GLR has finite support (PSF P x P spatially; the spectral convolution runs over the whole
z axis), so the float64 oracle evaluated on a haloed spatial WINDOW of the device's own input
cube is exact wherever the window's cut edges are >= P//2 away; at true field edges the
window's edge IS the field's edge and the zero padding / border normalisation of
``fftconvolve(mode='same')`` (reference lib_origin.py:1037-1041) is the same.  The greedy PCA is
local to an area (lib_origin.py:806-819), so the oracle on one whole area is exact.

Used by tests/test_baseline_sizes.py and by ``bench.py --check`` (the checker beside the
measurement, never the thing measured).
"""
import numpy as np

from . import cpu_ref


# ---------------------------------------------------------------------------------- GLR
def glr_windows(Ny, Nx, out=48, halo=24, which=("corner", "edge", "interior")):
    """Windows (name, y0, y1, x0, x1) of (out + halo) spaxels a side cut at a field corner, in
    the middle of the top edge and in the interior (off the 100-px area grid and off the
    32/64-spaxel tile grids of the kernels)."""
    w = out + halo
    w_y, w_x = min(w, Ny), min(w, Nx)
    spec = {"corner": (0, 0),
            "edge": (0, max(0, min(Nx - w_x, Nx // 2 - w_x // 2 + 7))),
            "interior": (max(0, min(Ny - w_y, Ny // 2 - w_y // 2 + 13)),
                         max(0, min(Nx - w_x, Nx // 3 - w_x // 2 + 5))),
            "far_corner": (Ny - w_y, Nx - w_x)}
    return [(n, spec[n][0], spec[n][0] + w_y, spec[n][1], spec[n][1] + w_x) for n in which]


def _valid(lo, hi, N, c):
    """Part of [lo, hi) whose outputs are exact: cut edges lose c, true field edges nothing."""
    return (lo + (c if lo > 0 else 0)) - lo, (hi - (c if hi < N else 0)) - lo


def check_glr_window(faint, dev_out, mask, psf, profiles, window, pcut=1e-8, pmeansub=True,
                     nthreads=1, tol=1e-4, tol_argmax=1e-4, tol_rms=None, tol_scale_T=None):
    """Oracle GLR on one window of the device's input cube against the device's outputs.

    tol_scale_T : None, or the largest |T| of a window up to which ``tol`` is an absolute bound;
              for a window with a brighter source the bound grows in proportion,
              ``|dT| <= tol * max(1, max_window |T_ref| / tol_scale_T)``.  Used for the bf16
              arithmetic only: its rounding error is relative to the SUMMANDS (inputs rounded to
              8 significant bits: 0.2 % of the brightest line around, also at voxels where a
              mismatched profile makes the sum itself moderate), and SURVEY 8c's 5e-2 was set
              on a field with T in [-3.6, 19.5] (2.5e-3 of its maximum); the synthetic fields
              reach T = 28-45 on their brightest sources.

    faint   : DeviceArray (Nz, Ny, Nx), the cube the device GLR ran on
    dev_out : dict of DeviceArrays correl / correl_min / profile (+ maxmap, minmap or None)
    mask    : DeviceArray uint8 or None (``correl[mask] = 0``, ``profile[mask] = 0``,
              reference steps.py:781,788)
    Returns a dict of errors; ``ok`` says whether every bound holds."""
    name, y0, y1, x0, x1 = window
    Nz, Ny, Nx = faint.shape
    c = psf.shape[-1] // 2
    cube = faint.window(y0, y1, x0, x1).astype(np.float64)
    correl, profile, correl_min = cpu_ref.Correlation_GLR_test(
        cube, psf, None, profiles, nthreads=nthreads, pcut=pcut, pmeansub=pmeansub)
    if mask is not None:
        m = mask.window(y0, y1, x0, x1).astype(bool)
        correl[m] = 0
        profile[m] = 0
    a0, a1 = _valid(y0, y1, Ny, c)
    b0, b1 = _valid(x0, x1, Nx, c)
    sl = (slice(None), slice(a0, a1), slice(b0, b1))
    got = {k: dev_out[k].window(y0, y1, x0, x1)[sl] for k in ("correl", "correl_min", "profile")}
    res = dict(window=name, box=[int(v) for v in (y0 + a0, y0 + a1, x0 + b0, x0 + b1)],
               voxels=int(got["correl"].size))
    tmax = float(max(np.max(np.abs(correl[sl])), np.max(np.abs(correl_min[sl]))))
    tol_w = tol if tol_scale_T is None else tol * max(1.0, tmax / tol_scale_T)
    res["bound"] = float(tol_w)
    if tol_scale_T is None:
        local = None
    else:
        # the scaled bound holds only where a bright line is AROUND: the rounding error of a sum is
        # relative to its summands, i.e. to the brightest |T| within the reach of a profile along
        # z (+- 32 channels) and of the PSF across the field -- elsewhere (faint lines away from
        # bright ones) the absolute tolerance stays
        from scipy import ndimage as ndi
        env = np.maximum(np.abs(correl), np.abs(correl_min))
        env = ndi.maximum_filter(env, size=(65, 2 * c + 1, 2 * c + 1), mode="nearest")
        local = tol * np.maximum(1.0, env[sl] / tol_scale_T)
        res["voxels_at_the_absolute_bound"] = float(np.mean(local <= tol))

    def bound(ref):
        return tol_w if local is None else local

    res["correl"] = float(np.max(np.abs(got["correl"] - correl[sl])))
    res["correl_min"] = float(np.max(np.abs(got["correl_min"] - correl_min[sl])))
    res["correl_rms"] = float(np.sqrt(np.mean((got["correl"] - correl[sl]) ** 2)))
    res["argmax_mismatch"] = float(np.mean(got["profile"] != profile[sl]))
    res["T_range"] = [float(correl_min[sl].min()), float(correl[sl].max())]
    # worst error in units of the window's bound (<= 1 passes)
    res["correl_over_bound"] = float(np.max(np.abs(got["correl"] - correl[sl]) / bound(correl[sl])))
    res["correl_min_over_bound"] = float(np.max(np.abs(got["correl_min"] - correl_min[sl]) /
                                                bound(correl_min[sl])))
    ok = (res["correl_over_bound"] <= 1.0 and res["correl_min_over_bound"] <= 1.0
          and res["argmax_mismatch"] <= tol_argmax)
    if tol_rms is not None:
        ok = ok and res["correl_rms"] <= tol_rms
    for key, ref_map in (("maxmap", correl[sl].max(axis=0)), ("minmap", correl_min[sl].min(axis=0))):
        d = dev_out.get(key)
        if d is not None:
            dm = d.to_host()[y0 + a0:y0 + a1, x0 + b0:x0 + b1]
            res[key] = float(np.max(np.abs(dm - ref_map)))
            # (a map entry is one of the column's voxels: the column's loosest bound applies)
            bmap = tol_w if local is None else local.max(axis=0)
            ok = ok and bool(np.all(np.abs(dm - ref_map) <= bmap))
    # compute_local_max (lib_origin.py:1220-1256) of the DEVICE's correl / correl_min on this window
    # against the device's local maxima: index work on identical float32 inputs, bit exact.  A
    # 3x3x3 window needs its neighbours: one more spaxel is dropped at the window's cut edges.
    if dev_out.get("local_max") is not None and dev_out.get("local_min") is not None:
        dc = dev_out["correl"].window(y0, y1, x0, x1).astype(np.float64)
        dm = dev_out["correl_min"].window(y0, y1, x0, x1).astype(np.float64)
        mk = mask.window(y0, y1, x0, x1).astype(bool) if mask is not None else \
            np.zeros(dc.shape, bool)
        rmax, rmin = cpu_ref.compute_local_max(dc, dm, mk, 3)
        i0, i1 = (1 if y0 > 0 else 0), dc.shape[1] - (1 if y1 < Ny else 0)
        j0, j1 = (1 if x0 > 0 else 0), dc.shape[2] - (1 if x1 < Nx else 0)
        inner = (slice(None), slice(i0, i1), slice(j0, j1))
        gmax = dev_out["local_max"].window(y0, y1, x0, x1)[inner]
        gmin = dev_out["local_min"].window(y0, y1, x0, x1)[inner]
        res["local_max_mismatch"] = int(np.count_nonzero(gmax != rmax[inner]) +
                                        np.count_nonzero(gmin != rmin[inner]))
        res["local_maxima"] = int(np.count_nonzero(gmax))
        ok = ok and res["local_max_mismatch"] == 0 and res["local_maxima"] > 0
    res["ok"] = bool(ok)
    return res


# ---------------------------------------------------------------------------------- PCA
def check_pca_area(cube_std, cube_faint, mapO2, spx, thresO2, area, Noise_population=50,
                   itermax=100, tol_fro=2e-6, tol_abs=1e-4):
    """Oracle ``Compute_GreedyPCA`` (reference lib_origin.py:848-954) on one whole area of the
    device's cube_std against the device's cube_faint / mapO2 there.  ``spx``: flat spaxel
    indices of the area in the column order of ``cube[:, areamap == i]``."""
    Nz, Ny, Nx = cube_std.shape
    ys, xs = np.unravel_index(spx, (Ny, Nx))
    y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
    box = cube_std.window(y0, y1, x0, x1)
    X = box[:, ys - y0, xs - x0].astype(np.float64)
    del box
    test = cpu_ref.O2test(X)
    trace = []
    faint, m, nstop = cpu_ref.Compute_GreedyPCA(X, test, float(thresO2), Noise_population,
                                                itermax, trace=trace)
    got = cube_faint.window(y0, y1, x0, x1)[:, ys - y0, xs - x0]
    d = got - faint
    res = dict(area=int(area), spaxels=int(len(spx)), iterations=len(trace),
               n_nuisance_first=int(trace[0][0]) if trace else 0, nstop=int(nstop),
               rel_fro=float(np.linalg.norm(d) / max(np.linalg.norm(faint), 1e-300)),
               max_abs=float(np.max(np.abs(d))),
               mapO2_mismatch=int(np.count_nonzero(np.asarray(mapO2).reshape(-1)[spx] != m)))
    res["ok"] = bool(res["rel_fro"] <= tol_fro and res["max_abs"] <= tol_abs and
                     res["mapO2_mismatch"] == 0)
    return res


# ---------------------------------------------------------------------------------- DCT
def check_dct_window(raw, var, mask, cube_std, cont_dct, window, order=10, approx=False,
                     tol=1e-5):
    """Oracle ``dct_residual`` (reference lib_origin.py:150-240) on a small window against the
    device's cont_dct (= continuum / sqrt(var), steps.py:440,463), and cube_std up to the
    per-channel mean of the whole field (steps.py:442), which a window cannot know: the
    quantity ``(raw - cont) - cube_std * std`` must be the SAME number for every unmasked
    spaxel of a channel (it is that mean); its spread over the window is reported and bounded,
    and its value returned so that a caller holding the whole field can compare it."""
    name, y0, y1, x0, x1 = window
    r = raw.window(y0, y1, x0, x1).astype(np.float64)
    v = var.window(y0, y1, x0, x1).astype(np.float64)
    m = mask.window(y0, y1, x0, x1).astype(bool)
    cont = cpu_ref.dct_residual(r, order, v, approx, m)
    std = np.sqrt(v)
    want = cont / std
    got = cont_dct.window(y0, y1, x0, x1).astype(np.float64)
    e_cont = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    cs = cube_std.window(y0, y1, x0, x1).astype(np.float64)
    resid = np.where(m, np.nan, (r - cont) - cs * std)
    zmean = np.nanmean(resid, axis=(1, 2))
    spread = np.nanmax(np.abs(resid - zmean[:, None, None]) / (std * np.maximum(1.0, np.abs(cs))))
    res = dict(window=name, cont_dct=float(e_cont.max()), cube_std_spread=float(spread),
               masked_zero=bool(np.all(cs[m] == 0)))
    res["ok"] = bool(res["cont_dct"] <= tol and res["cube_std_spread"] <= 2 * tol and
                     res["masked_zero"])
    res["_zmean"] = zmean
    return res
