"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the ORIGIN hot path.

A NumPy/SciPy restatement of the reference's algorithm for the path named by
BASELINE.json (DCT continuum fit -> standardise -> O2 / PCA threshold ->
greedy PCA -> GLR correlation -> maxmap, plus compute_local_max).  It keeps
the reference's *call structure* (per-spaxel ``multi_dot`` loop, ``svds`` per
PCA iteration, ``fftconvolve`` per channel, ``rfftn/irfftn`` per profile) so
that its wall time is a faithful proxy for the reference on the same host.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``origin_amd``) never does; it fails
loudly when the HIP library is missing.

PINNING: every function below is checked against outputs of the reference's own
``muse_origin/lib_origin.py`` (imported unmodified through ``oracle/ref_import.py``
in the build container) by ``oracle/gen_golden.py``; the resulting vectors are
committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.

All ``file:line`` citations are relative to ``/root/reference/muse_origin/``.
"""
import os

import numpy as np
from numpy import fft
from numpy.linalg import inv, multi_dot
from scipy import optimize, stats
from scipy.ndimage import maximum_filter
from scipy.signal import fftconvolve
from scipy.sparse.linalg import svds

try:  # the reference's only parallel knob is joblib (lib_origin.py:1130,1185,1204)
    from joblib import Parallel, delayed
except ImportError:  # pragma: no cover
    Parallel = None


# --------------------------------------------------------------------------
# DCT continuum  (lib_origin.py:127-240)
# --------------------------------------------------------------------------
def DCTMAT(nl, order):
    """DCT-II orthonormal atoms, shape (nl, order+1).  lib_origin.py:127-146."""
    yy, xx = np.mgrid[:nl, : order + 1]
    D0 = np.sqrt(2 / nl) * np.cos((yy + 0.5) * (np.pi / nl) * xx)
    D0[:, 0] *= 1 / np.sqrt(2)
    return D0


def dct_residual(w_raw, order, var, approx, mask):
    """Continuum estimated per spaxel from `order`+1 DCT atoms.

    lib_origin.py:150-240.  approx -> D0 D0^T s (:191-194); else for a spaxel
    without any masked voxel (:226) the weighted LSQ D0 (D0^T S^-1 D0)^-1 D0^T S^-1 s
    (:233-235), and D0 D0^T s for the others (:237).  Returns the continuum cube.
    """
    nl = w_raw.shape[0]
    D0 = DCTMAT(nl, order)
    shape = w_raw.shape[1:]
    if approx:
        cont = [multi_dot([D0, D0.T, w_raw[:, y, x]]) for y, x in np.ndindex(shape)]
    else:
        w_raw_var = w_raw / var
        D0T = D0.T
        valid = ~np.any(mask, axis=0)
        cont = []
        for y, x in np.ndindex(shape):
            if valid[y, x]:
                res = multi_dot(
                    [D0, inv(np.dot(D0T / var[:, y, x], D0)), D0T, w_raw_var[:, y, x]]
                )
            else:
                res = multi_dot([D0, D0.T, w_raw[:, y, x]])
            cont.append(res)
    return np.stack(cont).T.reshape(w_raw.shape)


def preprocessing(cube_raw, var, mask, dct_order=10, dct_approx=False):
    """Dense part of ``Preprocessing.run`` (steps.py:431-450, :463-465).

    Returns dict(cube_std, ima_std, cont_dct (float32), ima_dct).
    """
    cont_dct = dct_residual(cube_raw, dct_order, var, dct_approx, mask)
    data = cube_raw - cont_dct  # steps.py:434
    data[mask] = np.nan  # :435
    std = np.sqrt(var)  # :439
    cont_dct /= std  # :440
    with np.errstate(invalid="ignore"):
        mean = np.nanmean(data, axis=(1, 2))  # :442
    data -= mean[:, np.newaxis, np.newaxis]  # :444
    data /= std  # :445
    data[mask] = 0  # :446
    cont32 = cont_dct.astype(np.float32)  # :463
    return dict(
        cube_std=data,
        ima_std=data.mean(axis=0),  # :450
        cont_dct=cont32,
        ima_dct=cont32.mean(axis=0),  # :465
    )


# --------------------------------------------------------------------------
# O2 test and threshold  (lib_origin.py:824-845, 957-1024)
# --------------------------------------------------------------------------
def O2test(arr):
    """mean(arr**2, axis=0).  lib_origin.py:957-974."""
    return np.mean(arr ** 2, axis=0)


def _sigma_clip_compressed(data, sigma, maxiters=5):
    """astropy.stats.sigma_clip(data, sigma).compressed() with astropy's defaults
    (cenfunc=median, stdfunc=std, maxiters=5), as called at lib_origin.py:1000-1001."""
    filtered = np.asarray(data).ravel()
    filtered = filtered[np.isfinite(filtered)]
    lo, hi = -np.inf, np.inf
    nchanged, it = 1, 0
    while nchanged != 0 and it < maxiters:
        it += 1
        size = filtered.size
        cen = np.median(filtered)
        sd = np.std(filtered)
        lo, hi = cen - sd * sigma, cen + sd * sigma
        filtered = filtered[(filtered >= lo) & (filtered <= hi)]
        nchanged = size - filtered.size
    d = np.asarray(data).ravel()
    return d[(d >= lo) & (d <= hi)]


def _gauss1d_lm_fit(x, y, amplitude, mean, stddev):
    """astropy ``LevMarLSQFitter()(Gaussian1D(...), x, y)`` (lib_origin.py:1014-1018):
    scipy.optimize.leastsq with the analytic Gaussian Jacobian, xtol=1e-7,
    maxfev=100, col_deriv, stddev bounded below by float eps."""
    tiny = float(np.finfo(np.float32).tiny)

    def clip(p):
        a, m, s = p
        return a, m, max(s, tiny)

    def resid(p):
        a, m, s = clip(p)
        return a * np.exp(-0.5 * (x - m) ** 2 / s ** 2) - y

    def jac(p):
        a, m, s = clip(p)
        d_a = np.exp(-0.5 / s ** 2 * (x - m) ** 2)
        d_m = a * d_a * (x - m) / s ** 2
        d_s = a * d_a * (x - m) ** 2 / s ** 3
        return [d_a, d_m, d_s]

    p, _ = optimize.leastsq(
        resid,
        [amplitude, mean, stddev],
        Dfun=jac,
        col_deriv=True,
        maxfev=100,
        epsfcn=np.sqrt(np.finfo(float).eps),
        xtol=1e-7,
    )
    a, m, s = clip(p)
    return a, m, s


def compute_thresh_gaussfit(data, pfa, bins="fd", sigclip=10):
    """lib_origin.py:977-1024.  Returns histO2, frecO2, thresO2, mea, std."""
    data = data[data > 0]
    data = _sigma_clip_compressed(data, sigclip)
    histO2, frecO2 = np.histogram(data, bins=bins, density=True)
    ind = np.argmax(histO2)
    mod = frecO2[ind]
    ind2 = np.argmin((histO2[ind] / 2 - histO2[:ind]) ** 2)
    fwhm = mod - frecO2[ind2]
    sigma = fwhm / np.sqrt(2 * np.log(2))

    coef = stats.norm.ppf(pfa)
    x = (frecO2[1:] + frecO2[:-1]) / 2
    sigma_to_fwhm = 2.0 * np.sqrt(2.0 * np.log(2.0))
    xcut = mod + sigma_to_fwhm * sigma / 2
    ksel = x < xcut
    _, mea, std = _gauss1d_lm_fit(x[ksel], histO2[ksel], histO2.max(), mod, sigma)
    thresO2 = float(mea - std * coef)
    return histO2, frecO2, thresO2, mea, std


def Compute_PCA_threshold(faint, pfa):
    """lib_origin.py:824-845."""
    test = O2test(faint)
    histO2, frecO2, thresO2, mea, std = compute_thresh_gaussfit(test, pfa)
    return test, histO2, frecO2, thresO2, mea, std


# --------------------------------------------------------------------------
# Greedy PCA  (lib_origin.py:76-88, 769-954)
# --------------------------------------------------------------------------
def orthogonal_projection(a, b):
    """a a^T b, *without* (a^T a)^-1.  lib_origin.py:76-88."""
    if a.ndim == 1:
        a = a[:, None]
    return multi_dot([a, a.T, b])


def Compute_GreedyPCA(cube_in, test, thresO2, Noise_population, itermax, svd="svds",
                      trace=None):
    """lib_origin.py:848-954.  Returns faint, mapO2, nstop.

    svd='svds' follows the reference call (ARPACK, tol=0); svd='dense' uses a full
    LAPACK SVD (same vector to ~1e-14, SURVEY section 7 hard part 1).  `trace`, if a
    list, receives (n_nuisance, nb_background) per iteration.
    """
    pypx = np.where(test > thresO2)[0]  # :889
    faint = cube_in.copy()  # :892
    mapO2 = np.zeros(faint.shape[1])
    nstop = 0
    nbiter = 0
    while len(pypx) > 0:  # :899
        nbiter += 1
        mapO2[pypx] += 1
        if nbiter > itermax:  # :902
            nstop += 1
            break
        test_v = np.ravel(test)  # :908
        test_v = test_v[test_v > 0]
        nind = np.where(test_v <= thresO2)[0]
        sortind = np.argsort(test_v[nind])
        nb = 1 + int(len(nind) / Noise_population)  # :914
        b = np.mean(faint[:, nind[sortind[:nb]]], axis=1)  # :917
        x_red = faint[:, pypx]  # :920
        x_red -= orthogonal_projection(b, x_red)  # :923
        x_red /= np.nansum(b ** 2)  # :924
        if trace is not None:
            trace.append((len(pypx), nb))
        if x_red.shape[1] == 1:  # :927
            break
        if svd == "svds":
            U, s, V = svds(x_red, k=1)  # :940
        else:
            U, s, V = np.linalg.svd(x_red, full_matrices=False)
        faint -= orthogonal_projection(U[:, 0], faint)  # :943
        test = O2test(faint)  # :946
        pypx = np.where(test > thresO2)[0]  # :949
    return faint, mapO2, nstop


def Compute_GreedyPCA_area(NbArea, cube_std, areamap, Noise_population, threshold_test,
                           itermax, testO2, svd="svds"):
    """lib_origin.py:769-821."""
    cube_faint = cube_std.copy()
    mapO2 = np.zeros(cube_std.shape[1:])
    nstop = 0
    for area_ind in range(1, NbArea + 1):
        ksel = areamap == area_ind
        cube_temp = cube_std[:, ksel]
        thr = threshold_test[area_ind - 1]
        test = testO2[area_ind - 1]
        cube_faint[:, ksel], mO2, kstop = Compute_GreedyPCA(
            cube_temp, test, thr, Noise_population, itermax, svd=svd
        )
        mapO2[ksel] = mO2
        nstop += kstop
    return cube_faint, mapO2, nstop


def pca_threshold_areas(cube_std, areamap, nbAreas, pfa_test=0.01):
    """``ComputePCAThreshold.run`` body (steps.py:610-631)."""
    results = []
    for area_ind in range(1, nbAreas + 1):
        ksel = areamap == area_ind
        results.append(Compute_PCA_threshold(cube_std[:, ksel], pfa_test))
    testO2, histO2, binO2, thresO2, meaO2, stdO2 = zip(*results)
    return testO2, histO2, binO2, thresO2, meaO2, stdO2


# --------------------------------------------------------------------------
# GLR correlation  (lib_origin.py:1027-1217)
# --------------------------------------------------------------------------
def _convolve_fsf(psf, cube, weights=None):
    """lib_origin.py:1027-1043 (one channel)."""
    ones = np.ones_like(cube)
    if weights is not None:
        cube = cube * weights
        ones *= weights
    psf = np.ascontiguousarray(psf[::-1, ::-1])
    psf -= psf.mean()
    cube_fsf = fftconvolve(cube, psf, mode="same")
    psf **= 2
    norm_fsf = fftconvolve(ones, psf, mode="same")
    return cube_fsf, norm_fsf


def next_fast_len(target):
    """Smallest 5-smooth integer >= target (what scipy.fftpack.helper.next_fast_len
    returns at lib_origin.py:1174)."""
    if target <= 6:
        return target
    best = None
    p5 = 1
    while p5 < 2 * target:
        p35 = p5
        while p35 < 2 * target:
            n = p35
            while n < target:
                n *= 2
            if best is None or n < best:
                best = n
            p35 *= 3
        p5 *= 5
    return best


def _convolve_spectral(parallel, nslices, arr, shape, func):
    """lib_origin.py:1063-1066."""
    arr = np.array_split(arr, nslices, axis=-1)
    if parallel is None:
        out = [func(chunk, shape, axes=(0,)) for chunk in arr]
    else:
        out = parallel(delayed(func)(chunk, shape, axes=(0,)) for chunk in arr)
    return np.concatenate(out, axis=-1)


def _convolve_profile(Dico, cube_fft, norm_fft, fshape, n_jobs, parallel):
    """lib_origin.py:1046-1060."""
    dico_fft = fft.rfftn(Dico, fshape, axes=(0,))[:, None] * cube_fft
    cube_profile = _convolve_spectral(parallel, n_jobs, dico_fft, fshape, func=fft.irfftn)
    dico_fft = fft.rfftn(Dico ** 2, fshape, axes=(0,))[:, None] * norm_fft
    norm_profile = _convolve_spectral(parallel, n_jobs, dico_fft, fshape, func=fft.irfftn)
    norm_profile[norm_profile <= 0] = np.inf
    np.sqrt(norm_profile, out=norm_profile)
    cube_profile /= norm_profile
    return cube_profile


def prepare_profiles(profiles, pcut=None, pmeansub=True):
    """Trim / normalise / mean-subtract the dictionary.  lib_origin.py:1155-1165."""
    prof_cut = []
    for prof in profiles:
        prof = np.array(prof, dtype=float)
        if pcut is not None:
            lpeak = prof.argmax()
            lw = np.max(np.abs(np.where(prof >= pcut)[0][[0, -1]] - lpeak))
            prof = prof[lpeak - lw: lpeak + lw + 1]
        prof /= np.linalg.norm(prof)
        if pmeansub:
            prof -= prof.mean()
        prof_cut.append(prof)
    return prof_cut


def Correlation_GLR_test(cube, fsf, weights, profiles, nthreads=1, pcut=None, pmeansub=True):
    """lib_origin.py:1070-1217.  Returns correl, profile (uint8), correl_min."""
    Nz, Ny, Nx = cube.shape
    if weights is None:
        fsf = [fsf]
        weights = [None]
    nfields = len(fsf)
    cube = np.array(cube).astype(float)

    use_joblib = Parallel is not None and nthreads != 1
    for nf in range(nfields):
        if use_joblib:
            with Parallel(n_jobs=nthreads) as parallel:
                res = parallel(
                    delayed(_convolve_fsf)(fsf[nf][i], cube[i], weights=weights[nf])
                    for i in range(Nz)
                )
        else:
            res = [_convolve_fsf(fsf[nf][i], cube[i], weights=weights[nf]) for i in range(Nz)]
        res = [np.stack(arr) for arr in zip(*res)]
        if nf == 0:
            cube_fsf, norm_fsf = res
        else:
            cube_fsf += res[0]
            norm_fsf += res[1]

    prof_cut = prepare_profiles(profiles, pcut, pmeansub)

    s1 = np.array(cube_fsf.shape)
    s2 = np.array([(d.shape[0], 1, 1) for d in prof_cut])
    fftshape = s1 + s2 - 1
    fshape = [next_fast_len(int(d)) for d in fftshape.max(axis=0)[:1]]
    startind = (fftshape - s1) // 2
    endind = startind + s1
    cslice = [slice(startind[k, 0], endind[k, 0]) for k in range(len(endind))]

    def run(parallel):
        cube_fft = _convolve_spectral(parallel, nthreads, cube_fsf, fshape, func=fft.rfftn)
        norm_fft = _convolve_spectral(parallel, nthreads, norm_fsf, fshape, func=fft.rfftn)
        cube_fft = cube_fft.reshape(cube_fft.shape[0], -1)
        norm_fft = norm_fft.reshape(norm_fft.shape[0], -1)
        profile = np.zeros((Nz, Ny * Nx), dtype=np.uint8)  # reference: np.empty (:1197)
        correl = np.full((Nz, Ny * Nx), -np.inf)
        correl_min = np.full((Nz, Ny * Nx), np.inf)
        for k in range(len(prof_cut)):
            cube_profile = _convolve_profile(
                prof_cut[k], cube_fft, norm_fft, fshape, nthreads, parallel
            )
            cube_profile = cube_profile[cslice[k]]
            profile[cube_profile > correl] = k  # strict '>' : first maximum wins (:1210)
            np.maximum(correl, cube_profile, out=correl)
            np.minimum(correl_min, cube_profile, out=correl_min)
        return correl, profile, correl_min

    if use_joblib:
        with Parallel(n_jobs=nthreads, backend="threading") as parallel:
            correl, profile, correl_min = run(parallel)
    else:
        correl, profile, correl_min = run(None)
    return (
        correl.reshape(Nz, Ny, Nx),
        profile.reshape(Nz, Ny, Nx),
        correl_min.reshape(Nz, Ny, Nx),
    )


def Correlation_GLR_test_direct(cube, fsf, weights, profiles, pcut=None, pmeansub=True):
    """Direct (non-FFT) evaluation of the GLR algebra of SURVEY.md section 8(a):

        cube_fsf[z,y,x] = sum_f sum_{dy,dx} k_fz[dy,dx] (w_f cube)[z, y+dy-c, x+dx-c]
        norm_fsf[z,y,x] = sum_f sum_{dy,dx} k_fz[dy,dx]^2 w_f[y+dy-c, x+dx-c]
        num_k[z] = sum_j p_k[j] cube_fsf[z + lw_k - j]
        den_k[z] = sum_j p_k[j]^2 norm_fsf[z + lw_k - j]
        T_k = num_k / sqrt(den_k)   (den_k <= 0 -> 0)

    with k = PSF - mean(PSF), c = P//2, zero outside the cube.  Small cubes only
    (pure NumPy shifts); this is the formulation the HIP kernels implement.
    """
    Nz, Ny, Nx = cube.shape
    if weights is None:
        fsf = [fsf]
        weights = [None]
    cube = np.asarray(cube, dtype=float)
    cube_fsf = np.zeros((Nz, Ny, Nx))
    norm_fsf = np.zeros((Nz, Ny, Nx))
    for f, w in zip(fsf, weights):
        f = np.asarray(f, dtype=float)
        P = f.shape[1]
        c = P // 2
        k = f - f.mean(axis=(1, 2), keepdims=True)
        wmap = np.ones((Ny, Nx)) if w is None else np.asarray(w, dtype=float)
        src = np.zeros((Nz, Ny + 2 * c, Nx + 2 * c))
        src[:, c: c + Ny, c: c + Nx] = cube * wmap
        wsrc = np.zeros((Ny + 2 * c, Nx + 2 * c))
        wsrc[c: c + Ny, c: c + Nx] = wmap
        for dy in range(P):
            for dx in range(P):
                cube_fsf += k[:, dy, dx][:, None, None] * src[:, dy: dy + Ny, dx: dx + Nx]
                norm_fsf += (k[:, dy, dx] ** 2)[:, None, None] * wsrc[dy: dy + Ny, dx: dx + Nx]
    prof_cut = prepare_profiles(profiles, pcut, pmeansub)
    correl = np.full((Nz, Ny, Nx), -np.inf)
    correl_min = np.full((Nz, Ny, Nx), np.inf)
    profile = np.zeros((Nz, Ny, Nx), dtype=np.uint8)
    for kk, p in enumerate(prof_cut):
        L = len(p)
        lw = (L - 1) // 2
        # 'same'-centred true convolution: out[z] = sum_j p[j] in[z + lw - j]
        num = np.zeros((Nz, Ny, Nx))
        den = np.zeros((Nz, Ny, Nx))
        for j in range(L):
            sh = lw - j  # reads in[z + sh]
            z0, z1 = max(0, -sh), min(Nz, Nz - sh)
            if z1 > z0:
                num[z0:z1] += p[j] * cube_fsf[z0 + sh: z1 + sh]
                den[z0:z1] += p[j] ** 2 * norm_fsf[z0 + sh: z1 + sh]
        with np.errstate(divide="ignore", invalid="ignore"):
            T = np.where(den > 0, num / np.sqrt(np.where(den > 0, den, 1.0)), 0.0)
        profile[T > correl] = kk
        np.maximum(correl, T, out=correl)
        np.minimum(correl_min, T, out=correl_min)
    return correl, profile, correl_min


def compute_TGLR(cube_faint, PSF, wfields, profiles, mask, ncpu=1, pcut=1e-8, pmeansub=True,
                 size=3):
    """Dense part of ``ComputeTGLR.run`` (steps.py:770-802)."""
    correl, profile, correl_min = Correlation_GLR_test(
        cube_faint, PSF, wfields, profiles, nthreads=ncpu, pcut=pcut, pmeansub=pmeansub
    )
    correl[mask] = 0  # :781
    profile[mask] = 0  # :788
    maxmap = np.amax(correl, axis=0)  # :792
    minmap = np.amin(correl_min, axis=0)  # :793
    local_max, local_min = compute_local_max(correl, correl_min, mask, size)  # :796
    return dict(cube_correl=correl, cube_profile=profile, cube_correl_min=correl_min,
                maxmap=maxmap, minmap=minmap, cube_local_max=local_max,
                cube_local_min=local_min)


# --------------------------------------------------------------------------
# Local maxima  (lib_origin.py:1220-1256)
# --------------------------------------------------------------------------
def compute_local_max(correl, correl_min, mask, size=3):
    """lib_origin.py:1220-1256."""
    if np.isscalar(size):
        size = (size, size, size)
    local_max = maximum_filter(correl, size=size)
    local_mask = correl == local_max
    local_mask[mask] = False
    local_max *= local_mask
    minus_correl_min = -correl_min
    local_min = maximum_filter(minus_correl_min, size=size)
    local_mask = minus_correl_min == local_min
    local_mask[mask] = False
    local_min *= local_mask
    return local_max, local_min


# --------------------------------------------------------------------------
# Purity threshold  (lib_origin.py:1391-1479)
# --------------------------------------------------------------------------
def Compute_threshold_purity(purity, cube_local_max, cube_local_min, segmap=None, threshlist=None):
    """lib_origin.py:1391-1479 without the astropy Table: returns (threshold, columns) where
    columns = dict(Tval_r, Pval_r, Det_m, Det_M) sorted by Tval_r like ``res.sort('Tval_r')``."""
    L1 = np.prod(cube_local_min.shape[1:])                                    # :1425
    if segmap is not None:                                                    # :1428-1435
        segmask = segmap == 0
        cube_local_min = cube_local_min * segmask
        L0 = np.count_nonzero(segmask)
    else:
        L0 = L1
    if threshlist is None:                                                    # :1437-1442
        threshmax = min(cube_local_min.max(), cube_local_max.max())
        threshmin = np.median(np.amax(cube_local_max, axis=0)) * 1.1
        threshlist = np.linspace(threshmin, threshmax, 50)
    else:
        threshmin = np.min(threshlist)
    locM = cube_local_max[cube_local_max > threshmin]                         # :1444-1445
    locm = cube_local_min[cube_local_min > threshmin]
    n0, n1 = [], []
    for thresh in threshlist:                                                 # :1447-1450
        n1.append(np.count_nonzero(locM > thresh))
        n0.append(np.count_nonzero(locm > thresh))
    n0 = np.array(n0) * (L1 / L0)                                             # :1452
    n1 = np.array(n1)
    with np.errstate(divide="ignore", invalid="ignore"):
        est_purity = 1 - n0 / n1                                              # :1454
    tval = np.asarray(threshlist, dtype=float)
    order = np.argsort(tval, kind="stable")                                   # res.sort('Tval_r')
    cols = dict(Tval_r=tval[order], Pval_r=est_purity[order],
                Det_m=n0.astype(int)[order], Det_M=n1[order])
    if est_purity[-1] < purity:                                               # :1464-1468
        threshold = np.inf
    else:
        threshold = np.interp(purity, cols["Pval_r"], cols["Tval_r"])         # :1470
    return float(threshold), cols


# --------------------------------------------------------------------------
# Thresholding of step 7  (steps.py:935-939, :956-994: the inline NumPy of Detection.run, no
# lib_origin function -- nothing to import there, so this restatement is pinned by construction:
# it IS the reference's expression)
# --------------------------------------------------------------------------
def detection_threshold(cube_local_max, cube_profile, cube_std_local_max, threshold_correl,
                        threshold_std, maxdist_lines=2.5):
    """Cat0 columns of Detection.run (steps.py:956-981) as a dict of arrays in vstack order
    (correl detections, then std detections), and the indices of the std detections that
    survive the merging of :983-994 (those farther than maxdist_lines from every correl
    detection), sorted."""
    from scipy.spatial import cKDTree

    z, y, x = np.where(cube_local_max > threshold_correl)                      # :958
    t_glr = cube_local_max[z, y, x]                                            # :962
    prof = cube_profile[z, y, x]                                               # :963
    zs, ys, xs = np.where(cube_std_local_max > threshold_std)                  # :968
    std = cube_std_local_max[zs, ys, xs]                                       # :971
    n, m = len(z), len(zs)
    cat0 = dict(
        x0=np.concatenate([x, xs]), y0=np.concatenate([y, ys]), z0=np.concatenate([z, zs]),
        comp=np.concatenate([np.zeros(n, int), np.ones(m, int)]),              # :960, :970
        STD=np.concatenate([np.full(n, np.nan), std]),                         # :961, :971
        T_GLR=np.concatenate([t_glr, np.full(m, np.nan)]),                     # :962, :972
        profile=np.concatenate([prof, np.zeros(m, prof.dtype)]))               # :963, :973
    if n and m:                                                                # :984-992
        kdt_cor = cKDTree(np.array([x, y, z]).T)
        kdt_std = cKDTree(np.array([xs, ys, zs]).T)
        matched = set()
        for lst in kdt_cor.query_ball_tree(kdt_std, maxdist_lines):
            matched.update(lst)
    else:
        matched = set()
    unmatched = sorted(set(range(m)) - matched)
    return cat0, np.array(unmatched, dtype=int)


# --------------------------------------------------------------------------
# Whole chain in Step order (substitute for BASELINE config 0)
# --------------------------------------------------------------------------
def run_chain(cube_raw, var, mask, PSF, wfields, profiles, areamap, nbAreas,
              dct_order=10, dct_approx=False, pfa_test=0.01, Noise_population=50,
              itermax=100, ncpu=1, pcut=1e-8, pmeansub=True, svd="svds", timings=None):
    """preprocessing -> PCA threshold -> greedy PCA -> TGLR, as steps 1,3,4,5 do."""
    import time

    t0 = time.time()
    pre = preprocessing(cube_raw, var, mask, dct_order, dct_approx)
    t1 = time.time()
    testO2, histO2, binO2, thresO2, meaO2, stdO2 = pca_threshold_areas(
        pre["cube_std"], areamap, nbAreas, pfa_test)
    t2 = time.time()
    faint, mapO2, nstop = Compute_GreedyPCA_area(
        nbAreas, pre["cube_std"], areamap, Noise_population, thresO2, itermax, testO2, svd=svd)
    t3 = time.time()
    glr = compute_TGLR(faint, PSF, wfields, profiles, mask, ncpu=ncpu, pcut=pcut,
                       pmeansub=pmeansub)
    t4 = time.time()
    if timings is not None:
        timings.update(preprocessing=t1 - t0, pca_threshold=t2 - t1, greedy_pca=t3 - t2,
                       tglr=t4 - t3)
    out = dict(pre)
    out.update(testO2=testO2, thresO2=np.array(thresO2), meaO2=np.array(meaO2),
               stdO2=np.array(stdO2), cube_faint=faint, mapO2=mapO2, nstop=nstop)
    out.update(glr)
    return out
