"""TEST INFRASTRUCTURE ONLY -- seeded inputs of the golden cases G1..G7.

The same functions are used by ``oracle/gen_golden.py`` (conda python3.9 + the real
reference, build container only) and by ``tests/`` (python3.10, anywhere) so inputs
never need to be stored: ``numpy.random.default_rng`` streams are identical in
numpy 1.26 and 2.2; every golden file carries a sha256 of its inputs which the
tests re-check before trusting the stored outputs.
"""
import hashlib
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(os.path.dirname(_HERE), "tests", "golden")


def _load_synth():
    spec = importlib.util.spec_from_file_location(
        "_origin_synth", os.path.join(os.path.dirname(_HERE), "origin_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


synth = _load_synth()


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


# ---------------------------------------------------------------- G1 / G2
def g1_inputs():
    """DCT / preprocessing: Nz=256, 12x12, a few masked voxels, one fully masked spaxel."""
    f = synth.SyntheticField(256, 12, 12, seed=101, psf_size=7, nprof=3,
                             blob_density=1 / 30, emitter_density=1 / 70, area_size=6)
    raw, var, mask = f.arrays()
    mask[40:43, 2, 3] = True
    mask[100, 7, 1] = True
    mask[:, 5, 5] = True
    mask[250:, 9, 9] = True
    raw[mask] = 0.0
    var[mask] = np.inf
    return dict(raw=raw, var=var, mask=mask, order=10)


# ---------------------------------------------------------------- G3
def g3_inputs():
    """O2 vectors for compute_thresh_gaussfit: chi2-like bulk + a tail of outliers."""
    out = {}
    for name, n, seed in (("a", 2304, 31), ("b", 10000, 32), ("c", 400, 33)):
        rng = np.random.default_rng(seed)
        t = (rng.standard_normal((n, 300)) ** 2).mean(axis=1)
        nout = max(3, n // 40)
        idx = np.argsort(rng.random(n))[:nout]  # (choice/chisquare streams vary with numpy)
        t[idx] *= np.exp(rng.uniform(0.1, 4.0, nout))
        if name == "a":
            t[5] = 0.0  # exercised by data[data > 0]  (lib_origin.py:999)
        out[name] = t
    return out


# ---------------------------------------------------------------- G4
def _std_cube(Nz, Ny, Nx, seed, area_size, blob_density=1 / 25, emitter_density=1 / 200):
    """A standardised cube (float32-representable float64) from a synthetic field."""
    f = synth.SyntheticField(Nz, Ny, Nx, seed=seed, psf_size=7, nprof=3,
                             blob_density=blob_density, emitter_density=emitter_density,
                             area_size=area_size)
    raw, var, mask = f.arrays()
    # cheap standardisation (no DCT needed for a PCA input): remove a per-spaxel
    # quadratic continuum so that only part of the blobs' spectra remains
    z = np.linspace(-1, 1, Nz)
    V = np.stack([np.ones(Nz), z, z * z], axis=1)
    r = raw.reshape(Nz, -1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(V, r, rcond=None)
    std = ((r - V @ coef) / np.sqrt(var.reshape(Nz, -1))).reshape(Nz, Ny, Nx)
    return f, std.astype(np.float32).astype(np.float64)


def g4_inputs():
    """Greedy PCA: three single-area cases (Nz, S) and one 2x2-area cube."""
    cases = {}
    for name, (Nz, Ny, Nx, seed, dens) in dict(a=(240, 20, 20, 41, 1 / 40),
                                               b=(300, 18, 25, 42, 1 / 80),
                                               c=(200, 24, 24, 43, 1 / 40)).items():
        _, std = _std_cube(Nz, Ny, Nx, seed, area_size=max(Ny, Nx), blob_density=dens)
        cases[name] = std.reshape(Nz, Ny * Nx)
    f, std = _std_cube(120, 32, 36, 44, area_size=16, blob_density=1 / 60)
    cases["area_cube"] = std
    cases["areamap"] = f.areamap
    cases["nbAreas"] = f.nbAreas
    return cases


# ---------------------------------------------------------------- G5
def _weights2(Ny, Nx):
    x = np.linspace(0, 1, Nx)[None, :] * np.ones((Ny, 1))
    w0 = (0.15 + 0.7 * x).astype(np.float32).astype(np.float64)
    return [w0, 1.0 - w0]


def g5_inputs():
    cases = {}

    def faint(Nz, Ny, Nx, seed):
        rng = np.random.default_rng(seed)
        c = rng.standard_normal((Nz, Ny, Nx), dtype=np.float32)
        # a couple of emitters so T has a real dynamic range
        for _ in range(3):
            z, y, x = rng.integers(10, Nz - 10), rng.integers(0, Ny), rng.integers(0, Nx)
            zz = np.arange(Nz)[:, None, None]
            yy = np.arange(Ny)[None, :, None]
            xx = np.arange(Nx)[None, None, :]
            c += (6 * np.exp(-0.5 * ((zz - z) / 2.0) ** 2)
                  * np.exp(-0.5 * ((yy - y) ** 2 + (xx - x) ** 2) / 1.5 ** 2)).astype(np.float32)
        return c.astype(np.float64)

    # a) small PSF, 3 profiles
    cases["a"] = dict(cube=faint(150, 17, 19, 51), fsf=synth.moffat_psf(150, 7).astype(float),
                      weights=None, profiles=synth.dico_fwhm(3), pcut=1e-8, pmeansub=True)
    # b) PSF larger than the field, 20 profiles
    cases["b"] = dict(cube=faint(200, 20, 20, 52), fsf=synth.moffat_psf(200, 25).astype(float),
                      weights=None, profiles=synth.dico_fwhm(20), pcut=1e-8, pmeansub=True)
    # c) two fields with weight maps
    p0 = synth.moffat_psf(150, 7, fwhm0=3.6, fwhm1=3.0).astype(float)
    p1 = synth.moffat_psf(150, 7, fwhm0=2.6, fwhm1=3.3).astype(float)
    cases["c"] = dict(cube=faint(150, 17, 19, 53), fsf=[p0, p1], weights=_weights2(17, 19),
                      profiles=synth.dico_fwhm(3), pcut=1e-8, pmeansub=True)
    # d) deliberately asymmetric PSF and profiles, no pcut, no mean subtraction
    rng = np.random.default_rng(54)
    pa = synth.moffat_psf(150, 7).astype(float)
    pa = pa * (1.0 + 0.5 * rng.random(pa.shape))
    pa /= pa.sum(axis=(1, 2), keepdims=True)
    pa = pa.astype(np.float32).astype(np.float64)
    profs = []
    for i, p in enumerate(synth.dico_fwhm(3)):
        q = p[80:121].copy() * (1.0 + 0.6 * np.linspace(-1, 1, 41))
        profs.append(q)
    cases["d"] = dict(cube=faint(150, 26, 29, 55), fsf=pa, weights=None, profiles=profs,
                      pcut=None, pmeansub=False)
    # e) production PSF size with a field that contains an interior
    cases["e"] = dict(cube=faint(100, 26, 27, 56), fsf=synth.moffat_psf(100, 25).astype(float),
                      weights=None, profiles=synth.dico_fwhm(3), pcut=1e-8, pmeansub=True)
    return cases


def g5_mask(shape, seed=57):
    rng = np.random.default_rng(seed)
    m = rng.random(shape) < 0.01
    m[:, 0, :] = True
    return m


# ---------------------------------------------------------------- G7
def g7_inputs():
    """Minicube-shaped chain (1100 x 65 x 80), 4 areas, Dico_3FWHM, PSF 25x25."""
    f = synth.SyntheticField(1100, 65, 80, seed=71, psf_size=25, nprof=3,
                             blob_density=1 / 200, emitter_density=1 / 400, area_size=40,
                             masked_border=0)
    # grid_areamap(65, 80, 40) -> 1 x 2; force a 2 x 2 grid as the reference test has 4 areas
    ey = (np.arange(65) >= 32).astype(int)
    ex = (np.arange(80) >= 40).astype(int)
    areamap = (ey[:, None] * 2 + ex[None, :] + 1).astype(np.int32)
    raw, var, mask = f.arrays()
    return dict(raw=raw, var=var, mask=mask, PSF=f.PSF.astype(float), profiles=f.profiles,
                areamap=areamap, nbAreas=4)


# ---------------------------------------------------------------- input hand-over
# libm-level differences (exp/pow) between the numpy of the conda python3.9 that can
# import the reference and the system python3.10 that runs the tests would make the
# float inputs differ in the last bit.  So the inputs are always *produced* by the
# system python (this file run as a script) and handed to gen_golden.py as an .npz.
def g8_inputs():
    """Purity threshold: local-maximum cubes as compute_local_max leaves them
    (mostly zeros, positive maxima), a segmentation map, an explicit threshold list."""
    rng = np.random.default_rng(88)
    shape = (120, 30, 34)

    def sparse_maxima(scale, frac):
        v = np.abs(rng.standard_normal(shape)) * scale + 2.0
        v[rng.random(shape) > frac] = 0.0
        return v.astype(np.float32).astype(np.float64)

    lmax = sparse_maxima(3.0, 0.03)
    lmax[40:44, 10:13, 12:16] += 9.0 * (lmax[40:44, 10:13, 12:16] > 0)   # real detections
    lmin = sparse_maxima(2.0, 0.03)
    segmap = np.zeros(shape[1:], dtype=int)
    segmap[8:15, 10:18] = 1
    segmap[20:24, 3:9] = 2
    return dict(lmax=lmax, lmin=lmin, segmap=segmap,
                threshlist=np.array([9.5, 4.0, 6.25, 5.0, 7.75, 3.0, 12.0]))


def g10_inputs():
    """Area construction: exposure map with a ragged unexposed border and a hole,
    merged segmentation map with compact sources (each with >= 3 non-collinear pixels, what
    ConvexHull needs), sized so that the 3 x 3 grid of squares needs merging and growing.
    Two cases: sources in most squares / few sources (areas without one are dropped and the
    others grow over them)."""
    rng = np.random.default_rng(1010)
    Ny, Nx = 132, 150
    yy, xx = np.mgrid[:Ny, :Nx]

    def field(nsrc, seed):
        r = np.random.default_rng(seed)
        exp = np.ones((Ny, Nx), dtype=bool)
        exp[:3 + (xx[0] % 7 == 0).astype(int).max(), :] = False
        exp[:, :4] = False
        exp[-5:, :] = False
        exp[(yy > 118) & (xx > 120 + (yy - 118) * 2)] = False      # ragged corner
        exp[(yy - 70) ** 2 + (xx - 95) ** 2 < 36] = False          # hole
        seg = np.zeros((Ny, Nx), dtype=np.int32)
        k = 0
        while k < nsrc:
            cy, cx = r.integers(8, Ny - 10), r.integers(8, Nx - 8)
            ry, rx = r.integers(2, 6), r.integers(2, 7)
            blob = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            if (seg[blob] > 0).any() or not exp[blob].all() or blob.sum() < 6:
                continue
            k += 1
            seg[blob] = k
        return exp, seg

    e1, s1 = field(14, 5)
    e2, s2 = field(4, 6)
    del rng
    # cube masks (True = masked): a few channels, exposure = any unmasked channel
    m1 = np.broadcast_to(~e1, (3, Ny, Nx)).copy()
    m2 = np.broadcast_to(~e2, (3, Ny, Nx)).copy()
    m1[1] = False                      # channel 1 fully valid except ...
    m1[1][~e1] = True
    return dict(many=dict(mask=m1, segmap=s1, minsize=40, maxsize=None, pfa=0.2),
                few=dict(mask=m2, segmap=s2, minsize=45, maxsize=70, pfa=0.2))


def _flatten(prefix, obj, out):
    if isinstance(obj, dict):
        for k, v in obj.items():
            _flatten(f"{prefix}/{k}", v, out)
    elif isinstance(obj, (list, tuple)):
        out[f"{prefix}/__len__"] = np.array(len(obj))
        for i, v in enumerate(obj):
            _flatten(f"{prefix}/{i}", v, out)
    elif obj is None:
        out[f"{prefix}/__none__"] = np.array(0)
    else:
        out[prefix] = np.asarray(obj)


def _unflatten(flat):
    tree = {}
    for key, val in flat.items():
        parts = key.strip("/").split("/")
        node = tree
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = val

    def fix(node):
        if not isinstance(node, dict):
            return node[()] if getattr(node, "ndim", 1) == 0 else node
        if "__none__" in node:
            return None
        if "__len__" in node:
            return [fix(node[str(i)]) for i in range(int(node["__len__"]))]
        return {k: fix(v) for k, v in node.items()}

    return fix(tree)


def all_inputs():
    return dict(g1=g1_inputs(), g3=g3_inputs(), g4=g4_inputs(), g5=g5_inputs(), g7=g7_inputs(),
                g8=g8_inputs(), g10=g10_inputs())


def dump_inputs(path):
    flat = {}
    _flatten("", all_inputs(), flat)
    np.savez(path, **flat)


def load_inputs(path):
    with np.load(path) as z:
        return _unflatten({k: z[k] for k in z.files})


if __name__ == "__main__":
    import sys

    dump_inputs(sys.argv[1])


def g9_inputs():
    """FITS dump / load: one array per file type the steps write, with the
    special values a byte-level codec can get wrong, and the world-coordinate cards a MUSE
    cube carries."""
    rng = np.random.default_rng(99)
    cube64 = rng.standard_normal((7, 5, 6)) * 1e3
    cube64[0, 0, :6] = [np.nan, np.inf, -np.inf, -0.0, 5e-324, 1.7976931348623157e308]
    cube64[1, 2, 3] = np.float64(np.float32(1.0) / np.float32(3.0))
    cube32 = (rng.standard_normal((7, 5, 6)) * 50).astype(np.float32)
    cube32[3, 1, :4] = [np.nan, np.inf, -0.0, 1e-45]
    prof8 = rng.integers(0, 20, size=(7, 5, 6)).astype(np.uint8)
    prof8[6, 4, 5] = 255
    area64 = rng.integers(0, 37, size=(5, 6)).astype(np.int64)
    area64[0, 0] = -3
    img64 = rng.standard_normal((5, 6)) * 40
    table = dict(Tval_r=np.linspace(3.0, 12.0, 9), Pval_r=np.linspace(0.2, 1.0, 9),
                 Det_m=np.arange(9, dtype=np.int64) * 3, Det_M=np.arange(9, dtype=np.int64)[::-1] * 7 + 1)
    wcs = dict(CRPIX1=3.5, CRPIX2=3.0, CRVAL1=53.16, CRVAL2=-27.78, CD1_1=-5.55555555555556e-05,
               CD1_2=0.0, CD2_1=0.0, CD2_2=5.55555555555556e-05, CTYPE1='RA---TAN',
               CTYPE2='DEC--TAN', CUNIT1='deg', CUNIT2='deg')
    wave = dict(CRPIX3=1.0, CRVAL3=4750.0, CD3_3=1.25, CTYPE3='AWAV', CUNIT3='Angstrom')
    return dict(cube64=cube64, cube32=cube32, prof8=prof8, area64=area64, img64=img64,
                table=table, wcs=wcs, wave=wave)
