"""Deterministic synthetic MUSE-like inputs for the hot path.

Everything is float32-representable so that the float64 CPU oracle and the fp32
HIP path see *identical* input values.  The cube is produced in fixed chunks of
``ZCHUNK`` wavelength planes (``default_rng([seed, chunk])``) so a 3681x600x600
cube (5.3 GB per array) never needs more than one chunk on the host; a spatial
window (tile) of a larger field can be generated without building the field.
"""
import numpy as np

ZCHUNK = 32
SIGMA_TO_FWHM = 2.0 * np.sqrt(2.0 * np.log(2.0))


def dico_fwhm(nprof=20, fwhm_min=2.0, fwhm_max=12.0, size=201):
    """Analytic equivalent of the reference's ``Dico_FWHM_2_12.fits`` (20 HDUs PROFnn,
    float64[201]): L2-normalised Gaussians centred on index 100 with
    FWHM = linspace(2, 12, 20) (reference origin.py:516-527).
    ``dico_fwhm(3)`` gives ``Dico_3FWHM.fits`` (FWHM 2, 6.7368.., 12 = indices 0, 9, 19)."""
    fwhms = np.linspace(fwhm_min, fwhm_max, 20)
    if nprof == 3:
        fwhms = fwhms[[0, 9, 19]]
    elif nprof != 20:
        fwhms = np.linspace(fwhm_min, fwhm_max, nprof)
    x = np.arange(size, dtype=np.float64) - size // 2
    out = []
    for f in fwhms:
        s = f / SIGMA_TO_FWHM
        p = np.exp(-0.5 * (x / s) ** 2)
        out.append(p / np.linalg.norm(p))
    return out


def moffat_psf(Nz, size=25, beta=2.8, fwhm0=3.6, fwhm1=3.0):
    """Per-channel Moffat PSF cube (Nz, size, size), unit sum per channel, float32.
    FWHM varies linearly from fwhm0 (blue) to fwhm1 (red) pixels."""
    c = size // 2
    yy, xx = np.mgrid[:size, :size]
    r2 = ((yy - c) ** 2 + (xx - c) ** 2).astype(np.float64)
    fwhm = np.linspace(fwhm0, fwhm1, Nz)
    alpha = fwhm / (2.0 * np.sqrt(2.0 ** (1.0 / beta) - 1.0))
    psf = (1.0 + r2[None] / alpha[:, None, None] ** 2) ** (-beta)
    psf /= psf.sum(axis=(1, 2), keepdims=True)
    return psf.astype(np.float32)


def grid_areamap(Ny, Nx, area_size=100):
    """Regular grid of area_size x area_size squares labelled 1..n (what
    ``CreateAreas`` with minsize=100 yields on a source-free field, steps.py:524-559)."""
    ny = max(1, Ny // area_size)
    nx = max(1, Nx // area_size)
    ey = np.minimum(np.arange(Ny) // area_size, ny - 1)
    ex = np.minimum(np.arange(Nx) // area_size, nx - 1)
    areamap = (ey[:, None] * nx + ex[None, :] + 1).astype(np.int32)
    return areamap, ny * nx


class SyntheticField:
    """Scene description for a (Nz, Ny, Nx) field; planes are rendered on demand."""

    def __init__(self, Nz, Ny, Nx, seed=None, psf_size=25, nprof=20, masked_border=0,
                 blob_density=1.0 / 400, emitter_density=1.0 / 900, area_size=100):
        self.Nz, self.Ny, self.Nx = Nz, Ny, Nx
        self.seed = 20260000 + Nx if seed is None else seed
        self.masked_border = masked_border
        rng = np.random.default_rng([self.seed, 0xC0FFEE])
        self.PSF = moffat_psf(Nz, psf_size)
        self.profiles = dico_fwhm(nprof)
        self.areamap, self.nbAreas = grid_areamap(Ny, Nx, area_size)
        zeta = np.arange(Nz) / Nz

        # continuum nuisances: Gaussian blobs (sigma 1.5 px) with smooth spectra
        nblob = int(round(Ny * Nx * blob_density))
        self.blob_y = rng.uniform(0, Ny, nblob)
        self.blob_x = rng.uniform(0, Nx, nblob)
        self.blob_amp = np.exp(rng.uniform(np.log(2.0), np.log(40.0), nblob))
        a = rng.uniform(-0.5, 0.5, nblob)
        om = rng.uniform(3.0, 30.0, nblob)
        ph = rng.uniform(0, 2 * np.pi, nblob)
        self.blob_spec = (1.0 + a[:, None] * zeta[None] + 0.2 * np.cos(om[:, None] * zeta[None]
                                                                        + ph[:, None]))
        self.blob_sigma = 1.5

        # emitters: point source x PSF(z) x Gaussian line
        nem = int(round(Ny * Nx * emitter_density))
        self.em_y = rng.integers(0, Ny, nem)
        self.em_x = rng.integers(0, Nx, nem)
        self.em_z = rng.uniform(30, Nz - 30, nem)
        fw = np.linspace(2.0, 12.0, 20)
        self.em_sigma = fw[rng.integers(0, 20, nem)] / SIGMA_TO_FWHM
        self.em_snr = rng.uniform(3.0, 8.0, nem)

    # -- rendering -----------------------------------------------------------
    def _sources(self, z0, z1, y0, y1, x0, x1):
        """Noise-free signal for planes [z0,z1) in window [y0,y1)x[x0,x1), float64."""
        nz = z1 - z0
        out = np.zeros((nz, y1 - y0, x1 - x0))
        r = int(np.ceil(5 * self.blob_sigma))
        for by, bx, amp, spec in zip(self.blob_y, self.blob_x, self.blob_amp, self.blob_spec):
            cy, cx = int(np.floor(by)), int(np.floor(bx))
            ya, yb = max(y0, cy - r), min(y1, cy + r + 1)
            xa, xb = max(x0, cx - r), min(x1, cx + r + 1)
            if ya >= yb or xa >= xb:
                continue
            yy = np.arange(ya, yb)[:, None] + 0.5 - by
            xx = np.arange(xa, xb)[None, :] + 0.5 - bx
            g = amp * np.exp(-0.5 * (yy ** 2 + xx ** 2) / self.blob_sigma ** 2)
            out[:, ya - y0: yb - y0, xa - x0: xb - x0] += spec[z0:z1, None, None] * g[None]
        P = self.PSF.shape[1]
        c = P // 2
        zz = np.arange(z0, z1)
        for ey, ex, ez, es, snr in zip(self.em_y, self.em_x, self.em_z, self.em_sigma,
                                       self.em_snr):
            if ez + 4 * es < z0 or ez - 4 * es >= z1:
                continue
            ya, yb = max(y0, ey - c), min(y1, ey + c + 1)
            xa, xb = max(x0, ex - c), min(x1, ex + c + 1)
            if ya >= yb or xa >= xb:
                continue
            line = snr * np.exp(-0.5 * ((zz - ez) / es) ** 2)
            psf = self.PSF[z0:z1, ya - ey + c: yb - ey + c, xa - ex + c: xb - ex + c]
            peak = self.PSF[z0:z1, c, c][:, None, None]
            out[:, ya - y0: yb - y0, xa - x0: xb - x0] += line[:, None, None] * psf / peak
        return out

    def chunk(self, ic, window=None):
        """Planes [ic*ZCHUNK, min(Nz,(ic+1)*ZCHUNK)) -> (raw f32, var f32, mask u8)."""
        z0 = ic * ZCHUNK
        z1 = min(self.Nz, z0 + ZCHUNK)
        y0, y1, x0, x1 = window or (0, self.Ny, 0, self.Nx)
        rng = np.random.default_rng([self.seed, ic])
        shape = (z1 - z0, self.Ny, self.Nx)
        var = (1.0 + 0.2 * rng.random(shape, dtype=np.float32)).astype(np.float32)
        noise = rng.standard_normal(shape, dtype=np.float32)
        var = var[:, y0:y1, x0:x1]
        raw = noise[:, y0:y1, x0:x1] * np.sqrt(var)
        raw = (raw + self._sources(z0, z1, y0, y1, x0, x1)).astype(np.float32)
        mask = np.zeros(raw.shape, dtype=np.uint8)
        b = self.masked_border
        if b > 0:
            gy = np.arange(y0, y1)
            gx = np.arange(x0, x1)
            my = (gy < b) | (gy >= self.Ny - b)
            mx = (gx < b) | (gx >= self.Nx - b)
            m2 = my[:, None] | mx[None, :]
            mask[:, m2] = 1
            raw[:, m2] = 0.0  # cube_raw = data.filled(0)      (origin.py:265)
            var[:, m2] = np.inf  # var = var.filled(inf)        (origin.py:274)
        return raw, np.ascontiguousarray(var), mask

    @property
    def nchunks(self):
        return (self.Nz + ZCHUNK - 1) // ZCHUNK

    def arrays(self, window=None):
        """Whole (windowed) cube on the host: raw, var (float32) and mask (bool)."""
        parts = [self.chunk(ic, window) for ic in range(self.nchunks)]
        raw = np.concatenate([p[0] for p in parts])
        var = np.concatenate([p[1] for p in parts])
        mask = np.concatenate([p[2] for p in parts]).astype(bool)
        return raw, var, mask


def small_case(Nz=200, Ny=24, Nx=28, seed=7, psf_size=9, nprof=3, masked_border=0,
               area_size=12, blob_density=1.0 / 60, emitter_density=1.0 / 150):
    """A seconds-scale case with the same statistics as the big configs."""
    f = SyntheticField(Nz, Ny, Nx, seed=seed, psf_size=psf_size, nprof=nprof,
                       masked_border=masked_border, blob_density=blob_density,
                       emitter_density=emitter_density, area_size=area_size)
    raw, var, mask = f.arrays()
    return f, raw, var, mask
