"""Device-level stage functions: thin, typed wrappers over the C ABI working on
``DeviceArray`` s.  Everything here is asynchronous on the context's stream unless it
returns host values."""
import ctypes as C

import numpy as np

from . import _capi
from .device import DeviceArray

NULL = C.c_void_p(0)


def _p(a):
    return NULL if a is None else a.p


# ------------------------------------------------------------------------- DCT / O2
def dct_fit(ctx, raw, var, mask, order=10, approx=False, coef=None):
    Nz, Ny, Nx = raw.shape
    if coef is None:
        coef = ctx.empty((order + 1, Ny, Nx), np.float64)
    _capi.call("origin_dct_fit", ctx.handle, raw.p, var.p, mask.p, Nz, Ny, Nx, int(order),
               int(bool(approx)), coef.p)
    return coef


def dct_fit_sums(ctx, raw, var, mask, order=10, approx=False, coef=None, zsum=None, zcnt=None):
    """``dct_fit`` and ``dct_resid_sums`` in one call: the per-channel sums of raw over the
    unmasked spaxels are taken inside the moments pass, which reads raw and mask anyway (the
    5 B/voxel plane pass becomes a mask-only visit of the spaxel groups that have masked voxels).
    Returns (coef, zsum, zcnt)."""
    Nz, Ny, Nx = raw.shape
    if coef is None:
        coef = ctx.empty((order + 1, Ny, Nx), np.float64)
    zsum = ctx.empty((Nz,), np.float64) if zsum is None else zsum
    zcnt = ctx.empty((Nz,), np.float64) if zcnt is None else zcnt
    _capi.call("origin_dct_fit_sums", ctx.handle, raw.p, var.p, mask.p, Nz, Ny, Nx, int(order),
               int(bool(approx)), coef.p, zsum.p, zcnt.p)
    return coef, zsum, zcnt


def dct_continuum(ctx, coef, Nz, out=None):
    na, Ny, Nx = coef.shape
    if out is None:
        out = ctx.empty((Nz, Ny, Nx), np.float32)
    _capi.call("origin_dct_continuum", ctx.handle, coef.p, Nz, Ny, Nx, na - 1, out.p)
    return out


def dct_resid_sums(ctx, raw, mask, coef, zsum=None, zcnt=None):
    Nz, Ny, Nx = raw.shape
    zsum = ctx.empty((Nz,), np.float64) if zsum is None else zsum
    zcnt = ctx.empty((Nz,), np.float64) if zcnt is None else zcnt
    _capi.call("origin_dct_resid_sums", ctx.handle, raw.p, mask.p, coef.p, Nz, Ny, Nx,
               coef.shape[0] - 1, zsum.p, zcnt.p)
    return zsum, zcnt


def dct_standardize(ctx, raw, var, mask, coef, zsum, zcnt, cube_std=None, cont_dct=None,
                    want_cont=True, want_images=True, o2=None, ima_std=None):
    Nz, Ny, Nx = raw.shape
    cube_std = ctx.empty((Nz, Ny, Nx), np.float32) if cube_std is None else cube_std
    if cont_dct is None and want_cont:
        cont_dct = ctx.empty((Nz, Ny, Nx), np.float32)
    if ima_std is None and want_images:
        ima_std = ctx.empty((Ny, Nx), np.float32)
    ima_dct = ctx.empty((Ny, Nx), np.float32) if (want_images and cont_dct is not None) else None
    if o2 is None and want_images:
        o2 = ctx.empty((Ny, Nx), np.float64)
    _capi.call("origin_dct_standardize", ctx.handle, raw.p, var.p, mask.p, coef.p, zsum.p,
               zcnt.p, Nz, Ny, Nx, coef.shape[0] - 1, cube_std.p, _p(cont_dct), _p(ima_std),
               _p(ima_dct), _p(o2))
    return dict(cube_std=cube_std, cont_dct=cont_dct, ima_std=ima_std, ima_dct=ima_dct, o2=o2)


def dct_cont_std(ctx, var, coef, cont_dct=None, want_image=True, ima_dct=None, aux=False):
    """cont_dct = continuum / sqrt(var) and its mean image, as a pass of its own (what
    ``dct_standardize(want_cont=False)`` leaves out).  Give ``cont_dct`` / ``ima_dct`` buffers to
    keep the call free of allocations (an allocation can wait for the device).  ``aux``: on the
    context's low-priority auxiliary stream -- ``ctx.aux_join()`` or ``ctx.sync()`` before the
    outputs are read or ``var`` / ``coef`` rewritten."""
    Nz, Ny, Nx = var.shape
    cont_dct = ctx.empty((Nz, Ny, Nx), np.float32) if cont_dct is None else cont_dct
    if ima_dct is None and want_image:
        ima_dct = ctx.empty((Ny, Nx), np.float32)
    _capi.call("origin_dct_cont_std_async" if aux else "origin_dct_cont_std", ctx.handle, var.p,
               coef.p, Nz, Ny, Nx, coef.shape[0] - 1, cont_dct.p, _p(ima_dct))
    return dict(cont_dct=cont_dct, ima_dct=ima_dct)


def o2test(ctx, cube, out=None):
    Nz = cube.shape[0]
    S = cube.size // Nz
    out = ctx.empty(cube.shape[1:], np.float64) if out is None else out
    _capi.call("origin_o2", ctx.handle, cube.p, Nz, S, out.p)
    return out


# ------------------------------------------------------------------------- GLR
def prepare_profiles(profiles, pcut=None, pmeansub=True):
    """Trim / normalise / mean-subtract the dictionary exactly as the reference does
    (lib_origin.py:1155-1165), in float64 on the host (K small vectors)."""
    out = []
    for prof in profiles:
        prof = np.array(prof, dtype=np.float64)
        if pcut is not None:
            lpeak = prof.argmax()
            lw = np.max(np.abs(np.where(prof >= pcut)[0][[0, -1]] - lpeak))
            prof = prof[lpeak - lw: lpeak + lw + 1]
        prof /= np.linalg.norm(prof)
        if pmeansub:
            prof -= prof.mean()
        out.append(prof)
    return out


class GLRPlan:
    """Device-side constants of a GLR run: zero-mean PSFs, weights, prepared profiles and
    normalisation tables (include/origin_hip.h: origin_glr_plan)."""

    PRECISIONS = ("f32", "f16x2", "bf16")   # origin_glr_plan_set_precision codes 0, 1, 2

    def __init__(self, ctx, shape, fsf, weights, profiles, pcut=None, pmeansub=True,
                 precision=None):
        """``precision``: None = library default ("f16x2" where eligible), "f16x2" = both GLR stages
        on the matrix cores with a two-term f16 split (fp32 accumulation, fp32-class results),
        "bf16" = spectral stage with ONE bf16 MFMA per product (BASELINE config 4: SURVEY 8c
        bf16 tolerances), "f32" = fp32 FMA kernels.  ``self.precision`` tells what the plan will
        run (plans with weight maps or very wide profiles only have "f32")."""
        Nz, Ny, Nx = (int(s) for s in shape)
        self.ctx, self.shape = ctx, (Nz, Ny, Nx)
        if weights is None:  # one FSF                         (lib_origin.py:1112-1114)
            fsf_list, w = [fsf], None
        else:
            fsf_list, w = list(fsf), [np.asarray(x, dtype=np.float64) for x in weights]
            if len(fsf_list) != len(w):
                raise ValueError("need one weight map per FSF")
        psf = np.ascontiguousarray(np.stack([np.asarray(f, dtype=np.float64) for f in fsf_list]))
        if psf.ndim != 4 or psf.shape[1] != Nz or psf.shape[2] != psf.shape[3]:
            raise ValueError(f"FSF must be (Nz, P, P) per field, got {psf.shape[1:]}")
        self.P = psf.shape[2]
        wptr = NULL
        if w is not None:
            warr = np.ascontiguousarray(np.stack(w))
            if warr.shape != (len(fsf_list), Ny, Nx):
                raise ValueError("weight maps must be (Ny, Nx)")
            wptr = warr.ctypes.data_as(C.c_void_p)
        prof = prepare_profiles(profiles, pcut, pmeansub)
        self.K = len(prof)
        self.tap_lengths = [len(p) for p in prof]
        off = np.zeros(self.K + 1, dtype=np.int32)
        off[1:] = np.cumsum(self.tap_lengths)
        taps = np.ascontiguousarray(np.concatenate(prof))
        self._h = C.c_void_p()
        _capi.call("origin_glr_plan_create", ctx.handle, Nz, Ny, Nx, len(fsf_list), self.P,
                   psf.ctypes.data_as(C.c_void_p), wptr, self.K,
                   taps.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                   C.byref(self._h))
        n = C.c_size_t()
        _capi.call("origin_glr_work_elems", self._h, C.byref(n))
        self.work_elems = n.value
        _capi.call("origin_glr_plan_bytes", self._h, C.byref(n))
        self.nbytes = n.value
        self._work = None
        if precision is not None:
            if precision not in self.PRECISIONS:
                raise ValueError("precision must be None, 'f32', 'f16x2' or 'bf16'")
            _capi.call("origin_glr_plan_set_precision", self._h, self.PRECISIONS.index(precision))
        got = C.c_int()
        _capi.call("origin_glr_plan_get_precision", self._h, C.byref(got))
        self.precision = self.PRECISIONS[got.value]
        # mirrors origin_spatial_mfma_ok (csrc/glr_spatial_mfma.hip): which spatial kernel runs
        # (weighted mosaics included: per-field accumulation on the matrix cores; their spectral
        # stage convolves the norm cube next to the data and stays in fp32)
        self.spatial_on_matrix_cores = self.precision != "f32" and 5 <= self.P <= 25
        # (a weighted plan's spectral stage -- second Toeplitz product for the denominator,
        # csrc/glr_spectral_norm_mfma.hip -- exists for the f16 split only)
        self.spectral_on_matrix_cores = self.precision != "f32" and (
            w is None or (self.precision == "f16x2" and len(self.tap_lengths) <= 26
                          and max(self.tap_lengths) <= 65))

    # -- the same run in row bands (include/origin_hip.h origin_glr_run_rows) ---------------
    def rows_supported(self):
        """Whether the plan's two stages run the table kernels on the matrix cores (one field,
        no weight maps, PSF 5..25, half widths <= 32): only those plans run in row bands."""
        ok = C.c_int()
        _capi.call("origin_glr_rows_supported", self._h, C.byref(ok))
        return bool(ok.value)

    def run_rows(self, cube, mask, correl, profile, correl_min, y0, y1, first=False, side=False):
        """The rows [y0, y1) of a run (y0 a multiple of 64, y1 one or Ny): spatial stage (reads
        cube rows [y0 - P//2, y1 + P//2)), spectral stage, partial maps.  ``first``: the first band
        of a run; ``side``: enqueue on the context's side stream (all CUs but a reserve), behind
        what the main stream holds now."""
        assert cube.shape == self.shape and cube.dtype == np.float32
        _capi.call("origin_glr_run_rows", self.ctx.handle, self._h, cube.p, _p(mask),
                   self.workspace().p, correl.p, profile.p, correl_min.p, int(y0), int(y1),
                   (1 if first else 0) | (2 if side else 0))

    def run_rect(self, cube, mask, correl, profile, correl_min, y0, y1, x0, x1, first=False,
                 side=False):
        """Rows [y0, y1) x columns [x0, x1) of a run (each a multiple of 64 or the field's end);
        see ``run_rows``.  With a proper column range the results agree with ``run`` to rounding
        (the spectral stage's waves hold 32 columns of one row), not bit for bit."""
        assert cube.shape == self.shape and cube.dtype == np.float32
        _capi.call("origin_glr_run_rect", self.ctx.handle, self._h, cube.p, _p(mask),
                   self.workspace().p, correl.p, profile.p, correl_min.p, int(y0), int(y1), int(x0),
                   int(x1), (1 if first else 0) | (2 if side else 0))

    def run_finish(self, want_maps=True):
        """Ends a run in row bands: the main stream waits for the side bands; returns
        (maxmap, minmap) or (None, None)."""
        Nz, Ny, Nx = self.shape
        maxmap = self.ctx.empty((Ny, Nx), np.float32) if want_maps else None
        minmap = self.ctx.empty((Ny, Nx), np.float32) if want_maps else None
        _capi.call("origin_glr_run_finish", self.ctx.handle, self._h, self.workspace().p,
                   _p(maxmap), _p(minmap))
        return maxmap, minmap

    def mfma_count(self):
        """(spatial, spectral): matrix-core instructions (32768 flop each) one run issues per
        stage -- rocprofv3's SQ_INSTS_MFMA per launch; 0 for a stage on the fp32 kernels."""
        a, b = C.c_long(), C.c_long()
        _capi.call("origin_glr_plan_mfma_count", self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def fold_eps(self):
        """(eps, active): FOLD of the matrix-core spectral stage -- the plan's measured
        max |1/(a_k sqrt(den_k)) / s - 1| away from the cube's ends and whether the stage uses it
        (eps <= 2e-6, include/origin_hip.h origin_glr_plan_fold_eps)."""
        e, a = C.c_float(), C.c_int()
        _capi.call("origin_glr_plan_fold_eps", self._h, C.byref(e), C.byref(a))
        return e.value, bool(a.value)

    def close(self):
        if self._h is not None and self._h.value:
            # (a plan outliving its context -- interpreter exit closes contexts first -- is not
            # destroyed through the dangling context pointer it holds: the process is ending)
            if self.ctx.handle.value:
                _capi.load().origin_glr_plan_destroy(self._h)
            self._h = C.c_void_p()
        self._work = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace(self):
        if self._work is None:
            self._work = self.ctx.empty((self.work_elems,), np.float32)
        return self._work

    def run(self, cube, mask=None, correl=None, profile=None, correl_min=None, want_maps=True):
        ctx = self.ctx
        Nz, Ny, Nx = self.shape
        assert cube.shape == self.shape and cube.dtype == np.float32
        correl = ctx.empty(self.shape, np.float32) if correl is None else correl
        correl_min = ctx.empty(self.shape, np.float32) if correl_min is None else correl_min
        profile = ctx.empty(self.shape, np.uint8) if profile is None else profile
        maxmap = ctx.empty((Ny, Nx), np.float32) if want_maps else None
        minmap = ctx.empty((Ny, Nx), np.float32) if want_maps else None
        _capi.call("origin_glr_run", ctx.handle, self._h, cube.p, _p(mask), self.workspace().p,
                   correl.p, profile.p, correl_min.p, _p(maxmap), _p(minmap))
        return dict(correl=correl, profile=profile, correl_min=correl_min, maxmap=maxmap,
                    minmap=minmap)


def local_max(ctx, correl, correl_min, mask, size=3, out_max=None, out_min=None):
    Nz, Ny, Nx = correl.shape
    out_max = ctx.empty(correl.shape, np.float32) if out_max is None else out_max
    out_min = ctx.empty(correl.shape, np.float32) if out_min is None else out_min
    _capi.call("origin_local_max", ctx.handle, correl.p, correl_min.p, _p(mask), Nz, Ny, Nx,
               int(size), out_max.p, out_min.p)
    return out_max, out_min


def zmax_map(ctx, cube, keep=None):
    """Per-spaxel maximum over z of a float32 device cube -> host float64 (Ny, Nx);
    ``keep`` (uint8 DeviceArray [Ny*Nx], 0 = excluded) zeroes spaxels first."""
    Nz, Ny, Nx = cube.shape
    m = ctx.empty((Ny, Nx), np.float32)
    _capi.call("origin_zmax_map", ctx.handle, cube.p, _p(keep), Nz, Ny * Nx, m.p)
    return m.to_host().astype(np.float64)


def where_above(ctx, cube, threshold, aux=None, cap=1 << 20):
    """``z, y, x = np.where(cube > threshold)`` on a float32 device cube, in NumPy's order, with
    the values there (and those of the uint8 device cube ``aux``): Detection.run's thresholding
    (steps.py:956-974).  Only the detections cross PCIe.  Returns a dict of host arrays
    ``z, y, x`` (int64), ``value`` (float64) and, with ``aux``, ``aux`` (uint8)."""
    Nz, Ny, Nx = cube.shape
    if cube.dtype != np.float32:
        raise TypeError("where_above needs a float32 device cube")
    if aux is not None and (aux.dtype != np.uint8 or aux.shape != cube.shape):
        raise ValueError("aux must be a uint8 device cube of the same shape")
    count = C.c_long(0)
    cap = max(int(cap), 1)
    while True:
        idx = ctx.empty((3, cap), np.int32)
        val = ctx.empty((cap,), np.float32)
        ax = ctx.empty((cap,), np.uint8) if aux is not None else None
        _capi.call("origin_where_above", ctx.handle, cube.p, _p(aux), Nz, Ny, Nx,
                   float(threshold), cap, idx.view(0, (cap,)).p, idx.view(cap, (cap,)).p,
                   idx.view(2 * cap, (cap,)).p, val.p, _p(ax), C.byref(count))
        n = count.value
        if n <= cap:
            break
        cap = n   # more detections than expected: once more with room for all of them
    def head(a, off, dtype):   # the first n entries only
        if n == 0:
            return np.empty(0, dtype=dtype)
        return a.view(off, (n,)).to_host().astype(dtype)

    out = dict(z=head(idx, 0, np.int64), y=head(idx, cap, np.int64),
               x=head(idx, 2 * cap, np.int64), value=head(val, 0, np.float64))
    if aux is not None:
        out["aux"] = head(ax, 0, np.uint8)
    return out


def count_above(ctx, cube, thresholds, keep=None):
    """counts[t] = #{voxels of cube (x keep) with value > thresholds[t]} (int64, host)."""
    Nz, Ny, Nx = cube.shape
    thr = np.ascontiguousarray(thresholds, dtype=np.float64)
    out = np.zeros(thr.size, dtype=np.int64)
    _capi.call("origin_count_above", ctx.handle, cube.p, _p(keep), Nz, Ny * Nx, thr.size,
               thr.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out
