"""Local-maximum cubes in sparse form.

``compute_local_max`` (reference lib_origin.py:1220-1256) returns two cubes that are zero except
at the 3x3x3 maxima of correl / of -correl_min -- one voxel in ~70 of a smoothed cube -- and its
consumers only ever count or pick those voxels (``Compute_threshold_purity`` :1391-1479,
``Detection.run`` steps.py:935-974).  In the device-resident chain the pass therefore appends
(linear index, value) pairs to per-wave segments (``origin_local_max_sparse``,
csrc/localmax.hip) instead of writing 8 B per voxel of zeros; ``SparseCube`` is what a DataObj
holds then.  It answers the consumers' questions from the lists and becomes the dense array the
reference's interface promises when somebody asks for one (``to_host_f64`` -- ``._data`` of the
DataObj --, ``dense()`` on the device); values and support are bit for bit those of the dense pass.
"""
import ctypes as C

import numpy as np

from . import _capi
from .device import DeviceArray

NULL = C.c_void_p(0)


def plan(ctx, shape):
    """(nseg, seg_cap) of the sparse pass for a cube shape; nseg 0 = no sparse form (Nx % 4)."""
    Nz, Ny, Nx = (int(v) for v in shape)
    nseg, cap = C.c_long(0), C.c_int(0)
    _capi.call("origin_local_max_sparse_plan", ctx.handle, Nz, Ny, Nx, C.byref(nseg), C.byref(cap))
    return nseg.value, cap.value


class SparseBuffers:
    """The segment arrays of one pass (both cubes), reusable from step to step."""

    def __init__(self, ctx, shape):
        self.ctx, self.shape = ctx, tuple(int(v) for v in shape)
        self.nseg, self.seg_cap = plan(ctx, shape)
        if self.nseg == 0:
            raise ValueError(f"no sparse local-maximum form for shape {self.shape}")
        n = self.nseg * self.seg_cap
        self.idx = [ctx.empty((n,), np.int64), ctx.empty((n,), np.int64)]
        self.val = [ctx.empty((n,), np.float32), ctx.empty((n,), np.float32)]
        self.counts = ctx.empty((2 * self.nseg,), np.int32)


def local_max_sparse(ctx, correl, correl_min, mask, bufs=None):
    """The 3x3x3 local maxima of correl and of -correl_min as two ``SparseCube`` s (asynchronous;
    nothing is read back).  ``bufs``: a ``SparseBuffers`` to write into (the cubes returned then
    alias it and are overwritten by the next pass that uses it)."""
    if bufs is None:
        bufs = SparseBuffers(ctx, correl.shape)
    Nz, Ny, Nx = correl.shape
    _capi.call("origin_local_max_sparse", ctx.handle, correl.p, correl_min.p,
               NULL if mask is None else mask.p, Nz, Ny, Nx, bufs.nseg, bufs.seg_cap,
               bufs.idx[0].p, bufs.val[0].p, bufs.idx[1].p, bufs.val[1].p, bufs.counts.p)
    return SparseCube(bufs, 0), SparseCube(bufs, 1)


def local_max(ctx, correl, correl_min, mask, size=3, check=True):
    """``compute_local_max`` on device cubes in the cheapest form that is exact: the sparse pass
    where it exists (size 3, Nx % 4 == 0, aligned cubes) -- ``check``: read the segment counts
    back (64 KB, one synchronisation) and fall back to the dense pass if a segment overflowed --,
    the dense kernels otherwise.  Returns two cubes: ``SparseCube`` s or float32 DeviceArrays."""
    from . import kernels
    ok = (int(size) == 3 and correl.shape[2] % 4 == 0 and correl.ptr % 16 == 0
          and correl_min.ptr % 16 == 0 and (mask is None or mask.ptr % 4 == 0))
    if ok and plan(ctx, correl.shape)[0] > 0:
        a, b = local_max_sparse(ctx, correl, correl_min, mask)
        if not check:
            return a, b
        try:
            a.counts(), b.counts()
            return a, b
        except SparseOverflow:
            pass
    return kernels.local_max(ctx, correl, correl_min, mask, size)


class SparseOverflow(RuntimeError):
    """A segment held more entries than it has room for (a cube with far more local maxima than
    one voxel in 8): the caller runs the dense pass instead."""


class SparseCube:
    """One of the two cubes of a sparse local-maximum pass.  ``shape`` / ``dtype`` / ``to_host`` /
    ``to_host_f64`` read like a float32 DeviceArray's; ``count_above`` / ``zmax_map`` /
    ``where_above`` are the reductions steps 6 and 7 make of it."""

    dtype = np.dtype(np.float32)

    def __init__(self, bufs, which):
        self.bufs, self.which, self.ctx = bufs, which, bufs.ctx
        self.shape = bufs.shape
        self.size = int(np.prod(self.shape))
        self._counts = None
        self._dense = None

    # -- the lists ---------------------------------------------------------------------
    def _args(self):
        b = self.bufs
        cnt = b.counts.view(self.which * b.nseg, (b.nseg,))
        return b.idx[self.which].p, b.val[self.which].p, cnt.p, b.nseg, b.seg_cap

    def counts(self):
        """Entries per segment (host int32; checks for overflow)."""
        if self._counts is None:
            b = self.bufs
            c = b.counts.to_host()[self.which * b.nseg:(self.which + 1) * b.nseg]
            if c.size and c.max() > b.seg_cap:
                raise SparseOverflow(f"{int(c.max())} local maxima in a segment of {b.seg_cap}")
            self._counts = c
        return self._counts

    @property
    def nnz(self):
        return int(self.counts().sum())

    def entries(self):
        """(linear indices int64, values float32) of the non-zero voxels on the host, sorted by
        index (C order: np.where's).  Compacted on the device (every entry is above -inf): only
        the entries cross PCIe, not the segments' empty tails."""
        idx, val, _ = self._above(-np.inf, None, max(self.nnz, 1))
        return idx, val

    def _above(self, threshold, aux, cap):
        ctx, count = self.ctx, C.c_long(0)
        cap = max(int(cap), 1)
        while True:
            oi, ov = ctx.empty((cap,), np.int64), ctx.empty((cap,), np.float32)
            oa = ctx.empty((cap,), np.uint8) if aux is not None else None
            _capi.call("origin_sparse_where_above", ctx.handle, *self._args(), float(threshold),
                       NULL if aux is None else aux.p, cap, oi.p, ov.p,
                       NULL if oa is None else oa.p, C.byref(count))
            n = count.value
            if n <= cap:
                break
            cap = n
        if n == 0:
            return (np.zeros(0, np.int64), np.zeros(0, np.float32),
                    np.zeros(0, np.uint8) if aux is not None else None)
        idx = oi.view(0, (n,)).to_host()
        order = np.argsort(idx, kind="stable")
        return (idx[order], ov.view(0, (n,)).to_host()[order],
                oa.view(0, (n,)).to_host()[order] if aux is not None else None)

    # -- dense forms -------------------------------------------------------------------
    def dense(self):
        """The dense float32 cube on the device (made once, kept)."""
        if self._dense is None:
            self.counts()
            d = self.ctx.empty(self.shape, np.float32)
            _capi.call("origin_sparse_to_dense", self.ctx.handle, *self._args(), d.p, d.size)
            self._dense = d
        return self._dense

    def to_host(self, out=None, dtype=np.float32):
        """Dense host array: zeros plus the entries (only the lists cross PCIe)."""
        idx, val = self.entries()
        if out is None:
            out = np.zeros(self.shape, dtype)
        else:
            out[...] = 0
        out.reshape(-1)[idx] = val
        return out

    def to_host_f64(self, out=None):
        return self.to_host(out, np.float64)

    def gathered(self, ctx):
        return self.dense()

    # -- the consumers' reductions --------------------------------------------------------
    def _keep(self, keep):
        if keep is None:
            return NULL, None
        k = keep if isinstance(keep, DeviceArray) else self.ctx.to_device(
            np.ascontiguousarray(keep, dtype=np.uint8).reshape(-1))
        return k.p, k

    def count_above(self, thresholds, keep=None):
        """counts[t] = #{voxels (x keep) with value > thresholds[t]} (int64, host)."""
        self.counts()
        Nz, Ny, Nx = self.shape
        thr = np.ascontiguousarray(thresholds, dtype=np.float64)
        neg = thr < 0
        if neg.any():   # the zeros of the dense cube are above a negative threshold too
            thr = np.concatenate([thr, [-np.inf]])
        out = np.zeros(thr.size, dtype=np.int64)
        kp, _hold = self._keep(keep)
        _capi.call("origin_sparse_count_above", self.ctx.handle, *self._args(), kp, Ny * Nx,
                   thr.size, thr.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        if neg.any():
            kept = Ny * Nx if keep is None else int(np.count_nonzero(
                keep.to_host() if isinstance(keep, DeviceArray) else np.asarray(keep)))
            zeros = Nz * kept - int(out[-1])
            out = out[:-1]
            out[neg] += zeros
        return out

    def zmax_map(self, keep=None):
        """max over z per spaxel -> host float64 (Ny, Nx); keep == 0 spaxels give 0."""
        self.counts()
        Nz, Ny, Nx = self.shape
        m = self.ctx.empty((Ny, Nx), np.float32)
        kp, _hold = self._keep(keep)
        _capi.call("origin_sparse_zmax_map", self.ctx.handle, *self._args(), kp, Ny * Nx, m.p)
        return m.to_host().astype(np.float64)

    def where_above(self, threshold, aux=None, cap=1 << 20):
        """``z, y, x = np.where(cube > threshold)`` in NumPy's order with the values (and those of
        the uint8 device cube ``aux``): kernels.where_above's contract."""
        self.counts()
        Nz, Ny, Nx = self.shape
        if aux is not None and (aux.dtype != np.uint8 or tuple(aux.shape) != self.shape):
            raise ValueError("aux must be a uint8 device cube of the same shape")
        if not threshold >= 0:      # zeros qualify (or NaN: nothing does): the dense cube answers
            from . import kernels
            return kernels.where_above(self.ctx, self.dense(), threshold, aux=aux, cap=cap)
        idx, val, ax = self._above(threshold, aux, cap)
        z, rem = np.divmod(idx, Ny * Nx)
        y, x = np.divmod(rem, Nx)
        res = dict(z=z, y=y, x=x, value=val.astype(np.float64))
        if aux is not None:
            res["aux"] = ax
        return res
