"""Greedy PCA driver: ``Compute_GreedyPCA`` / ``Compute_GreedyPCA_area`` (reference
muse_origin/lib_origin.py:769-954) over a cube that stays resident in HBM.

The whole loop lives in liborigin_hip.so (``origin_pca_run``, csrc/pca.hip): all areas
advance in lock step, nuisance / background selection, Gram matrix (float64 MFMA), leading
eigenvector (repeated squaring / Lanczos), deflation and the O2 test run on the device; this module only marshals the
area lists and thresholds.
"""
import ctypes as C

import numpy as np

from . import _capi


class GreedyPCA:
    """Runs the greedy PCA of all areas of a (tile of a) cube in place on the device."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.trace = []      # per lock-step iteration: (areas iterating, nuisance spaxels)
        self.iterations = 0

    def prepare(self, area_spx, S):
        """Upload the area lists once; they are reused by every later ``run``."""
        lens = np.array([len(s) for s in area_spx], dtype=np.int64)
        off = np.zeros(len(area_spx) + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        spx = (np.concatenate([np.asarray(s, dtype=np.int32) for s in area_spx])
               if off[-1] else np.zeros(0, np.int32))
        if spx.size and (spx.min() < 0 or spx.max() >= S):
            raise ValueError("spaxel index outside the cube")
        self._area_spx = area_spx
        self._off = off
        self._d_spx = self.ctx.to_device(spx if spx.size else np.zeros(1, np.int32))
        self._S = S
        return self

    def run(self, F, area_spx, tests, thresholds, Noise_population=50, itermax=100,
            test_map=None, want_map=True, src=None, into=None):
        """F: DeviceArray (Nz, Ny, Nx) float32 receiving cube_faint; ``src`` (cube_std) is
        read instead of F when given (out of place), else F is updated in place.
        ``into``: None, or ``(ext, top, left)`` -- cube_faint goes into the box of the larger
        DeviceArray ``ext`` (Nz, Ny_e, Nx_e) that starts at row ``top``, column ``left``
        (origin_pca_run_into; the tiled path's halo-extended tile); F is then None and ``src``
        the (Nz, Ny, Nx) input.
        area_spx: per area, int32 flat spaxel indices in C order (``areamap == i``).
        tests: per area float64 O2 values in the same order (``testO2``); alternatively
        ``test_map``: a float64 DeviceArray [Ny*Nx] holding the O2 test of every spaxel (what
        dct_standardize produces), which avoids a host round trip.
        Returns (mapO2 per area as float64 arrays, nstop); ``want_map="full"``: the int32 map
        over all S spaxels instead of the per-area arrays."""
        ctx = self.ctx
        ref = src if into is not None else F
        Nz = ref.shape[0]
        S = ref.size // Nz
        na = len(area_spx)
        if na == 0:
            return [], 0
        if getattr(self, "_area_spx", None) is not area_spx or getattr(self, "_S", None) != S:
            self.prepare(area_spx, S)
        off, d_spx = self._off, self._d_spx
        if test_map is not None:
            d_test = test_map
        else:
            test0 = np.zeros(S, dtype=np.float64)
            for s, t in zip(area_spx, tests):
                t = np.asarray(t, dtype=np.float64).reshape(-1)
                if len(t) != len(s):
                    raise ValueError("testO2 and area size differ")
                test0[np.asarray(s)] = t
            d_test = ctx.to_device(test0)
        thr = np.ascontiguousarray(thresholds, dtype=np.float64)
        d_map = ctx.empty((S,), np.int32)
        nstop, iters = C.c_int(0), C.c_int(0)
        cap = int(itermax) + 2
        trace = np.zeros(2 * cap, dtype=np.int64)
        if into is not None:
            ext, top, left = into
            _, ny_e, nx_e = ext.shape
            nx = ref.shape[2]
            if src is None or ext.shape[0] != Nz or top + ref.shape[1] > ny_e or left + nx > nx_e:
                raise ValueError("`into` needs a separate source cube and a box inside `ext`")
            first = C.c_void_p(ext.ptr + (top * nx_e + left) * 4)
            _capi.call("origin_pca_run_into", ctx.handle, src.p, first, Nz, S, na, d_spx.p,
                       off.ctypes.data_as(C.c_void_p), d_test.p, thr.ctypes.data_as(C.c_void_p),
                       float(Noise_population), int(itermax), d_map.p, C.byref(nstop),
                       C.byref(iters), trace.ctypes.data_as(C.c_void_p), cap, nx, nx_e,
                       ny_e * nx_e)
        else:
            _capi.call("origin_pca_run", ctx.handle, (src.p if src is not None else F.p), F.p, Nz,
                       S, na, d_spx.p,
                       off.ctypes.data_as(C.c_void_p), d_test.p, thr.ctypes.data_as(C.c_void_p),
                       float(Noise_population), int(itermax), d_map.p, C.byref(nstop),
                       C.byref(iters), trace.ctypes.data_as(C.c_void_p), cap)
        self.iterations = iters.value
        self.trace = [(int(trace[2 * i]), int(trace[2 * i + 1]))
                      for i in range(min(iters.value, cap))]
        self.map_dev = d_map
        if not want_map:
            return None, nstop.value
        hmap = d_map.to_host()
        if want_map == "full":  # the map over all spaxels (zero outside the areas), no regrouping
            return hmap, nstop.value
        return [hmap[np.asarray(s)].astype(np.float64) for s in area_spx], nstop.value
