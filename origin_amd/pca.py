"""Greedy PCA driver: the control flow of ``Compute_GreedyPCA`` (reference
muse_origin/lib_origin.py:848-954) over a cube that stays resident in HBM.

All areas advance in lock step (they are independent, lib_origin.py:806-819), so every
kernel launch is batched over the areas that are still iterating.  Per iteration the dense
work -- background mean, nuisance gather + projection, Gram matrix (float64 MFMA), u = Xp v,
deflation of the whole area and the new O2 test -- runs on the GPU; the host keeps only the
threshold logic on one float64 per spaxel (which spaxels are nuisances, which background
spectra feed the mean: argsort on <= 1e4 numbers per area) and the leading eigenvector of
the small Gram matrices.
"""
import ctypes as C

import numpy as np

from . import _capi
from .device import DeviceArray


class _Pool:
    """Grow-only device buffer."""

    def __init__(self, ctx, dtype):
        self.ctx, self.dtype, self.arr = ctx, np.dtype(dtype), None

    def get(self, n):
        n = max(int(n), 1)
        if self.arr is None or self.arr.size < n:
            if self.arr is not None:
                self.arr.free()
            self.arr = self.ctx.empty((int(n * 1.25) + 16,), self.dtype)
        return self.arr


def leading_eigvec(G, n):
    """Leading eigenvector of the symmetric PSD matrix G[:n,:n] (float64), converged to
    machine precision (the reference asks ARPACK for tol=0, lib_origin.py:940)."""
    A = G[:n, :n]
    if n <= 96:
        w, V = np.linalg.eigh(A)
        return V[:, -1]
    from scipy.sparse.linalg import eigsh
    try:
        w, V = eigsh(A, k=1, which="LA", tol=0, v0=np.ones(n), ncv=min(n, 32))
        return V[:, 0]
    except Exception:  # no convergence: dense solve
        w, V = np.linalg.eigh(A)
        return V[:, -1]


class GreedyPCA:
    """Runs the greedy PCA of all areas of a (tile of a) cube in place on the device."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.pools = {k: _Pool(ctx, dt) for k, dt in dict(
            i32=np.int32, i64=np.int64, Xp=np.float64, G=np.float64, c=np.float64,
            v=np.float64, b=np.float64, u=np.float64).items()}
        self.trace = []

    def run(self, F, area_spx, tests, thresholds, Noise_population=50, itermax=100):
        """F: DeviceArray (Nz, Ny, Nx) float32, updated in place (cube_faint).
        area_spx: per area, int32 flat spaxel indices in C order (``areamap == i``).
        tests: per area float64 O2 values in the same order (``testO2``).
        Returns (mapO2 per area, nstop)."""
        ctx = self.ctx
        Nz = F.shape[0]
        S = F.size // Nz
        na_all = len(area_spx)
        area_spx = [np.ascontiguousarray(s, dtype=np.int32) for s in area_spx]
        tests = [np.array(t, dtype=np.float64) for t in tests]
        mapO2 = [np.zeros(len(s)) for s in area_spx]
        nbiter = [0] * na_all
        active = [len(s) > 0 for s in area_spx]
        nstop = 0
        d_test = ctx.empty((S,), np.float64)
        h_test = np.zeros(S, dtype=np.float64)
        self.trace = []
        first = True
        while any(active):
            if not first:
                d_test.to_host(h_test)
                for a in range(na_all):
                    if active[a]:
                        tests[a] = h_test[area_spx[a]]
            first = False
            work = []
            for a in range(na_all):
                if not active[a]:
                    continue
                test, thr = tests[a], thresholds[a]
                pypx = np.where(test > thr)[0]                       # lib :889, :949
                if len(pypx) == 0:                                   # while len(pypx) > 0
                    active[a] = False
                    continue
                nbiter[a] += 1
                mapO2[a][pypx] += 1                                  # :901
                if nbiter[a] > itermax:                              # :902-905
                    nstop += 1
                    active[a] = False
                    continue
                test_v = test[test > 0]                              # :908-909
                nind = np.where(test_v <= thr)[0]                    # :910 (indices into the
                sortind = np.argsort(test_v[nind])                   #  filtered vector, used on
                nb = 1 + int(len(nind) / Noise_population)           #  unfiltered columns: kept)
                bg = nind[sortind[:nb]]                              # :917
                if len(pypx) == 1:                                   # :927-928  break
                    active[a] = False
                    continue
                work.append((a, pypx, bg))
            if not work:
                break
            self.trace.append([(a, len(p), len(g)) for a, p, g in work])
            self._iterate(F, Nz, S, area_spx, work, d_test)
        return mapO2, nstop

    # ------------------------------------------------------------------
    def _iterate(self, F, Nz, S, area_spx, work, d_test):
        ctx = self.ctx
        na = len(work)
        n = np.array([len(p) for _, p, _ in work], dtype=np.int32)
        ld = ((n + 15) // 16 * 16).astype(np.int32)
        nuis = np.concatenate([area_spx[a][p] for a, p, _ in work]).astype(np.int32)
        bgl = np.concatenate([area_spx[a][g] for a, _, g in work]).astype(np.int32)
        spx = np.concatenate([area_spx[a] for a, _, _ in work]).astype(np.int32)

        def offsets(lengths):
            o = np.zeros(len(lengths) + 1, dtype=np.int64)
            o[1:] = np.cumsum(np.asarray(lengths, dtype=np.int64))
            return o

        nuis_off = offsets(n)
        bg_off = offsets([len(g) for _, _, g in work])
        spx_off = offsets([len(area_spx[a]) for a, _, _ in work])
        xp_off = offsets(ld.astype(np.int64) * Nz)
        c_off = offsets(ld)
        g_off = offsets(ld.astype(np.int64) ** 2)
        ti, tj, ta = [], [], []
        for k in range(na):
            T = (int(ld[k]) + 31) // 32
            iu, ju = np.triu_indices(T)
            ti.append(iu)
            tj.append(ju)
            ta.append(np.full(len(iu), k))
        ti = np.concatenate(ti).astype(np.int32)
        tj = np.concatenate(tj).astype(np.int32)
        ta = np.concatenate(ta).astype(np.int32)

        # one upload for all int32 / int64 descriptors
        i32_parts = [nuis, bgl, spx, n, ld, ti, tj, ta]
        i64_parts = [nuis_off, bg_off, spx_off, xp_off, c_off, g_off]
        i32_all = np.concatenate(i32_parts)
        i64_all = np.concatenate(i64_parts)
        d_i32 = self.pools["i32"].get(i32_all.size)
        d_i64 = self.pools["i64"].get(i64_all.size)
        _capi.call("origin_h2d", ctx.handle, d_i32.p, i32_all.ctypes.data_as(C.c_void_p),
                   i32_all.nbytes)
        _capi.call("origin_h2d", ctx.handle, d_i64.p, i64_all.ctypes.data_as(C.c_void_p),
                   i64_all.nbytes)

        def sub(base, parts, itemsize):
            out, o = [], 0
            for p in parts:
                out.append(C.c_void_p(base.ptr + o * itemsize))
                o += p.size
            return out

        (p_nuis, p_bg, p_spx, p_n, p_ld, p_ti, p_tj, p_ta) = sub(d_i32, i32_parts, 4)
        (p_nuis_off, p_bg_off, p_spx_off, p_xp_off, p_c_off, p_g_off) = sub(d_i64, i64_parts, 8)

        g_total = int(g_off[-1])
        Xp = self.pools["Xp"].get(int(xp_off[-1]))
        G = self.pools["G"].get(g_total)
        cvec = self.pools["c"].get(int(c_off[-1]))
        vvec = self.pools["v"].get(int(c_off[-1]))
        b = self.pools["b"].get(na * Nz)
        u = self.pools["u"].get(na * Nz)

        h = ctx.handle
        _capi.call("origin_pca_bmean", h, F.p, Nz, S, p_bg, p_bg_off, na, b.p)
        _capi.call("origin_pca_build_xp", h, F.p, Nz, S, p_nuis, p_nuis_off, na, int(ld.max()),
                   b.p, Xp.p, p_xp_off, p_ld, cvec.p, p_c_off)
        _capi.call("origin_pca_gram", h, Xp.p, p_xp_off, p_ld, Nz, int(len(ti)), p_ti, p_tj, p_ta,
                   g_total, G.p, p_g_off)
        # leading eigenvector of each Gram matrix (host, float64, converged)
        hG = np.empty(g_total, dtype=np.float64)
        _capi.call("origin_d2h", h, hG.ctypes.data_as(C.c_void_p), G.p, hG.nbytes)
        hv = np.zeros(int(c_off[-1]), dtype=np.float64)
        for k in range(na):
            Gk = hG[g_off[k]: g_off[k + 1]].reshape(int(ld[k]), int(ld[k]))
            hv[c_off[k]: c_off[k] + n[k]] = leading_eigvec(Gk, int(n[k]))
        _capi.call("origin_h2d", h, vvec.p, hv.ctypes.data_as(C.c_void_p), hv.nbytes)
        _capi.call("origin_pca_uvec", h, Xp.p, p_xp_off, p_ld, p_n, na, Nz, vvec.p, p_c_off, u.p)
        _capi.call("origin_pca_deflate", h, F.p, Nz, S, p_spx, p_spx_off, na, int(spx.size),
                   int(max(len(area_spx[a]) for a, _, _ in work)), u.p, d_test.p)
