"""Several GPUs behind the Step seam: ONE process, one context and one worker thread per device.

The reference's session object builds its steps in one process and calls them one after the
other (``ORIGIN.__init__``, origin.py:193-208; ``Step.__call__``, steps.py:242-281), so a
drop-in cannot be an SPMD program: the user calls ``orig.step04_compute_greedy_PCA()`` once, in
one interpreter.  ``DeviceGroup`` therefore keeps N contexts (one per entry of ``devices``; the
same device may appear twice: two contexts on one card, strips staged through the host) and N
threads -- ctypes releases the GIL for the duration of a library call, and every entry point of
liborigin_hip.so selects its context's device and keeps its error text per thread, so the ranks
run side by side.  Their host-side meeting point is ``ThreadGroup`` (the five methods of
``rendezvous.HostGroup`` on a ``threading.Barrier``); cubes travel over RCCL, each context with a
communicator of its own (``TileComm.attach`` from the rank's thread: ``ncclCommInitRank`` of N
ranks in one process), or through the host group when devices repeat.

``TiledSession`` is what the run bodies of ``origin_amd.steps`` call when a session has more
than one device:

* step 1 (Preprocessing, steps.py:431-465) knows no areas yet (CreateAreas is step 2): the field
  is cut into bands of whole rows (``OwnerTiling.row_bands``); one all-reduce of the per-channel
  sums, a one-spaxel halo for the 3x3x3 local maxima of ``cube_std``;
* step 4 (ComputeGreedyPCA, steps.py:681-704) has the area map: areas go to ranks as wholes
  (``OwnerTiling.from_areamap``), ``cube_std`` is re-distributed from the row bands to the
  bounding boxes of the areas (lists of spaxel columns), every rank runs its areas and writes
  ``cube_faint`` straight into its halo-extended box;
* step 5 (ComputeTGLR, steps.py:770-802) exchanges the halo and runs the GLR and the local
  maxima per box (``multigpu.TiledGLR``).

Cubes stay on their devices between the steps as ``TiledCube`` s -- the parts, their boxes and
ownership maps; the host array the reference's interface promises is stitched together when
somebody reads it (``._data`` of the DataObj).  The reductions of the later steps (purity
curves, thresholding) run per part and combine on the host.
"""
import queue
import threading

import numpy as np

from . import _capi, kernels, multigpu, pipeline, sparse
from .device import Context, DeviceArray
from .pca import GreedyPCA


# ------------------------------------------------------------------------------- host group
class _Shared:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}


class ThreadGroup:
    """``rendezvous.HostGroup`` for ranks that are threads of one process: broadcast, allreduce
    (sum | max | min, reduced in rank order: deterministic), barrier, exchange, close.  A rank
    that fails calls ``abort()``: the others leave their barrier with ``BrokenBarrierError``
    instead of waiting for it."""

    TIMEOUT = 300.0

    def __init__(self, shared, rank):
        self._s, self.rank, self.world = shared, rank, shared.world

    def _wait(self):
        self._s.barrier.wait(self.TIMEOUT)

    def broadcast(self, data, src=0):
        if self.rank == src:
            self._s.slots[src] = bytes(data)
        self._wait()
        out = self._s.slots[src]
        self._wait()
        return out

    def allreduce(self, arr, op="sum"):
        a = np.ascontiguousarray(arr, dtype=np.float64).copy()
        if self.world == 1:
            return a
        self._s.slots[self.rank] = a
        self._wait()
        f = {"sum": np.add, "max": np.maximum, "min": np.minimum}[op]
        out = self._s.slots[0].copy()
        for r in range(1, self.world):
            f(out, self._s.slots[r], out=out)
        self._wait()
        return out

    def barrier(self):
        self._wait()

    def exchange(self, sends, recvs):
        for peer, arr in sends:
            self._s.boxes[(self.rank, peer)].put(np.array(arr, copy=True))
        for peer, out in recvs:
            box, waited = self._s.boxes[(peer, self.rank)], 0.0
            while True:
                try:
                    got = box.get(timeout=0.2)
                    break
                except queue.Empty:
                    waited += 0.2
                    if self._s.barrier.broken or waited > self.TIMEOUT:   # the peer has failed
                        raise threading.BrokenBarrierError() from None
            out[...] = got.reshape(out.shape)

    def abort(self):
        self._s.barrier.abort()

    def close(self):
        pass


# ------------------------------------------------------------------------------- devices
class DeviceGroup:
    """N contexts, N worker threads, N ``TileComm`` s.  ``devices``: device ordinals, one per
    rank; ``backend``: "rccl" (default when the ordinals differ), "host" (default when a device
    appears twice: RCCL refuses two ranks on one card)."""

    def __init__(self, devices, backend=None):
        self.devices = [int(d) for d in devices]
        self.world = len(self.devices)
        if self.world < 1:
            raise ValueError("no device")
        if backend is None:
            backend = "rccl" if len(set(self.devices)) == self.world else "host"
        self.backend = backend
        shared = _Shared(self.world)
        self.ctxs = [Context(d) for d in self.devices]
        self.comms = [multigpu.TileComm(r, self.world, self.devices[r], backend,
                                        group=ThreadGroup(shared, r)) for r in range(self.world)]
        self._jobs = [queue.Queue() for _ in range(self.world)]
        self._threads = [threading.Thread(target=self._loop, args=(r,), daemon=True,
                                          name=f"origin-rank{r}") for r in range(self.world)]
        for t in self._threads:
            t.start()
        self._attached = False

    def _loop(self, rank):
        while True:
            job = self._jobs[rank].get()
            if job is None:
                return
            fn, done, out = job
            try:
                out[rank] = (True, fn(rank))
            except BaseException as exc:   # noqa: BLE001 -- re-raised by run()
                out[rank] = (False, exc)
                self.comms[rank].group.abort()   # nobody waits for this rank any more
            done.release()

    def run(self, fn):
        """``fn(rank)`` on every rank's thread at once; the list of results in rank order.  The
        first rank that raised is re-raised (the others' barriers are broken, not left
        waiting)."""
        done, out = threading.Semaphore(0), [None] * self.world
        for q in self._jobs:
            q.put((fn, done, out))
        for _ in range(self.world):
            done.acquire()
        errs = [v for ok, v in out if not ok]
        if errs:
            # every thread is back: the group can be used again
            sh = self.comms[0].group._s
            sh.barrier.reset()
            for q in sh.boxes.values():
                while not q.empty():
                    q.get_nowait()
            first = [e for e in errs if not isinstance(e, threading.BrokenBarrierError)]
            raise (first or errs)[0]
        return [v for _, v in out]

    def attach(self):
        """RCCL communicators (collective, once); a no-op for the host backend."""
        if not self._attached:
            self.run(lambda r: self.comms[r].attach(self.ctxs[r]))
            self._attached = True

    def close(self):
        for q in self._jobs:
            q.put(None)
        for c in self.comms:
            try:
                c.close()
            except Exception:   # noqa: BLE001
                pass


# ------------------------------------------------------------------------------- cubes
class TiledCube:
    """A cube of the field that lives in pieces on the ranks' devices.  ``parts[r]`` =
    (DeviceArray a, (by, bx), (y0, y1, x0, x1), owned): the box of ``a`` that starts at (by, bx)
    covers the field window [y0:y1, x0:x1]; ``owned`` (bool (y1-y0, x1-x0) or None = all) marks
    the spaxels of that window that are this part's.  Reads like a DeviceArray where the Step
    seam needs one (``shape``, ``dtype``, ``to_host``, ``to_host_f64``)."""

    def __init__(self, group, shape, dtype, parts):
        self.group, self.shape, self.dtype, self.parts = group, tuple(shape), np.dtype(dtype), parts
        self.size = int(np.prod(self.shape))

    def _gather(self, out, convert):
        three = len(self.shape) == 3

        def one(rank):
            a, (by, bx), (y0, y1, x0, x1), owned = self.parts[rank]
            ny, nx = y1 - y0, x1 - x0
            if three:
                # the part's box as one contiguous block: crop on the device (one strided copy
                # kernel), then one large copy through pinned staging -- a strided device-to-host
                # copy is one hipMemcpy2D per channel
                if hasattr(a, "entries"):      # sparse.SparseCube: zeros + its entries, on the host
                    blk = a.to_host()[:, by:by + ny, bx:bx + nx]
                elif (by, bx) == (0, 0) and a.shape[1:] == (ny, nx):
                    blk = a.to_host()
                else:
                    ctx = self.group.ctxs[rank]
                    tmp = ctx.empty((a.shape[0], ny, nx), a.dtype)
                    multigpu._copy_box(ctx, tmp, tmp.shape, (0, 0, 0), a, a.shape, (0, by, bx),
                                       (a.shape[0], ny, nx))
                    blk = tmp.to_host()
                    tmp.free()
                dst = out[:, y0:y1, x0:x1]
                if owned is None:
                    dst[...] = convert(blk)
                else:
                    dst[:, owned] = convert(blk[:, owned])
            else:
                blk = a.to_host()[by:by + ny, bx:bx + nx]
                if owned is None:
                    out[y0:y1, x0:x1] = convert(blk)
                else:
                    out[y0:y1, x0:x1][owned] = convert(blk[owned])
        self.group.run(one)
        return out

    def to_host(self, out=None):
        out = np.empty(self.shape, self.dtype) if out is None else out
        return self._gather(out, lambda b: b)

    def to_host_f64(self, out=None):
        out = np.empty(self.shape, np.float64) if out is None else out
        return self._gather(out, lambda b: b.astype(np.float64))

    def gathered(self, ctx):
        """The whole cube as one DeviceArray on ``ctx`` (through the host: dump / hand-over to
        single-device code)."""
        return ctx.to_device(self.to_host())

    # -- reductions of the later steps, per part ----------------------------------------
    def _keep(self, rank, keep):
        """uint8 keep map over the spaxels of part ``rank``'s ARRAY: 1 where the spaxel is inside
        the part's box, owned, and kept by the caller's field map."""
        a, (by, bx), (y0, y1, x0, x1), owned = self.parts[rank]
        k = np.zeros(a.shape[1:], np.uint8)
        w = np.ones((y1 - y0, x1 - x0), bool) if owned is None else owned.copy()
        if keep is not None:
            w &= np.asarray(keep).reshape(self.shape[1:])[y0:y1, x0:x1] != 0
        k[by:by + y1 - y0, bx:bx + x1 - x0] = w
        return k

    def zmax_map(self, keep=None):
        """max over z per spaxel -> host float64 (Ny, Nx); spaxels with keep == 0 count as 0
        (kernels.zmax_map)."""
        out = np.zeros(self.shape[1:], np.float64)

        def one(rank):
            a, (by, bx), (y0, y1, x0, x1), owned = self.parts[rank]
            ctx = self.group.ctxs[rank]
            full = a.zmax_map(None) if hasattr(a, "entries") else kernels.zmax_map(ctx, a, None)
            m = full[by:by + y1 - y0, bx:bx + x1 - x0]
            if keep is not None:   # max_z (cube * keep) = keep ? max_z cube : 0
                m = np.where(np.asarray(keep).reshape(self.shape[1:])[y0:y1, x0:x1] != 0, m, 0.0)
            if owned is None:
                out[y0:y1, x0:x1] = m
            else:
                out[y0:y1, x0:x1][owned] = m[owned]
        self.group.run(one)
        return out

    def count_above(self, thresholds, keep=None):
        """counts[t] = #{kept voxels > thresholds[t]} over the whole field (int64)."""
        def one(rank):
            ctx = self.group.ctxs[rank]
            k = ctx.to_device(self._keep(rank, keep).reshape(-1))
            a = self.parts[rank][0]
            if hasattr(a, "entries"):
                return a.count_above(thresholds, k)
            return kernels.count_above(ctx, a, thresholds, k)
        return np.sum(self.group.run(one), axis=0)

    def where_above(self, threshold, aux=None):
        """``np.where(cube > threshold)`` of the stitched cube, in NumPy's order, with the values
        (and those of the uint8 TiledCube ``aux`` with the same parts layout)."""
        def one(rank):
            a, (by, bx), (y0, y1, x0, x1), owned = self.parts[rank]
            ctx = self.group.ctxs[rank]
            ax = None if aux is None else aux.parts[rank][0]
            w = (a.where_above(threshold, aux=ax) if hasattr(a, "entries")
                 else kernels.where_above(ctx, a, threshold, aux=ax))
            yy, xx = w["y"] - by, w["x"] - bx
            ok = (yy >= 0) & (yy < y1 - y0) & (xx >= 0) & (xx < x1 - x0)
            if owned is not None:
                ok[ok] &= owned[yy[ok], xx[ok]]
            res = dict(z=w["z"][ok], y=yy[ok] + y0, x=xx[ok] + x0, value=w["value"][ok])
            if aux is not None:
                res["aux"] = w["aux"][ok]
            return res
        parts = self.group.run(one)
        cat = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        order = np.lexsort((cat["x"], cat["y"], cat["z"]))
        return {k: v[order] for k, v in cat.items()}


# ------------------------------------------------------------------------------- moves
def redistribute(ctx, comm, src_tiling, dst_tiling, rank, src, dst, bufs):
    """Columns of a cube from one partition of the field to another: ``src`` is this rank's
    (Nz, ny, nx) array over its bounding box in ``src_tiling`` (its owned spaxels valid), ``dst``
    the array over its box in ``dst_tiling``; afterwards the spaxels this rank owns in
    ``dst_tiling`` hold the cube's values, wherever they lived before.  Lists of spaxel columns
    through the same gather / exchange / scatter as the halo of an ``OwnerTiling``."""
    Nz = src.shape[0]
    st, dt = src_tiling.tile(rank), dst_tiling.tile(rank)
    s_nx, d_nx = st.x1 - st.x0, dt.x1 - dt.x0
    mine_src = src_tiling.owner == rank
    mine_dst = dst_tiling.owner == rank
    es = src.dtype.itemsize
    sends, recvs, local = [], [], None
    for other in range(src_tiling.world):
        ys, xs = np.nonzero(mine_src & (dst_tiling.owner == other))
        if len(ys):
            ix = ((ys - st.y0) * s_nx + (xs - st.x0)).astype(np.int32)
            if other == rank:
                local = [ix]
            else:
                sends.append((other, ix))
        ys, xs = np.nonzero(mine_dst & (src_tiling.owner == other))
        if len(ys):
            ix = ((ys - dt.y0) * d_nx + (xs - dt.x0)).astype(np.int32)
            if other == rank:
                local.append(ix)
            else:
                recvs.append((other, ix))

    def pack(ix):
        d_ix = ctx.to_device(ix)
        b = ctx.empty((Nz, ix.size), src.dtype)
        _capi.call("origin_gather_columns", ctx.handle, src.p, Nz, src.size // Nz, d_ix.p, ix.size,
                   es, b.p)
        return b

    def unpack(ix, b):
        d_ix = ctx.to_device(ix)
        _capi.call("origin_scatter_columns", ctx.handle, dst.p, Nz, dst.size // Nz, d_ix.p, ix.size,
                   es, b.p)
        bufs.append((d_ix, b))     # (alive until the stream has run the scatter)
    if local is not None:
        unpack(local[1], pack(local[0]))
    out = [(peer, pack(ix)) for peer, ix in sends]
    inn = [(peer, ctx.empty((Nz, ix.size), src.dtype)) for peer, ix in recvs]
    comm.exchange(ctx, out, inn)
    for (peer, ix), (_, b) in zip(recvs, inn):
        unpack(ix, b)
    bufs.extend(b for _, b in out)
    return dst


# ------------------------------------------------------------------------------- session
class TiledSession:
    """The hot steps of one session spread over the devices of a ``DeviceGroup``; created by
    ``origin_amd.steps`` for sessions with more than one device."""

    def __init__(self, group):
        self.group = group
        self.world = group.world
        self.p1 = None            # row bands (steps 1-3)
        self.p2 = None            # areas (steps 4-5)
        self.rk = [dict() for _ in range(self.world)]   # per-rank device state

    # -- helpers -------------------------------------------------------------------
    def _cube(self, name, shape, dtype, tiling, halo_box=False):
        """TiledCube over the per-rank arrays ``self.rk[r][name]``: tile-shaped (halo_box False) or
        extended-box-shaped."""
        parts = []
        for r in range(self.world):
            t = tiling.tile(r)
            (_, _, _, _), (top, _, left, _) = tiling.extended(r)
            parts.append((self.rk[r][name], (top, left) if halo_box else (0, 0),
                          (t.y0, t.y1, t.x0, t.x1), tiling.owned_tile(r)))
        return TiledCube(self.group, shape, dtype, parts)

    def _stitch_image(self, tiling, tiles, dtype=np.float64):
        out = np.zeros((tiling.Ny, tiling.Nx), dtype)
        for r, img in enumerate(tiles):
            t = tiling.tile(r)
            own = tiling.owned_tile(r)
            out[t.y0:t.y1, t.x0:t.x1][own] = np.asarray(img).reshape(own.shape)[own]
        return out

    def distribute(self, host, tiling, name, dtype=np.float32):
        """A host (or TiledCube / DeviceArray) cube into per-rank tiles of ``tiling`` (the boxes;
        what other ranks own inside a box is there too, harmlessly)."""
        if isinstance(host, TiledCube) or isinstance(host, DeviceArray):
            host = host.to_host()
        host = np.asarray(getattr(host, "_data", host))

        def one(r):
            t = tiling.tile(r)
            self.rk[r][name] = self.group.ctxs[r].to_device(
                np.ascontiguousarray(host[:, t.y0:t.y1, t.x0:t.x1]), dtype)
        self.group.run(one)

    # -- step 1 --------------------------------------------------------------------
    def preprocess(self, raw, var, mask, dct_order=10, dct_approx=False, local_max_size=3):
        """``Preprocessing.run`` dense part (steps.py:431-465) on row bands.  raw / var / mask:
        host arrays of the whole field.  Returns the TiledCubes cube_std, cont_dct,
        cube_std_local_max / _min and the host images ima_std, ima_dct, o2 (O2 map of cube_std),
        cont_o2 (mean_z cont_dct^2)."""
        raw = np.asarray(getattr(raw, "_data", raw))
        var = np.asarray(getattr(var, "_data", var))
        Nz, Ny, Nx = raw.shape
        if mask is None or mask is np.ma.nomask:
            mask = np.zeros(raw.shape, np.uint8)
        mask = np.asarray(getattr(mask, "_data", mask))
        self.shape, self.host_mask = (Nz, Ny, Nx), mask
        halo = int(local_max_size) // 2
        p1 = self.p1 = multigpu.OwnerTiling.row_bands(Ny, Nx, self.world, halo)
        self.group.attach()

        def one(r):
            ctx, comm, st = self.group.ctxs[r], self.group.comms[r], self.rk[r]
            t = p1.tile(r)
            sl = (slice(None), slice(t.y0, t.y1), slice(t.x0, t.x1))
            d_raw = ctx.to_device(np.ascontiguousarray(raw[sl]), np.float32)
            d_var = ctx.to_device(np.ascontiguousarray(var[sl]), np.float32)
            d_mask = ctx.to_device(np.ascontiguousarray(mask[sl]), np.uint8)
            st.update(raw=d_raw, var=d_var, mask=d_mask)
            if comm.device_p2p:
                pre = pipeline.preprocess(ctx, d_raw, d_var, d_mask, dct_order, dct_approx,
                                          allreduce_dev=comm.allreduce_sum_device)
            else:
                pre = pipeline.preprocess(ctx, d_raw, d_var, d_mask, dct_order, dct_approx,
                                          allreduce=comm.allreduce_sum)
            st.update(cube_std=pre["cube_std"], cont_dct=pre["cont_dct"])
            # 3x3x3 local maxima of cube_std (steps.py:453): one spaxel of halo, cube and mask
            strips = {}
            e_std = multigpu.exchange_halo(ctx, comm, p1, r, pre["cube_std"], None, strips)
            e_msk = multigpu.exchange_halo(ctx, comm, p1, r, d_mask, None, strips)
            # (lists of the non-zero voxels where the pass has a sparse form, as on one device)
            lmax, lmin = sparse.local_max(ctx, e_std, e_std, e_msk, local_max_size)
            st.update(std_lmax=lmax, std_lmin=lmin)
            cont_o2 = kernels.o2test(ctx, pre["cont_dct"]).to_host()
            return dict(ima_std=pre["ima_std"].to_host(), ima_dct=pre["ima_dct"].to_host(),
                        o2=pre["o2_host"], cont_o2=cont_o2)
        res = self.group.run(one)
        out = {k: self._stitch_image(p1, [x[k] for x in res]) for k in res[0]}
        out["cube_std"] = self._cube("cube_std", self.shape, np.float32, p1)
        out["cont_dct"] = self._cube("cont_dct", self.shape, np.float32, p1)
        out["cube_std_local_max"] = self._cube("std_lmax", self.shape, np.float32, p1, True)
        out["cube_std_local_min"] = self._cube("std_lmin", self.shape, np.float32, p1, True)
        return out

    # -- step 4 --------------------------------------------------------------------
    def _areas(self, areamap, halo):
        areamap = np.asarray(getattr(areamap, "_data", areamap)).astype(np.int64)
        p2 = self.p2
        if p2 is None or p2.halo != halo or not np.array_equal(self._areamap, areamap):
            p2 = self.p2 = multigpu.OwnerTiling.from_areamap(areamap, self.world, halo)
            self._areamap = areamap
            for st in self.rk:
                for k in ("ext", "glr", "pca", "std2"):
                    st.pop(k, None)
        return p2, areamap

    def greedy_pca(self, cube_std, areamap, nbAreas, thresholds, testO2, halo,
                   Noise_population=50, itermax=100):
        """``ComputeGreedyPCA.run`` (steps.py:681-704): areas to ranks as wholes, cube_std moved
        from its partition to the areas' boxes, every rank's areas in lock step on its device,
        cube_faint written into the halo-extended box the GLR will read.  ``halo``: spaxels the
        GLR (and its local maxima) will look beyond a rank's own.  Returns (cube_faint TiledCube,
        mapO2 (Ny, Nx) float64, nstop)."""
        p2, areamap = self._areas(areamap, halo)
        if not (isinstance(cube_std, TiledCube) and cube_std.group is self.group
                and self.p1 is not None and "cube_std" in self.rk[0]):
            # (a session reloaded from its files: cube_std comes from the host)
            self.shape = tuple(cube_std.shape)
            Nz, Ny, Nx = self.shape
            self.p1 = multigpu.OwnerTiling.row_bands(Ny, Nx, self.world, 1)
            self.distribute(cube_std, self.p1, "cube_std")
        p1 = self.p1
        Nz, Ny, Nx = self.shape
        self.group.attach()

        def one(r):
            ctx, comm, st = self.group.ctxs[r], self.group.comms[r], self.rk[r]
            t = p2.tile(r)
            ny, nx = t.y1 - t.y0, t.x1 - t.x0
            keep = []
            std2 = st.get("std2")
            if std2 is None:
                std2 = st["std2"] = ctx.zeros((Nz, ny, nx), np.float32)
            redistribute(ctx, comm, p1, p2, r, st["cube_std"], std2, keep)
            owned = p2.owned_tile(r)
            amap = np.where(owned, areamap[t.y0:t.y1, t.x0:t.x1], 0)
            labels = np.unique(amap[amap > 0])
            lmap = np.where(amap > 0, np.searchsorted(labels, amap) + 1, 0)
            spx = pipeline.area_lists(lmap, len(labels))
            (ey0, ey1, ex0, ex1), (top, _, left, _) = p2.extended(r)
            ext = st.get("ext")
            if ext is None:
                ext = st["ext"] = ctx.zeros((Nz, ey1 - ey0, ex1 - ex0), np.float32)
            drv = st.setdefault("pca", GreedyPCA(ctx))
            _, mapO2, nstop, _ = pipeline.greedy_pca(
                ctx, std2, lmap, len(labels), [thresholds[l - 1] for l in labels],
                [testO2[l - 1] for l in labels], Noise_population, itermax, spx=spx, driver=drv,
                into=(ext, top, left))
            ctx.sync()
            del keep
            return mapO2, nstop
        res = self.group.run(one)
        mapO2 = self._stitch_image(p2, [m for m, _ in res])
        faint = self._cube("ext", self.shape, np.float32, p2, True)
        return faint, mapO2, int(sum(n for _, n in res))

    # -- step 5 --------------------------------------------------------------------
    def tglr(self, cube_faint, areamap, PSF, wfields, profiles, size=3, pcut=1e-8, pmeansub=True):
        """``ComputeTGLR.run`` dense part (steps.py:770-802) on the areas' boxes: halo exchange,
        GLR, mask glue, maps, local maxima.  Returns the TiledCubes correl, correl_min, profile,
        local_max, local_min and the host maps maxmap, minmap."""
        psf0 = PSF[0] if isinstance(PSF, (list, tuple)) else PSF
        halo = int(np.asarray(psf0).shape[-1]) // 2 + int(size) // 2
        p2, areamap = self._areas(areamap, halo)
        Nz, Ny, Nx = self.shape = tuple(cube_faint.shape)
        mine = (isinstance(cube_faint, TiledCube) and cube_faint.group is self.group
                and all("ext" in st and cube_faint.parts[r][0] is st["ext"]
                        for r, st in enumerate(self.rk)))
        if not mine:   # cube_faint from elsewhere (reloaded session): into the boxes' interiors
            host = np.asarray(getattr(cube_faint, "_data", None) if hasattr(cube_faint, "_data")
                              else cube_faint.to_host())
        mask = getattr(self, "host_mask", None)
        self.group.attach()

        def one(r):
            ctx, comm, st = self.group.ctxs[r], self.group.comms[r], self.rk[r]
            (ey0, ey1, ex0, ex1), (top, _, left, _) = p2.extended(r)
            t = p2.tile(r)
            if not mine:
                st["ext"] = ctx.to_device(np.ascontiguousarray(host[:, ey0:ey1, ex0:ex1]),
                                          np.float32)
            old = st.pop("glr", None)
            if old is not None:
                old.plan.close()
            glr = st["glr"] = multigpu.TiledGLR(ctx, comm, p2, r, Nz, PSF, profiles, pcut,
                                                pmeansub, weights=wfields, ext=st["ext"])
            m = None
            if mask is not None:
                # the true mask of the whole extended box, straight from the host's copy
                m = ctx.to_device(np.ascontiguousarray(mask[:, ey0:ey1, ex0:ex1]), np.uint8)
                glr.set_ext_mask(m)
            lm_form = "sparse" if (int(size) == 3 and sparse.plan(ctx, glr.eshape)[0] > 0) else True
            o = glr.run(None, m, None, None, None, local_max=lm_form, size=size)
            st.update(correl=o["correl"], correl_min=o["correl_min"], profile=o["profile"],
                      lmax=o["local_max"], lmin=o["local_min"])
            ctx.sync()
            return o["maxmap"].to_host(), o["minmap"].to_host()
        res = self.group.run(one)
        out = dict(maxmap=self._stitch_image(p2, [a for a, _ in res]),
                   minmap=self._stitch_image(p2, [b for _, b in res]))
        for name, key, dt in (("correl", "correl", np.float32), ("correl_min", "correl_min",
                                                                 np.float32),
                              ("profile", "profile", np.uint8), ("local_max", "lmax", np.float32),
                              ("local_min", "lmin", np.float32)):
            out[name] = self._cube(key, self.shape, dt, p2, True)
        return out

    def close(self):
        for st in self.rk:
            g = st.pop("glr", None)
            if g is not None:
                try:
                    g.plan.close()
                except Exception:   # noqa: BLE001
                    pass
            st.clear()
        self.group.close()
