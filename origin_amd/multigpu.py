"""Spatial tiling of the hot path over the GPUs of one node.

One process per GPU.  The field is cut into a gy x gx grid of tiles whose boundaries follow
the PCA area grid, so every area lives on exactly one GPU and the greedy PCA needs no
communication.  Two exchanges remain:

* one all-reduce of 2*Nz float64 (per-channel sum and count of the DCT residual) for the
  ``nanmean`` over the *whole* field in the standardisation (reference steps.py:442);
* one halo exchange of ``cube_faint`` before the GLR: strips of P//2 spaxels from the (up to)
  8 neighbours, done in two phases (x, then y including the fresh x halos, which delivers
  the corners without diagonal messages).  Outside the true field nothing is exchanged:
  the kernels zero-pad there exactly as the reference's ``fftconvolve(..., 'same')`` does.

The GLR then runs on the halo-extended tile as if it were a field of its own: spatial sums
of kept spaxels only touch data inside the extension, and their border class is the class
with respect to the *true* field border because kept spaxels are at least P//2 away from any
internal cut.  Only the interior tile is kept.

Strips and the all-reduce go through RCCL called natively on the library's own stream
(``origin_comm_*``, csrc/comm.hip); the host-side rendezvous that carries the RCCL unique id,
barriers and a few scalars is ``rendezvous.HostGroup`` (plain sockets) -- see ``TileComm``.
"""
import os
from collections import namedtuple

import numpy as np

from . import _capi, kernels

Tile = namedtuple("Tile", "rank ty tx y0 y1 x0 x1")


def _split(n_units, parts):
    """Split n_units into `parts` contiguous groups whose sizes differ by at most one."""
    base, rem = divmod(n_units, parts)
    sizes = [base + (1 if i < rem else 0) for i in range(parts)]
    edges = np.concatenate([[0], np.cumsum(sizes)])
    return edges


def grid_shape(world):
    """gy x gx with gy*gx == world, as square as possible, gx >= gy."""
    gy = int(np.floor(np.sqrt(world)))
    while world % gy:
        gy -= 1
    return gy, world // gy


class Tiling:
    """Tile grid aligned to the area grid (areas are area_size x area_size squares, the last
    one absorbs the remainder, as synth.grid_areamap builds them)."""

    def __init__(self, Ny, Nx, world, area_size=100, halo=12):
        self.Ny, self.Nx, self.world, self.halo = Ny, Nx, world, halo
        self.gy, self.gx = grid_shape(world)
        nay, nax = max(1, Ny // area_size), max(1, Nx // area_size)
        if self.gy > nay or self.gx > nax:
            raise ValueError(f"{world} tiles do not fit a {nay}x{nax} area grid")
        ey = _split(nay, self.gy) * area_size
        ex = _split(nax, self.gx) * area_size
        ey[-1], ex[-1] = Ny, Nx
        self.ey, self.ex = ey, ex

    def tile(self, rank):
        ty, tx = divmod(rank, self.gx)
        return Tile(rank, ty, tx, int(self.ey[ty]), int(self.ey[ty + 1]), int(self.ex[tx]),
                    int(self.ex[tx + 1]))

    def neighbour(self, rank, dy, dx):
        ty, tx = divmod(rank, self.gx)
        ny_, nx_ = ty + dy, tx + dx
        if 0 <= ny_ < self.gy and 0 <= nx_ < self.gx:
            return ny_ * self.gx + nx_
        return None

    def extended(self, rank):
        """Extent of the tile plus its halo, clipped to the field: (y0, y1, x0, x1) and the
        halo widths (top, bottom, left, right) actually present."""
        t, h = self.tile(rank), self.halo
        top = h if t.ty > 0 else 0
        bot = h if t.ty < self.gy - 1 else 0
        left = h if t.tx > 0 else 0
        right = h if t.tx < self.gx - 1 else 0
        return (t.y0 - top, t.y1 + bot, t.x0 - left, t.x1 + right), (top, bot, left, right)


# ------------------------------------------------------------------------------- comm
class TileComm:
    """Exchange layer of the tiled path.

    Host side: a ``rendezvous.HostGroup`` (plain sockets: the RCCL unique id from rank 0,
    barriers, a handful of host scalars) -- or any object with the same five methods
    (``broadcast, allreduce, barrier, exchange, close``; the tests also run a
    gloo adaptor of their own through here).  The package itself has no such dependency.
    The cubes are this library's, so the device side is this library's too:

    * ``backend="rccl"`` (default): a native RCCL communicator on the context's stream
      (``origin_comm_*`` in include/origin_hip.h), created at the first device operation (the
      context exists only then); its unique id travels over the host group.  Strips go GPU to
      GPU over xGMI; nothing is synchronised on the host.  If the communicator cannot be
      created on ANY rank, EVERY rank raises: there is no silent change of transport.
    * ``backend="host"``: strips are staged through host memory and the host group (CPU
      tests; one-GPU rehearsal with several ranks on the same card, which RCCL refuses).
      Only on request.  (``"gloo"``, the name rounds 1-2 used, is accepted as an alias.)
    """

    def __init__(self, rank, world, local_rank, backend=None, group=None):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL
        backend = {None: "rccl", "gloo": "host"}.get(backend, backend)
        if backend not in ("rccl", "host"):
            raise ValueError(f"backend must be 'rccl' or 'host', not {backend!r}")
        self.note = ""
        if group is None:
            from . import rendezvous
            group = rendezvous.HostGroup(rank, world)
        self.group = group
        self._want_rccl = backend == "rccl"
        self._native = None          # origin_comm* once attached
        self._ctx = None
        self.backend = backend

    @property
    def device_p2p(self):
        return self._native is not None

    # -- native communicator -----------------------------------------------------
    def attach(self, ctx):
        """Create the RCCL communicator on ``ctx`` (collective; no-op for the host backend or
        when already attached).  If any rank fails, every rank raises."""
        if not self._want_rccl or self._native is not None:
            return
        import ctypes as C
        err = ""
        # ncclCommInitRank is collective: a rank that cannot even load librccl would leave the
        # others waiting inside it.  Every rank first proves it can reach the library (a
        # throw-away unique id); only if all can is the communicator created.
        probe = 1.0
        try:
            _capi.call("origin_comm_unique_id", C.create_string_buffer(_capi.COMM_ID_BYTES))
        except Exception as exc:  # noqa: BLE001 -- re-raised on every rank below
            probe, err = 0.0, str(exc)
        if self.group.allreduce(np.array([probe]), "min")[0] != 1.0:
            raise _capi.OriginHipError(-5, "RCCL is not reachable on every rank (" +
                                       (err or "failed on another rank") + ")")
        ident = b""
        if self.rank == 0:
            buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
            try:
                _capi.call("origin_comm_unique_id", buf)
                ident = bytes(buf.raw)
            except Exception as exc:  # noqa: BLE001
                err = str(exc)
        ident = self.group.broadcast(ident, src=0)
        handle, ok = C.c_void_p(), 0.0
        if len(ident) == _capi.COMM_ID_BYTES:
            try:
                _capi.call("origin_comm_create", ctx.handle, ident, self.rank, self.world,
                           C.byref(handle))
                ok = 1.0
            except Exception as exc:  # noqa: BLE001
                err = str(exc)
        if self.group.allreduce(np.array([ok]), "min")[0] == 1.0:
            self._native, self._ctx = handle, ctx
            return
        if ok:
            _capi.call("origin_comm_destroy", handle)
        raise _capi.OriginHipError(-5, "the RCCL communicator could not be created on every rank (" +
                                   (err or "failed on another rank") + "); host-staged strips "
                                   "are only used when asked for (backend='host')")

    # -- small host collectives ------------------------------------------------
    def allreduce_sum(self, arr):
        return self.group.allreduce(arr, "sum")

    def allreduce_sum_device(self, ctx, arrays):
        """In-place sum over ranks of float64 DeviceArrays (per-channel sum and count)."""
        self.attach(ctx)
        if self._native is not None:
            for a in arrays:
                _capi.call("origin_comm_allreduce_f64", self._native, a.ptr, a.size)
            return
        host = self.allreduce_sum(np.concatenate([a.to_host().reshape(-1) for a in arrays]))
        o = 0
        for a in arrays:
            a.upload(host[o: o + a.size].reshape(a.shape))
            o += a.size

    def max_float(self, x):
        return float(self.group.allreduce(np.array([float(x)]), "max")[0])

    def barrier(self):
        self.group.barrier()

    def close(self):
        if self._native is not None:
            _capi.call("origin_comm_destroy", self._native)
            self._native = None
        if self.group is not None:
            self.group.close()
            self.group = None

    # -- strip exchange ----------------------------------------------------------
    def exchange(self, ctx, sends, recvs):
        """sends / recvs: lists of (peer_rank, DeviceArray) -- contiguous strips.  Every
        rank posts its receives and sends together (grouped point-to-point)."""
        if not sends and not recvs:
            return
        self.attach(ctx)
        if self._native is not None:
            import ctypes as C

            def pack(items):
                n = len(items)
                return (n, (C.c_int * n)(*[p for p, _ in items]),
                        (C.c_void_p * n)(*[b.ptr for _, b in items]),
                        (C.c_long * n)(*[b.nbytes for _, b in items]))
            ns, sp, sb, sl = pack(sends)
            nr, rp, rb, rl = pack(recvs)
            _capi.call("origin_comm_exchange", self._native, ns, sp, sb, sl, nr, rp, rb, rl)
            return
        host_recv = [(peer, np.empty(buf.shape, np.float32), buf) for peer, buf in recvs]
        self.group.exchange([(peer, buf.to_host()) for peer, buf in sends],
                            [(peer, h) for peer, h, _ in host_recv])
        for _, h, buf in host_recv:
            buf.upload(h)


def init_comm(rank, world, local_rank, backend=None, group=None):
    return TileComm(rank, world, local_rank, backend, group)


# names of rounds 1-2, kept as aliases (deprecated: use TileComm / Tiling)
TorchComm = TileComm


# ------------------------------------------------------------------------------- halo
def _copy_box(ctx, dst, dst_shape, dst_off, src, src_shape, src_off, box):
    """device->device copy of an (nz, ny, nx) box between float32 cubes of the given shapes."""
    nz, ny, nx = box
    es = src.dtype.itemsize
    sp = src.ptr + ((src_off[0] * src_shape[1] + src_off[1]) * src_shape[2] + src_off[2]) * es
    dp = dst.ptr + ((dst_off[0] * dst_shape[1] + dst_off[1]) * dst_shape[2] + dst_off[2]) * es
    import ctypes as C
    _capi.call("origin_copy_box", ctx.handle, 2, C.c_void_p(dp), dst_shape[2],
               dst_shape[1] * dst_shape[2], C.c_void_p(sp), src_shape[2],
               src_shape[1] * src_shape[2], nz, ny, nx, es)


def halo_plan(tiling, rank, ny, nx):
    """The two exchange phases of one rank as plain index boxes (no data):
    [(phase, peer, send_src, recv_dst, (by, bx)), ...] where phase 0 strips are cut from the
    bare tile (y, x offsets in tile coordinates) and phase 1 strips from the extended tile
    (offsets in extended coordinates); recv_dst is always in extended coordinates."""
    (_, _, _, _), (top, bot, left, right) = tiling.extended(rank)
    h = tiling.halo
    nx_e = nx + left + right
    plan = []
    for dx, src_x, dst_x in ((-1, 0, 0), (1, nx - h, left + nx)):
        peer = tiling.neighbour(rank, 0, dx)
        if peer is not None:
            plan.append((0, peer, (0, src_x), (top, dst_x), (ny, h)))
    for dy, src_y, dst_y in ((-1, top, 0), (1, top + ny - h, top + ny)):
        peer = tiling.neighbour(rank, dy, 0)
        if peer is not None:
            plan.append((1, peer, (src_y, 0), (dst_y, 0), (h, nx_e)))
    return plan


def exchange_halo(ctx, comm, tiling, rank, cube, ext=None, bufs=None):
    """Build the halo-extended copy of this rank's (Nz, ny, nx) device tile.  Returns the
    extended DeviceArray (Nz, ny + top + bot, nx + left + right).  ``bufs``: a dict the
    caller keeps between calls so that the strip buffers are allocated once."""
    Nz, ny, nx = cube.shape
    (_, _, _, _), (top, bot, left, right) = tiling.extended(rank)
    eshape = (Nz, ny + top + bot, nx + left + right)
    if ext is None:
        ext = ctx.empty(eshape, np.float32)
    _copy_box(ctx, ext, eshape, (0, top, left), cube, cube.shape, (0, 0, 0), (Nz, ny, nx))
    plan = halo_plan(tiling, rank, ny, nx)
    for phase in (0, 1):
        src, sshape = (cube, cube.shape) if phase == 0 else (ext, eshape)
        sends, recvs, unpack = [], [], []
        for ph, peer, (sy, sx), (dy, dx), (by, bx) in plan:
            if ph != phase:
                continue
            key = (phase, peer, Nz, by, bx)
            if bufs is not None and key in bufs:
                sbuf, rbuf = bufs[key]
            else:
                sbuf = ctx.empty((Nz, by, bx), np.float32)
                rbuf = ctx.empty((Nz, by, bx), np.float32)
                if bufs is not None:
                    bufs[key] = (sbuf, rbuf)
            _copy_box(ctx, sbuf, sbuf.shape, (0, 0, 0), src, sshape, (0, sy, sx), (Nz, by, bx))
            sends.append((peer, sbuf))
            recvs.append((peer, rbuf))
            unpack.append((rbuf, (0, dy, dx), (Nz, by, bx)))
        comm.exchange(ctx, sends, recvs)
        for rbuf, off, box in unpack:
            _copy_box(ctx, ext, eshape, off, rbuf, rbuf.shape, (0, 0, 0), box)
    return ext


def exchange_halo_host(comm, tiling, rank, tile):
    """Same exchange on host ndarrays (float64 allowed) -- used by the CPU tests to check the
    tiling arithmetic against the untiled oracle."""
    Nz, ny, nx = tile.shape
    (_, _, _, _), (top, bot, left, right) = tiling.extended(rank)
    ext = np.zeros((Nz, ny + top + bot, nx + left + right), dtype=tile.dtype)
    ext[:, top: top + ny, left: left + nx] = tile
    plan = halo_plan(tiling, rank, ny, nx)
    for phase in (0, 1):
        src = tile if phase == 0 else ext
        sends, recvs, pending = [], [], []
        for ph, peer, (sy, sx), (dy, dx), (by, bx) in plan:
            if ph != phase:
                continue
            sends.append((peer, np.ascontiguousarray(src[:, sy: sy + by, sx: sx + bx])))
            rbuf = np.empty((Nz, by, bx), dtype=tile.dtype)
            recvs.append((peer, rbuf))
            pending.append((rbuf, dy, dx, by, bx))
        if sends:
            comm.group.exchange(sends, recvs)
        for rbuf, dy, dx, by, bx in pending:
            ext[:, dy: dy + by, dx: dx + bx] = rbuf
    return ext


class TiledGLR:
    """GLR of one tile of a tiled field: halo exchange + plan on the extended tile + crop."""

    def __init__(self, ctx, comm, tiling, rank, Nz, PSF, profiles, pcut=1e-8, pmeansub=True,
                 weights=None):
        """``weights``: None or the mosaic's weight maps (one (Ny, Nx) array per field of the WHOLE
        field, origin.py:600-609): every rank crops them to its halo-extended tile, no exchange."""
        self.ctx, self.comm, self.tiling, self.rank, self.Nz = ctx, comm, tiling, rank, Nz
        # a kept spaxel must see real neighbour data over the whole PSF footprint, and a strip
        # must come from ONE neighbour: halo >= P//2 and every tile at least a halo wide
        psf0 = PSF[0] if isinstance(PSF, (list, tuple)) else PSF
        need = int(np.asarray(psf0).shape[-1]) // 2
        if tiling.halo < need:
            raise ValueError(f"tiling halo {tiling.halo} is smaller than the PSF half width {need}")
        for r in range(tiling.world):
            t_ = tiling.tile(r)
            if min(t_.y1 - t_.y0, t_.x1 - t_.x0) < tiling.halo:
                raise ValueError(f"tile {r} ({t_.y1 - t_.y0}x{t_.x1 - t_.x0}) is narrower than "
                                 f"the halo {tiling.halo}")
        (y0, y1, x0, x1), self.halos = tiling.extended(rank)
        self.eshape = (Nz, y1 - y0, x1 - x0)
        wext = None
        if weights is not None:
            wext = [np.ascontiguousarray(np.asarray(w, dtype=np.float64)[y0:y1, x0:x1])
                    for w in weights]
        self.plan = kernels.GLRPlan(ctx, self.eshape, PSF, wext, profiles, pcut, pmeansub)
        t = tiling.tile(rank)
        self.shape = (Nz, t.y1 - t.y0, t.x1 - t.x0)
        self.ext = ctx.empty(self.eshape, np.float32)
        self.emask = ctx.zeros(self.eshape, np.uint8)
        self.out = dict(correl=ctx.empty(self.eshape, np.float32),
                        correl_min=ctx.empty(self.eshape, np.float32),
                        profile=ctx.empty(self.eshape, np.uint8))
        self.maps = dict(maxmap=ctx.empty(self.shape[1:], np.float32),
                         minmap=ctx.empty(self.shape[1:], np.float32))
        self._mask_set = False
        self._strips = {}

    def run(self, cube_faint, mask, correl, profile, correl_min):
        """Returns the caller's correl / profile / correl_min and the tile's maxmap / minmap.  The
        two maps are DeviceArrays OWNED BY THIS OBJECT and rewritten by the next call: copy them
        (``.to_host()`` / ``.copy()``) to keep them across steps."""
        ctx = self.ctx
        top, bot, left, right = self.halos
        Nz, ny, nx = self.shape
        exchange_halo(ctx, self.comm, self.tiling, self.rank, cube_faint, self.ext, self._strips)
        if mask is not None and not self._mask_set:  # halo spaxels are discarded: mask 0 there
            _copy_box(ctx, self.emask, self.eshape, (0, top, left), mask, mask.shape, (0, 0, 0),
                      (Nz, ny, nx))
            self._mask_set = True
        o = self.plan.run(self.ext, mask=self.emask if mask is not None else None,
                          correl=self.out["correl"], profile=self.out["profile"],
                          correl_min=self.out["correl_min"], want_maps=True)
        for name, dst in (("correl", correl), ("correl_min", correl_min), ("profile", profile)):
            _copy_box(ctx, dst, dst.shape, (0, 0, 0), o[name], self.eshape, (0, top, left),
                      (Nz, ny, nx))
        # the maps of the kept spaxels, cropped on the device (no host round trip per step)
        e_ny, e_nx = self.eshape[1:]
        for name in ("maxmap", "minmap"):
            _copy_box(ctx, self.maps[name], (1, ny, nx), (0, 0, 0), o[name], (1, e_ny, e_nx),
                      (0, top, left), (1, ny, nx))
        return dict(correl=correl, profile=profile, correl_min=correl_min,
                    maxmap=self.maps["maxmap"], minmap=self.maps["minmap"])
