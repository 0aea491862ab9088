"""Spatial tiling of the hot path over the GPUs of one node.

One process per GPU.  The field is cut into one rectangle per GPU along the PCA area grid (row
bands with their own column cuts, so that the fullest tile holds as few areas as possible), so
every area lives on exactly one GPU and the greedy PCA needs no communication.  Two exchanges
remain:

* one all-reduce of 2*Nz float64 (per-channel sum and count of the DCT residual) for the
  ``nanmean`` over the *whole* field in the standardisation (reference steps.py:442);
* one halo exchange of ``cube_faint`` before the GLR, in ONE phase: every rank sends each
  other rank the part of its tile that lies inside that rank's halo-extended box (edge strips
  and corner blocks; P//2 spaxels wide, + 1 when the local maxima are computed on the tiles).
  Outside the true field nothing is exchanged: the kernels zero-pad there exactly as the
  reference's ``fftconvolve(..., 'same')`` does.

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

Tile = namedtuple("Tile", "rank ty tx y0 y1 x0 x1")   # ty: row band, tx: position in the band


def _split(n_units, parts):
    """Split n_units into `parts` contiguous groups whose sizes differ by at most one."""
    base, rem = divmod(n_units, parts)
    sizes = [base + (1 if i < rem else 0) for i in range(parts)]
    edges = np.concatenate([[0], np.cumsum(sizes)])
    return edges


def _compositions(total, parts, lo=1):
    """All ways to write `total` as an ordered sum of `parts` integers >= lo."""
    if parts == 1:
        if total >= lo:
            yield (total,)
        return
    for first in range(lo, total - lo * (parts - 1) + 1):
        for rest in _compositions(total - first, parts - 1, lo):
            yield (first,) + rest


def grid_shape(world, nay=None, nax=None):
    """gy x gx with gy*gx == world.  Without an area grid: as square as possible, gx >= gy.
    With one (nay x nax areas): the factorisation whose fullest tile holds the fewest areas,
    then the squarest, then gx >= gy."""
    if nay is None or nax is None:
        gy = int(np.floor(np.sqrt(world)))
        while world % gy:
            gy -= 1
        return gy, world // gy
    best = None
    for gy in range(1, world + 1):
        if world % gy:
            continue
        gx = world // gy
        if gy > nay or gx > nax:
            continue
        worst = int(np.ceil(nay / gy)) * int(np.ceil(nax / gx))
        key = (worst, abs(gy - gx), gy > gx)
        if best is None or key < best[0]:
            best = (key, (gy, gx))
    if best is None:
        raise ValueError(f"{world} tiles do not fit a {nay}x{nax} area grid")
    return best[1]


def band_layout(world, nay, nax):
    """Row bands, each cut into its own number of columns: (rows per band, ranks per band) with
    the smallest number of areas in the fullest tile -- the greedy PCA of a rank costs at least its
    share of the areas, and DCT / GLR go with the spaxels.  A regular gy x gx grid is the special
    case of equal bands; when world does not divide the area grid evenly the bands win: 9 x 9
    areas on 8 ranks are 15 areas at most as a 2 x 4 grid, 12 as bands of 4 / 3 / 2 rows with
    3 / 3 / 2 ranks (the bound is ceil(81 / 8) = 11).  Ties: fewer bands, then the most even
    rows."""
    best = None
    for nb in range(1, min(world, nay) + 1):
        for ranks in _compositions(world, nb):
            if max(ranks) > nax:
                continue
            for rows in _compositions(nay, nb):
                worst = max(h * int(np.ceil(nax / r)) for h, r in zip(rows, ranks))
                key = (worst, nb, max(rows) - min(rows), max(ranks) - min(ranks), rows, ranks)
                if best is None or key < best[0]:
                    best = (key, (rows, ranks))
    if best is None:
        raise ValueError(f"{world} tiles do not fit a {nay}x{nax} area grid")
    return best[1]


class Tiling:
    """Partition of the field into one rectangle per rank, cut along the area grid (areas are
    area_size x area_size squares, the last one of a row / column absorbs the remainder, as
    synth.grid_areamap builds them), so that every PCA area lives on exactly one GPU.

    layout="bands" (default): row bands with their own column cuts (band_layout);
    layout="grid": a regular gy x gx grid (grid_shape) -- what rounds 1-2 used."""

    def __init__(self, Ny, Nx, world, area_size=100, halo=12, layout="bands"):
        self.Ny, self.Nx, self.world, self.halo = Ny, Nx, world, halo
        nay, nax = max(1, Ny // area_size), max(1, Nx // area_size)
        self.nay, self.nax, self.area_size = nay, nax, area_size
        if layout == "grid":
            gy, gx = grid_shape(world, nay, nax)
            rows = tuple(int(v) for v in np.diff(_split(nay, gy)))
            ranks = (gx,) * gy
        elif layout == "bands":
            rows, ranks = band_layout(world, nay, nax)
        else:
            raise ValueError("layout must be 'bands' or 'grid'")
        self.rows, self.ranks = rows, ranks
        self.tiles, self._areas = [], []
        ay0 = 0
        for ty, (h, nr) in enumerate(zip(rows, ranks)):
            y0 = ay0 * area_size
            y1 = Ny if ay0 + h == nay else (ay0 + h) * area_size
            ex = _split(nax, nr)
            for tx in range(nr):
                x0 = int(ex[tx]) * area_size
                x1 = Nx if ex[tx + 1] == nax else int(ex[tx + 1]) * area_size
                self.tiles.append(Tile(len(self.tiles), ty, tx, y0, y1, x0, x1))
                self._areas.append(h * int(ex[tx + 1] - ex[tx]))
            ay0 += h
        # a regular grid is described by (gy, gx); other band layouts have no single gx
        self.gy = len(rows)
        self.gx = ranks[0] if len(set(ranks)) == 1 else None

    def tile(self, rank):
        return self.tiles[rank]

    def balance(self):
        """max / mean over the ranks of the tile's spaxels (DCT, GLR work) and of its areas (the
        PCA's work, to first order): what the tiling costs against an even split."""
        spx = [float((t.y1 - t.y0) * (t.x1 - t.x0)) for t in self.tiles]
        ar = [float(a) for a in self._areas]
        return dict(spaxels=float(max(spx) / np.mean(spx)), areas=float(max(ar) / np.mean(ar)),
                    rows_per_band=[int(v) for v in self.rows],
                    ranks_per_band=[int(v) for v in self.ranks])

    def extended(self, rank):
        """Extent of the tile plus its halo, clipped to the field: (y0, y1, x0, x1) and the
        halo widths (top, bottom, left, right) actually present."""
        t, h = self.tiles[rank], self.halo
        top = min(h, t.y0)
        bot = min(h, self.Ny - t.y1)
        left = min(h, t.x0)
        right = min(h, self.Nx - t.x1)
        return (t.y0 - top, t.y1 + bot, t.x0 - left, t.x1 + right), (top, bot, left, right)

    def min_tile_side(self):
        return min(min(t.y1 - t.y0, t.x1 - t.x0) for t in self.tiles)

    def check_halo(self):
        # a strip must come from ONE neighbour: every tile at least a halo wide
        if self.min_tile_side() < self.halo:
            raise ValueError(f"a tile is narrower than the halo {self.halo}: a halo must come "
                             "from the tiles next to this one, not from beyond them")

    def owned_ext(self, rank):
        """None: every spaxel of the tile (the interior of the extended box) is this rank's."""
        return None

    def local_regions(self, rank, reach, R=64):
        """Bool array over the R x R regions of the extended tile: True where the region's GLR
        reads no spaxel of another rank (interior_regions)."""
        (y0, y1, x0, x1), halos = self.extended(rank)
        return interior_regions(y1 - y0, x1 - x0, halos, reach, R)


class OwnerTiling:
    """Partition of the field by an OWNER MAP: ``owner[y, x]`` = the rank that keeps spaxel
    (y, x).  What a field cut along irregular PCA areas needs (reference steps.py:492-569: areas
    come out of a segmentation, convex hulls and growing, not off a grid): every area goes to one
    rank as a whole, so the greedy PCA needs no communication, and a rank works on the BOUNDING
    BOX of its spaxels plus ``halo`` spaxels on every side (clipped to the field).  Inside that
    extended box it owns some spaxels; of the others it needs those within ``halo`` (Chebyshev
    distance) of an owned one -- they arrive as lists of spaxel columns from their owners
    (``column_plan`` / ``exchange_columns``); the rest of the box never influences a kept result.

    Same accessors as ``Tiling`` (tile, extended, halo, balance) so that ``TiledGLR`` takes either.
    """

    def __init__(self, owner, world, halo, n_areas=None):
        owner = np.ascontiguousarray(owner).astype(np.int32)
        if owner.ndim != 2 or owner.min() < 0 or owner.max() >= world:
            raise ValueError("owner must be a (Ny, Nx) map of ranks 0 .. world - 1")
        self.owner, self.world, self.halo = owner, int(world), int(halo)
        self.Ny, self.Nx = owner.shape
        self.tiles, self._count = [], []
        for r in range(world):
            rows = np.flatnonzero((owner == r).any(axis=1))
            cols = np.flatnonzero((owner == r).any(axis=0))
            if not len(rows):
                raise ValueError(f"rank {r} owns no spaxel: fewer areas than ranks?")
            self.tiles.append(Tile(r, 0, r, int(rows[0]), int(rows[-1]) + 1, int(cols[0]),
                                   int(cols[-1]) + 1))
            self._count.append(int(np.count_nonzero(owner == r)))
        self._areas = list(n_areas) if n_areas is not None else [1] * world
        self._need = {}

    # -- constructors ------------------------------------------------------------
    @classmethod
    def row_bands(cls, Ny, Nx, world, halo):
        """Bands of whole rows with (almost) the same number of rows each: the partition of the
        steps that know no areas yet (Preprocessing runs before CreateAreas, origin.py:193-208)."""
        if world > Ny:
            raise ValueError(f"{world} ranks for {Ny} rows")
        owner = np.repeat(np.arange(Ny, dtype=np.int64) * world // Ny, Nx).reshape(Ny, Nx)
        return cls(owner, world, halo)

    @classmethod
    def from_areamap(cls, areamap, world, halo):
        """Areas to ranks as wholes, balanced by spaxel count, compact in space: the labelled
        areas are split recursively along the longer side of the box of their centroids, at the
        cut that comes closest to the share of spaxels the ranks on either side stand for
        (world = 5: 2 + 3 ranks, shares 2/5 and 3/5).  Label 0 (spaxels no exposure covers,
        steps.py:560-563: cube_faint keeps cube_std there, lib_origin.py:799) goes with the
        nearest labelled spaxel."""
        from scipy import ndimage as ndi
        amap = np.asarray(getattr(areamap, "_data", areamap)).astype(np.int64)
        labels = np.unique(amap)
        labels = labels[labels > 0]
        if len(labels) < world:
            raise ValueError(f"{len(labels)} areas cannot be spread over {world} ranks")
        idx = np.arange(1, len(labels) + 1)
        comp = np.searchsorted(labels, amap) + 1          # labels -> 1 .. n (0 stays 0 below)
        comp[amap <= 0] = 0
        size = ndi.sum(np.ones_like(comp), comp, idx)
        cy, cx = np.array(ndi.center_of_mass(np.ones_like(comp), comp, idx)).T
        rank_of = np.zeros(len(labels), dtype=np.int32)

        def split(members, r0, nr):
            if nr == 1:
                rank_of[members] = r0
                return
            nl = nr // 2
            ys, xs = cy[members], cx[members]
            key = ys if (ys.max() - ys.min()) >= (xs.max() - xs.min()) else xs
            order = members[np.argsort(key, kind="stable")]
            csum = np.cumsum(size[order])
            want = csum[-1] * nl / nr
            # at least nl areas left of the cut and nr - nl right of it
            lo, hi = nl, len(order) - (nr - nl)
            cut = int(np.argmin(np.abs(csum[lo - 1:hi] - want))) + lo
            split(order[:cut], r0, nl)
            split(order[cut:], r0 + nl, nr - nl)

        split(np.arange(len(labels)), 0, world)
        table = np.concatenate([[0], rank_of])
        owner = table[comp]
        if np.any(comp == 0):   # label 0: with the nearest labelled spaxel
            _, (iy, ix) = ndi.distance_transform_edt(comp == 0, return_indices=True)
            owner = owner[iy, ix]
        n_areas = [int(np.count_nonzero(rank_of == r)) for r in range(world)]
        t = cls(owner, world, halo, n_areas)
        t.rank_of_label = {int(l): int(r) for l, r in zip(labels, rank_of)}
        return t

    # -- Tiling's accessors --------------------------------------------------------
    def tile(self, rank):
        return self.tiles[rank]

    def extended(self, rank):
        t, h = self.tiles[rank], self.halo
        top, bot = min(h, t.y0), min(h, self.Ny - t.y1)
        left, right = min(h, t.x0), min(h, self.Nx - t.x1)
        return (t.y0 - top, t.y1 + bot, t.x0 - left, t.x1 + right), (top, bot, left, right)

    def check_halo(self):
        pass    # (columns come from whoever owns them, however thin a rank's share is)

    def balance(self):
        """max / mean over the ranks of the owned spaxels (what the areas cost the PCA, to first
        order), of the areas, and of the extended boxes (what DCT-free stages -- GLR, local
        maxima -- run on: a box also holds spaxels of other ranks)."""
        own = np.array(self._count, dtype=float)
        box = np.array([float((e[1] - e[0]) * (e[3] - e[2]))
                        for e in (self.extended(r)[0] for r in range(self.world))])
        ar = np.array(self._areas, dtype=float)
        return dict(spaxels=float(own.max() / own.mean()), areas=float(ar.max() / ar.mean()),
                    boxes=float(box.max() / box.mean()),
                    box_over_owned=float(box.sum() / own.sum()),
                    owned=[int(v) for v in own], areas_per_rank=[int(v) for v in ar])

    def owned_tile(self, rank):
        """Bool (ny, nx): the spaxels of the rank's bounding box that are its own."""
        t = self.tiles[rank]
        return self.owner[t.y0:t.y1, t.x0:t.x1] == rank

    def owned_ext(self, rank):
        (y0, y1, x0, x1), _ = self.extended(rank)
        return self.owner[y0:y1, x0:x1] == rank

    def needed(self, rank):
        """Bool (Ny, Nx): spaxels of OTHER ranks within ``halo`` of one of this rank's."""
        got = self._need.get(rank)
        if got is None:
            from scipy import ndimage as ndi
            own = self.owner == rank
            (y0, y1, x0, x1), _ = self.extended(rank)
            grown = np.zeros_like(own)
            k = 2 * self.halo + 1
            grown[y0:y1, x0:x1] = ndi.maximum_filter(own[y0:y1, x0:x1].astype(np.uint8), size=k,
                                                     mode="constant", cval=0) > 0
            got = self._need[rank] = grown & ~own
        return got

    def local_regions(self, rank, reach, R=64):
        """Bool array over the R x R regions of the extended box: True where everything the
        region's GLR reads (the region grown by ``reach``, clipped to the field) is this rank's."""
        own = self.owned_ext(rank)
        e_ny, e_nx = own.shape
        nry, nrx = (e_ny + R - 1) // R, (e_nx + R - 1) // R
        (y0, y1, x0, x1), _ = self.extended(rank)
        ok = np.zeros((nry, nrx), bool)
        for ry in range(nry):
            a, b = R * ry - reach, min(e_ny, R * ry + R) + reach
            if (a < 0 and y0 > 0) or (b > e_ny and y1 < self.Ny):
                continue    # would read beyond the box where the field goes on
            for rx in range(nrx):
                c, d = R * rx - reach, min(e_nx, R * rx + R) + reach
                if (c < 0 and x0 > 0) or (d > e_nx and x1 < self.Nx):
                    continue
                ok[ry, rx] = own[max(a, 0):b, max(c, 0):d].all()
        return ok


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
        host_recv = [(peer, np.empty(buf.shape, buf.dtype), buf) for peer, buf in recvs]
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


def halo_plan(tiling, rank):
    """The halo exchange of one rank as plain index boxes (no data), ONE phase for any partition
    into rectangles: rank r sends to every other rank t the part of its own tile that lies inside
    t's halo-extended box, and receives from every u the part of u's tile inside its own extended
    box (edge strips and corner blocks alike; a tile of another band can cover part of an edge).
    Returns (sends, recvs): lists of (peer, (y, x) offset, (by, bx) size), offsets in this
    rank's TILE coordinates for sends and in its EXTENDED coordinates for receives, both ordered by
    peer -- what rank r sends to t is, box for box, what t receives from r."""
    me = tiling.tile(rank)
    (ey0, ey1, ex0, ex1), _ = tiling.extended(rank)
    sends, recvs = [], []
    for other in range(tiling.world):
        if other == rank:
            continue
        o = tiling.tile(other)
        (oy0, oy1, ox0, ox1), _ = tiling.extended(other)
        # my interior inside the other's extended box
        y0, y1 = max(me.y0, oy0), min(me.y1, oy1)
        x0, x1 = max(me.x0, ox0), min(me.x1, ox1)
        if y1 > y0 and x1 > x0:
            sends.append((other, (y0 - me.y0, x0 - me.x0), (y1 - y0, x1 - x0)))
        # the other's interior inside my extended box
        y0, y1 = max(o.y0, ey0), min(o.y1, ey1)
        x0, x1 = max(o.x0, ex0), min(o.x1, ex1)
        if y1 > y0 and x1 > x0:
            recvs.append((other, (y0 - ey0, x0 - ex0), (y1 - y0, x1 - x0)))
    return sends, recvs


def exchange_halo(ctx, comm, tiling, rank, cube, ext=None, bufs=None):
    """Fill the halo of this rank's extended device tile (Nz, ny + top + bot, nx + left + right).
    ``cube``: the bare (Nz, ny, nx) tile, copied into the interior first -- or None when the
    interior of ``ext`` already holds it (the greedy PCA can write there directly).  ``bufs``: a
    dict the caller keeps between calls so that the strip buffers are allocated once.  Any
    element type (float32 cubes, the uint8 mask); with an ``OwnerTiling`` the halo is the list of
    spaxel columns of ``column_plan`` instead of strips."""
    t = tiling.tile(rank)
    ny, nx = t.y1 - t.y0, t.x1 - t.x0
    (_, _, _, _), (top, bot, left, right) = tiling.extended(rank)
    Nz = (cube if cube is not None else ext).shape[0]
    eshape = (Nz, ny + top + bot, nx + left + right)
    dtype = (cube if cube is not None else ext).dtype
    if ext is None:
        ext = ctx.zeros(eshape, dtype)
    if cube is not None:
        _copy_box(ctx, ext, eshape, (0, top, left), cube, cube.shape, (0, 0, 0), (Nz, ny, nx))
    if isinstance(tiling, OwnerTiling):
        return exchange_columns(ctx, comm, tiling, rank, ext, bufs)
    plan_s, plan_r = halo_plan(tiling, rank)
    sends, recvs, unpack = [], [], []
    for kind, items in (("s", plan_s), ("r", plan_r)):
        for peer, (oy, ox), (by, bx) in items:
            key = (kind, peer, Nz, by, bx, dtype.str)
            buf = bufs.get(key) if bufs is not None else None
            if buf is None:
                buf = ctx.empty((Nz, by, bx), dtype)
                if bufs is not None:
                    bufs[key] = buf
            if kind == "s":   # cut from the interior of ext (tile coordinates + halo offset)
                _copy_box(ctx, buf, buf.shape, (0, 0, 0), ext, eshape, (0, top + oy, left + ox),
                          (Nz, by, bx))
                sends.append((peer, buf))
            else:
                recvs.append((peer, buf))
                unpack.append((buf, (0, oy, ox), (Nz, by, bx)))
    comm.exchange(ctx, sends, recvs)
    for rbuf, off, box in unpack:
        _copy_box(ctx, ext, eshape, off, rbuf, rbuf.shape, (0, 0, 0), box)
    return ext


def column_plan(tiling, rank):
    """The exchange of one rank of an ``OwnerTiling`` as index lists (no data): rank r sends every
    other rank t the columns of its own spaxels that t needs (``tiling.needed(t)``) and receives
    from every u the columns u owns among the ones it needs itself.  Returns (sends, recvs): lists
    of (peer, flat int32 indices into THIS rank's extended box), ordered by peer, indices in C
    order of the field -- what r sends to t is, column for column, what t receives from r."""
    (ey0, ey1, ex0, ex1), _ = tiling.extended(rank)
    e_nx = ex1 - ex0
    own = tiling.owner == rank
    mine = tiling.needed(rank)
    sends, recvs = [], []
    for other in range(tiling.world):
        if other == rank:
            continue
        ys, xs = np.nonzero(tiling.needed(other) & own)
        if len(ys):
            sends.append((other, ((ys - ey0) * e_nx + (xs - ex0)).astype(np.int32)))
        ys, xs = np.nonzero(mine & (tiling.owner == other))
        if len(ys):
            recvs.append((other, ((ys - ey0) * e_nx + (xs - ex0)).astype(np.int32)))
    return sends, recvs


def exchange_columns(ctx, comm, tiling, rank, ext, bufs=None):
    """Halo exchange of an ``OwnerTiling``: gather the columns the peers need from this rank's
    extended device tile, exchange, scatter what arrives (origin_gather_columns /
    origin_scatter_columns).  The owned part of ``ext`` must be in place."""
    Nz, e_ny, e_nx = ext.shape
    bufs = {} if bufs is None else bufs
    plan = bufs.get(("plan", rank))
    if plan is None:
        ps, pr = column_plan(tiling, rank)
        plan = bufs[("plan", rank)] = (
            [(peer, ctx.to_device(ix)) for peer, ix in ps],
            [(peer, ctx.to_device(ix)) for peer, ix in pr])
    es = ext.dtype.itemsize
    S = e_ny * e_nx

    def packed(kind, peer, n):
        key = (kind, peer, Nz, n, ext.dtype.str)
        b = bufs.get(key)
        if b is None:
            b = bufs[key] = ctx.empty((Nz, n), ext.dtype)
        return b
    sends, recvs = [], []
    for peer, ix in plan[0]:
        b = packed("cs", peer, ix.size)
        _capi.call("origin_gather_columns", ctx.handle, ext.p, Nz, S, ix.p, ix.size, es, b.p)
        sends.append((peer, b))
    for peer, ix in plan[1]:
        recvs.append((peer, packed("cr", peer, ix.size)))
    comm.exchange(ctx, sends, recvs)
    for (peer, ix), (_, b) in zip(plan[1], recvs):
        _capi.call("origin_scatter_columns", ctx.handle, ext.p, Nz, S, ix.p, ix.size, es, b.p)
    return ext


def exchange_halo_host(comm, tiling, rank, tile):
    """Same exchange on host ndarrays (float64 allowed) -- used by the CPU tests to check the
    tiling arithmetic against the untiled oracle."""
    Nz, ny, nx = tile.shape
    (_, _, _, _), (top, bot, left, right) = tiling.extended(rank)
    ext = np.zeros((Nz, ny + top + bot, nx + left + right), dtype=tile.dtype)
    ext[:, top: top + ny, left: left + nx] = tile
    if isinstance(tiling, OwnerTiling):
        ps, pr = column_plan(tiling, rank)
        flat = ext.reshape(Nz, -1)
        sends = [(peer, np.ascontiguousarray(flat[:, ix])) for peer, ix in ps]
        recvs = [(peer, np.empty((Nz, len(ix)), dtype=tile.dtype)) for peer, ix in pr]
        if sends or recvs:
            comm.group.exchange(sends, recvs)
        for (peer, ix), (_, rbuf) in zip(pr, recvs):
            flat[:, ix] = rbuf
        return ext
    plan_s, plan_r = halo_plan(tiling, rank)
    sends = [(peer, np.ascontiguousarray(tile[:, oy: oy + by, ox: ox + bx]))
             for peer, (oy, ox), (by, bx) in plan_s]
    recvs = [(peer, np.empty((Nz, by, bx), dtype=tile.dtype)) for peer, _, (by, bx) in plan_r]
    if sends or recvs:
        comm.group.exchange(sends, recvs)
    for (peer, (oy, ox), (by, bx)), (_, rbuf) in zip(plan_r, recvs):
        ext[:, oy: oy + by, ox: ox + bx] = rbuf
    return ext


def region_rects(ok, Ny, Nx, R=64):
    """Rectangles (y0, y1, x0, x1) in pixels that cover exactly the True cells of ``ok`` (a bool
    array over the R x R regions of a (Ny, Nx) field): runs of cells per region row, merged with
    the row above when they span the same columns."""
    ok = np.asarray(ok, bool)
    nry, nrx = ok.shape
    rects, open_ = [], {}     # open_: (rx0, rx1) -> index of a rect that ended on the row above
    for ry in range(nry):
        runs, rx = [], 0
        while rx < nrx:
            if ok[ry, rx]:
                e = rx
                while e < nrx and ok[ry, e]:
                    e += 1
                runs.append((rx, e))
                rx = e
            else:
                rx += 1
        now = {}
        for run in runs:
            if run in open_:
                i = open_[run]
                rects[i][1] = ry + 1
            else:
                i = len(rects)
                rects.append([ry, ry + 1, run[0], run[1]])
            now[run] = i
        open_ = now
    return [(R * a, min(Ny, R * b), R * c, min(Nx, R * d)) for a, b, c, d in rects]


def interior_regions(Ny, Nx, halos, reach, R=64):
    """Bool array over the R x R regions of a halo-extended tile: True where the region's GLR
    needs no halo data -- its spatial stage reads rows / columns within ``reach`` (P // 2) of the
    region, none of them in a halo strip.  ``halos`` = (top, bottom, left, right) widths."""
    top, bot, left, right = halos
    nry, nrx = (Ny + R - 1) // R, (Nx + R - 1) // R

    def clear(n, N, lo_h, hi_h):
        out = np.ones(n, bool)
        for r in range(n):
            a, b = R * r - reach, min(N, R * r + R) + reach     # rows read: [a, b)
            if lo_h and a < lo_h:
                out[r] = False
            if hi_h and b > N - hi_h:
                out[r] = False
        return out
    return np.outer(clear(nry, Ny, top, bot), clear(nrx, Nx, left, right))


class TiledGLR:
    """GLR of one tile of a tiled field: halo exchange + plan on the extended tile + crop."""

    def __init__(self, ctx, comm, tiling, rank, Nz, PSF, profiles, pcut=1e-8, pmeansub=True,
                 weights=None, ext=None):
        """``weights``: None or the mosaic's weight maps (one (Ny, Nx) array per field of the WHOLE
        field, origin.py:600-609): every rank crops them to its halo-extended tile, no exchange.
        ``ext``: an extended tile made earlier (a greedy PCA that ran before the GLR's parameters
        were known wrote cube_faint into it) instead of a fresh one."""
        self.ctx, self.comm, self.tiling, self.rank, self.Nz = ctx, comm, tiling, rank, Nz
        # a kept spaxel must see real neighbour data over the whole PSF footprint, and a strip
        # must come from ONE neighbour: halo >= P//2 and every tile at least a halo wide
        psf0 = PSF[0] if isinstance(PSF, (list, tuple)) else PSF
        need = int(np.asarray(psf0).shape[-1]) // 2
        if tiling.halo < need:
            raise ValueError(f"tiling halo {tiling.halo} is smaller than the PSF half width {need}")
        tiling.check_halo()
        (y0, y1, x0, x1), self.halos = tiling.extended(rank)
        self.eshape = (Nz, y1 - y0, x1 - x0)
        wext = None
        if weights is not None:
            wext = [np.ascontiguousarray(np.asarray(w, dtype=np.float64)[y0:y1, x0:x1])
                    for w in weights]
        self.plan = kernels.GLRPlan(ctx, self.eshape, PSF, wext, profiles, pcut, pmeansub)
        t = tiling.tile(rank)
        self.shape = (Nz, t.y1 - t.y0, t.x1 - t.x0)
        # (zeros: with an OwnerTiling the box also holds spaxels that are neither this rank's nor
        # needed by it; nothing kept depends on them, but they should not be NaN patterns)
        if ext is not None and (ext.shape != self.eshape or ext.dtype != np.float32):
            raise ValueError(f"ext must be a float32 array of shape {self.eshape}")
        self.ext = ctx.zeros(self.eshape, np.float32) if ext is None else ext
        self.emask = ctx.zeros(self.eshape, np.uint8)
        self.out = dict(correl=ctx.empty(self.eshape, np.float32),
                        correl_min=ctx.empty(self.eshape, np.float32),
                        profile=ctx.empty(self.eshape, np.uint8))
        self.maps = dict(maxmap=ctx.empty(self.shape[1:], np.float32),
                         minmap=ctx.empty(self.shape[1:], np.float32))
        self._mask_set = False
        self._early_done = None   # regions whose GLR the tail hook has started (this step)
        self._strips = {}
        self._lm = None   # extended local-maxima cubes, allocated by the first run that wants them

    def faint_target(self):
        """``into`` argument of pipeline.greedy_pca: the interior of the extended tile.  A PCA run
        that writes there is followed by ``run(None, ...)`` -- no copy of the tile in between."""
        top, _, left, _ = self.halos
        return (self.ext, top, left)

    def set_ext_mask(self, emask):
        """Hand over the TRUE mask of the whole extended tile (uint8, extended shape) -- a caller
        that holds the whole field's mask on the host cuts it there; no exchange then."""
        if emask.shape != self.eshape or emask.dtype != np.uint8:
            raise ValueError(f"the extended mask must be uint8 of shape {self.eshape}")
        self.emask, self._mask_set = emask, True

    def _set_mask(self, mask):
        """The mask of the extended tile, made once: this rank's part copied in, the halo
        exchanged like the halo of the cube (COLLECTIVE: every rank passes a mask, or none
        does).  The halo's results are discarded, but the 3x3x3 local maxima of the tile's edge
        spaxels compare them with their neighbours beyond it, and ``correl[mask] = 0``
        (steps.py:781) holds there too: with mask 0 in the halo a masked voxel next to an
        internal cut kept its T_GLR and could out-rank a true local maximum on the edge row."""
        if mask is not None and not self._mask_set:
            exchange_halo(self.ctx, self.comm, self.tiling, self.rank, mask, self.emask,
                          self._strips)
            self._mask_set = True
        return self.emask if mask is not None else None

    def make_tail_hook(self, area_boxes, mask, early_budget=8.5e8):
        """A function for ``Context.set_pca_tail_hook`` around the greedy PCA that writes this
        tile (``into=faint_target()``): when few areas still iterate, the regions of the extended
        tile that read neither halo data nor a row / column of those areas start their GLR on the
        side stream, in the shadow of the PCA's tail (pipeline.greedy_pca_then_glr does the same
        on an untiled field).  ``area_boxes[a]`` = (ymin, ymax, xmin, xmax) of area ``a`` in TILE
        coordinates, inclusive; ``early_budget``: voxels of GLR started that way at most.  The
        next ``run(None, ...)`` finishes the step."""
        self._set_mask(mask)   # (collective: here, on every rank, not inside the hook)
        if not self.plan.rows_supported():
            return None
        top, _, left, _ = self.halos
        e_ny, e_nx = self.eshape[1:]
        reach = self.plan.P // 2

        def hook(areas):
            ok = self.tiling.local_regions(self.rank, reach)
            for a in areas:
                if area_boxes[a] is None:
                    continue
                ymin, ymax, xmin, xmax = area_boxes[a]
                r0, r1 = max(0, (ymin + top - reach) // 64), (ymax + top + reach) // 64
                c0, c1 = max(0, (xmin + left - reach) // 64), (xmax + left + reach) // 64
                ok[r0:r1 + 1, c0:c1 + 1] = False
            if early_budget is not None:     # the first regions in row order, up to the budget
                keep = max(1, int(early_budget / (self.Nz * 64 * 64)))
                flat = np.flatnonzero(ok)
                ok.reshape(-1)[flat[keep:]] = False
            rects = region_rects(ok, e_ny, e_nx)
            if not rects:
                return
            emask = self._set_mask(mask)
            oc, op, om = self.out["correl"], self.out["profile"], self.out["correl_min"]
            for i, (y0, y1, x0, x1) in enumerate(rects):
                self.plan.run_rect(self.ext, emask, oc, op, om, y0, y1, x0, x1, first=(i == 0),
                                   side=True)
            self._early_done = ok
        return hook

    def run(self, cube_faint, mask, correl, profile, correl_min, local_max=None, size=3,
            before_exchange=None):
        """Returns the caller's correl / profile / correl_min and the tile's maxmap / minmap.  The
        two maps are DeviceArrays OWNED BY THIS OBJECT and rewritten by the next call: copy them
        (``.to_host()`` / ``.copy()``) to keep them across steps.

        ``correl = profile = correl_min = None`` (and ``local_max=True``): NO crop.  The results
        stay in the halo-extended arrays this object owns -- ``res["correl"]`` etc. have the
        extended shape and ``res["box"] = (top, left, ny, nx)`` is the tile inside them (read it
        with ``DeviceArray.window`` or hand kernels the strides); cropping three cubes and
        the two local-maxima cubes on the device is 34 B/voxel of extra traffic per step, more
        than a third of what the whole path moves.

        ``before_exchange``: called (no arguments) right before the collective halo exchange,
        behind the regions that run ahead of it -- where a caller lets the ranks agree that all of
        them got this far (a rank whose PCA failed must not leave the others waiting in RCCL)
        without holding the interior regions of the fast ranks back.

        ``local_max``: None, ``True`` (dense, in the extended arrays, with correl=None),
        ``"sparse"`` (the same as lists of the non-zero voxels of the extended cubes:
        ``sparse.SparseCube`` s), or a pair of tile-shaped DeviceArrays that receive
        ``compute_local_max(correl, correl_min, mask, size)`` (reference steps.py:796).  The
        maximum filter looks size // 2 spaxels beyond the tile, so the tiling's halo must be at
        least P // 2 + size // 2: those neighbours are then exact values of the extended GLR."""
        ctx = self.ctx
        top, bot, left, right = self.halos
        Nz, ny, nx = self.shape
        emask = self._set_mask(mask)
        # Interior first: the regions of the extended tile whose GLR reads no halo data run on the
        # context's side stream WHILE the strips travel (and while this rank waits for a neighbour
        # that is still iterating); the regions along the halo follow the exchange on the main
        # stream.  Needs the tile in self.ext already (cube_faint None: the PCA wrote it there)
        # and a plan whose stages take rectangles; ORIGIN_TILED_INTERIOR_FIRST=0 turns it off.
        early, done = [], self._early_done   # (done: regions the tail hook started, this step)
        self._early_done = None
        if cube_faint is not None and done is not None:
            raise ValueError("the tail hook started this step's GLR on the extended tile: "
                             "run(None, ...) must finish it")
        if (cube_faint is None and self.plan.rows_supported()
                and (done is not None
                     or os.environ.get("ORIGIN_TILED_INTERIOR_FIRST", "1") != "0")):
            e_ny, e_nx = self.eshape[1:]
            ok = self.tiling.local_regions(self.rank, self.plan.P // 2)
            if os.environ.get("ORIGIN_TILED_INTERIOR_FIRST", "1") == "0":
                ok[:] = False                 # (only what the hook started runs ahead)
            if done is not None:
                ok |= done
            late = region_rects(~ok, e_ny, e_nx)
            early = region_rects(ok & ~done if done is not None else ok, e_ny, e_nx)
            if not early and done is not None:
                early = [None]                # (nothing more ahead, but the step is in rectangles)
        if early:
            oc, op, om = self.out["correl"], self.out["profile"], self.out["correl_min"]
            for i, rect in enumerate(r for r in early if r is not None):
                y0, y1, x0, x1 = rect
                self.plan.run_rect(self.ext, emask, oc, op, om, y0, y1, x0, x1,
                                   first=(i == 0 and done is None), side=True)
            early = [r for r in early if r is not None]
            if before_exchange is not None:
                before_exchange()
            exchange_halo(ctx, self.comm, self.tiling, self.rank, None, self.ext, self._strips)
            for y0, y1, x0, x1 in late:
                self.plan.run_rect(self.ext, emask, oc, op, om, y0, y1, x0, x1)
            maxmap, minmap = self.plan.run_finish(want_maps=True)
            o = dict(correl=oc, profile=op, correl_min=om, maxmap=maxmap, minmap=minmap)
        else:
            # cube_faint None: the greedy PCA wrote this step's tile straight into self.ext's
            # interior
            if before_exchange is not None:
                before_exchange()
            exchange_halo(ctx, self.comm, self.tiling, self.rank, cube_faint, self.ext,
                          self._strips)
            o = self.plan.run(self.ext, mask=emask, correl=self.out["correl"],
                              profile=self.out["profile"], correl_min=self.out["correl_min"],
                              want_maps=True)
        self.last_rects = (early, late if (early or done is not None) else [],
                           0 if done is None else int(done.sum()))
        crop = correl is not None
        if crop:
            for name, dst in (("correl", correl), ("correl_min", correl_min), ("profile", profile)):
                _copy_box(ctx, dst, dst.shape, (0, 0, 0), o[name], self.eshape, (0, top, left),
                          (Nz, ny, nx))
        else:
            if profile is not None or correl_min is not None:
                raise ValueError("pass all three output cubes or none of them")
            correl, correl_min, profile = o["correl"], o["correl_min"], o["profile"]
        # the maps of the kept spaxels, cropped on the device (no host round trip per step)
        e_ny, e_nx = self.eshape[1:]
        for name in ("maxmap", "minmap"):
            _copy_box(ctx, self.maps[name], (1, ny, nx), (0, 0, 0), o[name], (1, e_ny, e_nx),
                      (0, top, left), (1, ny, nx))
        res = dict(correl=correl, profile=profile, correl_min=correl_min,
                   maxmap=self.maps["maxmap"], minmap=self.maps["minmap"],
                   box=(top, left, ny, nx) if not crop else (0, 0, ny, nx),
                   # OwnerTiling: which spaxels of the box are this rank's (None: all of them)
                   owned=(None if self.tiling.owned_ext(self.rank) is None else
                          self.tiling.owned_ext(self.rank)[top:top + ny, left:left + nx]))
        if local_max is not None and local_max is not False:
            need = self.plan.P // 2 + int(size) // 2
            if self.tiling.halo < need:
                raise ValueError(f"local maxima of size {size} on tiles need a halo of {need} "
                                 f"spaxels, the tiling has {self.tiling.halo}")
            if local_max == "sparse":
                # (index, value) lists of the extended cubes' non-zero voxels instead of two dense
                # cubes (origin_amd/sparse.py): indices are those of the EXTENDED tile, entries
                # outside res["box"] / not owned belong to the neighbours
                from . import sparse
                if crop:
                    raise ValueError("local_max='sparse' goes with correl=None (no crop at all)")
                if int(size) != 3 or sparse.plan(ctx, self.eshape)[0] == 0:
                    raise ValueError("no sparse local-maximum form for this tile (size 3, Nx % 4)")
                if getattr(self, "_lm_sparse", None) is None:
                    self._lm_sparse = sparse.SparseBuffers(ctx, self.eshape)
                res["local_max"], res["local_min"] = sparse.local_max_sparse(
                    ctx, o["correl"], o["correl_min"], self.emask if mask is not None else None,
                    self._lm_sparse)
                return res
            if self._lm is None:
                self._lm = (ctx.empty(self.eshape, np.float32), ctx.empty(self.eshape, np.float32))
            kernels.local_max(ctx, o["correl"], o["correl_min"],
                              self.emask if mask is not None else None, size,
                              out_max=self._lm[0], out_min=self._lm[1])
            if local_max is True:   # no crop: the extended cubes (valid inside res["box"])
                if crop:
                    raise ValueError("local_max=True goes with correl=None (no crop at all)")
                res["local_max"], res["local_min"] = self._lm
            else:
                for src, dst in zip(self._lm, local_max):
                    _copy_box(ctx, dst, dst.shape, (0, 0, 0), src, self.eshape, (0, top, left),
                              (Nz, ny, nx))
                res["local_max"], res["local_min"] = local_max
        return res
