"""Host-side rendezvous of the ranks of one node -- sockets and NumPy only.

The tiled path (multigpu.py) needs very little from the host: the 128-byte RCCL unique id
from rank 0, barriers, max / sum of a few float64, and -- only for CPU tests and one-GPU
rehearsals -- host-staged strips between neighbours.  Every rank listens on an address
derived from MASTER_ADDR / MASTER_PORT and its rank, connects to all lower ranks and accepts
all higher ones (full mesh, world <= a few dozen); collectives go through rank 0.

Addresses: by default abstract unix sockets ``\\0origin-rdv-<MASTER_PORT>-<key>-r<rank>``
(one node, nothing on disk, nothing stale: the name dies with its process; a launcher's own
store can keep MASTER_PORT).  ``key`` is ORIGIN_RDV_KEY, else TORCHELASTIC_RUN_ID, else the
parent's pid (the launcher is the common parent of all ranks).  With
``ORIGIN_RDV_ADDR=tcp://host:port`` rank r listens on TCP ``port + r`` instead.
"""
import os
import socket
import struct
import threading
import time

import numpy as np

TIMEOUT = float(os.environ.get("ORIGIN_RDV_TIMEOUT", "300"))


def _addresses(world):
    spec = os.environ.get("ORIGIN_RDV_ADDR", "")
    if spec.startswith("tcp://"):
        host, port = spec[6:].rsplit(":", 1)
        return [(socket.AF_INET, (host, int(port) + r)) for r in range(world)]
    key = os.environ.get("ORIGIN_RDV_KEY") or os.environ.get("TORCHELASTIC_RUN_ID") or \
        str(os.getppid())
    port = os.environ.get("MASTER_PORT", "0")
    return [(socket.AF_UNIX, f"\0origin-rdv-{port}-{key}-r{r}") for r in range(world)]


def _job_tag():
    """16 bytes every rank of one job derives alike (key and port): the first thing a rank says
    when it connects, so that a stray connection cannot take a rank's place."""
    import hashlib
    key = os.environ.get("ORIGIN_RDV_KEY") or os.environ.get("TORCHELASTIC_RUN_ID") or \
        str(os.getppid())
    return hashlib.sha256(f"origin-rdv:{key}:{os.environ.get('MASTER_PORT', '0')}".encode()).digest()[:16]


def _hello(rank):
    return struct.pack("<16si", _job_tag(), rank)


def _recv_exact(sock, n, into=None):
    buf = into if into is not None else bytearray(n)
    view = memoryview(buf).cast("B")
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the rendezvous connection")
        got += k
    return buf


class HostGroup:
    """rank / world, broadcast, allreduce (sum | max | min), barrier, exchange, close."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.peers = {}
        self._listener = None
        if world == 1:
            return
        addrs = _addresses(world)
        fam, me = addrs[rank]
        ls = socket.socket(fam, socket.SOCK_STREAM)
        if fam == socket.AF_INET:
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            ls.bind(me)
        except OSError as exc:
            raise OSError(f"rank {rank}: rendezvous address {me!r} is taken -- another job with the "
                          "same MASTER_PORT and key runs on this node (set ORIGIN_RDV_KEY or "
                          f"another MASTER_PORT): {exc}") from exc
        ls.listen(world)
        ls.settimeout(TIMEOUT)
        self._listener = ls
        deadline = time.time() + TIMEOUT
        for r in range(rank):                       # connect to every lower rank
            fam_r, addr = addrs[r]
            while True:
                s = socket.socket(fam_r, socket.SOCK_STREAM)
                try:
                    s.connect(addr)
                    break
                except (ConnectionRefusedError, FileNotFoundError, OSError):
                    s.close()
                    if time.time() > deadline:
                        raise TimeoutError(f"rank {rank}: rank {r} never came up at {addr!r}")
                    time.sleep(0.02)
            s.sendall(_hello(rank))
            self._setup(s, fam_r)
            self.peers[r] = s
        while len(self.peers) < world - 1:          # accept every higher rank
            s, _a = ls.accept()
            s.settimeout(TIMEOUT)
            try:
                tag, r = struct.unpack("<16si", bytes(_recv_exact(s, 20)))
            except (ConnectionError, OSError, struct.error):
                s.close()
                continue
            # a connection that is not one of this job's higher ranks (wrong key, a rank out of
            # range or one that is already here) is dropped, not given a rank's place
            if tag != _job_tag() or not (rank < r < world) or r in self.peers:
                s.close()
                continue
            self._setup(s, fam)
            self.peers[r] = s

    @staticmethod
    def _setup(s, fam):
        s.settimeout(TIMEOUT)
        if fam == socket.AF_INET:
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    # -- collectives through rank 0 ---------------------------------------------
    def broadcast(self, data, src=0):
        """bytes from rank `src` to everyone (returns them on every rank)."""
        if self.world == 1:
            return bytes(data)
        if src != 0:                                 # hop through rank 0
            if self.rank == src:
                self.peers[0].sendall(struct.pack("<q", len(data)) + bytes(data))
            if self.rank == 0:
                (n,) = struct.unpack("<q", bytes(_recv_exact(self.peers[src], 8)))
                data = bytes(_recv_exact(self.peers[src], n))
        if self.rank == 0:
            msg = struct.pack("<q", len(data)) + bytes(data)
            for r in range(1, self.world):
                self.peers[r].sendall(msg)
            return bytes(data)
        (n,) = struct.unpack("<q", bytes(_recv_exact(self.peers[0], 8)))
        return bytes(_recv_exact(self.peers[0], n))

    def allreduce(self, arr, op="sum"):
        """float64 reduction over ranks, summed in rank order on rank 0 (deterministic)."""
        a = np.ascontiguousarray(arr, dtype=np.float64).copy()
        if self.world == 1:
            return a
        if self.rank == 0:
            f = {"sum": np.add, "max": np.maximum, "min": np.minimum}[op]
            tmp = np.empty_like(a)
            for r in range(1, self.world):
                _recv_exact(self.peers[r], a.nbytes, tmp.reshape(-1).view(np.uint8))
                f(a, tmp, out=a)
            for r in range(1, self.world):
                self.peers[r].sendall(a.tobytes())
            return a
        self.peers[0].sendall(a.tobytes())
        _recv_exact(self.peers[0], a.nbytes, a.reshape(-1).view(np.uint8))
        return a

    def barrier(self):
        self.allreduce(np.zeros(1))

    # -- point to point -----------------------------------------------------------
    def exchange(self, sends, recvs):
        """sends: [(peer, ndarray)], recvs: [(peer, ndarray to fill)].  All posted together:
        the sends to a peer run on a thread of their own, so no order of receives can dead-lock."""
        threads = []
        errs = []

        def push(sock, payloads):
            try:
                for payload in payloads:
                    sock.sendall(payload)
            except Exception as exc:  # noqa: BLE001 -- re-raised below
                errs.append(exc)

        by_peer = {}
        for peer, arr in sends:       # one thread per PEER: its messages stay in order
            by_peer.setdefault(peer, []).append(memoryview(np.ascontiguousarray(arr)).cast("B"))
        for peer, payloads in by_peer.items():
            t = threading.Thread(target=push, args=(self.peers[peer], payloads))
            t.start()
            threads.append(t)
        for peer, out in recvs:
            assert out.flags.c_contiguous
            _recv_exact(self.peers[peer], out.nbytes, out.reshape(-1).view(np.uint8))
        for t in threads:
            t.join()
        if errs:
            raise errs[0]

    def close(self):
        for s in self.peers.values():
            try:
                s.close()
            except OSError:
                pass
        self.peers = {}
        if self._listener is not None:
            self._listener.close()
            self._listener = None


def from_env():
    return HostGroup(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
