"""origin_amd.rendezvous.HostGroup: the socket rendezvous under the tiled path (CPU)."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tcp, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT=str(port),
                      ORIGIN_RDV_KEY="t")
    if tcp:
        os.environ["ORIGIN_RDV_ADDR"] = f"tcp://127.0.0.1:{port}"
    from origin_amd.rendezvous import HostGroup
    try:
        g = HostGroup(rank, world)
        ident = g.broadcast(bytes(range(128)) if rank == 0 else b"", src=0)
        assert ident == bytes(range(128))
        assert g.broadcast(b"from-two" if rank == 2 % world else b"", src=2 % world) == b"from-two"
        a = np.arange(6.0) * (rank + 1)
        assert np.array_equal(g.allreduce(a), np.arange(6.0) * sum(range(1, world + 1)))
        assert g.allreduce(np.array([float(rank)]), "max")[0] == world - 1
        assert g.allreduce(np.array([float(rank)]), "min")[0] == 0
        g.barrier()
        # ring + reverse ring with strips larger than any socket buffer, posted together
        n = 3_000_000
        nxt, prv = (rank + 1) % world, (rank - 1) % world
        out_a, out_b = np.empty(n, np.float32), np.empty((3, n // 3), np.float64)
        g.exchange([(nxt, np.full(n, rank, np.float32)), (prv, np.full((3, n // 3), -rank, np.float64))],
                   [(prv, out_a), (nxt, out_b)])   # (messages of one pair are FIFO)
        assert np.all(out_a == prv) and np.all(out_b == -nxt)
        g.barrier()
        g.close()
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        q.put((rank, repr(exc)))


@pytest.mark.parametrize("world,tcp", [(2, False), (3, False), (4, True)])
def test_host_group(world, tcp):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tcp, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == {r: "ok" for r in range(world)}, res


def test_world_one_needs_no_socket():
    from origin_amd.rendezvous import HostGroup
    g = HostGroup(0, 1)
    assert g.broadcast(b"abc") == b"abc"
    assert g.allreduce(np.array([2.0]), "max")[0] == 2.0
    g.barrier()
    g.close()


def test_package_does_not_import_torch():
    """north_star: host code is Python + ctypes, no PyTorch."""
    import re
    for dirpath, _, files in os.walk(os.path.join(ROOT, "origin_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+torch", txt, re.M), f


def test_stray_connection_does_not_take_a_rank():
    """A connection that does not open with this job's tag (or names a rank out of range, or one
    that is already connected) is dropped by the listener; the real rank still gets in."""
    import socket
    import struct
    import threading
    import time
    from origin_amd import rendezvous as rdv
    os.environ["ORIGIN_RDV_KEY"] = f"stray{os.getpid()}"
    os.environ["MASTER_PORT"] = "29611"
    groups = {}

    def rank(r):
        groups[r] = rdv.HostGroup(r, 2)
    t0 = threading.Thread(target=rank, args=(0,))
    t0.start()
    fam, addr = rdv._addresses(2)[0]
    deadline = time.time() + 20
    for payload in (struct.pack("<16si", b"x" * 16, 1),          # wrong tag
                    struct.pack("<16si", rdv._job_tag(), 7),      # rank out of range
                    b"\x01\x00"):                                 # short and gone
        while True:
            s = socket.socket(fam, socket.SOCK_STREAM)
            try:
                s.connect(addr)
                break
            except OSError:
                s.close()
                assert time.time() < deadline
                time.sleep(0.02)
        s.sendall(payload)
        s.close()
    t1 = threading.Thread(target=rank, args=(1,))
    t1.start()
    t0.join(30), t1.join(30)
    assert set(groups) == {0, 1} and set(groups[0].peers) == {1}

    def red(r, out):
        out[r] = groups[r].allreduce(np.array([float(r + 1)]))[0]
    out = {}
    ts = [threading.Thread(target=red, args=(r, out)) for r in (0, 1)]
    [t.start() for t in ts], [t.join(30) for t in ts]
    assert out == {0: 3.0, 1: 3.0}
    for g in groups.values():
        g.close()
