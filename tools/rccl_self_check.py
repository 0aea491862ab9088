"""Rehearsal of the RCCL path of origin_amd.multigpu on a ONE-GPU box.

    python tools/rccl_self_check.py            # world size 1: native communicator, self send

What it can prove: librccl.so opens next to the library's HIP context, the communicator
initialises from a unique id that went through the host group, all-reduce and grouped
send/recv run on the library's stream on its own buffers.  What it cannot prove: a transfer
over xGMI between two devices.  (Two ranks on ONE card: RCCL refuses, and every rank raises
-- there is no silent change of transport.)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")

from origin_amd import multigpu  # noqa: E402
from origin_amd.device import Context  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    comm = multigpu.init_comm(rank, world, 0, backend="rccl")
    ctx = Context(0)
    comm.attach(ctx)
    print(rank, "backend", comm.backend, "device_p2p", comm.device_p2p, comm.note, flush=True)
    a = ctx.to_device(np.arange(5.0) + rank)
    b = ctx.to_device(np.ones(3) * (rank + 1))
    comm.allreduce_sum_device(ctx, [a, b])
    want_a = sum(np.arange(5.0) + r for r in range(world))
    assert np.array_equal(a.to_host(), want_a), a.to_host()
    assert np.array_equal(b.to_host(), np.ones(3) * sum(range(1, world + 1)))
    assert comm.max_float(3.5 + rank) == 3.5 + world - 1
    comm.barrier()
    peer = (rank + 1) % world
    src = ctx.to_device(np.random.default_rng(rank).normal(size=(7, 5, 12)).astype(np.float32))
    dst = ctx.zeros((7, 5, 12), np.float32)
    comm.exchange(ctx, [(peer, src)], [((rank - 1) % world, dst)])
    want = np.random.default_rng((rank - 1) % world).normal(size=(7, 5, 12)).astype(np.float32)
    assert np.array_equal(dst.to_host(), want), "exchange changed the data"
    comm.barrier()
    comm.close()
    print(rank, "comm self check OK", flush=True)


if __name__ == "__main__":
    main()
