"""torch.distributed (gloo) behind the five methods of origin_amd.rendezvous.HostGroup -- test
infrastructure: the tiled path runs over the package's own socket rendezvous AND over gloo
(world_size 2 / 4 on CPU), and both must give the untiled result."""
import os

import numpy as np


class GlooGroup:
    def __init__(self, rank, world):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def broadcast(self, data, src=0):
        n = self.torch.tensor([len(data)], dtype=self.torch.int64)
        self.dist.broadcast(n, src=src)
        buf = self.torch.zeros(int(n[0]), dtype=self.torch.uint8)
        if self.rank == src and len(data):
            buf[:] = self.torch.frombuffer(bytearray(data), dtype=self.torch.uint8)
        if int(n[0]):
            self.dist.broadcast(buf, src=src)
        return bytes(buf.numpy().tobytes())

    def allreduce(self, arr, op="sum"):
        t = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
        ops = {"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX,
               "min": self.dist.ReduceOp.MIN}
        self.dist.all_reduce(t, op=ops[op])
        return t.numpy()

    def barrier(self):
        self.dist.barrier()

    def exchange(self, sends, recvs):
        torch, dist = self.torch, self.dist
        ops = [dist.P2POp(dist.irecv, torch.from_numpy(out), peer) for peer, out in recvs]
        ops += [dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(a)), peer)
                for peer, a in sends]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
