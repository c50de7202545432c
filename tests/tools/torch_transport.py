"""Test infrastructure: a ``torch.distributed`` process group (gloo on CPU) behind the transport interface
of ``shennong_amd.distributed`` (rank, world_size, all_gather_object, gather_features, allreduce).  The
product talks RCCL through the C ABI (``shennong_amd.comm.RcclComm``) and never imports torch; this class is
how the multi-process logic of ``shennong_amd.distributed`` is exercised on a box without GPUs, next to the
socket-backed stand-in of the ``snf_comm_*`` entry points in ``fake_comm.py``."""

import numpy as np


class TorchTransport:
    def __init__(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)

    def all_gather_object(self, obj):
        import torch.distributed as dist
        out = [None] * self.world_size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def gather_features(self, local, dst=0):
        """``{name: float32 [nframes, ndims]}`` of every rank merged on rank `dst` (None elsewhere): one
        contiguous float32 buffer per peer, point to point to the root"""
        import torch
        import torch.distributed as dist
        group = self.group
        rank, world = self.rank, self.world_size
        names = list(local.keys())
        shapes = [tuple(local[n].shape) for n in names]
        meta = self.all_gather_object((names, shapes))
        sizes = [sum(int(np.prod(s)) for s in m[1]) for m in meta]
        flat = np.concatenate([np.ascontiguousarray(local[n], dtype=np.float32).reshape(-1)
                               for n in names]) if names else np.zeros(0, np.float32)
        send = torch.from_numpy(flat)
        if rank == dst:
            bufs = {r: torch.empty(sizes[r], dtype=torch.float32)
                    for r in range(world) if r != dst and sizes[r] > 0}
            ops = [dist.P2POp(dist.irecv, buf, r, group) for r, buf in bufs.items()]
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            merged = {}
            for r in range(world):
                data = (send if r == dst else bufs.get(r))
                host = data.numpy() if data is not None else np.zeros(0, np.float32)
                pos = 0
                for name, shape in zip(*meta[r]):
                    n = int(np.prod(shape))
                    merged[name] = host[pos:pos + n].reshape(shape).copy()
                    pos += n
            return merged
        if sizes[rank] > 0:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, dst, group)]):
                req.wait()
        return None

    def allreduce(self, array, op='sum'):
        """Rank-ordered host sum (or max) of float64 blocks: one all-gather of the (tiny) blocks"""
        import torch
        import torch.distributed as dist
        stats = np.ascontiguousarray(array, dtype=np.float64)
        if self.world_size == 1:
            return stats.copy()
        send = torch.from_numpy(stats.reshape(-1))
        recv = torch.empty(self.world_size * send.numel(), dtype=torch.float64)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        parts = recv.numpy().reshape((self.world_size,) + stats.shape)
        if op == 'max':
            return parts.max(axis=0)
        total = np.zeros_like(stats)
        for r in range(self.world_size):
            total += parts[r]
        return total

    def barrier(self):
        import torch.distributed as dist
        dist.barrier(group=self.group)
