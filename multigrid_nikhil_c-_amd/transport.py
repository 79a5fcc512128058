"""An mgx_transport (include/mgx.h) over torch.distributed with host-staged buffers.

The multi-GPU driver is C++ (csrc/mgx_dist.hpp): with one process per GPU its built-in transport is
RCCL (ncclSend / ncclRecv / ncclAllGather / ncclAllReduce on the slab's stream).  RCCL refuses two
ranks on one device, so the 1-GPU development box cannot run it with more than one rank; this
module is the stand-in used there (`MGX_DIST_BACKEND=gloo`: bench.py's rehearsal mode and
tests/test_gpu_dist.py) - device rows are copied to the host (mgx_memcpy_d2h), moved with gloo and
copied back.  It exercises the same C++ executor, plans and slab kernels as the RCCL transport;
only the wire differs.  It is also the shape of what an MPI host application would plug in.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import binding as B


class StagedTransport:
    """callbacks for mgx_create_rank(transport=...); keep the object alive as long as the handle"""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.calls = {"sendrecv": 0, "allgather": 0, "allreduce": 0}
        self._cb = (B.SENDRECV_FN(self._sendrecv), B.ALLGATHER_FN(self._allgather), B.ALLREDUCE_FN(self._allreduce))
        self.struct = B.Transport(None, *self._cb)

    # int (*sendrecv)(void* ctx, int n, const mgx_xfer* x, void* stream)
    def _sendrecv(self, ctx, n, x, stream):
        try:
            L = B.lib()
            ops, recvs = [], []
            for i in range(n):
                t = torch.empty(x[i].bytes, dtype=torch.uint8)
                if x[i].send:
                    if L.mgx_memcpy_d2h(t.data_ptr(), x[i].ptr, x[i].bytes, stream) != 0:
                        return 1
                    ops.append(dist.P2POp(dist.isend, t, x[i].peer, self.group))
                else:
                    recvs.append((t, x[i].ptr, x[i].bytes))
                    ops.append(dist.P2POp(dist.irecv, t, x[i].peer, self.group))
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            for t, ptr, nbytes in recvs:
                if L.mgx_memcpy_h2d(ptr, t.data_ptr(), nbytes, stream) != 0:
                    return 1
            self.calls["sendrecv"] += 1
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            print(f"StagedTransport.sendrecv: {e!r}", flush=True)
            return 1

    # int (*allgather)(void* ctx, const void* send, void* recv, size_t bytes, void* stream)
    def _allgather(self, ctx, send, recv, nbytes, stream):
        try:
            L = B.lib()
            mine = torch.empty(nbytes, dtype=torch.uint8)
            if L.mgx_memcpy_d2h(mine.data_ptr(), send, nbytes, stream) != 0:
                return 1
            full = torch.empty(nbytes * self.world, dtype=torch.uint8)
            dist.all_gather_into_tensor(full, mine, group=self.group)
            if L.mgx_memcpy_h2d(recv, full.data_ptr(), full.numel(), stream) != 0:
                return 1
            self.calls["allgather"] += 1
            return 0
        except Exception as e:
            print(f"StagedTransport.allgather: {e!r}", flush=True)
            return 1

    # int (*allreduce_sum)(void* ctx, double* value)
    def _allreduce(self, ctx, value):
        try:
            t = torch.tensor([value[0]], dtype=torch.float64)
            dist.all_reduce(t, group=self.group)
            value[0] = float(t.item())
            self.calls["allreduce"] += 1
            return 0
        except Exception as e:
            print(f"StagedTransport.allreduce: {e!r}", flush=True)
            return 1


def broadcast_rccl_id(group=None) -> bytes:
    """rank 0 creates the ncclUniqueId (mgx_rccl_unique_id), every rank receives its 128 bytes
    (over the caller's control-plane process group, e.g. gloo)"""
    rank = dist.get_rank(group)
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.from_numpy(np.frombuffer(B.rccl_unique_id(), dtype=np.uint8).copy())
    dist.broadcast(t, src=0, group=group)
    return bytes(t.numpy().tobytes())
