"""An mgx_transport (include/mgx.h) with host-staged buffers over the job's TCP store (rendezvous.py).

The multi-GPU driver is C++ (csrc/mgx_dist.hpp): with one process per GPU its built-in transport is
RCCL (ncclSend / ncclRecv / ncclAllGather / ncclAllReduce on the slab's stream).  RCCL refuses two
ranks on one device, so the 1-GPU development box cannot run it with more than one rank; this
module is the stand-in used there, and ONLY on explicit request (`MGX_DIST_BACKEND=staged`: bench.py's
rehearsal mode and tests/test_gpu_dist.py) - device rows are copied to the host (mgx_memcpy_d2h), moved
through the store and copied back.  It exercises the same C++ executor, plans and slab kernels as the
RCCL transport; only the wire differs.  It is also the shape of what an MPI host application would
plug in.  No torch: a rank process runs libmgx on the ROCm stack it was built against.
"""
from __future__ import annotations

import ctypes as C
import struct

from . import binding as B


class StagedTransport:
    """callbacks for mgx_create_rank(transport=...); keep the object alive as long as the handle"""

    def __init__(self, store):
        self.store = store
        self.world = store.world
        self.rank = store.rank
        self.calls = {"sendrecv": 0, "allgather": 0, "allreduce": 0}
        self._cb = (B.SENDRECV_FN(self._sendrecv), B.ALLGATHER_FN(self._allgather), B.ALLREDUCE_FN(self._allreduce))
        self.struct = B.Transport(None, *self._cb)

    # int (*sendrecv)(void* ctx, int n, const mgx_xfer* x, void* stream)
    def _sendrecv(self, ctx, n, x, stream):
        try:
            L = B.lib()
            for i in range(n):                       # all sends first: nobody waits for a message not yet posted
                if x[i].send:
                    buf = C.create_string_buffer(x[i].bytes)
                    if L.mgx_memcpy_d2h(buf, x[i].ptr, x[i].bytes, stream) != 0:
                        return 1
                    self.store.send(x[i].peer, buf.raw)
            for i in range(n):
                if not x[i].send:
                    data = self.store.recv(x[i].peer)
                    if len(data) != x[i].bytes or L.mgx_memcpy_h2d(x[i].ptr, data, x[i].bytes, stream) != 0:
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
            mine = C.create_string_buffer(nbytes)
            if L.mgx_memcpy_d2h(mine, send, nbytes, stream) != 0:
                return 1
            full = b"".join(self.store.allgather(mine.raw))
            if len(full) != nbytes * self.world or L.mgx_memcpy_h2d(recv, full, len(full), stream) != 0:
                return 1
            self.calls["allgather"] += 1
            return 0
        except Exception as e:
            print(f"StagedTransport.allgather: {e!r}", flush=True)
            return 1

    # int (*allreduce_sum)(void* ctx, double* value)
    def _allreduce(self, ctx, value):
        try:
            value[0] = self.store.allreduce_sum(float(value[0]))
            self.calls["allreduce"] += 1
            return 0
        except Exception as e:
            print(f"StagedTransport.allreduce: {e!r}", flush=True)
            return 1


def broadcast_rccl_id(store) -> bytes:
    """rank 0 creates the ncclUniqueId (mgx_rccl_unique_id), every rank receives its 128 bytes"""
    return store.broadcast(B.rccl_unique_id() if store.rank == 0 else None, src=0)
