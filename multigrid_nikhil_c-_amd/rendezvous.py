"""Control plane of the one-process-per-GPU jobs (bench.py --gpus N, the rank-process tests): a small
key-value store over TCP on one node, and the handful of collectives the job needs on top of it -
the 128 bytes of rank 0's ncclUniqueId to every rank, barriers around the timed region, the
max over ranks of the elapsed time.  Pure sockets, no torch: a rank process must run libmgx on the
ROCm stack it was built against (/opt/rocm, via the library's RUNPATH), and importing the torch wheel
first would put the wheel's own libamdhip64 / libhsa-runtime64 / librccl (same SONAMEs) under it.
The data plane is RCCL inside libmgx (csrc/mgx_dist.hpp); nothing here moves grid data except the
host-staged rehearsal transport of transport.py.

Rendezvous: rank 0 hosts the store.  The launcher may name its port (MGX_RDZV_PORT); under
`python -m torch.distributed.run` MASTER_PORT belongs to the launcher's own store, so rank 0 binds an
ephemeral port and publishes it in a file named after the launcher's pid and MASTER_PORT (every
rank of a job is a child of the same launcher process on this one node).

Every blocking call has a deadline (default 600 s, MGX_RDZV_TIMEOUT: a cold start of the first rank - libraries paged in, a 16384^2 reference run - must fit): a rank that died never makes
the others wait for the launcher's outer timeout."""
from __future__ import annotations

import os
import socket
import struct
import tempfile
import threading
import time

_SET, _GET, _ADD = 1, 2, 3


class RendezvousError(RuntimeError):
    pass


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(sock: socket.socket, op: int, key: bytes, val: bytes) -> None:
    sock.sendall(struct.pack("<BIQ", op, len(key), len(val)) + key + val)


def _recv_msg(sock: socket.socket):
    op, klen, vlen = struct.unpack("<BIQ", _recv_exact(sock, 13))
    key = _recv_exact(sock, klen) if klen else b""
    val = _recv_exact(sock, vlen) if vlen else b""
    return op, key, val


class _Server(threading.Thread):
    """rank 0's store: SET key value / GET key (answers once the key exists) / ADD key delta"""

    def __init__(self, port: int):
        super().__init__(daemon=True)
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.sock.bind(("127.0.0.1", port))
        self.sock.listen(64)
        self.port = self.sock.getsockname()[1]
        self.data: dict[bytes, bytes] = {}
        self.cv = threading.Condition()
        self.stop = False

    def run(self):
        while not self.stop:
            try:
                conn, _ = self.sock.accept()
            except OSError:
                return
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            threading.Thread(target=self._serve, args=(conn,), daemon=True).start()

    def _serve(self, conn: socket.socket):
        try:
            while True:
                op, key, val = _recv_msg(conn)
                if op == _SET:
                    with self.cv:
                        self.data[key] = val
                        self.cv.notify_all()
                    _send_msg(conn, op, b"", b"")
                elif op == _GET:
                    (timeout,) = struct.unpack("<d", val)
                    deadline = time.monotonic() + timeout
                    with self.cv:
                        while key not in self.data and not self.stop:
                            left = deadline - time.monotonic()
                            if left <= 0:
                                break
                            self.cv.wait(min(left, 1.0))
                        found = self.data.get(key)
                    _send_msg(conn, op, b"1" if found is not None else b"0", found or b"")
                elif op == _ADD:
                    (delta,) = struct.unpack("<q", val)
                    with self.cv:
                        cur = struct.unpack("<q", self.data.get(key, struct.pack("<q", 0)))[0] + delta
                        self.data[key] = struct.pack("<q", cur)
                        self.cv.notify_all()
                    _send_msg(conn, op, b"", struct.pack("<q", cur))
        except (ConnectionError, OSError, struct.error):
            pass
        finally:
            conn.close()

    def close(self):
        self.stop = True
        with self.cv:
            self.cv.notify_all()
        try:
            self.sock.close()
        except OSError:
            pass


def _rdzv_file() -> str:
    tag = f"{os.getuid()}_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
    return os.path.join(tempfile.gettempdir(), f"mgx_rdzv_{tag}")


class Store:
    """One rank's view of the job's store; rank 0 also hosts it.  Collectives are built from SET / GET
    with a per-rank sequence number: every rank must call them in the same order."""

    def __init__(self, rank: int | None = None, world: int | None = None, timeout: float | None = None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.timeout = float(os.environ.get("MGX_RDZV_TIMEOUT", "600")) if timeout is None else timeout
        self._seq = 0
        self._p2p: dict[tuple[int, int], int] = {}
        self._server = None
        self._file = None
        port_env = os.environ.get("MGX_RDZV_PORT")
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            self._server = _Server(int(port_env) if port_env else 0)
            self._server.start()
            port = self._server.port
            if not port_env:
                self._file = _rdzv_file()
                tmp = self._file + f".{os.getpid()}"
                with open(tmp, "w") as fh:
                    fh.write(str(port))
                os.replace(tmp, self._file)          # atomic: a reader sees the whole port or no file
        self._sock = None
        last = None
        while self._sock is None:
            try:
                port = int(port_env) if port_env else int(open(_rdzv_file()).read())
                s = socket.create_connection(("127.0.0.1", port), timeout=5.0)
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                s.settimeout(self.timeout + 10.0)
                self._sock = s
            except (OSError, ValueError) as e:      # rank 0 not up yet (or a stale file): try again
                last = e
                if time.monotonic() > deadline:
                    raise RendezvousError(f"rank {self.rank}: no store to connect to after {self.timeout:.0f} s ({last})")
                time.sleep(0.05)
        # everybody is here (and talks to THIS job's store: a stale file's port would not count to `world`)
        self.barrier()

    # -- primitives ---------------------------------------------------------------------------
    def set(self, key: str, val: bytes) -> None:
        _send_msg(self._sock, _SET, key.encode(), bytes(val))
        _recv_msg(self._sock)

    def get(self, key: str, timeout: float | None = None) -> bytes:
        t = self.timeout if timeout is None else timeout
        _send_msg(self._sock, _GET, key.encode(), struct.pack("<d", t))
        try:
            _, ok, val = _recv_msg(self._sock)
        except (ConnectionError, OSError) as e:
            raise RendezvousError(f"rank {self.rank}: store connection lost while waiting for {key!r} ({e})")
        if ok != b"1":
            raise RendezvousError(f"rank {self.rank}: timed out after {t:.0f} s waiting for {key!r} (a rank died?)")
        return val

    def add(self, key: str, delta: int) -> int:
        _send_msg(self._sock, _ADD, key.encode(), struct.pack("<q", delta))
        return struct.unpack("<q", _recv_msg(self._sock)[2])[0]

    # -- collectives ---------------------------------------------------------------------------
    def allgather(self, val: bytes) -> list[bytes]:
        self._seq += 1
        self.set(f"c{self._seq}/{self.rank}", val)
        return [self.get(f"c{self._seq}/{r}") for r in range(self.world)]

    def barrier(self) -> None:
        self.allgather(b"")

    def broadcast(self, val: bytes | None, src: int = 0) -> bytes:
        self._seq += 1
        if self.rank == src:
            self.set(f"c{self._seq}/b", val)
        return self.get(f"c{self._seq}/b")

    def allreduce_max(self, x: float) -> float:
        return max(struct.unpack("<d", v)[0] for v in self.allgather(struct.pack("<d", x)))

    def allreduce_sum(self, x: float) -> float:
        # rank order: the same sum on every rank
        return sum(struct.unpack("<d", v)[0] for v in self.allgather(struct.pack("<d", x)))

    def allreduce_sum_int(self, x: int) -> int:
        return sum(struct.unpack("<q", v)[0] for v in self.allgather(struct.pack("<q", x)))

    # -- point to point (host-staged rehearsal transport) --------------------------------------------
    def send(self, peer: int, val: bytes) -> None:
        k = self._p2p.get((self.rank, peer), 0)
        self._p2p[(self.rank, peer)] = k + 1
        self.set(f"p/{self.rank}>{peer}/{k}", val)

    def recv(self, peer: int) -> bytes:
        k = self._p2p.get((peer, self.rank), 0)
        self._p2p[(peer, self.rank)] = k + 1
        return self.get(f"p/{peer}>{self.rank}/{k}")

    def close(self) -> None:
        try:
            if self._sock is not None:
                try:
                    # leave together: rank 0 must not tear the store down under a slower rank
                    self.barrier()
                    if self.rank != 0:
                        self.add("bye", 1)
                    else:
                        end = time.monotonic() + 5.0
                        while self.add("bye", 0) < self.world - 1 and time.monotonic() < end:
                            time.sleep(0.01)
                except Exception:      # noqa: BLE001 - a dead peer must not keep us from closing
                    pass
                self._sock.close()
        finally:
            self._sock = None
            if self._server is not None:
                self._server.close()
                self._server = None
            if self._file:
                try:
                    os.unlink(self._file)
                except OSError:
                    pass
                self._file = None
