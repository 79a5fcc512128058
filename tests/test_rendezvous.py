"""The control plane of the rank processes (multigrid_nikhil_c-_amd/rendezvous.py): a TCP key-value
store hosted by rank 0 and the collectives bench.py needs on it - no torch, deadlines on every wait."""
import multiprocessing as mp
import os
import socket
import struct
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _rank(rank, world, port, ret, mode):
    os.environ["MGX_RDZV_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.load_package()
    from multigrid_nikhil_c_amd.rendezvous import RendezvousError, Store

    if mode == "absent" and rank == world - 1:
        return                                           # this rank never shows up
    try:
        st = Store(rank, world, timeout=3.0 if mode != "ok" else 60.0)
    except RendezvousError as e:
        ret[f"err{rank}"] = str(e)
        return
    try:
        ret[f"bcast{rank}"] = st.broadcast(b"x" * 128 if rank == 0 else None)
        ret[f"gather{rank}"] = [bytes(v) for v in st.allgather(bytes([rank]) * (rank + 1))]
        ret[f"max{rank}"] = st.allreduce_max(float(rank) * 1.5)
        ret[f"sum{rank}"] = st.allreduce_sum(0.1 * (rank + 1))
        ret[f"n{rank}"] = st.allreduce_sum_int(1)
        # neighbour exchange, both directions, twice (what a halo exchange does)
        for it in range(2):
            if rank > 0:
                st.send(rank - 1, struct.pack("<ii", rank, it))
            if rank < world - 1:
                st.send(rank + 1, struct.pack("<ii", rank, it))
            got = []
            if rank > 0:
                got.append(struct.unpack("<ii", st.recv(rank - 1)))
            if rank < world - 1:
                got.append(struct.unpack("<ii", st.recv(rank + 1)))
            ret[f"p2p{rank}_{it}"] = got
        big = os.urandom(3 << 20) if rank == 0 else None             # a halo-sized message
        ret[f"big{rank}"] = len(st.broadcast(big))
        if mode == "dies" and rank == 1:
            os._exit(5)
        try:
            st.barrier()
            ret[f"done{rank}"] = True
        except RendezvousError as e:
            ret[f"err{rank}"] = str(e)
    finally:
        if not (mode == "dies"):
            st.close()


def _run(world, mode):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, ret, mode)) for r in range(world)]
    t0 = time.monotonic()
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
    return dict(ret), [p.exitcode for p in procs], time.monotonic() - t0


@pytest.mark.parametrize("world", [2, 4])
def test_collectives_over_the_store(world):
    ret, codes, _ = _run(world, "ok")
    assert codes == [0] * world
    for r in range(world):
        assert ret[f"bcast{r}"] == b"x" * 128
        assert ret[f"gather{r}"] == [bytes([q]) * (q + 1) for q in range(world)]
        assert ret[f"max{r}"] == 1.5 * (world - 1)
        assert ret[f"sum{r}"] == ret["sum0"] and abs(ret[f"sum{r}"] - 0.1 * world * (world + 1) / 2) < 1e-12
        assert ret[f"n{r}"] == world and ret[f"big{r}"] == 3 << 20 and ret[f"done{r}"]
        for it in range(2):
            want = ([(r - 1, it)] if r > 0 else []) + ([(r + 1, it)] if r < world - 1 else [])
            assert ret[f"p2p{r}_{it}"] == want


def test_a_missing_rank_times_out_instead_of_hanging():
    ret, codes, secs = _run(3, "absent")
    assert secs < 60
    assert "err0" in ret and "err1" in ret and "timed out" in ret["err0"]


def test_a_rank_that_dies_mid_job_releases_the_others():
    ret, codes, secs = _run(3, "dies")
    assert codes[1] == 5 and secs < 60
    assert "err0" in ret and "err2" in ret          # the survivors' barrier ends with an error, not a hang
