"""world_size 2 / 3 tests (CPU, no GPU) of the rank layer of libhemocell_amd.so (csrc/comm.hip): the TCP control plane
that boots a multi-GPU run and carries its barriers and reductions, and the neighbour routing rule that the data plane
(RCCL ncclSend / ncclRecv, or the mesh itself for ranks sharing a GPU) uses for lattice faces and particle records.
One of the tests cross-checks routing and reductions against torch.distributed's gloo backend."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port(salt):
    return 31000 + (os.getpid() * 7 + salt * 131) % 20000


def _lib():
    sys.path.insert(0, ROOT)
    from hemocell_amd import capi
    return capi.lib(), capi.check


def _exchange(lib, check, periodic, s_lo, s_hi, m_lo, m_hi):
    r_lo, r_hi = np.full(m_lo, -1.0), np.full(m_hi, -1.0)
    check(lib.hc_comm_exchange_host(int(periodic), s_lo.ctypes.data, s_lo.nbytes, s_hi.ctypes.data, s_hi.nbytes,
                                    r_lo.ctypes.data, r_lo.nbytes, r_hi.ctypes.data, r_hi.nbytes))
    return r_lo, r_hi


def _mesh_worker(rank, world, port, periodic, q):
    try:
        lib, check = _lib()
        check(lib.hc_comm_init(rank, world, rank, b"127.0.0.1", port, 0, 0))   # control plane only, no device
        r, w, t = C.c_int(), C.c_int(), C.c_int()
        check(lib.hc_comm_info(C.byref(r), C.byref(w), C.byref(t)))
        assert (r.value, w.value, t.value) == (rank, world, 0)
        # ---- neighbour routing: my low-face message must arrive as the low neighbour's high-halo message.  Messages are
        # large enough (8 MB) that both directions have to make progress together, and of different length per side.
        n_lo, n_hi = 1_000_003, 600_001
        s_lo = rank * 10.0 + 1.0 + np.arange(n_lo) * 1e-7      # "face" data tagged with the sender and the side
        s_hi = rank * 10.0 + 2.0 + np.arange(n_hi) * 1e-7
        r_lo, r_hi = _exchange(lib, check, periodic, s_lo, s_hi, n_hi, n_lo)
        lo = (rank - 1) % world if (periodic or rank > 0) else None
        hi = (rank + 1) % world if (periodic or rank < world - 1) else None
        if lo is not None:
            assert np.array_equal(r_lo, lo * 10.0 + 2.0 + np.arange(n_hi) * 1e-7), "low halo must hold the low neighbour's HIGH face"
        else:
            assert (r_lo == -1.0).all()
        if hi is not None:
            assert np.array_equal(r_hi, hi * 10.0 + 1.0 + np.arange(n_lo) * 1e-7), "high halo must hold the high neighbour's LOW face"
        else:
            assert (r_hi == -1.0).all()
        # ---- a ragged round: zero-length messages are skipped on both sides (empty envelope records)
        e = np.zeros(0)
        a, b = _exchange(lib, check, periodic, e, e, 0, 0)
        assert a.size == 0 and b.size == 0
        # ---- reductions are folded in rank order on rank 0: bit-identical on every rank
        v = np.array([0.1 * (rank + 1), float(rank), -float(rank)])
        s = v.copy(); check(lib.hc_comm_allreduce(s.ctypes.data_as(C.POINTER(C.c_double)), 3, 0))
        mn = v.copy(); check(lib.hc_comm_allreduce(mn.ctypes.data_as(C.POINTER(C.c_double)), 3, 1))
        mx = v.copy(); check(lib.hc_comm_allreduce(mx.ctypes.data_as(C.POINTER(C.c_double)), 3, 2))
        acc = 0.1
        for k in range(1, world):
            acc = acc + 0.1 * (k + 1)
        assert s[0] == acc and s[1] == sum(range(world)) and mn[2] == -(world - 1) and mx[1] == world - 1
        # ---- broadcast (how the RCCL unique id reaches the ranks)
        blob = np.arange(128, dtype=np.uint8) if rank == 0 else np.zeros(128, dtype=np.uint8)
        check(lib.hc_comm_bcast(blob.ctypes.data, 128, 0))
        assert np.array_equal(blob, np.arange(128, dtype=np.uint8))
        # ---- all-gather of a placement-sized block (hcp_slab_sync_placement gathers (type, id) pairs of rejected cells: with the
        # 15 500 cells of an 8 x 256^3 pipe that is up to 248 KB per rank), every rank sees every block in rank order
        mine = np.full(31_000, rank, dtype=np.int64) + np.arange(31_000) * 1000
        everyone = np.zeros(31_000 * world, dtype=np.int64)
        check(lib.hc_comm_allgather(mine.ctypes.data, mine.nbytes, everyone.ctypes.data))
        for k in range(world):
            assert np.array_equal(everyone[k * 31_000:(k + 1) * 31_000], np.full(31_000, k, dtype=np.int64) + np.arange(31_000) * 1000)
        check(lib.hc_comm_barrier())
        check(lib.hc_comm_finalize())
        q.put((rank, "ok"))
    except BaseException as e:   # noqa: BLE001 -- report to the parent instead of hanging the peers
        q.put((rank, "FAILED: %r" % (e,)))


@pytest.mark.parametrize("world,periodic", [(2, True), (3, True), (2, False), (3, False), (8, True)])   # 8: the node the scaling runs use
def test_mesh_routing_and_reductions(world, periodic):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port(world * 2 + periodic)
    ps = [ctx.Process(target=_mesh_worker, args=(r, world, port, periodic, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(30)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _gloo_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        lib, check = _lib()
        os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
        os.environ["HEMOCELL_PORT"] = str(port + 40)
        check(lib.hc_comm_init(rank, world, rank, b"127.0.0.1", port + 40, 0, 0))
        rng = np.random.default_rng(100 + rank)
        s_lo, s_hi = rng.standard_normal(4096), rng.standard_normal(4096)
        r_lo, r_hi = _exchange(lib, check, True, s_lo, s_hi, 4096, 4096)
        # the same ring exchange over gloo point-to-point
        lo, hi = (rank - 1) % world, (rank + 1) % world
        g_lo, g_hi = torch.empty(4096, dtype=torch.float64), torch.empty(4096, dtype=torch.float64)
        ops = [dist.P2POp(dist.isend, torch.from_numpy(s_lo), lo, tag=1), dist.P2POp(dist.isend, torch.from_numpy(s_hi), hi, tag=2),
               dist.P2POp(dist.irecv, g_hi, hi, tag=1), dist.P2POp(dist.irecv, g_lo, lo, tag=2)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        assert np.array_equal(r_lo, g_lo.numpy()) and np.array_equal(r_hi, g_hi.numpy())
        v = rng.standard_normal(5)
        mine = v.copy(); check(lib.hc_comm_allreduce(mine.ctypes.data_as(C.POINTER(C.c_double)), 5, 2))
        t = torch.from_numpy(v.copy()); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert np.array_equal(mine, t.numpy())
        mine = v.copy(); check(lib.hc_comm_allreduce(mine.ctypes.data_as(C.POINTER(C.c_double)), 5, 0))
        t = torch.from_numpy(v.copy()); dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert np.allclose(mine, t.numpy(), rtol=1e-14, atol=0)
        check(lib.hc_comm_finalize())
        dist.barrier(); dist.destroy_process_group()
        q.put((rank, "ok"))
    except BaseException as e:   # noqa: BLE001
        q.put((rank, "FAILED: %r" % (e,)))


def test_mesh_agrees_with_gloo_world3():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 3) % 300
    ps = [ctx.Process(target=_gloo_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in ps]
    for p in ps:
        p.join(30)
    assert sorted(res) == [(r, "ok") for r in range(3)], res


def test_env_bootstrap_single_rank_is_a_no_op():
    lib, check = _lib()
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HEMOCELL_WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE") if k in os.environ}
    try:
        check(lib.hc_comm_init_env())
        r, w, t = C.c_int(7), C.c_int(7), C.c_int(7)
        check(lib.hc_comm_info(C.byref(r), C.byref(w), C.byref(t)))
        assert (r.value, w.value, t.value) == (0, 1, 0)
    finally:
        os.environ.update(env)
