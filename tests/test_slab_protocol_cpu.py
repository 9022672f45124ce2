"""world_size-2/3 gloo tests (CPU) of the slab exchange protocol in hemocell_amd/exchange.py: routing of
the population halos (5 populations per face / full double planes, periodic seam included) and the particle
envelope synchronisation (replication near faces, ownership merge, migration, seam shift, dropping)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _speeds(periodic):
    # without periodicity the two cells next to the pipe ends move inward (nobody owns a vertex outside)
    return {0: 0.9, 1: 0.9, 2: -0.7, 3: 0.9 if periodic else -0.7, 4: -0.7 if periodic else 0.9, 5: 0.9}


def _worker(rank, world, port, periodic, steps, k_p, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_engine import CX, FakeEngine
    from hemocell_amd.exchange import NeighbourComm, SlabProtocol
    nx, plane, nv = 48, 3, 5
    nxg = nx * world
    e = FakeEngine(nx, rank * nx, nxg, plane, nv)
    # fluid: unique value per (q, global x, p) in the stored (post-collision) field
    for q in range(19):
        for x in range(nx):
            e.f[0][q, x + 2] = 1000.0 * q + (rank * nx + x) + 0.1 * np.arange(plane)
    # cells: ids 0..5 spread along the pipe, some near faces; speed +0.9 or -0.7 lu/step
    centres = {0: 10.0, 1: nx - 3.0, 2: nx + 2.5, 3: nxg - (2.0 if periodic else 5.0), 4: 1.5 if periodic else 4.0, 5: nxg / 2 + 7.0}
    speeds = _speeds(periodic)
    e.cell_speed = speeds
    offs = np.linspace(-2.0, 2.0, nv)
    x0, x1 = rank * nx, (rank + 1) * nx
    ids, pos = [], []
    for cid, c in centres.items():
        for shift in ((0.0, -nxg, nxg) if periodic else (0.0,)):
            xs = c + shift + offs
            owned = ((np.floor(xs + 0.5) >= x0) & (np.floor(xs + 0.5) < x1)).any()
            if owned or (xs.max() >= x0 - 4.0 and xs.min() < x1 + 4.0):
                p = np.zeros((nv, 3)); p[:, 0] = xs; p[:, 1] = cid
                ids.append(cid); pos.append(p); break
    e.ids = np.array(ids, np.int64); e.pos = np.array(pos).reshape(-1, nv, 3)
    e.vel = np.zeros_like(e.pos); e.frc = np.zeros_like(e.pos)
    comm = NeighbourComm(rank, world, periodic)
    proto = SlabProtocol(e, comm, k_p, nxg, periodic)
    proto.prepare()
    total_owned = []
    for k in range(steps):
        proto.step(more=k + 1 < steps)
        t = torch.tensor([e.owned_vertices()]); dist.all_reduce(t); total_owned.append(int(t))
        assert not np.isnan(e.pos).any(), "a vertex advanced with a velocity nobody owned"
    proto.halo_exchange_begin(2)()   # the post-stream view pulls from the halo planes: refresh them first
    res = dict(rank=rank, S=e.post_stream(), ids=e.ids.copy(), pos=e.pos.copy(), owned=e._owned(), total_owned=total_owned,
               stats=dict(proto.stats))
    torch.save(res, os.path.join(out, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,periodic,k_p", [(2, True, 1), (2, True, 3), (3, True, 2), (2, False, 1)])
def test_slab_protocol_gloo(tmp_path, world, periodic, k_p):
    steps = 40
    port = 29600 + (os.getpid() + world * 7 + k_p) % 300
    mp.spawn(_worker, args=(world, port, periodic, steps, k_p, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "r%d.pt" % r), weights_only=False) for r in range(world)]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_engine import CX
    nx, plane, nv = 48, 3, 5
    nxg = nx * world
    # ---- fluid: after n steps the post-stream value at global x for population q started at x - (n+1) c_q
    S = np.concatenate([r["S"] for r in res], axis=1)  # [19][nxg][plane]
    for q in range(19):
        for x in range(nxg):
            src = x - (steps + 1) * CX[q]
            if periodic:
                src %= nxg
                expect = 1000.0 * q + src + 0.1 * np.arange(plane)
            else:
                expect = (1000.0 * q + src + 0.1 * np.arange(plane)) if 0 <= src < nxg else np.zeros(plane)
            assert np.allclose(S[q, x], expect), (q, x, S[q, x], expect)
    # ---- cells
    speeds = _speeds(periodic)
    centres = {0: 10.0, 1: nx - 3.0, 2: nx + 2.5, 3: nxg - (2.0 if periodic else 5.0), 4: 1.5 if periodic else 4.0, 5: nxg / 2 + 7.0}
    offs = np.linspace(-2.0, 2.0, nv)
    if periodic:
        # every vertex is owned exactly once at every step
        for r in res:
            assert all(t == 6 * nv for t in r["total_owned"]), r["total_owned"]
    seen = {}
    for r in res:
        for s, cid in enumerate(r["ids"]):
            for i in range(nv):
                if r["owned"][s, i]:
                    seen.setdefault(int(cid), {})[i] = r["pos"][s, i, 0]
    # velocities are refreshed every k_p steps; a vertex moves with its cell's speed from step 0 on
    for cid, c in centres.items():
        expect = c + offs + steps * speeds[cid]
        got = seen.get(cid, {})
        if periodic:
            assert len(got) == nv, (cid, got)
        for i, x in got.items():
            d = x - expect[i]
            if periodic:
                d = (d + nxg / 2) % nxg - nxg / 2
            assert abs(d) < 1e-9, (cid, i, x, expect[i])
    assert sum(r["stats"]["cells_new"] for r in res) > 0 and sum(r["stats"]["cells_dropped"] for r in res) > 0
