"""two ranks (two processes sharing the one GPU of the test box, gloo transport) vs one rank: the slab
decomposition with halo exchange and particle envelopes must reproduce the single-domain run."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NXG, NY, NZ = 144, 34, 34
CELLS = [((10.0, 16.5, 16.5), (90, 0, 0)), ((46.5, 16.0, 17.0), (80, 20, 10)), ((141.0, 17.0, 16.0), (90, 0, 30)),
         ((70.0, 12.0, 20.0), (10, 20, 30)), ((96.5, 17.5, 16.0), (90, 10, 0)), ((57.5, 21.0, 14.0), (90, 0, 0))]
PLTS = [((47.0, 12.0, 21.0), (20, 40, 10)), ((120.0, 20.0, 13.0), (0, 0, 0)), ((143.5, 15.0, 21.5), (70, 0, 30))]   # platelets, two at faces
# fast flow: the cells travel ~6 lu in 250 iterations, so envelope copies are created and dropped at the faces.
# (Beyond ~300 iterations this strongly driven case amplifies a 1e-13 perturbation to 1e-5 even on a single
# domain, so longer runs cannot be compared position by position.)
STEPS, K_P, K_M = 250, 2, 4
FORCE = (3e-4, 0.0, 0.0)


REPULSION = dict(k=2e-6, cutoff_um=0.7, k_b=3e-6, b_cutoff_um=1.0)   # examples/pipeflow/config.xml:36-38 magnitudes


def _build(rank, world, rep=False, padded=False):
    from hemocell_amd import host
    from hemocell_amd.slab import SlabRunner
    host.check(host.capi.lib().hc_debug_force_plane_padding(1 if padded else 0))   # padded x-plane stride (hc_lattice::xs)
    P = host.base_parameters()
    r = SlabRunner(NXG // world, NY, NZ, rank, world, P, periodic=(True, False, False), particle_timescale=K_P,
                   material_timescale=K_M, deletion_check_every=1000000)
    mask, _ = host.pipe_mask(NXG, NY, NZ)
    r.define_bounce_back(mask)
    r.lattice.latticeEquilibrium(1.0, (0, 0, 0))
    r.lattice.setExternalVector(FORCE)
    r.add_cell_type(host.CellType.rbc(P))
    r.add_cell_type(host.CellType.plt(P))
    r.load_cells(0, [np.array(c) for c, _ in CELLS], [np.array(a) for _, a in CELLS])
    if world > 1:
        r.exchange.load_cells(1, [np.array(c) for c, _ in PLTS], [np.array(a) for _, a in PLTS], radius=3.0)
    else:
        for i, (c, a) in enumerate(PLTS):
            assert r.cells.addCell(1, np.array(c), np.array(a), cell_id=i)
    if rep:
        r.cells.setRepulsion(REPULSION["k"], REPULSION["cutoff_um"], K_P)
        r.cells.enableBoundaryParticles(REPULSION["k_b"], REPULSION["b_cutoff_um"], K_P)
    r.prepare()
    return r, mask


def _worker(rank, world, port, out, rep=False, padded=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from hemocell_amd import host
    host.init(0)
    r, _ = _build(rank, world, rep, padded)
    r.run(STEPS)
    cid, vid, pos = r.owned_vertex_table(0)
    pcid, pvid, ppos = r.owned_vertex_table(1)
    gstats = (r.fluid_stats(0), r.vertex_stats(1), r.vertex_stats(2))    # all-reduced: every rank gets the global numbers
    torch.save(dict(f=r.populations(), cid=cid, vid=vid, pos=pos, pcid=pcid, pvid=pvid, ppos=ppos, held=r.cells.counts()[1],
                    stats=r.exchange.protocol.stats, gstats=gstats),
               os.path.join(out, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _initial_x():
    from hemocell_amd import host
    r0, _ = _build(0, 1)
    x = r0.cells.positions[:, 0].copy()
    r0.lattice.destroy()
    return x


@pytest.mark.parametrize("world,rep,padded", [(2, False, False), (3, False, False), (2, True, False), (2, True, True)])
def test_slabs_match_single_domain(tmp_path, gpu, world, rep, padded):
    """rep: vertex-vertex and boundary-particle repulsion on (cell records then carry force_repulsion);
    padded: the slabs (not the single-domain reference run) use the padded x-plane stride"""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + 17 * world + 7 * rep + 3 * padded) % 400
    mp.spawn(_worker, args=(world, port, str(tmp_path), rep, padded), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "r%d.pt" % k), weights_only=False) for k in range(world)]
    ref, mask = _build(0, 1, rep)
    ref.run(STEPS)
    if rep:
        assert np.abs(ref.cells.repulsion_forces).max() > 0   # the repulsions do act in this case
    f_ref = ref.lattice.populations().reshape(NXG, NY * NZ, 19)
    f_two = np.concatenate([r["f"].reshape(NXG // world, NY * NZ, 19) for r in res], axis=0)
    fluid = (mask.reshape(NXG, NY * NZ) == 0)
    err_f = np.abs(f_two - f_ref)[fluid].max()
    assert err_f <= 1e-10, err_f
    allpos = ref.cells.positions
    nrbc = len(CELLS) * 642
    for key, p_ref in ((("cid", "vid", "pos"), allpos[:nrbc].reshape(len(CELLS), -1, 3)), (("pcid", "pvid", "ppos"), allpos[nrbc:].reshape(len(PLTS), -1, 3))):
        seen = np.zeros(p_ref.shape[:2], dtype=int)
        for r in res:
            for c, v, p in zip(r[key[0]], r[key[1]], r[key[2]]):
                d = p - p_ref[c, v]
                d[0] = (d[0] + NXG / 2) % NXG - NXG / 2
                assert np.abs(d).max() <= 1e-8, (key, c, v, d)
                seen[c, v] += 1
        assert (seen == 1).all()          # every vertex owned by exactly one rank
    assert sum(r["held"] for r in res) > len(CELLS) + len(PLTS)   # cells near the faces are replicated
    n_new, n_drop = sum(r["stats"]["cells_new"] for r in res), sum(r["stats"]["cells_dropped"] for r in res)
    assert n_new + n_drop > 0                      # envelope copies changed hands during the run
    if world == 2:
        assert n_new > 0 and n_drop > 0            # both a fresh copy and a dropped copy (faces at x = 72 and the seam)
    # diagnostics all-reduced over the slabs equal those of the single domain (same values on every rank)
    ref_stats = (ref.fluid_stats(0), ref.vertex_stats(1), ref.vertex_stats(2))
    for r in res:
        for got, want in zip(r["gstats"], ref_stats):
            assert got[3] == want[3]                                          # same number of nodes / owned vertices
            assert np.allclose(got[:3], want[:3], rtol=1e-6, atol=1e-18), (got, want)
    travelled = np.abs(allpos[:, 0] - _initial_x()).max()
    assert travelled > 5.0, travelled


def test_slab_protocol_over_rccl_matches_hc_iterate(gpu):
    """the real transport of N > 1 runs (torch.distributed backend nccl = RCCL) on the one GPU of the test box: a single
    rank that is its own periodic neighbour sends to and receives from itself; examples/rccl_selfloop.py asserts that
    the slab protocol then reproduces hc_iterate (own process: the process group and the stream binding stay there)"""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "rccl_selfloop.py"), "128", "30"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "max |df|" in r.stdout and "protocol over RCCL" in r.stdout
