"""Run from the repository root: python examples/loopback_protocol_overhead.py

protocol overhead on one GPU: the slab protocol with a loopback transport (the slab is its own periodic
neighbour; cells are kept away from the seam so that no record crosses) against hc_iterate on the same case"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hemocell_amd import host
from hemocell_amd.packing import pack_pipe_rbc
from hemocell_amd import exchange as X
host.init(0); lib = host.capi.lib()
n = 256
P = host.base_parameters()
mask, R = host.pipe_mask(n, n, n)
centres, angles = pack_pipe_rbc(n, n, n, 0.10)
keep = (centres[:, 0] > 24) & (centres[:, 0] < n - 24)
centres, angles = centres[keep], angles[keep]

def build(n_slabs):
    L = host.Lattice(n, n, n, (1, 0, 0), 1 / P.tau, x0=0, nx_global=n, n_slabs=n_slabs)
    L.defineBounceBack(mask); L.latticeEquilibrium(); L.setExternalVector((2e-6, 0, 0))
    h = host.HemoCell(L, P); h.cellfields.addCellType(host.CellType.rbc(P), 20)
    h.setParticleVelocityUpdateTimeScaleSeparation(5); h.deletion_check_every = 10**6
    for i, (c, a) in enumerate(zip(centres, angles)): h.cellfields.addCell(0, c, a, cell_id=i)
    h.cellfields.applyConstitutiveModel(0, True)
    return L, h

class Loopback:
    rank, world, lo, hi, backend = 0, 1, 0, 0, "loopback"
    def exchange(self, send_lo, send_hi, recv_lo, recv_hi):
        if send_lo is not None and send_lo.numel(): recv_hi.copy_(send_lo)   # my low face is my own high halo
        if send_hi is not None and send_hi.numel(): recv_lo.copy_(send_hi)
        return lambda: None

def timeit(fn, steps=100):
    fn(20); torch.cuda.synchronize(); lib.hc_synchronize()
    t0 = time.perf_counter(); fn(steps); torch.cuda.synchronize(); lib.hc_synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

L1, h1 = build(1)
t_iter = timeit(lambda k: h1.iterate(k))
h1.setParticleVelocityUpdateTimeScaleSeparation(10**9)
t_iter_nop = timeit(lambda k: h1.iterate(k))
L1.destroy()
L2, h2 = build(2)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); host.check(lib.hc_set_stream(stream.cuda_stream))
eng = X.HipEngine(L2, h2.cellfields, torch.device("cuda", 0))
proto = X.SlabProtocol(eng, Loopback(), 5, n, True)
proto.prepare()
t_full = timeit(lambda k: proto.run(k))
real_sync = proto.sync_cells
proto.sync_cells = lambda: None
t_nosync = timeit(lambda k: proto.run(k))
proto.sync_cells = real_sync
proto.k_p = 10**9
t_nop = timeit(lambda k: proto.run(k))
print("hc_iterate: %.4f (k_p=5)  %.4f (no particle update)" % (t_iter, t_iter_nop))
print("protocol  : %.4f (k_p=5)  %.4f (k_p=5, sync_cells skipped)  %.4f (no particle update)" % (t_full, t_nosync, t_nop), flush=True)
