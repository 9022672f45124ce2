"""Run from the repository root on a box with ONE GPU: python examples/rccl_selfloop.py [n]

The slab protocol over the real RCCL transport with a single rank: the slab is its own periodic neighbour, so every
halo / envelope message is an RCCL send to self + receive from self inside one group (torch.distributed backend
"nccl", the same NeighbourComm the N > 1 runs use).  Checks the result against hc_iterate on the same case and
reports the per-step cost next to it.  Cells are kept away from the seam so that no cell record crosses (a slab
cannot hold a cell and its own periodic image)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from hemocell_amd import exchange as X
from hemocell_amd import host
from hemocell_amd.packing import pack_pipe_rbc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
host.init(0); lib = host.capi.lib()
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
    for i, (c, a) in enumerate(zip(centres, angles)):
        h.cellfields.addCell(0, c, a, cell_id=i)
    h.cellfields.applyConstitutiveModel(0, True)
    return L, h


def timeit(fn, k):
    torch.cuda.synchronize(); lib.hc_synchronize()
    t0 = time.perf_counter(); fn(k); torch.cuda.synchronize(); lib.hc_synchronize()
    return (time.perf_counter() - t0) / k * 1e3


L1, h1 = build(1)
h1.iterate(20); t_iter = timeit(lambda k: h1.iterate(k), steps)
f_ref = L1.populations(); p_ref = h1.cellfields.positions.copy()
h1.setParticleVelocityUpdateTimeScaleSeparation(10**9)
h1.iterate(10); t_iter_nop = timeit(lambda k: h1.iterate(k), steps)
L1.destroy()


class Loopback:
    """same protocol, transport = two device copies on the compute stream (no RCCL)"""
    rank, world, lo, hi, backend = 0, 1, 0, 0, "loopback"

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi):
        if send_lo is not None and send_lo.numel():
            recv_hi.copy_(send_lo)
        if send_hi is not None and send_hi.numel():
            recv_lo.copy_(send_hi)
        return lambda: None


def issue_time(fn, k):
    """host time to enqueue k steps (the GPU is still busy when this returns unless the host is the bottleneck)"""
    torch.cuda.synchronize(); lib.hc_synchronize()
    t0 = time.perf_counter(); fn(k); t1 = time.perf_counter()
    torch.cuda.synchronize(); lib.hc_synchronize()
    return (t1 - t0) / k * 1e3

L2, h2 = build(2)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); host.check(lib.hc_set_stream(stream.cuda_stream))
eng = X.HipEngine(L2, h2.cellfields, torch.device("cuda", 0))
comm = X.NeighbourComm(0, 1, True)
assert comm.backend == "nccl" and comm.lo == 0 and comm.hi == 0
proto = X.SlabProtocol(eng, comm, 5, n, True)
proto.prepare()
import gc
gc.collect(); gc.freeze()   # as SlabExchange.prepare does: later collections only look at what the runs leave behind
proto.run(20); proto.stats["merge_host_s"] = 0.0
t_rccl = timeit(lambda k: proto.run(k), steps)
t_merge = proto.stats["merge_host_s"] / (steps / 5) * 1e3
proto.halo_exchange_begin(2)()
f_two = L2.populations(); p_two = h2.cellfields.positions
fluid = mask.reshape(-1) == 0
err_f = np.abs(f_two.reshape(-1, 19)[fluid] - f_ref.reshape(-1, 19)[fluid]).max()
err_p = np.abs(p_two - p_ref).max()
print("RCCL self-loop, %d^3 pipe, %d cells, %d steps: hc_iterate %.4f ms/step, slab protocol over RCCL %.4f ms/step (%+.1f %%)"
      % (n, len(centres), steps + 20, t_iter, t_rccl, (t_rccl / t_iter - 1) * 100))
print("max |df| = %.3e   max |dx| = %.3e lu;  host time of one envelope merge %.3f ms, %d cells sent" % (err_f, err_p, t_merge, proto.stats["cells_sent"]), flush=True)
assert err_f <= 1e-10 and err_p <= 1e-8, (err_f, err_p)
# where the difference comes from: without any particle update, and with a plain copy as the transport
proto.k_p = 10**9
proto.run(10); t_nop = timeit(lambda k: proto.run(k), steps)
t_issue = issue_time(lambda k: proto.run(k), steps)
proto.overlap = False
proto.run(10); t_nop_serial = timeit(lambda k: proto.run(k), steps)
proto.overlap = True
proto.comm = Loopback()
proto.run(10); t_nop_loop = timeit(lambda k: proto.run(k), steps)
t_issue_loop = issue_time(lambda k: proto.run(k), steps)
proto.k_p = 5
proto.run(10); t_loop = timeit(lambda k: proto.run(k), steps)
print("hc_iterate        : %.4f (k_p=5)  %.4f (no particle update)" % (t_iter, t_iter_nop))
print("protocol, loopback: %.4f (k_p=5)  %.4f (face messages only; host issue time %.4f)" % (t_loop, t_nop_loop, t_issue_loop))
print("protocol over RCCL: %.4f (k_p=5)  %.4f (face messages only; host issue time %.4f)  %.4f (face messages only, one stream, nothing in flight across phases)"
      % (t_rccl, t_nop, t_issue, t_nop_serial), flush=True)
dist.destroy_process_group()
