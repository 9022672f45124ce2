"""Run from the repository root on a box with ONE GPU: python examples/rccl_selfloop.py [n [steps [transport [nx]]]]
(pipe of nx x n x n nodes, nx = n unless given: 64 is the slab BASELINE config 3 gives each of 8 GPUs)

The native slab schedule (csrc/slab.hip) over the real RCCL data plane with a single rank: the slab is its own periodic
neighbour, so every face / envelope message is an ncclSend to self + ncclRecv from self inside one group, on the
library's side stream -- the same code path an N > 1 run takes.  Checks the result against hc_iterate (in-kernel wrap) on
the same case and reports the per-step cost next to it, plus the host time the schedule needs per step.  Cells are kept
away from the seam so that no cell record crosses (a slab cannot hold a cell and its own periodic image)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hemocell_amd import host, slab
from hemocell_amd.packing import pack_pipe_rbc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
transport = sys.argv[3] if len(sys.argv) > 3 else "rccl"
nx = int(sys.argv[4]) if len(sys.argv) > 4 else n
slab.comm_init(0, 1, port=30611 + os.getpid() % 2000, transport=transport)
lib = host.capi.lib()
P = host.base_parameters()
mask, R = host.pipe_mask(nx, n, n)
centres, angles = pack_pipe_rbc(nx, n, n, 0.10)
keep = (centres[:, 0] > 24) & (centres[:, 0] < nx - 24)
centres, angles = centres[keep], angles[keep]


def build(n_slabs):
    L = host.Lattice(nx, n, n, (1, 0, 0), 1 / P.tau, x0=0, nx_global=nx, n_slabs=n_slabs)
    L.defineBounceBack(mask); L.latticeEquilibrium(); L.setExternalVector((2e-6, 0, 0))
    h = host.HemoCell(L, P); h.cellfields.addCellType(host.CellType.rbc(P), 20)
    h.setParticleVelocityUpdateTimeScaleSeparation(5)
    for i, (c, a) in enumerate(zip(centres, angles)):
        h.cellfields.addCell(0, c, a, cell_id=i)
    h.cellfields.applyConstitutiveModel(0, True)
    return L, h


def timeit(fn, k):
    lib.hc_synchronize()
    t0 = time.perf_counter(); fn(k); lib.hc_synchronize()
    return (time.perf_counter() - t0) / k * 1e3


L1, h1 = build(1)
h1.iterate(20); t_iter = timeit(h1.iterate, steps)
f_ref = L1.populations(); p_ref = h1.cellfields.positions.copy()
L1.destroy()

L2, h2 = build(2)
h2.iterate(20)
o = np.zeros(8); host.check(lib.hc_slab_stats(L2.ptr, host.dptr(o), 1))
t_slab = timeit(h2.iterate, steps)
host.check(lib.hc_slab_stats(L2.ptr, host.dptr(o), 0))
f = L2.populations(); p = h2.cellfields.positions
err_f = np.abs(f - f_ref).max(); err_p = np.abs(p - p_ref).max()
print("%d x %d x %d pipe, %d RBC, %d + %d iterations" % (nx, n, n, len(centres), 20, steps))
print("hc_iterate (in-kernel wrap)      %.3f ms/step" % t_iter)
print("slab schedule over %s          %.3f ms/step (host: %.3f ms/step enqueueing; %.3f ms per velocity update waiting for cell extents and id "
      "headers, i.e. for the GPU to catch up with the queue)" % (transport.upper(), t_slab, (o[5] - o[6]) / max(o[4], 1) * 1e3, o[6] / max(o[7], 1) * 1e3))
print("max |df| = %.3e, max |dx| = %.3e lu vs hc_iterate" % (err_f, err_p))
assert err_f <= 1e-12 and err_p <= 1e-10, (err_f, err_p)
slab.comm_finalize()
