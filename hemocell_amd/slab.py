"""x-slab decomposition of the pipe across the GPUs of one node (one process per GPU).

Reference equivalent: Palabos atomic blocks + envelopes (core/hemoCell.cpp:142 fluid envelope,
core/hemoCellFields.cpp:377-499 syncEnvelopes).  Here every rank owns ONE slab along x (the pipe axis).  The whole
schedule -- faces of the 5 crossing populations every step, particle envelopes at every velocity update, both
travelling beside the interior collide -- runs inside libhemocell_amd.so (csrc/slab.hip over csrc/comm.hip: RCCL
point-to-point between x-neighbours, or the same messages through the library's TCP mesh when ranks share a GPU).
This module only creates the objects and calls hc_iterate / hcl_collide_stream; no step is scheduled from Python.

world == 1 runs the loop on one GPU with the periodic wrap done in-kernel.
"""
import ctypes as C
import os

import numpy as np

from . import host

TRANSPORT = {"none": 0, "rccl": 1, "tcp": 2, "auto": 3}


def comm_init_env():
    """connect the ranks named by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torch.distributed.run's
    variables) and select this rank's GPU; a no-op for a one-rank world.  Returns (rank, world)."""
    lib = host.capi.lib()
    host.check(lib.hc_comm_init_env())
    return comm_info()[:2]


def comm_init(rank, world, local_rank=None, addr="127.0.0.1", port=30417, transport="rccl", init_device=True):
    host.check(host.capi.lib().hc_comm_init(int(rank), int(world), int(rank if local_rank is None else local_rank),
                                           addr.encode(), int(port), TRANSPORT[transport], int(bool(init_device))))


def comm_info():
    r, w, t = C.c_int(), C.c_int(), C.c_int()
    host.check(host.capi.lib().hc_comm_info(C.byref(r), C.byref(w), C.byref(t)))
    return r.value, w.value, t.value


def barrier():
    host.check(host.capi.lib().hc_comm_barrier())


def allreduce(values, op="sum"):
    v = np.ascontiguousarray(values, dtype=np.float64).copy()
    host.check(host.capi.lib().hc_comm_allreduce(host.dptr(v), v.size, {"sum": 0, "min": 1, "max": 2}[op]))
    return v


def comm_finalize():
    host.check(host.capi.lib().hc_comm_finalize())


class SlabRunner:
    def __init__(self, nx_local, ny, nz, rank, world, P, periodic=(True, False, False), particle_timescale=5,
                 material_timescale=20, deletion_check_every=1, fluid_only=False, n_slabs=None):
        self.rank, self.world = rank, world
        self.nx, self.ny, self.nz = nx_local, ny, nz
        self.nx_global = nx_local * world
        self.x0 = rank * nx_local
        self.P = P
        self.periodic = periodic
        self.k_p, self.k_m = particle_timescale, material_timescale
        # n_slabs > 1 with world == 1: a rank that is its own periodic neighbour over the data plane (transport tests)
        self.n_slabs = world if n_slabs is None else n_slabs
        self.lattice = host.Lattice(nx_local, ny, nz, periodic, 1.0 / P.tau, x0=self.x0, nx_global=self.nx_global,
                                    n_slabs=self.n_slabs)
        self.fluid_only = fluid_only   # no membrane cells: the lattice then never touches the IBM force buffers
        if fluid_only:
            self.hemocell, self.cells = None, None
        else:
            self.hemocell = host.HemoCell(self.lattice, P)
            self.hemocell.setParticleVelocityUpdateTimeScaleSeparation(particle_timescale)
            self.hemocell.deletion_check_every = deletion_check_every
            self.cells = self.hemocell.cellfields

    def define_bounce_back(self, mask_global):
        self.lattice.defineBounceBack(mask_global)

    def add_cell_type(self, celltype):
        return self.cells.addCellType(celltype, self.k_m)

    def set_envelope(self, particle_envelope_lu):
        """<domain><particleEnvelope> of the configuration (core/hemoCell.cpp:139), after the cell types and before the cells;
        returns the distance from a slab face within which whole cells are replicated"""
        used = C.c_double()
        host.check(host.capi.lib().hcp_set_envelope(self.cells.ptr, float(particle_envelope_lu), C.byref(used)))
        return used.value

    def envelope(self):
        """(share in use [lu], copies that arrived late so far)"""
        used, late = C.c_double(), C.c_long()
        host.check(host.capi.lib().hcp_envelope(self.cells.ptr, C.byref(used), C.byref(late)))
        return used.value, late.value

    def load_cells(self, t, centres, angles, min_dist_um=0.0):
        """offer every cell to this rank (hcp_add_cell keeps what its slab has to hold, periodic images included);
        call sync_placement() after the last type.  Returns the number of cells kept on this rank."""
        n = 0
        for i, (c, a) in enumerate(zip(centres, angles)):
            n += bool(self.cells.addCell(t, c, a, min_dist_um, cell_id=i))
        return n

    def sync_placement(self):
        """drop everywhere the cells some rank rejected at a wall; distinct cells per type over all slabs"""
        out = np.zeros(max(len(self.cells.types), 1), dtype=np.int64)
        host.check(host.capi.lib().hcp_slab_sync_placement(self.cells.ptr, host.lptr(out)))
        return out[:len(self.cells.types)]

    def owned_vertices(self):
        if self.fluid_only:
            return 0
        n = C.c_long()
        host.check(host.capi.lib().hcp_owned_vertices(self.cells.ptr, C.byref(n)))
        return n.value

    def prepare(self):
        """what the drivers do before the loop: forces of the initial configuration"""
        if not self.fluid_only:
            self.cells.applyConstitutiveModel(0, True)

    def run(self, n):
        if self.fluid_only:
            self.lattice.collideAndStream(n)
        else:
            self.hemocell.iterate(n)

    def slab_stats(self, reset=False):
        """counters of the native slab schedule (hc_slab_stats)"""
        o = np.zeros(8)
        host.check(host.capi.lib().hc_slab_stats(self.lattice.ptr, host.dptr(o), int(reset)))
        names = ("cells_sent", "cells_new", "cells_dropped", "cells_deleted", "iterations", "host_s", "header_wait_s", "particle_steps")
        return dict(zip(names, o.tolist()))

    # ---- diagnostics across slabs: the reference gathers per-block statistics (HemoCellGatheringFunctional,
    # core/hemoCellFunctional.h:101-112); here a few scalars are reduced over the control plane at output cadence
    def _reduce(self, mn, mx, total, count):
        if self.world == 1:
            return mn, mx, total, count
        big = 1e300
        lo = allreduce([mn if count else big], "min")[0]
        hi = allreduce([mx if count else -big], "max")[0]
        sm = allreduce([total, float(count)], "sum")
        n = int(sm[1])
        return (float(lo) if n else 0.0), (float(hi) if n else 0.0), float(sm[0]), n

    def fluid_stats(self, what=0):
        """FluidInfo statistics over the whole domain: (min, max, mean, nodes)"""
        mn, mx, avg, n = self.lattice.fluid_stats(what)
        mn, mx, total, n = self._reduce(mn, mx, avg * n, n)
        return mn, mx, (total / n if n else 0.0), n

    def vertex_stats(self, what=2):
        """ParticleInfo statistics over all owned vertices of all slabs: (min, max, mean, vertices)"""
        mn, mx, avg, n = self.cells.vertex_stats(what) if not self.fluid_only else (0.0, 0.0, 0.0, 0)
        mn, mx, total, n = self._reduce(mn, mx, avg * n, n)
        return mn, mx, (total / n if n else 0.0), n

    # ---- inspection helpers (tests / output): gather-free, per rank
    def populations(self):
        """post-stream populations of this slab (the library refreshes the halo planes the view pulls from)"""
        return self.lattice.populations()

    def owned_vertex_table(self, t=0):
        """(cell id, vertex id, position) of the vertices whose nearest node lies in this slab"""
        pos = self.cells.positions
        ids = self.cells.cell_ids()
        f, n = self.cells.type_range(t)
        nv = self.cells.types[t].nv
        p = pos[f:f + n * nv].reshape(n, nv, 3)
        first = sum(self.cells.type_range(u)[1] for u in range(t))
        cid = ids[first:first + n]
        g = np.floor(p[:, :, 0] + 0.5).astype(np.int64) - self.x0
        own = (g >= 0) & (g < self.nx)
        ci, vi = np.nonzero(own)
        return cid[ci], vi, p[ci, vi]
