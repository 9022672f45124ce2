"""x-slab decomposition of the pipe across the GPUs of one node (one process per GPU).

Reference equivalent: Palabos atomic blocks + envelopes (core/hemoCell.cpp:142 fluid envelope,
core/hemoCellFields.cpp:377-499 syncEnvelopes).  Here every rank owns ONE slab along x (the pipe axis);
per step it exchanges the x-faces of the population field with its two neighbours (point-to-point, no
collective), and membrane cells that reach across a slab face are replicated on both ranks like the
reference's envelope copies, re-synchronised from their owner every stepParticleEvery steps.

world == 1 runs the whole loop inside the library (hc_iterate) with the periodic wrap done in-kernel.
"""
import numpy as np

from . import host


class SlabRunner:
    def __init__(self, nx_local, ny, nz, rank, world, P, periodic=(True, False, False), particle_timescale=5,
                 material_timescale=20, deletion_check_every=1, comm=None, fluid_only=False):
        self.rank, self.world = rank, world
        self.nx, self.ny, self.nz = nx_local, ny, nz
        self.nx_global = nx_local * world
        self.x0 = rank * nx_local
        self.P = P
        self.periodic = periodic
        self.k_p, self.k_m = particle_timescale, material_timescale
        self.lattice = host.Lattice(nx_local, ny, nz, periodic, 1.0 / P.tau, x0=self.x0, nx_global=self.nx_global,
                                    n_slabs=world)
        self.fluid_only = fluid_only   # no membrane cells: the lattice then never touches the IBM force buffers
        if fluid_only:
            self.hemocell, self.cells = None, None
        else:
            self.hemocell = host.HemoCell(self.lattice, P)
            self.hemocell.setParticleVelocityUpdateTimeScaleSeparation(particle_timescale)
            self.hemocell.deletion_check_every = deletion_check_every
            self.cells = self.hemocell.cellfields
        self.comm = comm
        if world > 1:
            from .exchange import SlabExchange
            self.exchange = SlabExchange(self, comm)
        else:
            self.exchange = None

    def define_bounce_back(self, mask_global):
        self.lattice.defineBounceBack(mask_global)

    def add_cell_type(self, celltype):
        return self.cells.addCellType(celltype, self.k_m)

    def load_cells(self, t, centres, angles, min_dist_um=0.0):
        """place the cells this rank has to hold; returns the number of cells placed on this rank"""
        if self.exchange is not None:
            return self.exchange.load_cells(t, centres, angles, min_dist_um)
        n = 0
        for i, (c, a) in enumerate(zip(centres, angles)):
            n += bool(self.cells.addCell(t, c, a, min_dist_um, cell_id=i))
        return n

    def owned_vertices(self):
        if self.fluid_only:
            return 0
        if self.exchange is not None:
            return self.exchange.owned_vertices()
        return self.cells.counts()[0]

    def prepare(self):
        """what the drivers do before the loop: forces of the initial configuration"""
        if not self.fluid_only:
            self.cells.applyConstitutiveModel(0, True)
        if self.exchange is not None:
            self.exchange.prepare()

    def run(self, n):
        if self.exchange is None:
            if self.fluid_only:
                self.lattice.collideAndStream(n)
            else:
                self.hemocell.iterate(n)
        else:
            self.exchange.run(n)

    # ---- diagnostics across slabs: the reference gathers per-block statistics (HemoCellGatheringFunctional,
    # core/hemoCellFunctional.h:101-112); here a few scalars are all-reduced at output cadence
    def _reduce(self, mn, mx, total, count):
        if self.world == 1:
            return mn, mx, total, count
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        big = 1e300
        lo = torch.tensor([mn if count else big], dtype=torch.float64, device=dev)
        hi = torch.tensor([mx if count else -big], dtype=torch.float64, device=dev)
        sm = torch.tensor([total, float(count)], dtype=torch.float64, device=dev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        n = int(sm[1].item())
        return (float(lo.item()) if n else 0.0), (float(hi.item()) if n else 0.0), float(sm[0].item()), n

    def fluid_stats(self, what=0):
        """FluidInfo statistics over the whole domain: (min, max, mean, nodes)"""
        if self.exchange is not None and what == 0:
            self.exchange.protocol.halo_exchange_begin(1)()      # velocities on the face planes pull from the halo planes
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
        """post-stream populations of this slab; the view pulls from the halo planes, so refresh them first"""
        if self.exchange is not None:
            self.exchange.protocol.halo_exchange_begin(2)()
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
