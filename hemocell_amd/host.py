"""Thin numpy-facing host layer over the C ABI.

Names follow the reference's interface for this path (hemocell.h:86-253,
core/hemoCellFields.h:103-158): latticeEquilibrium, collideAndStream,
spreadParticleForce, interpolateFluidVelocity, advanceParticles,
applyConstitutiveModel, iterate, setMaterialTimeScaleSeparation ...  Every
method is one call into libhemocell_amd.so; nothing is computed here.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import HcError, Material, Params, check, dptr, lptr

HALO = 2

MODEL_RBC_HO = 0
MODEL_PLT_SIMPLE = 1
RBC_FROM_SPHERE = 1        # config/constant_defaults.h:80
ELLIPSOID_FROM_SPHERE = 6  # config/constant_defaults.h:81

# examples/pipeflow/PLT.xml:14-38
PLT_INNER_EDGES = np.array([[60, 65], [62, 64], [37, 42], [54, 56], [34, 40], [25, 46], [50, 59], [29, 47],
                            [61, 63], [26, 45], [33, 43], [27, 35], [32, 39], [49, 51], [0, 4], [48, 52],
                            [6, 10], [53, 55], [19, 21], [57, 58], [15, 13]], dtype=np.int64)

_initialised = False


def init(device=0):
    """plb::plbInit equivalent: select the GPU; raises if no gfx950 device."""
    global _initialised
    check(capi.lib().hc_init(device))
    _initialised = True


def ensure_init():
    if not _initialised:
        init(0)


def base_parameters(dx=5e-7, dt=1e-7, nuP=1.1e-6, rhoP=1025.0, kBT=4.100531391e-21):
    """param::lbm_base_parameters(cfg) (mechanics/constantConversion.cpp:36-59)"""
    P = Params()
    check(capi.lib().hc_params_base(C.byref(P), dx, dt, nuP, rhoP, kBT))
    return P


class Lattice:
    """One x-slab of the MultiBlockLattice3D<T,DESCRIPTOR> with GuoExternalForceBGKdynamics."""

    def __init__(self, nx, ny, nz, periodic=(False, False, False), omega=1.0, x0=0, nx_global=None, n_slabs=1):
        ensure_init()
        self.lib = capi.lib()
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.x0 = int(x0)
        self.nx_global = int(nx_global if nx_global is not None else nx)
        self.n_slabs = int(n_slabs)
        self.periodic = tuple(bool(p) for p in periodic)
        per = (C.c_int * 3)(*[int(p) for p in periodic])
        self.ptr = C.c_void_p()
        check(self.lib.hcl_create(C.byref(self.ptr), self.nx, self.ny, self.nz, per, float(omega),
                                  self.x0, self.nx_global, self.n_slabs))
        self.n = self.nx * self.ny * self.nz

    # defineDynamics(lattice, flagMatrix, bbox, new BounceBack(1.), 0)
    def defineBounceBack(self, mask_global):
        """mask_global: uint8 [nx_global][ny][nz] (1 = BounceBack).  The slab's halo planes are filled
        from the global array (periodic wrap in x if enabled, otherwise wall)."""
        m = np.ascontiguousarray(mask_global, dtype=np.uint8)
        if m.shape != (self.nx_global, self.ny, self.nz):
            raise HcError("mask must have the global shape %s, got %s" % ((self.nx_global, self.ny, self.nz), m.shape))
        xs = np.arange(self.x0 - HALO, self.x0 + self.nx + HALO)
        if self.periodic[0]:
            local = m[np.mod(xs, self.nx_global)]
        else:
            local = np.ones((len(xs), self.ny, self.nz), np.uint8)   # outside a non-periodic pipe end: wall
            ok = (xs >= 0) & (xs < self.nx_global)
            local[ok] = m[xs[ok]]
        local = np.ascontiguousarray(local)
        check(self.lib.hcl_set_mask(self.ptr, local.ctypes.data))

    def latticeEquilibrium(self, rho=1.0, u=(0.0, 0.0, 0.0)):
        uu = np.array(u, dtype=np.float64)
        check(self.lib.hcl_init_equilibrium(self.ptr, float(rho), dptr(uu)))

    def setExternalVector(self, F):
        ff = np.array(F, dtype=np.float64)
        check(self.lib.hcl_set_body_force(self.ptr, dptr(ff)))

    def setExternalVectorBoxes(self, boxes, forces):
        """setExternalVector on sub-domains (cases/kolmogorovFlow/kolmogorovFlow.cpp:136-140): inclusive global node boxes
        (x0, x1, y0, y1, z0, z1), later ones override earlier ones; an empty list removes them"""
        bb = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 6)
        ff = np.ascontiguousarray(forces, dtype=np.float64).reshape(-1, 3)
        assert len(bb) == len(ff)
        check(self.lib.hcl_set_body_force_regions(self.ptr, len(bb), bb.ctypes.data_as(C.POINTER(C.c_int)), dptr(ff)))

    def setBoundaryVelocity(self, wall_class, u):
        """velocity of the nodes whose mask value is wall_class (3..6)"""
        uu = np.array(u, dtype=np.float64)
        check(self.lib.hcl_set_wall_velocity(self.ptr, int(wall_class), dptr(uu)))

    def collideAndStream(self, steps=1):
        check(self.lib.hcl_collide_stream(self.ptr, int(steps)))

    def collide_part(self, part):
        check(self.lib.hcl_collide_stream_part(self.ptr, int(part)))

    def fluid_stats(self, what=0):
        """FluidInfo::calculate{Velocity,Force}Statistics: (min, max, mean, n) of |u| (what 0) or |F| (what 1) over
        the non-boundary nodes, reduced on the device"""
        o = np.zeros(3); n = C.c_long()
        check(self.lib.hcl_fluid_stats(self.ptr, int(what), dptr(o), C.byref(n)))
        return o[0], o[1], (o[2] / n.value if n.value else 0.0), n.value

    def step_end(self):
        check(self.lib.hcl_step_end(self.ptr))

    def populations(self):
        """[n][19] post-stream populations (f - t_i), reference node order z + nz*(y + ny*x)"""
        out = np.empty((self.n, 19), dtype=np.float64)
        check(self.lib.hcl_download_populations(self.ptr, dptr(out)))
        return out

    def set_populations(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        if f.shape != (self.n, 19):
            raise HcError("populations must have shape (%d, 19), got %s" % (self.n, f.shape))
        check(self.lib.hcl_upload_populations(self.ptr, dptr(f)))

    def rho_u(self):
        rho = np.empty(self.n, dtype=np.float64)
        u = np.empty((self.n, 3), dtype=np.float64)
        check(self.lib.hcl_download_rho_u(self.ptr, dptr(rho), dptr(u)))
        return rho, u

    def pi_neq(self):
        """off-equilibrium momentum flux (xx, xy, xz, yy, yz, zz) per node"""
        pi = np.empty((self.n, 6), dtype=np.float64)
        check(self.lib.hcl_download_pi_neq(self.ptr, dptr(pi)))
        return pi

    def ibm_force(self):
        F = np.empty((self.n, 3), dtype=np.float64)
        check(self.lib.hcl_download_ibm_force(self.ptr, dptr(F)))
        return F

    def halo_doubles(self, width):
        return int(self.lib.hcl_halo_doubles(self.ptr, int(width)))

    def halo_pack(self, side, width, dev_ptr, next=False):
        fn = self.lib.hcl_halo_pack_next if next else self.lib.hcl_halo_pack
        check(fn(self.ptr, int(side), int(width), C.c_void_p(dev_ptr)))

    def halo_unpack(self, side, width, dev_ptr):
        check(self.lib.hcl_halo_unpack(self.ptr, int(side), int(width), C.c_void_p(dev_ptr)))

    def bytes_per_node(self):
        return float(self.lib.hcl_mlups_bytes_per_node(self.ptr))

    def destroy(self):
        if self.ptr:
            check(self.lib.hcl_destroy(self.ptr))
            self.ptr = C.c_void_p()


class CellType:
    """hemocell.addCellType<Mechanics>(name, constructType) for one type."""

    def __init__(self, P, model, shape, radius, min_triangles, kLink, kArea, kVolume, kBend, eta_m=0.0,
                 aspect_ratio=0.3, inner_edges=None):
        ensure_init()
        self.lib = capi.lib()
        M = Material()
        M.kLink, M.kArea, M.kVolume, M.kBend, M.eta_m = kLink, kArea, kVolume, kBend, eta_m
        M.radius, M.min_triangles, M.aspect_ratio = radius, int(min_triangles), aspect_ratio
        self._ie = None
        if inner_edges is not None and len(inner_edges):
            self._ie = np.ascontiguousarray(inner_edges, dtype=np.int64)
            M.inner_edges = lptr(self._ie)
            M.n_inner = len(self._ie)
        else:
            M.inner_edges = None
            M.n_inner = 0
        self.ptr = C.c_void_p()
        check(self.lib.hcp_celltype_create(C.byref(self.ptr), int(model), int(shape), C.byref(P), C.byref(M)))
        sz = (C.c_int * 4)()
        check(self.lib.hcp_celltype_sizes(self.ptr, sz))
        self.nv, self.nt, self.ne, self.nie = [int(x) for x in sz]
        self.model = model

    @classmethod
    def rbc(cls, P, **kw):
        """examples/pipeflow/RBC.xml with RbcHighOrderModel / RBC_FROM_SPHERE"""
        d = dict(radius=3.91e-6, min_triangles=600, kLink=15.0, kArea=5.0, kVolume=20.0, kBend=80.0, eta_m=0.0)
        d.update(kw)
        return cls(P, MODEL_RBC_HO, RBC_FROM_SPHERE, **d)

    @classmethod
    def plt(cls, P, **kw):
        """examples/pipeflow/PLT.xml with PltSimpleModel / ELLIPSOID_FROM_SPHERE"""
        d = dict(radius=1.25e-6, min_triangles=66, kLink=25.0, kArea=8.0, kVolume=100.0, kBend=250.0, eta_m=0.0,
                 aspect_ratio=0.434782608696, inner_edges=PLT_INNER_EDGES)
        d.update(kw)
        return cls(P, MODEL_PLT_SIMPLE, ELLIPSOID_FROM_SPHERE, **d)

    def tables(self):
        t = dict(vertices=np.empty((self.nv, 3)), triangles=np.empty((self.nt, 3), np.int64),
                 edges=np.empty((self.ne, 2), np.int64), edge_length_eq=np.empty(self.ne),
                 edge_angle_eq=np.empty(self.ne), triangle_area_eq=np.empty(self.nt),
                 vertex_vertexes=np.empty((self.nv, 6), np.int64), patch_dist_eq=np.empty(self.nv),
                 scalars=np.empty(9))
        check(self.lib.hcp_celltype_tables(self.ptr, dptr(t["vertices"]), lptr(t["triangles"]), lptr(t["edges"]),
                                           dptr(t["edge_length_eq"]), dptr(t["edge_angle_eq"]),
                                           dptr(t["triangle_area_eq"]), lptr(t["vertex_vertexes"]),
                                           dptr(t["patch_dist_eq"]), dptr(t["scalars"])))
        names = ("volume_eq", "area_mean_eq", "edge_mean_eq", "angle_mean_eq", "k_volume", "k_area", "k_link",
                 "k_bend", "eta_m")
        t.update({n: float(v) for n, v in zip(names, t["scalars"])})
        return t

    def destroy(self):
        if self.ptr:
            check(self.lib.hcp_celltype_destroy(self.ptr))
            self.ptr = C.c_void_p()


class Cells:
    """HemoCellFields: all membrane vertices on this GPU and the per-phase operations."""

    def __init__(self, lattice, P):
        self.lib = capi.lib()
        self.lattice = lattice
        self.P = P
        self.ptr = C.c_void_p()
        check(self.lib.hcp_create(C.byref(self.ptr), lattice.ptr, C.byref(P)))
        self.types = []
        self._next_id = 0
        self.rep_timescale = self.brep_timescale = 0   # cadences of the two repulsions (0 = off)

    def addCellType(self, celltype, material_timescale=1):
        idx = C.c_int()
        check(self.lib.hcp_add_type(self.ptr, celltype.ptr, int(material_timescale), C.byref(idx)))
        self.types.append(celltype)
        return idx.value

    def addCell(self, type_index, centre_lu, angles_deg=(0.0, 0.0, 0.0), min_dist_um=0.0, cell_id=None):
        """one line of a .pos file, already in lattice units; angles in degrees as in the file
        (io/readPositionsBloodCells.cpp:218-229: rad, then negated)"""
        c = np.array(centre_lu, dtype=np.float64)
        a = np.array(angles_deg, dtype=np.float64) * (3.14159265358979323846 / 180.0)
        a = a * -1.0
        placed = C.c_int()
        cid = self._next_id if cell_id is None else int(cell_id)
        self._next_id = max(self._next_id, cid + 1)
        check(self.lib.hcp_add_cell(self.ptr, int(type_index), cid, dptr(c), dptr(a), float(min_dist_um), C.byref(placed)))
        return bool(placed.value)

    def counts(self):
        nv, nc, nd = C.c_long(), C.c_long(), C.c_long()
        check(self.lib.hcp_counts(self.ptr, C.byref(nv), C.byref(nc), C.byref(nd)))
        return nv.value, nc.value, nd.value

    def type_range(self, t):
        f, n = C.c_long(), C.c_long()
        check(self.lib.hcp_type_range(self.ptr, int(t), C.byref(f), C.byref(n)))
        return f.value, n.value

    def _get(self, what):
        out = np.empty((self.counts()[0], 3), dtype=np.float64)
        check(self.lib.hcp_download(self.ptr, what, dptr(out)))
        return out

    def _set(self, what, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != (self.counts()[0], 3):
            raise HcError("vertex array must have shape (%d, 3), got %s" % (self.counts()[0], a.shape))
        check(self.lib.hcp_upload(self.ptr, what, dptr(a)))

    positions = property(lambda s: s._get(0), lambda s, a: s._set(0, a))
    velocities = property(lambda s: s._get(1), lambda s, a: s._set(1, a))
    forces = property(lambda s: s._get(2), lambda s, a: s._set(2, a))

    def cell_ids(self):
        ids = np.empty(self.counts()[1], dtype=np.int64)
        check(self.lib.hcp_download_cell_ids(self.ptr, lptr(ids)))
        return ids

    # HemoCellParticle::serializeValues_t (core/hemoCellParticle.h:45-63), 120 bytes per vertex
    SV_DTYPE = np.dtype({"names": ["v", "position", "force", "force_repulsion", "cellId", "vertexId", "restime", "celltype"],
                         "formats": [("<f8", 3), ("<f8", 3), ("<f8", 3), ("<f8", 3), "<i8", "<u2", "<u4", "u1"],
                         "offsets": [0, 24, 48, 72, 96, 104, 108, 112], "itemsize": 120})

    def records(self):
        """every particle as the reference's particle record (removed particles of incomplete cells are not listed)"""
        rec = np.zeros(self.counts()[0] - self.deletion_counts()[3], dtype=self.SV_DTYPE)
        check(self.lib.hcp_download_records(self.ptr, rec.ctypes.data, len(rec)))
        return rec

    def set_records(self, rec):
        """replace the whole population by these records (any order, complete cells)"""
        rec = np.ascontiguousarray(rec, dtype=self.SV_DTYPE)
        check(self.lib.hcp_upload_records(self.ptr, rec.ctypes.data, len(rec)))

    def addVertexForce(self, vertex_index, f):
        idx = np.ascontiguousarray(vertex_index, dtype=np.int64)
        ff = np.ascontiguousarray(f, dtype=np.float64).reshape(len(idx), 3)
        check(self.lib.hcp_add_vertex_force(self.ptr, lptr(idx), len(idx), dptr(ff)))

    def vertex_stats(self, what=2):
        """ParticleInfo::calculate{Velocity,Force}Statistics: (min, max, mean, n) of |v| (what 1) or
        |force + force_repulsion| (what 2) over the owned vertices, reduced on the device"""
        o = np.zeros(3); n = C.c_long()
        check(self.lib.hcp_vertex_stats(self.ptr, int(what), dptr(o), C.byref(n)))
        return o[0], o[1], (o[2] / n.value if n.value else 0.0), n.value

    def setRepulsion(self, r_const, r_cutoff_um, timescale=1):
        """hemocell.setRepulsion(k, cutoff [um]) + setRepulsionTimeScaleSeperation(timescale)"""
        check(self.lib.hcp_set_repulsion(self.ptr, float(r_const), float(r_cutoff_um) * (1e-6 / self.P.dx), int(timescale)))
        self.rep_timescale = int(timescale)

    def applyRepulsionForce(self):
        check(self.lib.hcp_repulsion(self.ptr))

    def enableBoundaryParticles(self, br_const, br_cutoff_um, timescale=1):
        """hemocell.enableBoundaryParticles(k, cutoff [um], timestep) (core/hemoCell.cpp:428-436)"""
        check(self.lib.hcp_set_boundary_repulsion(self.ptr, float(br_const), float(br_cutoff_um) * (1e-6 / self.P.dx), int(timescale)))
        self.brep_timescale = int(timescale)

    def applyBoundaryRepulsionForce(self):
        check(self.lib.hcp_boundary_repulsion(self.ptr))

    @property
    def repulsion_forces(self):
        out = np.empty((self.counts()[0], 3), dtype=np.float64)
        check(self.lib.hcp_download_repulsion(self.ptr, dptr(out)))
        return out

    def spreadParticleForce(self, force_limit=True):
        check(self.lib.hcp_spread(self.ptr, int(force_limit)))

    def interpolateFluidVelocity(self):
        check(self.lib.hcp_interpolate(self.ptr))

    def advanceParticles(self, check_deletions=True):
        check(self.lib.hcp_advance(self.ptr, int(check_deletions)))

    # what happens to a particle that reaches a wall: the reference removes that single particle
    # (core/hemoCellParticleField.cpp:566-588, "particle", default) -- or the whole cell goes at once ("cell")
    def setDeletionMode(self, mode):
        check(self.lib.hcp_set_deletion_mode(self.ptr, {"particle": 0, "cell": 1}[mode]))

    def deleteIncompleteCells(self):
        """HemoCellFields::deleteIncompleteCells (core/hemoCellParticleField.cpp:512-553); returns the cells removed"""
        n = C.c_long()
        check(self.lib.hcp_delete_incomplete_cells(self.ptr, C.byref(n)))
        return n.value

    def deletion_counts(self):
        """(cells removed entirely, particles removed, incomplete cells listed now, their missing particles)"""
        a, b, c, d = C.c_long(), C.c_long(), C.c_long(), C.c_long()
        check(self.lib.hcp_deletion_counts(self.ptr, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def alive(self):
        out = np.empty(self.counts()[0], dtype=np.uint8)
        check(self.lib.hcp_download_alive(self.ptr, out.ctypes.data))
        return out.astype(bool)

    def applyConstitutiveModel(self, iter_=0, forced=False):
        check(self.lib.hcp_mechanics(self.ptr, int(iter_), int(forced)))

    def force_components(self, t):
        f, n = self.type_range(t)
        nv = self.types[t].nv
        comp = np.empty((6, n * nv, 3), dtype=np.float64)
        check(self.lib.hcp_mechanics_components(self.ptr, int(t), dptr(comp)))
        return comp

    def cell_info(self, t):
        f, n = self.type_range(t)
        vol, area, bbox, cen = np.empty(n), np.empty(n), np.empty((n, 6)), np.empty((n, 3))
        check(self.lib.hcp_cell_info(self.ptr, int(t), dptr(vol), dptr(area), dptr(bbox), dptr(cen)))
        return dict(volume=vol, area=area, bbox=bbox, position=cen)

    def destroy(self):
        if self.ptr:
            check(self.lib.hcp_destroy(self.ptr))
            self.ptr = C.c_void_p()


class HemoCell:
    """hemo::HemoCell facade for one GPU (hemocell.h:68-253): owns lattice + cellfields, iterate()."""

    def __init__(self, lattice, P):
        self.lattice = lattice
        self.P = P
        self.cellfields = Cells(lattice, P)
        self.iter = 0
        self.particleVelocityUpdateTimescale = 1
        self.force_limit = True
        self.deletion_check_every = 1

    def setParticleVelocityUpdateTimeScaleSeparation(self, n):
        self.particleVelocityUpdateTimescale = int(n)

    def iterate(self, n=1):
        it = C.c_long(self.iter)
        check(capi.lib().hc_iterate(self.lattice.ptr, self.cellfields.ptr, C.byref(it), int(n),
                                    self.particleVelocityUpdateTimescale, int(self.force_limit),
                                    int(self.deletion_check_every)))
        self.iter = it.value

    def synchronize(self):
        check(capi.lib().hc_synchronize())


def pipe_mask(nx, ny, nz):
    """analytic cylinder along x replacing tube.stl: radius (ny-2)/2 centred at
    ((ny-1)/2,(nz-1)/2); node solid iff r > R"""
    y = np.arange(ny)[:, None] - (ny - 1) / 2.0
    z = np.arange(nz)[None, :] - (nz - 1) / 2.0
    R = (ny - 2) / 2.0
    solid = (y * y + z * z) > R * R
    return np.ascontiguousarray(np.broadcast_to(solid[None], (nx, ny, nz)).astype(np.uint8)), R
