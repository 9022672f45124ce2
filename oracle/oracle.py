"""ctypes binding of oracle/libhemo_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (hemocell_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhemo_oracle.so")

c_double_p = C.POINTER(C.c_double)
c_long_p = C.POINTER(C.c_long)
c_int_p = C.POINTER(C.c_int)


class Lattice(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("periodic", C.c_int * 3),
                ("omega", C.c_double), ("f", c_double_p), ("ftmp", c_double_p), ("force", c_double_p),
                ("mask", C.POINTER(C.c_ubyte)), ("nthreads", C.c_int), ("wall_u", (C.c_double * 3) * 4),
                ("fused", C.c_int)]


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("dx", "dt", "nu_p", "rho_p", "kBT_p", "tau", "nu_lbm", "dm", "df", "f_limit", "kBT_lbm")]


class CellType(C.Structure):
    _fields_ = [("model", C.c_int), ("nv", C.c_int), ("nt", C.c_int), ("ne", C.c_int), ("nie", C.c_int),
                ("vertices", c_double_p), ("triangles", c_long_p), ("edges", c_long_p),
                ("edge_length_eq", c_double_p), ("edge_angle_eq", c_double_p),
                ("edge_bending_triangles", c_long_p), ("edge_bending_outer", c_long_p),
                ("triangle_area_eq", c_double_p), ("vertex_vertexes", c_long_p),
                ("vertex_n_vertexes", c_int_p), ("patch_dist_eq", c_double_p),
                ("inner_edges", c_long_p), ("inner_edge_length_eq", c_double_p),
                ("volume_eq", C.c_double), ("area_mean_eq", C.c_double), ("edge_mean_eq", C.c_double),
                ("angle_mean_eq", C.c_double),
                ("k_volume", C.c_double), ("k_area", C.c_double), ("k_link", C.c_double),
                ("k_bend", C.c_double), ("eta_m", C.c_double), ("timescale", C.c_int)]

    def arr(self, name):
        shapes = {"vertices": (self.nv, 3), "triangles": (self.nt, 3), "edges": (self.ne, 2),
                  "edge_length_eq": (self.ne,), "edge_angle_eq": (self.ne,),
                  "edge_bending_triangles": (self.ne, 2), "edge_bending_outer": (self.ne, 2),
                  "triangle_area_eq": (self.nt,), "vertex_vertexes": (self.nv, 6),
                  "vertex_n_vertexes": (self.nv,), "patch_dist_eq": (self.nv,),
                  "inner_edges": (self.nie, 2), "inner_edge_length_eq": (self.nie,)}
        shp = shapes[name]
        if int(np.prod(shp)) == 0:
            return np.zeros(shp)
        return np.ctypeslib.as_array(getattr(self, name), shape=shp).copy()


class Sim(C.Structure):
    _fields_ = [("L", C.POINTER(Lattice)), ("P", Params), ("ntypes", C.c_int),
                ("types", C.POINTER(CellType) * 8), ("ncells", C.c_long * 8),
                ("particles", C.c_void_p), ("np", C.c_long),
                ("st_nodes", c_long_p), ("st_w", c_double_p), ("st_n", c_int_p),
                ("particle_velocity_timescale", C.c_int), ("force_limit_enabled", C.c_int),
                ("iter", C.c_long), ("cells_deleted", C.c_long), ("body_force", C.c_double * 3),
                ("rep_enabled", C.c_int), ("rep_timescale", C.c_int), ("rep_const", C.c_double), ("rep_cutoff", C.c_double),
                ("brep_enabled", C.c_int), ("brep_timescale", C.c_int), ("brep_const", C.c_double), ("brep_cutoff", C.c_double),
                ("deletion_mode", C.c_int), ("dead", C.c_void_p), ("particles_deleted", C.c_long),
                ("n_regions", C.c_int), ("region_box", (C.c_int * 6) * 4), ("region_force", (C.c_double * 3) * 4)]


def build():
    """(Re)build the oracle shared library with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    LP, TP, SP = C.POINTER(Lattice), C.POINTER(CellType), C.POINTER(Sim)
    sig = {
        "orc_lattice_create": (LP, [C.c_int, C.c_int, C.c_int, c_int_p, C.c_double]),
        "orc_lattice_destroy": (None, [LP]),
        "orc_lattice_set_mask": (None, [LP, C.c_void_p]),
        "orc_lattice_init_equilibrium": (None, [LP, C.c_double, c_double_p]),
        "orc_lattice_set_force_uniform": (None, [LP, c_double_p]),
        "orc_lattice_set_force_box": (None, [LP, C.POINTER(C.c_int), c_double_p]),
        "orc_collide_stream": (None, [LP]),
        "orc_collide_stream_fused": (None, [LP]),
        "orc_lattice_set_threads": (None, [LP, C.c_int]),
        "orc_lattice_set_wall_velocity": (None, [LP, C.c_int, c_double_p]),
        "orc_node_rho_u": (None, [LP, C.c_long, c_double_p, c_double_p]),
        "orc_params_base": (None, [C.POINTER(Params)] + [C.c_double] * 5),
        "orc_celltype_create": (TP, [C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, c_long_p, C.c_int]),
        "orc_celltype_destroy": (None, [TP]),
        "orc_celltype_set_moduli": (None, [TP, C.POINTER(Params)] + [C.c_double] * 5),
        "orc_mesh_surface": (C.c_double, [TP]),
        "orc_cell_forces": (None, [TP, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
        "orc_phi2_stencil": (C.c_int, [LP, c_double_p, c_long_p, c_double_p]),
        "orc_sim_create": (SP, [LP, C.POINTER(Params)]),
        "orc_sim_destroy": (None, [SP]),
        "orc_sim_add_type": (C.c_int, [SP, TP]),
        "orc_sim_add_cell": (C.c_int, [SP, C.c_int, c_double_p, c_double_p, C.c_double]),
        "orc_sim_spread": (None, [SP]),
        "orc_sim_interpolate": (None, [SP]),
        "orc_sim_advance": (None, [SP]),
        "orc_sim_mechanics": (None, [SP, C.c_int]),
        "orc_sim_iterate": (None, [SP]),
        "orc_sim_repulsion": (None, [SP, C.c_double, C.c_double]),
        "orc_sim_boundary_repulsion": (None, [SP, C.c_double, C.c_double]),
        "orc_sim_type_offset": (C.c_long, [SP, C.c_int]),
        "orc_sim_get": (None, [SP, C.c_int, c_double_p]),
        "orc_sim_set": (None, [SP, C.c_int, c_double_p]),
        "orc_sim_add_vertex_force": (None, [SP, C.c_long, c_double_p]),
        "orc_sim_delete_incomplete_cells": (C.c_long, [SP]),
        "orc_sim_get_alive": (None, [SP, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def lptr(a):
    return a.ctypes.data_as(c_long_p)


class OracleLattice:
    """numpy-facing wrapper of orc_lattice."""

    def __init__(self, lib, nx, ny, nz, periodic, omega):
        self.lib = lib
        per = (C.c_int * 3)(*[int(p) for p in periodic])
        self.ptr = lib.orc_lattice_create(nx, ny, nz, per, omega)
        self.nx, self.ny, self.nz = nx, ny, nz
        self.n = nx * ny * nz

    def set_mask(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
        assert m.size == self.n
        self.lib.orc_lattice_set_mask(self.ptr, m.ctypes.data)

    def init_equilibrium(self, rho=1.0, u=(0.0, 0.0, 0.0)):
        uu = np.array(u, dtype=np.float64)
        self.lib.orc_lattice_init_equilibrium(self.ptr, rho, dptr(uu))

    def set_force_uniform(self, F):
        ff = np.array(F, dtype=np.float64)
        self.lib.orc_lattice_set_force_uniform(self.ptr, dptr(ff))

    def set_force_box(self, box, F):
        """setExternalVector on a sub-domain; box = inclusive (x0, x1, y0, y1, z0, z1)"""
        bb = (C.c_int * 6)(*[int(b) for b in box]); ff = np.array(F, dtype=np.float64)
        self.lib.orc_lattice_set_force_box(self.ptr, bb, dptr(ff))

    def set_threads(self, n):
        self.lib.orc_lattice_set_threads(self.ptr, n)

    def set_wall_velocity(self, cls, u):
        uu = np.array(u, dtype=np.float64)
        self.lib.orc_lattice_set_wall_velocity(self.ptr, cls, dptr(uu))

    def collide_stream(self, steps=1):
        for _ in range(steps):
            self.lib.orc_collide_stream(self.ptr)

    @property
    def f(self):  # [n][19] view
        return np.ctypeslib.as_array(self.ptr.contents.f, shape=(self.n, 19))

    @property
    def force(self):  # [n][3] view
        return np.ctypeslib.as_array(self.ptr.contents.force, shape=(self.n, 3))

    def destroy(self):
        if self.ptr:
            self.lib.orc_lattice_destroy(self.ptr)
            self.ptr = None


def make_params(lib, dx=5e-7, dt=1e-7, nu_p=1.1e-6, rho_p=1025.0, kBT=4.100531391e-21):
    P = Params()
    lib.orc_params_base(C.byref(P), dx, dt, nu_p, rho_p, kBT)
    return P


# examples/pipeflow/PLT.xml:14-38
PLT_INNER_EDGES = np.array([[60, 65], [62, 64], [37, 42], [54, 56], [34, 40], [25, 46], [50, 59], [29, 47],
                            [61, 63], [26, 45], [33, 43], [27, 35], [32, 39], [49, 51], [0, 4], [48, 52],
                            [6, 10], [53, 55], [19, 21], [57, 58], [15, 13]], dtype=np.int64)


def make_rbc(lib, P, radius=3.91e-6, min_tri=600, kLink=15.0, kArea=5.0, kVolume=20.0, kBend=80.0, eta_m=0.0):
    """examples/pipeflow/RBC.xml"""
    T = lib.orc_celltype_create(0, 1, radius / P.dx, min_tri, 0.3, None, 0)
    lib.orc_celltype_set_moduli(T, C.byref(P), kLink, kArea, kVolume, kBend, eta_m)
    return T


def make_plt(lib, P, radius=1.25e-6, min_tri=66, aspect=0.434782608696, kLink=25.0, kArea=8.0, kVolume=100.0,
             kBend=250.0, eta_m=0.0):
    """examples/pipeflow/PLT.xml"""
    ie = np.ascontiguousarray(PLT_INNER_EDGES)
    T = lib.orc_celltype_create(1, 6, radius / P.dx, min_tri, aspect, lptr(ie), len(ie))
    lib.orc_celltype_set_moduli(T, C.byref(P), kLink, kArea, kVolume, kBend, eta_m)
    return T
