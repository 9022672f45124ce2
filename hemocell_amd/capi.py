"""ctypes binding of hemocell_amd/lib/libhemocell_amd.so (the C ABI declared in
include/hemocell_amd.h).  No compute happens in Python; if the HIP library is
missing or no gfx950 device is usable every call raises -- there is no CPU
fallback in the product path."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhemocell_amd.so")

c_double_p = C.POINTER(C.c_double)
c_long_p = C.POINTER(C.c_long)
c_int_p = C.POINTER(C.c_int)


class HcError(RuntimeError):
    pass


class Params(C.Structure):
    """hc_params == Parameters::lbm_base_parameters (mechanics/constantConversion.cpp:36-59)"""
    _fields_ = [(n, C.c_double) for n in
                ("dx", "dt", "nu_p", "rho_p", "kBT_p", "tau", "nu_lbm", "dm", "df", "f_limit", "kBT_lbm")]


class Material(C.Structure):
    """hc_material == <MaterialModel> of RBC.xml / PLT.xml"""
    _fields_ = [("kLink", C.c_double), ("kArea", C.c_double), ("kVolume", C.c_double), ("kBend", C.c_double),
                ("eta_m", C.c_double), ("radius", C.c_double), ("min_triangles", C.c_int),
                ("aspect_ratio", C.c_double), ("inner_edges", c_long_p), ("n_inner", C.c_int)]


# every symbol include/hemocell_amd.h declares: name -> (restype, argtypes)
VP = C.c_void_p
SIGNATURES = {
    "hc_last_error": (C.c_char_p, []),
    "hc_init": (C.c_int, [C.c_int]),
    "hc_device_count": (C.c_int, [c_int_p]),
    "hc_set_stream": (C.c_int, [VP]),
    "hc_synchronize": (C.c_int, []),
    "hc_fork": (C.c_int, []),
    "hc_route": (C.c_int, [C.c_int]),
    "hc_join": (C.c_int, []),
    "hc_side_stream": (C.c_int, [C.POINTER(VP)]),
    "hc_set_overlap": (C.c_int, [C.c_int]),
    "hc_comm_init_env": (C.c_int, []),
    "hc_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "hc_comm_finalize": (C.c_int, []),
    "hc_comm_info": (C.c_int, [c_int_p, c_int_p, c_int_p]),
    "hc_comm_barrier": (C.c_int, []),
    "hc_comm_allreduce": (C.c_int, [c_double_p, C.c_int, C.c_int]),
    "hc_comm_bcast": (C.c_int, [VP, C.c_size_t, C.c_int]),
    "hc_comm_allgather": (C.c_int, [VP, C.c_size_t, VP]),
    "hc_comm_exchange_host": (C.c_int, [C.c_int, VP, C.c_size_t, VP, C.c_size_t, VP, C.c_size_t, VP, C.c_size_t]),
    "hc_slab_stats": (C.c_int, [VP, c_double_p, C.c_int]),
    "hcl_slab_refresh_halos": (C.c_int, [VP, C.c_int]),
    "hcp_slab_sync_placement": (C.c_int, [VP, c_long_p]),
    "hcp_set_envelope": (C.c_int, [VP, C.c_double, c_double_p]),
    "hcp_envelope": (C.c_int, [VP, c_double_p, c_long_p]),
    "hcp_set_deletion_mode": (C.c_int, [VP, C.c_int]),
    "hcp_delete_incomplete_cells": (C.c_int, [VP, c_long_p]),
    "hcp_deletion_counts": (C.c_int, [VP, c_long_p, c_long_p, c_long_p, c_long_p]),
    "hcp_download_alive": (C.c_int, [VP, VP]),
    "hc_profile_enable": (C.c_int, [C.c_int]),
    "hc_profile_read": (C.c_int, [C.c_char_p, c_double_p, c_long_p]),
    "hc_profile_reset": (C.c_int, []),
    "hc_build_tag": (C.c_char_p, []),
    "hc_measure_copy_bandwidth": (C.c_int, [C.c_size_t, C.c_int, c_double_p]),
    "hcl_node_counts": (C.c_int, [VP, c_long_p]),
    "hc_debug_ibm_per_vertex": (C.c_int, [C.c_int]),
    "hc_set_reproducible_spread": (C.c_int, [C.c_int]),
    "hc_debug_force_plane_padding": (C.c_int, [C.c_int]),
    "hcl_create": (C.c_int, [C.POINTER(VP), C.c_int, C.c_int, C.c_int, c_int_p, C.c_double, C.c_int, C.c_int, C.c_int]),
    "hcl_destroy": (C.c_int, [VP]),
    "hcl_set_mask": (C.c_int, [VP, VP]),
    "hcl_init_equilibrium": (C.c_int, [VP, C.c_double, c_double_p]),
    "hcl_set_body_force_regions": (C.c_int, [VP, C.c_int, C.POINTER(C.c_int), c_double_p]),
    "hcl_set_body_force": (C.c_int, [VP, c_double_p]),
    "hcl_set_wall_velocity": (C.c_int, [VP, C.c_int, c_double_p]),
    "hcl_collide_stream": (C.c_int, [VP, C.c_int]),
    "hcl_collide_stream_part": (C.c_int, [VP, C.c_int]),
    "hcl_step_end": (C.c_int, [VP]),
    "hcl_download_populations": (C.c_int, [VP, c_double_p]),
    "hcl_upload_populations": (C.c_int, [VP, c_double_p]),
    "hcl_download_rho_u": (C.c_int, [VP, c_double_p, c_double_p]),
    "hcl_download_pi_neq": (C.c_int, [VP, c_double_p]),
    "hcl_download_ibm_force": (C.c_int, [VP, c_double_p]),
    "hcl_zero_ibm_force": (C.c_int, [VP]),
    "hcl_halo_doubles": (C.c_size_t, [VP, C.c_int]),
    "hcl_halo_pack": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "hcl_halo_unpack": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "hcl_halo_pack_next": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "hcl_face_velocity_pack": (C.c_int, [VP, C.c_int, VP]),
    "hcl_face_velocity_pack_both": (C.c_int, [VP, VP, VP]),
    "hcl_halo_pack_both": (C.c_int, [VP, VP, VP, C.c_int]),
    "hcl_halo_unpack_both": (C.c_int, [VP, VP, VP]),
    "hcl_zero_force_halos": (C.c_int, [VP]),
    "hcl_download_face_velocity": (C.c_int, [VP, C.c_int, c_double_p]),
    "hcl_dims": (C.c_int, [VP, c_int_p]),
    "hcl_mlups_bytes_per_node": (C.c_double, [VP]),
    "hc_params_base": (C.c_int, [C.POINTER(Params)] + [C.c_double] * 5),
    "hcp_celltype_create": (C.c_int, [C.POINTER(VP), C.c_int, C.c_int, C.POINTER(Params), C.POINTER(Material)]),
    "hcp_celltype_destroy": (C.c_int, [VP]),
    "hcp_celltype_sizes": (C.c_int, [VP, c_int_p]),
    "hcp_celltype_tables": (C.c_int, [VP, c_double_p, c_long_p, c_long_p, c_double_p, c_double_p, c_double_p,
                                      c_long_p, c_double_p, c_double_p]),
    "hcp_celltype_tables2": (C.c_int, [VP, c_long_p, c_long_p, c_long_p, c_double_p, c_int_p]),
    "hcp_create": (C.c_int, [C.POINTER(VP), VP, C.POINTER(Params)]),
    "hcp_destroy": (C.c_int, [VP]),
    "hcp_add_type": (C.c_int, [VP, VP, C.c_int, c_int_p]),
    "hcp_add_cell": (C.c_int, [VP, C.c_int, C.c_long, c_double_p, c_double_p, C.c_double, c_int_p]),
    "hcp_add_cell_unchecked": (C.c_int, [VP, C.c_int, C.c_long, c_double_p, c_double_p]),
    "hcp_counts": (C.c_int, [VP, c_long_p, c_long_p, c_long_p]),
    "hcp_type_range": (C.c_int, [VP, C.c_int, c_long_p, c_long_p]),
    "hcp_download": (C.c_int, [VP, C.c_int, c_double_p]),
    "hcp_upload": (C.c_int, [VP, C.c_int, c_double_p]),
    "hcp_download_cell_ids": (C.c_int, [VP, c_long_p]),
    "hcp_download_records": (C.c_int, [VP, C.c_void_p, C.c_long]),
    "hcp_upload_records": (C.c_int, [VP, C.c_void_p, C.c_long]),
    "hcp_add_vertex_force": (C.c_int, [VP, c_long_p, C.c_int, c_double_p]),
    "hcp_set_repulsion": (C.c_int, [VP, C.c_double, C.c_double, C.c_int]),
    "hcp_repulsion": (C.c_int, [VP]),
    "hcp_download_repulsion": (C.c_int, [VP, c_double_p]),
    "hcl_fluid_stats": (C.c_int, [VP, C.c_int, c_double_p, C.POINTER(C.c_long)]),
    "hcp_vertex_stats": (C.c_int, [VP, C.c_int, c_double_p, C.POINTER(C.c_long)]),
    "hcp_set_boundary_repulsion": (C.c_int, [VP, C.c_double, C.c_double, C.c_int]),
    "hcp_boundary_repulsion": (C.c_int, [VP]),
    "hcp_spread": (C.c_int, [VP, C.c_int]),
    "hcp_interpolate": (C.c_int, [VP]),
    "hcp_interpolate_cells": (C.c_int, [VP, C.c_int, c_int_p, C.c_int]),
    "hcp_advance": (C.c_int, [VP, C.c_int]),
    "hcp_mechanics": (C.c_int, [VP, C.c_long, C.c_int]),
    "hcp_mechanics_components": (C.c_int, [VP, C.c_int, c_double_p]),
    "hc_iterate": (C.c_int, [VP, VP, c_long_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hcp_cell_extents": (C.c_int, [VP, C.c_int, c_double_p]),
    "hcp_cell_extents_begin": (C.c_int, [VP, C.c_int]),
    "hcp_cell_extents_end": (C.c_int, [VP, C.c_int, c_double_p]),
    "hcp_record_doubles": (C.c_size_t, [VP, C.c_int]),
    "hcp_pack_cells": (C.c_int, [VP, C.c_int, c_int_p, C.c_int, C.c_double, VP]),
    "hcp_unpack_cells": (C.c_int, [VP, C.c_int, c_int_p, c_long_p, c_int_p, C.c_int, VP]),
    "hcp_remove_cells": (C.c_int, [VP, C.c_int, c_int_p, C.c_int]),
    "hcp_owned_vertices": (C.c_int, [VP, c_long_p]),
    "hcp_cell_info": (C.c_int, [VP, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
}

_lib = None


def build():
    """Compile the HIP library in-tree with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    """Load the shared library and bind every declared symbol."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HcError("libhemocell_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                          "the product path has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise HcError(lib().hc_last_error().decode("utf-8", "replace"))


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def lptr(a):
    return a.ctypes.data_as(c_long_p)
