"""ctypes binding of libwavehip.so (the C ABI of include/wavehip.h)."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# WAVEHIP_LIB selects another build of the same library (the diagnostic builds of tools/diag_build.sh
# and tools/dense_trace.sh); there is still no fallback if the file is missing.
LIB_PATH = os.environ.get("WAVEHIP_LIB") or os.path.join(HERE, "libwavehip.so")


class WavehipError(RuntimeError):
    """Mirrors the std::runtime_error the reference throws on device failures
    (common/cuda/array.hpp:15-17, mass.hpp:91-92, utils.hpp:30-34)."""


class Tuning(ctypes.Structure):
    """wf_tuning: explicit kernel selection / tuning (all zero = the library's choice)."""
    _fields_ = [("kernel", c_int), ("variant", c_int), ("lz", c_int), ("lz0", c_int),
                ("bx", c_int), ("by", c_int), ("bz", c_int), ("keep_cell_order", c_int), ("orient", c_int)]


class OpDesc(ctypes.Structure):
    _fields_ = [
        ("kind", c_int), ("degree", c_int), ("ncells", c_int), ("ndofs", c_int),
        ("h_dofmap", POINTER(c_int32)), ("h_perm", POINTER(c_int32)),
        ("h_G", POINTER(c_double)), ("h_detJ", POINTER(c_double)),
        ("nverts", c_int), ("h_xverts", POINTER(c_double)), ("h_geom_dofmap", POINTER(c_int32)),
        ("c0", c_double), ("flags", c_int),
        ("nq1", c_int), ("h_phi1", POINTER(c_double)),
        ("h_qpts1", POINTER(c_double)), ("h_qwts1", POINTER(c_double)),
        ("tuning", POINTER(Tuning)),
    ]


class DenseDesc(ctypes.Structure):
    _fields_ = [
        ("nd", c_int), ("nq", c_int), ("ncells", c_int), ("ndofs", c_int),
        ("h_dofmap", POINTER(c_int32)), ("h_dphi", POINTER(c_double)), ("h_weights", POINTER(c_double)),
        ("nverts", c_int), ("h_xverts", POINTER(c_double)), ("h_geom_dofmap", POINTER(c_int32)),
        ("c0", c_double), ("flags", c_int),
    ]


class OpInfo(ctypes.Structure):
    _fields_ = [
        ("kind", c_int), ("degree", c_int), ("num_cells", c_int), ("num_dofs_cell", c_int),
        ("num_quads", c_int), ("ndofs", c_int), ("structured", c_int),
        ("flops", c_double), ("alg_bytes", c_double), ("device_bytes", c_size_t),
        ("items_interior", c_int), ("items_interface", c_int),
        ("kernel", c_int), ("plan_items", c_int), ("plan_patterns", c_int), ("plan_lz", c_int),
        ("plan_reoriented", c_int), ("plan_fill", c_double),
    ]


class UpdaterDesc(ctypes.Structure):
    _fields_ = [
        ("ndofs", c_int),
        ("num_send_neighbors", c_int), ("send_neighbors", POINTER(c_int)),
        ("send_offsets", POINTER(c_int32)), ("send_indices", POINTER(c_int32)),
        ("num_recv_neighbors", c_int), ("recv_neighbors", POINTER(c_int)),
        ("recv_offsets", POINTER(c_int32)), ("ghost_positions", POINTER(c_int32)),
        ("flags", c_int),
    ]


MATVEC_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_void_p)


class CGDesc(ctypes.Structure):
    _fields_ = [
        ("n", c_int64), ("op", c_void_p), ("matvec", MATVEC_FN), ("user", c_void_p),
        ("updater", c_void_p), ("comm", c_void_p), ("kmax", c_int), ("rtol", c_double),
    ]


WF_QUAD_GLL, WF_QUAD_GAUSS_JACOBI = 0, 1
WF_VARIANT_GLL_WARPED, WF_VARIANT_EQUISPACED = 0, 1
WF_MAX_QUAD_POINTS = 16
WF_COMM_ID_BYTES = 128
WF_SUM, WF_MAX = 0, 1
WF_UPDATER_DEFAULT, WF_UPDATER_INLINE, WF_UPDATER_CHAIN_ON_SIDE = 0, 1, 2
(WF_KERNEL_NONE, WF_KERNEL_MARCH_BOX, WF_KERNEL_MARCH_IDX, WF_KERNEL_BATCH_UNIQUE, WF_KERNEL_BOX_BLOCK, WF_KERNEL_DIAGONAL,
 WF_KERNEL_MASS_DENSE_ANY, WF_KERNEL_DENSE_SIMPLEX, WF_KERNEL_ELEMENTWISE) = range(9)
(WF_KERNEL_AUTO, WF_KERNEL_FORCE_BATCH, WF_KERNEL_FORCE_BOX_BLOCK, WF_KERNEL_FORCE_MASS_ANY, WF_KERNEL_FORCE_ELEMENTWISE,
 WF_KERNEL_FORCE_MARCH) = range(6)
WF_OP_STIFFNESS, WF_OP_MASS_LUMPED, WF_OP_MASS_DENSE = 0, 1, 2
WF_FLAG_NONE, WF_FLAG_NO_FABS, WF_FLAG_NO_CLAMP, WF_FLAG_MASS_ELEMENTWISE, WF_FLAG_TENSOR_X_SLOWEST = 0, 1, 2, 4, 8
WF_PART_ALL, WF_PART_INTERIOR, WF_PART_INTERFACE, WF_PART_INTERIOR_A, WF_PART_INTERIOR_B = 0, 1, 2, 3, 4

# every symbol include/wavehip.h declares: name -> (restype, argtypes)
_dp, _ip, _vp = POINTER(c_double), POINTER(c_int32), c_void_p
SIGNATURES = {
    "wf_last_error": (c_char_p, []),
    "wf_version": (c_char_p, []),
    "wf_device_count": (c_int, [POINTER(c_int)]),
    "wf_set_device": (c_int, [c_int]),
    "wf_device_info": (c_int, [c_int, c_char_p, c_size_t, POINTER(c_size_t), POINTER(c_int)]),
    "wf_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "wf_free": (c_int, [c_void_p]),
    "wf_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "wf_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),
    "wf_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "wf_sync": (c_int, [c_void_p]),
    "wf_tabulate_gll": (c_int, [c_int, _dp, _dp, _dp]),
    "wf_tabulate_dense": (c_int, [c_int, _dp]),
    "wf_quadrature_1d": (c_int, [c_int, c_int, POINTER(c_int), _dp, _dp]),
    "wf_tabulate_1d": (c_int, [c_int, c_int, c_int, _dp, c_int, _dp]),
    "wf_geometry_hex_rule": (c_int, [c_int, c_int, _dp, _ip, c_int, _dp, _dp, c_int, c_int, _dp, _dp]),
    "wf_reorder_dofmap": (c_int, [c_int, c_int, _ip, _ip, _ip]),
    "wf_lattice_numbering": (c_int, [c_int, ctypes.c_int64, ctypes.c_int32, _ip, _ip]),
    "wf_geometry_hex": (c_int, [c_int, c_int, c_int, _dp, _ip, c_int, c_int, _dp, _dp]),
    "wf_op_create": (c_int, [POINTER(OpDesc), POINTER(c_void_p)]),
    "wf_op_create_box": (c_int, [c_int, c_int, c_int, c_int, c_int, _dp, c_double, c_int, POINTER(c_void_p)]),
    "wf_op_create_box_tuned": (c_int, [c_int, c_int, c_int, c_int, c_int, _dp, c_double, c_int, POINTER(Tuning),
                                       POINTER(c_void_p)]),
    "wf_op_set_ghost_dofs": (c_int, [c_void_p, _ip, c_int32]),
    "wf_op_create_dense_simplex": (c_int, [POINTER(DenseDesc), POINTER(c_void_p)]),
    "wf_op_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_op_set_ghost_faces": (c_int, [c_void_p, c_int, c_int, c_int]),
    "wf_op_apply_part": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "wf_op_info": (c_int, [c_void_p, POINTER(OpInfo)]),
    "wf_op_destroy": (c_int, [c_void_p]),
    "wf_gather": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_scatter_add": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_scatter_set": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_transform1": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_tsmm": (c_int, [c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_copy": (c_int, [c_int64, c_void_p, c_void_p, c_void_p]),
    "wf_fill": (c_int, [c_int64, c_double, c_void_p, c_void_p]),
    "wf_axpy": (c_int, [c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_scale": (c_int, [c_int64, c_double, c_void_p, c_void_p]),
    "wf_pointwise_div": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_pointwise_mult_add": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_dot": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_rk4_stage": (c_int, [c_int64, c_double, c_double, c_int] + [c_void_p] * 12),
    "wf_boundary_create": (c_int, [c_int64, c_int32, _ip, _dp, c_int32, _ip, _dp, POINTER(c_void_p)]),
    "wf_boundary_destroy": (c_int, [c_void_p]),
    "wf_boundary_apply_plan": (c_int, [c_void_p, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "wf_rk4_stage_bc": (c_int, [c_int64, c_double, c_double, c_int] + [c_void_p] * 11 + [c_void_p, c_double, c_double, c_void_p]),
    "wf_comm_unique_id": (c_int, [c_char_p]),
    "wf_comm_create": (c_int, [c_char_p, c_int, c_int, POINTER(c_void_p)]),
    "wf_comm_rendezvous_file": (c_int, [c_char_p, c_int, c_double, c_char_p]),
    "wf_comm_create_from_file": (c_int, [c_char_p, c_int, c_int, c_double, POINTER(c_void_p)]),
    "wf_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "wf_comm_allreduce": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "wf_comm_barrier": (c_int, [c_void_p, c_void_p]),
    "wf_comm_destroy": (c_int, [c_void_p]),
    "wf_updater_create": (c_int, [c_void_p, POINTER(UpdaterDesc), POINTER(c_void_p)]),
    "wf_updater_fwd_begin": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wf_updater_fwd_end": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wf_updater_fwd": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wf_updater_rev_begin": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wf_updater_rev_end": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wf_updater_rev": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wf_updater_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "wf_updater_destroy": (c_int, [c_void_p]),
    "wf_op_apply_overlapped": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wf_mesh_open": (c_int, [c_char_p, c_char_p, POINTER(c_void_p)]),
    "wf_mesh_sizes": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    "wf_mesh_read": (c_int, [c_void_p, _dp, _ip]),
    "wf_mesh_tags_size": (c_int, [c_void_p, c_char_p, POINTER(c_int64)]),
    "wf_mesh_read_tags": (c_int, [c_void_p, c_char_p, _ip, _ip]),
    "wf_mesh_close": (c_int, [c_void_p]),
    "wf_mesh_write": (c_int, [c_char_p, c_char_p, c_int64, _dp, c_int64, _ip, c_char_p, c_int64, _ip, _ip]),
    "wf_markers_enable": (c_int, [c_int]),
    "wf_marker_push": (c_int, [c_char_p]),
    "wf_marker_pop": (c_int, []),
    "wf_marker_mark": (c_int, [c_char_p]),
    "wf_fs_build": (c_int, [c_int, c_int64, _dp, c_int64, _ip, POINTER(c_int64), _ip, _dp, c_int64]),
    "wf_fs_locate_facets": (c_int, [c_int64, _ip, c_int64, _ip, _ip, _ip, _ip]),
    "wf_fs_facet_mass": (c_int, [c_int, c_int64, _dp, c_int64, _ip, _ip, c_int64, _ip, _ip, _ip, POINTER(c_int64), _ip, _dp]),
    "wf_fs_min_cell_diameter": (c_int, [c_int64, _dp, c_int64, _ip, POINTER(c_double)]),
    "wf_cg": (c_int, [POINTER(CGDesc), c_void_p, c_void_p, POINTER(c_int), POINTER(c_double), c_void_p]),
    "wf_boundary_apply": (c_int, [c_int32, c_void_p, c_void_p, c_double, c_int32, c_void_p, c_void_p, c_double,
                                  c_void_p, c_void_p, c_void_p]),
}

_LIB = None


def lib():
    """Load libwavehip.so.  Raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise WavehipError(
                f"{LIB_PATH} not found: build it with `python -m wave_fenics_amd.build` "
                "(there is no CPU fallback for the operator path)")
        # PyTorch-ROCm bundles its own HIP runtime; if libwavehip pulled in the system
        # libamdhip64 first, torch would later start a second runtime in the same
        # process and see no GPU.  Load torch's runtime first whenever torch is there
        # (libwavehip then binds to the already-loaded libamdhip64 by SONAME).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc: int):
    if rc != 0:
        msg = lib().wf_last_error().decode(errors="replace")
        raise WavehipError(f"libwavehip error {rc}: {msg}")
