"""
ctypes binding of libnimrud_hip.so (include/nimrud_hip.h).  nothing here computes: it declares the C
ABI and maps status codes to Python exceptions.  the library must be present - there is no fallback.

build it with `make -C nimrud_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NIMRUD_HIP_LIBRARY selects another build of the same ABI (diagnostic builds); default: in-tree
LIBRARY_PATH = os.environ.get("NIMRUD_HIP_LIBRARY") or os.path.join(_HERE, "libnimrud_hip.so")

NM_OK = 0
NM_ERR_INVALID = -1
NM_ERR_LATTICE = -2
NM_ERR_WORKSPACE = -3
NM_ERR_HIP = -4
NM_ERR_RADIUS = -5
NM_ERR_COMM = -6
NM_COMM_ID_BYTES = 128

ABI_VERSION = 6

c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int32
c_f64 = ctypes.c_double
c_ptr = ctypes.c_void_p
c_size = ctypes.c_size_t


class NmLattice(ctypes.Structure):
    """struct nm_lattice"""
    _fields_ = [("min_corner", c_f64 * 3), ("edge", c_f64), ("widths", c_i32 * 3),
                ("shifts", c_i32 * 2)]


class NmForest(ctypes.Structure):
    """struct nm_forest"""
    _fields_ = [("d_left", c_ptr), ("d_right", c_ptr), ("d_feature", c_ptr), ("d_threshold", c_ptr),
                ("d_value", c_ptr), ("d_roots", c_ptr), ("n_nodes", c_i32), ("n_trees", c_i32),
                ("n_classes", c_i32), ("n_features", c_i32), ("d_packed", c_ptr),
                ("d_leaf_value", c_ptr), ("d_packed_roots", c_ptr), ("n_leaves", c_i32),
                ("leaf_stride", c_i32), ("d_packed8", c_ptr)]


_LATP = ctypes.POINTER(NmLattice)

# name -> (restype, argtypes); every function declared in include/nimrud_hip.h
SIGNATURES = {
    "nm_create": (ctypes.c_int, [ctypes.POINTER(c_ptr), ctypes.c_int]),
    "nm_destroy": (None, [c_ptr]),
    "nm_last_error": (ctypes.c_char_p, [c_ptr]),
    "nm_abi_version": (ctypes.c_int, []),
    "nm_check": (ctypes.c_int, [c_ptr, ctypes.c_int]),
    "nm_clear_error": (ctypes.c_int, [c_ptr]),
    "nm_set_knn_fallback": (ctypes.c_int, [c_ptr, ctypes.c_int, c_f64]),
    "nm_profile_begin": (ctypes.c_int, [c_ptr]),
    "nm_set_overlap": (ctypes.c_int, [c_ptr, ctypes.c_int]),
    "nm_set_covariance_output": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64]),
    "nm_set_normal_output": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64]),
    "nm_profile_end": (ctypes.c_int, [c_ptr, ctypes.POINTER(c_f64 * 4), ctypes.POINTER(c_i64)]),
    "nm_bounds": (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "nm_spatial_order_workspace_bytes": (c_size, [c_i64]),
    "nm_spatial_order": (ctypes.c_int,
                         [c_ptr, c_ptr, c_i64, c_i64, _LATP, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "nm_voxelize_workspace_bytes": (c_size, [c_i64]),
    "nm_voxelize": (ctypes.c_int,
                    [c_ptr, c_ptr, c_i64, c_i64, _LATP, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "nm_coordinate_to_address": (ctypes.c_int,
                                 [c_ptr, c_ptr, c_i64, c_i64, _LATP, c_ptr, c_ptr, c_ptr]),
    "nm_address_to_coordinate": (ctypes.c_int, [c_ptr, c_ptr, c_i64, _LATP, c_ptr, c_ptr]),
    "nm_scale_workspace_bytes": (c_size, [c_i64, c_i64, _LATP]),
    "nm_scale_features": (ctypes.c_int,
                          [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, _LATP, c_f64, c_ptr,
                           c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "nm_multiscale_workspace_bytes": (c_size, [c_i64, c_i64, _LATP, c_i32]),
    "nm_multiscale_features": (ctypes.c_int,
                               [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, _LATP,
                                ctypes.POINTER(c_f64), c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_size,
                                c_ptr]),
    "nm_set_fuse_scales": (ctypes.c_int, [c_ptr, ctypes.c_int]),
    "nm_ladder_workspace_bytes": (c_size, [c_i64, c_i64, c_i32]),
    "nm_ladder_workspace_bytes_for": (c_size, [c_i64, c_i64, ctypes.POINTER(c_f64), c_i32]),
    "nm_ladder_features": (ctypes.c_int,
                           [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, ctypes.POINTER(c_f64),
                            ctypes.POINTER(c_f64), c_i32, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_size,
                            c_ptr]),
    "nm_set_forest_output": (ctypes.c_int, [c_ptr, ctypes.POINTER(NmForest), c_ptr, c_i64, c_ptr]),
    "nm_set_forest_mode": (ctypes.c_int, [c_ptr, ctypes.c_int]),
    "nm_field_workspace_bytes": (c_size, [c_i64, c_i64, _LATP, c_i32]),
    "nm_field_mean": (ctypes.c_int,
                      [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i32, _LATP, c_f64,
                       c_ptr, c_i64, c_ptr, c_size, c_ptr]),
    "nm_scale_neighbors": (ctypes.c_int,
                           [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, _LATP, c_f64, c_ptr, c_ptr,
                            c_ptr, c_ptr]),
    "nm_neighborhood_features": (ctypes.c_int,
                                 [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    "nm_halo_count": (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i32, c_i32, c_ptr, c_ptr]),
    "nm_halo_pack": (ctypes.c_int,
                     [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nm_copy_xyz": (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "nm_halo_cellset_workspace_bytes": (c_size, []),
    "nm_halo_cellset": (ctypes.c_int,
                        [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_f64, c_ptr, c_ptr, c_size, c_ptr]),
    "nm_halo_count_cells": (ctypes.c_int,
                            [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_f64, c_ptr, c_i32, c_i32, c_ptr,
                             c_ptr]),
    "nm_halo_pack_cells": (ctypes.c_int,
                           [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_f64, c_ptr, c_i32, c_i32, c_ptr,
                            c_ptr, c_ptr, c_ptr]),
    "nm_comm_unique_id": (ctypes.c_int, [c_ptr]),
    "nm_comm_create": (ctypes.c_int, [c_ptr, c_i32, c_i32, c_ptr, ctypes.POINTER(c_ptr)]),
    "nm_comm_destroy": (ctypes.c_int, [c_ptr, c_ptr]),
    "nm_halo_workspace_bytes": (c_size, [c_i64, c_i32]),
    "nm_halo_stats": (ctypes.c_int, [c_ptr, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "nm_halo_plan_from_matrix": (ctypes.c_int,
                                 [ctypes.POINTER(c_i64), c_i32, c_i32, ctypes.POINTER(c_i64),
                                  ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                  ctypes.POINTER(c_i32)]),
    "nm_halo_exchange": (ctypes.c_int,
                         [c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_i64, c_i64, c_f64, c_i32, c_ptr, c_i64,
                          ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_ptr, c_ptr, c_size, c_ptr]),
    "nm_descriptors": (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_i32, c_i64, c_ptr, c_i64, c_ptr]),
    "nm_forest_eval": (ctypes.c_int,
                       [c_ptr, ctypes.POINTER(NmForest), c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
}

_lib = None


class NimrudHipError(RuntimeError):
    """a HIP runtime failure or an internal error of libnimrud_hip.so"""


def load():
    """dlopen libnimrud_hip.so and declare its signatures.  raises ImportError when the library is
    not built: the product path has no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY_PATH):
        raise ImportError(
            "libnimrud_hip.so is not built (expected at %s). build it with "
            "`make -C nimrud_amd/csrc`; nimrud_amd has no CPU fallback." % LIBRARY_PATH)
    # torch first: the process must hold ONE HIP runtime, the one torch brings.  (loaded the other way round,
    # this library pulls in /opt/rocm's libamdhip64 and torch then brings its own; on the pool's boxes the
    # first one reports "no ROCm-capable device" - seen with `python __graft_entry__.py smoke`, which loads
    # the library in build() before smoke() imports torch.)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header and library out of sync
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.nm_abi_version() != ABI_VERSION:
        raise ImportError("libnimrud_hip.so ABI %d != binding ABI %d - rebuild the library"
                          % (lib.nm_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(lib, ctx, rc):
    """status -> exception.  lattice/argument/radius problems are ValueError like the reference's
    VoxelFilter (geometry.py:30-35,60,92-97); everything else is NimrudHipError."""
    if rc == NM_OK:
        return
    msg = lib.nm_last_error(ctx)
    msg = msg.decode("utf-8", "replace") if msg else "status %d" % rc
    if rc in (NM_ERR_INVALID, NM_ERR_LATTICE, NM_ERR_RADIUS):
        raise ValueError(msg)
    raise NimrudHipError("%s (status %d)" % (msg, rc))
