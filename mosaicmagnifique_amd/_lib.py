"""ctypes binding of ``libmosaic_hip.so`` (include/mosaic_hip.h).

There is no CPU fallback: if the library is missing or cannot be loaded this
module raises, and every product path above it fails with it.
"""
import ctypes
import os
import sys

import numpy as np

from . import _build

c_int_p = ctypes.POINTER(ctypes.c_int)


class MMCell(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("x", "y", "row_min", "row_max", "col_min", "col_max", "mask_id", "grid_x", "grid_y")]


class MMTiming(ctypes.Structure):
    _fields_ = [("stage_ms", ctypes.c_double), ("score_ms", ctypes.c_double),
                ("select_ms", ctypes.c_double), ("resolve_ms", ctypes.c_double),
                ("total_ms", ctypes.c_double), ("score_launches", ctypes.c_int64),
                ("pair_evals", ctypes.c_int64), ("exact_rescored", ctypes.c_int64),
                ("dense_rows", ctypes.c_int64), ("rescore_ms", ctypes.c_double),
                ("dense_cells", ctypes.c_int64), ("edge_pairs", ctypes.c_int64),
                ("seed_ms", ctypes.c_double), ("build_ms", ctypes.c_double),
                ("spec_rounds", ctypes.c_int64), ("deepened_cells", ctypes.c_int64),
                ("resolve_sweeps", ctypes.c_int64), ("visited_groups", ctypes.c_int64),
                ("total_groups", ctypes.c_int64), ("bound_dropped", ctypes.c_int64),
                ("band_alarms", ctypes.c_int64), ("filter_tile", ctypes.c_int64),
                ("host_syncs", ctypes.c_int64), ("fast_path", ctypes.c_int64),
                ("preprocess_ms", ctypes.c_double), ("filter_off_range", ctypes.c_int64), ("pretier", ctypes.c_int64), ("audit_ms", ctypes.c_double),
                ("voided_launches", ctypes.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MMStageTiming(ctypes.Structure):
    _fields_ = [("create_ms", ctypes.c_double), ("alloc_ms", ctypes.c_double), ("pinned_ms", ctypes.c_double),
                ("h2d_ms", ctypes.c_double), ("preprocess_ms", ctypes.c_double), ("planes_ms", ctypes.c_double),
                ("allocations", ctypes.c_int64), ("allocated_bytes", ctypes.c_int64), ("reused_engine", ctypes.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


ABI_VERSION = 4          # MM_ABI_VERSION of include/mosaic_hip.h

PROGRESS_CB = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_void_p)

MM_OK, MM_CANCELLED, MM_INVALID_ARGUMENT, MM_OUT_OF_MEMORY, MM_DEVICE_ERROR, MM_NOT_READY = range(6)
STATUS_NAMES = ("MM_OK", "MM_CANCELLED", "MM_INVALID_ARGUMENT", "MM_OUT_OF_MEMORY",
                "MM_DEVICE_ERROR", "MM_NOT_READY")

# Every symbol include/mosaic_hip.h and include/mosaic_hip_tuning.h declare: (restype, argtypes).
_VP = ctypes.c_void_p
SIGNATURES = {
    "mm_abi_version": (ctypes.c_int, []),
    "mm_debug_has_pretier": (ctypes.c_int, []),
    "mm_device_count": (ctypes.c_int, [c_int_p]),
    "mm_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_VP)]),
    "mm_destroy": (None, [_VP]),
    "mm_trim_cache": (ctypes.c_int, []),
    "mm_last_error": (ctypes.c_char_p, [_VP]),
    "mm_set_library": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int]),
    "mm_set_masks": (ctypes.c_int, [_VP, _VP, ctypes.c_int]),
    "mm_set_target": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int]),
    "mm_set_library_u8": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mm_set_target_u8": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mm_set_library_u8_sized": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mm_set_cell_resize": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int]),
    "mm_hue_rotate_u8": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_float, _VP]),
    "mm_shrink_library": (ctypes.c_int, [_VP, ctypes.c_int]),
    "mm_set_cell_scale": (ctypes.c_int, [_VP, ctypes.c_int]),
    "mm_library_shape": (ctypes.c_int, [_VP, c_int_p, c_int_p]),
    "mm_get_library": (ctypes.c_int, [_VP, _VP]),
    "mm_get_target": (ctypes.c_int, [_VP, _VP]),
    "mm_generate": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, _VP, _VP, _VP, _VP, _VP]),
    "mm_candidates_per_cell": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mm_score_cells": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP,
                                      ctypes.c_int, _VP, _VP, _VP]),
    "mm_speculative_candidates": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mm_score_cells_partial": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP,
                                              ctypes.c_int, _VP, _VP, _VP]),
    "mm_resolve_verify": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP, ctypes.c_int,
                                         _VP, _VP, _VP, _VP]),
    "mm_resolve": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP, ctypes.c_int, _VP, _VP]),
    "mm_set_library_bgr": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int]),
    "mm_build_photomosaic": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP,
                                            ctypes.c_int, ctypes.c_int, _VP]),
    "mm_build_photomosaic_step": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP]),
    "mm_shrink_library_bgr": (ctypes.c_int, [_VP, ctypes.c_int]),
    "mm_set_option": (ctypes.c_int, [_VP, ctypes.c_char_p, ctypes.c_int]),
    "mm_set_main_image_bgr": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int]),
    "mm_cell_entropy": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, _VP, _VP]),
    "mm_group_create": (ctypes.c_int, [c_int_p, ctypes.c_int, ctypes.POINTER(_VP)]),
    "mm_group_destroy": (None, [_VP]),
    "mm_group_last_error": (ctypes.c_char_p, [_VP]),
    "mm_group_size": (ctypes.c_int, [_VP]),
    "mm_group_uses_rccl": (ctypes.c_int, [_VP]),
    "mm_group_engine": (_VP, [_VP, ctypes.c_int]),
    "mm_group_set_library": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int]),
    "mm_group_set_library_u8": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mm_group_set_masks": (ctypes.c_int, [_VP, _VP, ctypes.c_int]),
    "mm_group_set_target": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int]),
    "mm_group_set_target_u8": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mm_group_set_option": (ctypes.c_int, [_VP, ctypes.c_char_p, ctypes.c_int]),
    "mm_group_generate": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, _VP, _VP, _VP, _VP, _VP]),
    "mm_get_timing": (ctypes.c_int, [_VP, ctypes.POINTER(MMTiming)]),
    "mm_get_stage_timing": (ctypes.c_int, [_VP, ctypes.POINTER(MMStageTiming)]),
    "mm_colour_diff_batch": (ctypes.c_int, [_VP, ctypes.c_int, _VP, _VP, ctypes.c_size_t, _VP]),
    "mm_filter_diff_batch": (ctypes.c_int, [_VP, ctypes.c_int, _VP, _VP, ctypes.c_size_t, _VP]),
    "mm_debug_visit_counters": (ctypes.c_int, [_VP, _VP]),
    "mm_debug_filter_band_sweep": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                                  ctypes.POINTER(ctypes.c_double)]),
    "mm_debug_stream_read": (ctypes.c_int, [_VP, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def library_path():
    # MOSAIC_HIP_LIB: an alternative build of the same engine (kernel tuning experiments)
    return os.environ.get("MOSAIC_HIP_LIB") or _build.LIB


def load():
    """Load (once) and type the C ABI.  Raises if the HIP engine is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "HIP engine %s is not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % path)
    # One HIP runtime per process: torch ships its own libamdhip64 (same SONAME
    # as /opt/rocm's).  Importing torch first makes the loader hand this library
    # the runtime torch already mapped, so device pointers / streams are shared
    # with torch.distributed (RCCL) in the multi-GPU path.
    if "torch" not in sys.modules and not os.environ.get("MOSAIC_HIP_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = missing export: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mm_abi_version() != ABI_VERSION:
        raise RuntimeError("libmosaic_hip.so ABI version %d != %d" % (lib.mm_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    # (the array interface is several times cheaper than a.ctypes.data_as: it matters to sub-millisecond calls)
    return ctypes.c_void_p(a.__array_interface__["data"][0])


def as_cells(records):
    """numpy structured records (grid.CELL_DTYPE) -> contiguous buffer matching mm_cell."""
    rec = np.ascontiguousarray(records)
    assert rec.dtype.itemsize == ctypes.sizeof(MMCell), (rec.dtype.itemsize, ctypes.sizeof(MMCell))
    return rec
