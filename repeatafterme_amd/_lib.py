"""ctypes loader for libramx.so (include/ramx.h).  Fails loudly: there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAMX_LIB") or os.path.join(_PKG, "libramx.so")   # RAMX_LIB: A/B builds (tools/build_variant.sh)
CLI_PATH = os.path.join(_PKG, "RAMExtend")

# every symbol include/ramx.h declares (tests/test_cabi.py checks the .so exports all of them)
EXPORTS = [
    "ramx_set_runtime", "ramx_extend_alignment", "ramx_extend_flat", "ramx_invalidate_library", "ramx_preload_library", "ramx_resolve_flanks", "ramx_last_error", "ramx_device_count",
    "ramx_dev_create", "ramx_dev_destroy", "ramx_dev_load_library", "ramx_dev_begin_direction",
    "ramx_dev_run_direction", "ramx_dev_download", "ramx_dev_peek_state", "ramx_dev_peek_family_state", "ramx_dev_set_row_trace", "ramx_dev_run_families", "ramx_extend_batch", "ramx_comm_unique_id",
    "ramx_dev_comm_init", "ramx_dev_comm_size", "ramx_dev_set_allreduce_cb", "ramx_dev_peer_export", "ramx_dev_peer_import",
    "ramx_dev_peer_selftest", "ramx_dev_peer_enable", "ramx_dev_hostbox_attach", "ramx_hostbox_unlink", "ramx_get_matrix", "ramx_get_matrix_using_gap_penalties",
    "ramx_get_repeatscout_matrix", "ramx_free_scoring_system", "ramx_calculate_lambda",
    "ramx_load_sequence_subset_minimal", "ramx_load_sequence_subset_packed", "ramx_packed_decode", "ramx_dev_load_library_packed", "ramx_preload_library_packed", "ramx_free_library", "ramx_overlap_avoidance",
    "ramx_print_core_edges", "ramx_allocate_score", "ramx_free_score", "ramx_cli_main",
]


class RamxError(RuntimeError):
    pass


class FlatCores(C.Structure):
    _fields_ = [("n", C.c_int32)] + [(k, C.c_void_p) for k in
                ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext", "right_ext",
                 "left_len", "right_len", "score")]


class Params(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("bandwidth", "cappenalty", "minimprovement", "L", "when_to_stop",
                                          "l", "gapopen", "gapextn")] + [("matrix", C.c_void_p)]


class RunInfo(C.Structure):
    _fields_ = [("ret", C.c_int32), ("rows_executed", C.c_int32), ("limit_warning", C.c_int32),
                ("overflow32", C.c_int32), ("n_extendable", C.c_int32), ("launches", C.c_int32),
                ("loop_ms", C.c_double), ("kernel_ms_avg", C.c_double), ("kernel_samples", C.c_int32),
                ("prep_ms", C.c_double), ("persistent", C.c_int32), ("lanes_per_flank", C.c_int32),
                ("respeculated_rows", C.c_int32), ("packed_rows", C.c_int32), ("lean_rows", C.c_int32)]


class Family(C.Structure):
    _fields_ = [("cores", FlatCores), ("sequence", C.c_void_p), ("seq_len", C.c_uint64), ("master", C.c_void_p)]


class Flank(C.Structure):
    _fields_ = [("start", C.c_int64), ("t_lo", C.c_int32), ("t_hi", C.c_int32), ("step", C.c_int8),
                ("compl_", C.c_int8), ("pad_", C.c_int8 * 6)]


ALLREDUCE_CB = C.CFUNCTYPE(None, C.POINTER(C.c_longlong), C.c_void_p)


def build(force: bool = False) -> None:
    """Compile libramx.so + RAMExtend in-tree (hipcc --offload-arch=gfx950, gcc)."""
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(CLI_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc"), "all"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RamxError(f"{LIB_PATH} is missing: build it with `make -C repeatafterme_amd/csrc` "
                            "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.ramx_last_error.restype = C.c_char_p
        L.ramx_set_runtime.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ramx_set_runtime.restype = None
        L.ramx_extend_flat.argtypes = [C.c_int, C.POINTER(FlatCores), C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.POINTER(Params), C.POINTER(RunInfo)]
        L.ramx_extend_flat.restype = C.c_int
        L.ramx_resolve_flanks.argtypes = [C.c_int, C.POINTER(FlatCores), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ramx_resolve_flanks.restype = C.c_int
        L.ramx_extend_batch.argtypes = [C.c_int, C.POINTER(Family), C.c_int32, C.POINTER(Params), C.POINTER(RunInfo)]
        L.ramx_extend_batch.restype = C.c_int
        L.ramx_device_count.restype = C.c_int
        L.ramx_dev_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.ramx_dev_destroy.argtypes = [C.c_void_p]
        L.ramx_dev_destroy.restype = None
        L.ramx_dev_load_library.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.ramx_dev_begin_direction.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(Params)]
        L.ramx_dev_run_direction.argtypes = [C.c_void_p, C.POINTER(RunInfo)]
        L.ramx_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.ramx_dev_peek_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32)]
        L.ramx_dev_peek_family_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ramx_comm_unique_id.argtypes = [C.c_void_p]
        L.ramx_dev_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ramx_dev_comm_size.argtypes = [C.c_void_p]
        L.ramx_dev_set_allreduce_cb.argtypes = [C.c_void_p, ALLREDUCE_CB, C.c_void_p]
        L.ramx_dev_peer_export.argtypes = [C.c_void_p, C.c_void_p]
        L.ramx_dev_peer_import.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ramx_dev_peer_selftest.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.ramx_dev_peer_enable.argtypes = [C.c_void_p, C.c_int]
        L.ramx_dev_hostbox_attach.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.ramx_hostbox_unlink.argtypes = [C.c_char_p]
        if hasattr(L, "ramx_cli_main"):
            L.ramx_cli_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        _lib = L
    return _lib


def check(rc: int, what: str) -> int:
    if rc < 0:
        raise RamxError(f"{what} failed ({rc}): {lib().ramx_last_error().decode()}")
    return rc
