"""Python host-side mirror of the reference's extension-loop interface, bound to libramx.so.

``extend_alignment`` has the argument meaning, return value and side effects of the reference's
``extend_alignment`` (ram_extend.c:859-1258): it updates ``master`` and the cores' extension
lengths / scores in place and returns ``max_extension_score_row_idx + 1``.  All computation runs
in the HIP kernels behind the C-ABI; a missing library or GPU raises ``RamxError``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .datamodel import CoreSet, ExtendParams


@dataclass
class RunInfo:
    ret: int
    rows_executed: int
    limit_warning: int
    overflow32: int
    n_extendable: int
    launches: int
    loop_ms: float
    kernel_ms_avg: float
    kernel_samples: int
    prep_ms: float
    persistent: int = 0
    lanes_per_flank: int = 1
    respeculated_rows: int = 0
    packed_rows: int = 0
    lean_rows: int = 0


def _params(p: ExtendParams):
    mat = np.ascontiguousarray(p.matrix, dtype=np.int32)
    assert mat.size == 100 * 100
    cp = _lib.Params(p.bandwidth, p.cappenalty, p.minimprovement, p.L, p.when_to_stop, p.l, p.gapopen,
                     p.gapextn, mat.ctypes.data)
    return cp, mat


def _info(ci: _lib.RunInfo) -> RunInfo:
    return RunInfo(**{k: getattr(ci, k) for k, _ in _lib.RunInfo._fields_})


def extend_alignment(direction: int, cores: CoreSet, sequence: np.ndarray, master: np.ndarray,
                     p: ExtendParams) -> RunInfo:
    """direction: 1 = right, 0 = left (reference ram_extend.c:424,506)."""
    L = _lib.lib()
    assert sequence.dtype == np.int8 and sequence.flags.c_contiguous
    assert master.dtype == np.int8 and len(master) >= 2 * p.L + p.l + 1
    fc = _lib.FlatCores(cores.n, *[getattr(cores, k).ctypes.data for k in
                                   ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext",
                                    "right_ext", "left_len", "right_len", "score")])
    cp, _keep = _params(p)
    ci = _lib.RunInfo()
    rc = L.ramx_extend_flat(int(direction), C.byref(fc), sequence.ctypes.data, len(sequence),
                            master.ctypes.data, C.byref(cp), C.byref(ci))
    _lib.check(rc, "ramx_extend_flat")
    return _info(ci)


def extend_batch(direction: int, families, p: ExtendParams):
    """Many families in one launch (C-ABI ramx_extend_batch).  `families` is a list of (cores, sequence, master);
    every family is updated in place exactly like extend_alignment does for one.  Returns one RunInfo per family."""
    L = _lib.lib()
    n = len(families)
    arr = (_lib.Family * max(n, 1))()
    keep = []
    for i, (cores, sequence, master) in enumerate(families):
        assert sequence.dtype == np.int8 and sequence.flags.c_contiguous and master.dtype == np.int8
        arr[i].cores = _lib.FlatCores(cores.n, *[getattr(cores, k).ctypes.data for k in
                                                 ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext",
                                                  "right_ext", "left_len", "right_len", "score")])
        arr[i].sequence = sequence.ctypes.data
        arr[i].seq_len = len(sequence)
        arr[i].master = master.ctypes.data
        keep.append((cores, sequence, master))
    cp, _keep = _params(p)
    infos = (_lib.RunInfo * max(n, 1))()
    _lib.check(L.ramx_extend_batch(int(direction), arr, n, C.byref(cp), infos), "ramx_extend_batch")
    return [_info(infos[i]) for i in range(n)]
