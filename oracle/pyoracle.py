"""ctypes bindings for the CPU oracle and (when present) the compiled reference.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (repeatafterme_amd) never
imports this module.

Two checkers are exposed with one calling convention:

* ``oracle_extend``  -> oracle/libramx_oracle.so, our plain-C restatement
  (oracle/ramx_oracle.c) of ram_extend.c:859-1258 + bnw_extend.c:750-1048.
* ``ref_extend``     -> oracle/_ref/libramref.so, the reference itself compiled
  from /root/reference by oracle/Makefile (``make ref``).  Only available where
  that file exists (build container; it also travels to the GPU box as a built
  artefact).  ``have_ref()`` tells.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libramx_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libramref.so")
REF_CLI = os.path.join(_HERE, "_ref", "RAMExtend_ref")
MSIZE = 100


def build(ref: bool = True) -> None:
    """Compile the oracle (and the reference when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.exists("/root/reference/ram_extend.c"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref", "-j4"])


def have_ref() -> bool:
    return os.path.exists(_REF_SO)


# ---------------------------------------------------------------- data model (flat)

@dataclass
class Cores:
    """Flat mirror of the reference's coreAlignment list (common.h:80-98)."""
    left_pos: np.ndarray
    right_pos: np.ndarray
    lower: np.ndarray
    upper: np.ndarray
    orient: np.ndarray
    left_ext: np.ndarray
    right_ext: np.ndarray
    seq_idx: Optional[np.ndarray] = None
    left_len: np.ndarray = field(default=None)
    right_len: np.ndarray = field(default=None)
    score: np.ndarray = field(default=None)

    def __post_init__(self):
        n = len(self.left_pos)
        self.left_pos = np.ascontiguousarray(self.left_pos, dtype=np.int64)
        self.right_pos = np.ascontiguousarray(self.right_pos, dtype=np.int64)
        self.lower = np.ascontiguousarray(self.lower, dtype=np.int64)
        self.upper = np.ascontiguousarray(self.upper, dtype=np.int64)
        self.orient = np.ascontiguousarray(self.orient, dtype=np.int8)
        self.left_ext = np.ascontiguousarray(self.left_ext, dtype=np.int8)
        self.right_ext = np.ascontiguousarray(self.right_ext, dtype=np.int8)
        if self.seq_idx is None:
            self.seq_idx = np.arange(n, dtype=np.int32)
        self.seq_idx = np.ascontiguousarray(self.seq_idx, dtype=np.int32)
        for name in ("left_len", "right_len", "score"):
            v = getattr(self, name)
            setattr(self, name, np.zeros(n, np.int32) if v is None else np.ascontiguousarray(v, dtype=np.int32))

    @property
    def n(self) -> int:
        return len(self.left_pos)

    def copy(self) -> "Cores":
        return Cores(**{k: (None if getattr(self, k) is None else getattr(self, k).copy())
                        for k in self.__dataclass_fields__})


@dataclass
class Params:
    bandwidth: int = 14
    cappenalty: int = -90
    minimprovement: int = 27
    L: int = 10000
    when_to_stop: int = 100
    l: int = 1
    gapopen: int = -28
    gapextn: int = -5
    matrix: np.ndarray = None  # int32 [100*100]

    @staticmethod
    def named(matrix: str, **kw) -> "Params":
        """CLI defaults per matrix, ram_extend.c:280-344."""
        if matrix == "repeatscout":
            m, go, ge = get_repeatscout_matrix(kw.pop("match", 1), kw.pop("mismatch", -1), kw.pop("gap", -5))
            d = dict(minimprovement=3, cappenalty=-20)
        else:
            m, go, ge = get_matrix(matrix)
            d = dict(minimprovement=24 if matrix == "25p43g" else 27, cappenalty=-90)
        d.update(gapopen=go, gapextn=ge, matrix=m)
        d.update(kw)
        return Params(**d)


@dataclass
class Result:
    ret: int
    master: np.ndarray
    left_len: np.ndarray
    right_len: np.ndarray
    score: np.ndarray
    rows_executed: int = -1
    limit_warning: int = -1
    col_sums: Optional[np.ndarray] = None
    col_base: Optional[np.ndarray] = None
    col_score: Optional[np.ndarray] = None
    row_best: Optional[np.ndarray] = None
    row_best_idx: Optional[np.ndarray] = None


def new_master(L: int, l: int = 1) -> np.ndarray:
    """master = malloc(2L+l+1); master[L..L+l) = 99 (ram_extend.c:347-353,415-416)."""
    m = np.zeros(2 * L + l + 1, np.int8)
    m[L:L + l] = 99
    return m


# ---------------------------------------------------------------- oracle (.so of ours)

class _OCores(C.Structure):
    _fields_ = [("n", C.c_int32)] + [(k, C.c_void_p) for k in
                ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext", "right_ext",
                 "left_len", "right_len", "score")]


class _OParams(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("bandwidth", "cappenalty", "minimprovement", "L", "when_to_stop",
                                          "l", "gapopen", "gapextn")] + [("matrix", C.c_void_p)]


class _OTrace(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("col_sums", "col_base", "col_score", "rows_executed",
                                          "limit_warning", "row_best", "row_best_idx")]


_olib = None


def _oracle():
    global _olib
    if _olib is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        _olib = C.CDLL(_ORACLE_SO)
        _olib.ramx_oracle_extend.restype = C.c_int
        _olib.ramx_oracle_extend.argtypes = [C.c_int, C.POINTER(_OCores), C.c_void_p, C.c_void_p,
                                             C.POINTER(_OParams), C.POINTER(_OTrace)]
        _olib.ramx_oracle_nw_row.restype = C.c_int
        _olib.ramx_oracle_nw_row.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                             C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                             C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_int, C.c_int]
        _olib.ramx_oracle_get_matrix.restype = C.c_int
        _olib.ramx_oracle_get_matrix.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _olib.ramx_oracle_get_repeatscout_matrix.restype = None
        _olib.ramx_oracle_get_repeatscout_matrix.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                             C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return _olib


def get_matrix(name: str):
    m = np.zeros(MSIZE * MSIZE, np.int32)
    go, ge = C.c_int(), C.c_int()
    if _oracle().ramx_oracle_get_matrix(name.encode(), m.ctypes.data, C.byref(go), C.byref(ge)) != 0:
        raise ValueError(f"unknown matrix {name}")
    return m, go.value, ge.value


def get_repeatscout_matrix(match=1, mismatch=-1, gap=-5):
    m = np.zeros(MSIZE * MSIZE, np.int32)
    go, ge = C.c_int(), C.c_int()
    _oracle().ramx_oracle_get_repeatscout_matrix(match, mismatch, gap, m.ctypes.data, C.byref(go), C.byref(ge))
    return m, go.value, ge.value


def oracle_nw_row(direction, row_idx, n, n_align, cons_base, left_pos, right_pos, orient, score,
                  lower, upper, sequence, matrix, gapopen, gapextn, bandwidth):
    idx = C.c_int(0)
    assert score.dtype == np.int32 and score.flags.c_contiguous
    best = _oracle().ramx_oracle_nw_row(direction, row_idx, n, n_align, cons_base, left_pos, right_pos, orient,
                                        score.ctypes.data, lower, upper, sequence.ctypes.data, C.byref(idx),
                                        matrix.ctypes.data, gapopen, gapextn, bandwidth)
    return best, idx.value


def oracle_extend(direction: int, cores: Cores, sequence: np.ndarray, master: np.ndarray, p: Params,
                  trace: bool = False, row_trace: bool = False) -> Result:
    """Runs the oracle; updates cores.left_len/right_len/score and master in place (like the reference)."""
    lib = _oracle()
    sequence = np.ascontiguousarray(sequence, dtype=np.int8)
    assert master.dtype == np.int8 and len(master) >= 2 * p.L + p.l + 1
    oc = _OCores(cores.n, *[getattr(cores, k).ctypes.data for k in
                            ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext", "right_ext",
                             "left_len", "right_len", "score")])
    mat = np.ascontiguousarray(p.matrix, dtype=np.int32)
    op = _OParams(p.bandwidth, p.cappenalty, p.minimprovement, p.L, p.when_to_stop, p.l, p.gapopen, p.gapextn,
                  mat.ctypes.data)
    rows = np.zeros(1, np.int32)
    warn = np.zeros(1, np.int32)
    res = Result(0, master, cores.left_len, cores.right_len, cores.score)
    tr = _OTrace(None, None, None, rows.ctypes.data, warn.ctypes.data, None, None)
    if trace:
        res.col_sums = np.zeros((p.L, 4), np.int64)
        res.col_base = np.zeros(p.L, np.int8)
        res.col_score = np.zeros(p.L, np.int32)
        tr.col_sums, tr.col_base, tr.col_score = (res.col_sums.ctypes.data, res.col_base.ctypes.data,
                                                  res.col_score.ctypes.data)
    if row_trace:
        res.row_best = np.zeros((p.L, max(cores.n, 1)), np.int32)
        res.row_best_idx = np.zeros((p.L, max(cores.n, 1)), np.int32)
        tr.row_best, tr.row_best_idx = res.row_best.ctypes.data, res.row_best_idx.ctypes.data
    res.ret = lib.ramx_oracle_extend(direction, C.byref(oc), sequence.ctypes.data, master.ctypes.data,
                                     C.byref(op), C.byref(tr))
    res.rows_executed = int(rows[0])
    res.limit_warning = int(warn[0])
    return res


# ---------------------------------------------------------------- the compiled reference

class _RefCore(C.Structure):
    pass


_RefCore._fields_ = [  # common.h:80-98
    ("next", C.POINTER(_RefCore)), ("seqIdx", C.c_int), ("leftSeqPos", C.c_uint64), ("rightSeqPos", C.c_uint64),
    ("leftExtendable", C.c_char), ("rightExtendable", C.c_char), ("lowerSeqBound", C.c_uint64),
    ("upperSeqBound", C.c_uint64), ("lowerSeqBoundFlag", C.c_int), ("upperSeqBoundFlag", C.c_int),
    ("leftExtensionLen", C.c_int), ("rightExtensionLen", C.c_int), ("score", C.c_int), ("orient", C.c_char)]


class _RefSeqLib(C.Structure):  # sequence.h:36-46
    _fields_ = [("sequence", C.c_void_p), ("identifiers", C.POINTER(C.c_char_p)), ("boundaries", C.c_void_p),
                ("offsets", C.c_void_p), ("length", C.c_uint64), ("count", C.c_int),
                ("markov_chain_order", C.c_int), ("markov_chain_prob_tables", C.c_void_p)]


class _RefScoring(C.Structure):  # score_system.h:7-17
    _fields_ = [("name", C.c_char_p), ("matrix", C.POINTER(C.POINTER(C.c_int))), ("msize", C.c_int),
                ("alphabet", C.c_char_p), ("m_lambda", C.c_double), ("m_bg_freqs", C.c_double * 4),
                ("gapopen", C.c_int), ("gapextn", C.c_int)]


_rlib = None


def _ref():
    global _rlib
    if _rlib is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libramref.so not built (run `make -C oracle ref` where /root/reference exists)")
        _rlib = C.CDLL(_REF_SO)
        _rlib.extend_alignment.restype = C.c_int
        _rlib.extend_alignment.argtypes = [C.c_int, C.POINTER(_RefCore), C.c_void_p, C.POINTER(_RefSeqLib),
                                           C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(_RefScoring), C.c_void_p]
        _rlib.allocate_score.restype = C.c_void_p
        _rlib.allocate_score.argtypes = [C.c_int, C.c_int]
        _rlib.free_score.restype = None
        _rlib.free_score.argtypes = [C.c_int, C.c_int, C.c_void_p]
        _rlib.getMatrix.restype = C.POINTER(_RefScoring)
        _rlib.getMatrix.argtypes = [C.c_char_p]
        _rlib.getRepeatScoutMatrix.restype = C.POINTER(_RefScoring)
        _rlib.getRepeatScoutMatrix.argtypes = [C.c_int, C.c_int, C.c_int]
        _rlib.compute_nw_row.restype = C.c_int
    return _rlib


def ref_matrix_values(name: str, match=1, mismatch=-1, gap=-5):
    """Reads the reference's own matrix for the defined index set ([0..7,99] x [0..7,99])."""
    lib = _ref()
    sp = lib.getRepeatScoutMatrix(match, mismatch, gap) if name == "repeatscout" else lib.getMatrix(name.encode())
    s = sp.contents
    m = np.zeros(MSIZE * MSIZE, np.int32)
    idx = list(range(8)) + [99]
    for i in idx:
        for j in idx:
            m[i * MSIZE + j] = s.matrix[i][j]
    return m, s.gapopen, s.gapextn


def ref_extend(direction: int, cores: Cores, sequence: np.ndarray, master: np.ndarray, p: Params,
               boundaries: Optional[np.ndarray] = None, verbose: int = 0) -> Result:
    """Calls the reference's extend_alignment (ram_extend.c:859) on in-memory structures."""
    lib = _ref()
    n = cores.n
    sequence = np.ascontiguousarray(sequence, dtype=np.int8)
    # globals read inside the hot path (ram_extend.c:40,52,61)
    C.c_int.in_dll(lib, "VERBOSE").value = verbose
    C.c_int.in_dll(lib, "WHEN_TO_STOP").value = p.when_to_stop
    C.c_int.in_dll(lib, "l").value = p.l
    arr = (_RefCore * max(n, 1))()
    for i in range(n):
        c = arr[i]
        c.next = C.pointer(arr[i + 1]) if i + 1 < n else None
        c.seqIdx = int(cores.seq_idx[i])
        c.leftSeqPos = int(cores.left_pos[i]); c.rightSeqPos = int(cores.right_pos[i])
        c.leftExtendable = bytes([int(cores.left_ext[i])]); c.rightExtendable = bytes([int(cores.right_ext[i])])
        c.lowerSeqBound = int(cores.lower[i]); c.upperSeqBound = int(cores.upper[i])
        c.leftExtensionLen = int(cores.left_len[i]); c.rightExtensionLen = int(cores.right_len[i])
        c.score = int(cores.score[i]); c.orient = bytes([int(cores.orient[i])])
    nseq = int(cores.seq_idx.max()) + 1 if n else 1
    if boundaries is None:
        boundaries = np.zeros(nseq + 1, np.uint64)
    boundaries = np.ascontiguousarray(boundaries, dtype=np.uint64)
    offsets = np.zeros(nseq + 1, np.uint64)
    idents = (C.c_char_p * (nseq + 1))(*([b"seq%d" % i for i in range(nseq)] + [None]))
    sl = _RefSeqLib(sequence.ctypes.data, idents, boundaries.ctypes.data, offsets.ctypes.data, len(sequence), nseq, 0, None)
    # scoring system: malloc'ed by the reference, then overwritten with p.matrix on the defined index set
    sp = lib.getMatrix(b"20p43g")
    idx = list(range(8)) + [99]
    for i in idx:
        for j in idx:
            sp.contents.matrix[i][j] = int(p.matrix[i * MSIZE + j])
    sp.contents.gapopen = p.gapopen
    sp.contents.gapextn = p.gapextn
    score = lib.allocate_score(max(n, 1), p.bandwidth)
    assert master.dtype == np.int8
    ret = lib.extend_alignment(direction, arr if n else None, score, C.byref(sl), master.ctypes.data, p.bandwidth,
                               p.cappenalty, p.minimprovement, p.L, n, sp, None)
    lib.free_score(max(n, 1), p.bandwidth, score)
    for i in range(n):
        cores.left_len[i] = arr[i].leftExtensionLen
        cores.right_len[i] = arr[i].rightExtensionLen
        cores.score[i] = arr[i].score
    return Result(ret, master, cores.left_len, cores.right_len, cores.score)
