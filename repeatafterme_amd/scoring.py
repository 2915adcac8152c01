"""Scoring systems through the C-ABI (reference score_system.h:23-37): same names, same numbers."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .datamodel import ExtendParams, MSIZE


class _Scoring(C.Structure):   # score_system.h:7-17
    _fields_ = [("name", C.c_char_p), ("matrix", C.POINTER(C.POINTER(C.c_int))), ("msize", C.c_int),
                ("alphabet", C.c_char_p), ("m_lambda", C.c_double), ("m_bg_freqs", C.c_double * 4),
                ("gapopen", C.c_int), ("gapextn", C.c_int)]


def _flatten(sp):
    s = sp.contents
    m = np.zeros(MSIZE * MSIZE, np.int32)
    idx = list(range(8)) + [99]
    for i in idx:
        for j in idx:
            m[i * MSIZE + j] = s.matrix[i][j]
    return m, s.gapopen, s.gapextn, s.m_lambda


def get_matrix(name: str):
    """getMatrix (score_system.c:182): -> (int32[100*100], gapopen, gapextn)."""
    L = _lib.lib()
    L.ramx_get_matrix.restype = C.POINTER(_Scoring)
    L.ramx_get_matrix.argtypes = [C.c_char_p]
    L.ramx_free_scoring_system.argtypes = [C.POINTER(_Scoring)]
    if name not in ("14p43g", "18p43g", "20p43g", "25p43g"):
        raise ValueError(f"{name} is not an internally coded matrix")   # the C entry exits(1) like the reference
    sp = L.ramx_get_matrix(name.encode())
    m, go, ge, _ = _flatten(sp)
    L.ramx_free_scoring_system(sp)
    return m, go, ge


def get_repeatscout_matrix(match: int = 1, mismatch: int = -1, gap: int = -5):
    L = _lib.lib()
    L.ramx_get_repeatscout_matrix.restype = C.POINTER(_Scoring)
    L.ramx_get_repeatscout_matrix.argtypes = [C.c_int, C.c_int, C.c_int]
    L.ramx_free_scoring_system.argtypes = [C.POINTER(_Scoring)]
    sp = L.ramx_get_repeatscout_matrix(match, mismatch, gap)
    m, go, ge, _ = _flatten(sp)
    L.ramx_free_scoring_system(sp)
    return m, go, ge


def matrix_lambda(name: str) -> float:
    L = _lib.lib()
    L.ramx_get_matrix.restype = C.POINTER(_Scoring)
    L.ramx_get_matrix.argtypes = [C.c_char_p]
    sp = L.ramx_get_matrix(name.encode())
    lam = sp.contents.m_lambda
    L.ramx_free_scoring_system(sp)
    return lam


def named_params(matrix: str, **kw) -> ExtendParams:
    """CLI defaults per matrix (reference ram_extend.c:280-344)."""
    if matrix == "repeatscout":
        m, go, ge = get_repeatscout_matrix(kw.pop("match", 1), kw.pop("mismatch", -1), kw.pop("gap", -5))
        d = dict(minimprovement=3, cappenalty=-20)
    else:
        m, go, ge = get_matrix(matrix)
        d = dict(minimprovement=24 if matrix == "25p43g" else 27, cappenalty=-90)
    d.update(gapopen=go, gapextn=ge, matrix=m, matrix_name=matrix)
    d.update(kw)
    return ExtendParams(**d)
