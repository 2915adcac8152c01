"""Flat host-side mirror of the reference's in-memory data model.

``CoreSet``  <-> linked list of ``struct coreAlignment`` (reference common.h:80-98)
``FlankSet`` <-> ``struct sequenceLibrary`` + the core list (reference sequence.h:36-46)
``ExtendParams`` <-> the arguments / globals of ``extend_alignment`` (reference
ram_extend.c:859-864, globals :40,:52,:61) plus the scoring system
(score_system.h:7-17).

Base codes are the reference's (sequence.h:7-15): A,C,G,T = 0..3, a,c,g,t = 4..7, N = 99.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

SYM_N = 99
MSIZE = 100  # scoring matrix is [100][100], index [consensus][sequence]


@dataclass
class CoreSet:
    left_pos: np.ndarray      # leftSeqPos   (index into FlankSet.sequence, 0-based closed)
    right_pos: np.ndarray     # rightSeqPos
    lower: np.ndarray         # lowerSeqBound
    upper: np.ndarray         # upperSeqBound
    orient: np.ndarray        # 1 = reverse strand
    left_ext: np.ndarray      # leftExtendable
    right_ext: np.ndarray     # rightExtendable
    seq_idx: Optional[np.ndarray] = None
    left_len: Optional[np.ndarray] = None    # leftExtensionLen  (out)
    right_len: Optional[np.ndarray] = None   # rightExtensionLen (out)
    score: Optional[np.ndarray] = None       # score (accumulates over both directions)
    lower_flag: Optional[np.ndarray] = None  # enum CoreBoundFlag (common.h:10)
    upper_flag: Optional[np.ndarray] = None

    def __post_init__(self):
        n = len(self.left_pos)
        for k in ("left_pos", "right_pos", "lower", "upper"):
            setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=np.int64))
        for k in ("orient", "left_ext", "right_ext"):
            setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=np.int8))
        if self.seq_idx is None:
            self.seq_idx = np.arange(n, dtype=np.int32)
        for k in ("seq_idx", "left_len", "right_len", "score", "lower_flag", "upper_flag"):
            v = getattr(self, k)
            setattr(self, k, np.zeros(n, np.int32) if v is None else np.ascontiguousarray(v, dtype=np.int32))

    @property
    def n(self) -> int:
        return len(self.left_pos)

    def copy(self) -> "CoreSet":
        return CoreSet(**{k: getattr(self, k).copy() for k in self.__dataclass_fields__})

    def subset(self, sl) -> "CoreSet":
        return CoreSet(**{k: getattr(self, k)[sl].copy() for k in self.__dataclass_fields__})


@dataclass
class FlankSet:
    sequence: np.ndarray                 # int8 codes, all windows concatenated
    boundaries: np.ndarray               # uint64 cumulative ends, 0-terminated (sequence.h:24-31)
    cores: CoreSet
    offsets: Optional[np.ndarray] = None  # genomic start of each window
    identifiers: Optional[List[str]] = None

    def __post_init__(self):
        self.sequence = np.ascontiguousarray(self.sequence, dtype=np.int8)
        self.boundaries = np.ascontiguousarray(self.boundaries, dtype=np.uint64)
        nseq = len(self.boundaries) - 1
        if self.offsets is None:
            self.offsets = np.zeros(nseq + 1, np.uint64)
        self.offsets = np.ascontiguousarray(self.offsets, dtype=np.uint64)
        if self.identifiers is None:
            self.identifiers = ["s%06d" % i for i in range(nseq)]


@dataclass
class ExtendParams:
    bandwidth: int = 14          # -bandwidth  (ram_extend.c:247)
    cappenalty: int = -90        # per-matrix default (ram_extend.c:301-344)
    minimprovement: int = 27
    L: int = 10000               # -L (ram_extend.c:246)
    when_to_stop: int = 100      # -stopafter (ram_extend.c:249)
    l: int = 1                   # ram_extend.c:40
    gapopen: int = -28
    gapextn: int = -5
    matrix: np.ndarray = field(default=None)   # int32 [100*100] row-major [cons][base]
    matrix_name: str = ""


def new_master(L: int, l: int = 1) -> np.ndarray:
    """master = malloc(2L+l+1) with the l-long N spacer at master[L] (ram_extend.c:347-353,415-416)."""
    m = np.zeros(2 * L + l + 1, np.int8)
    m[L:L + l] = SYM_N
    return m
