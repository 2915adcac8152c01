"""BED-6 ranges + .2bit -> FlankSet through the C-ABI loader (reference sequence.c:505-923),
plus small writers for the two input formats (used to put synthetic sets on disk for the CLI)."""
from __future__ import annotations

import ctypes as C
import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib
from .datamodel import CoreSet, FlankSet


class _Core(C.Structure):
    pass


_Core._fields_ = [  # common.h:80-98
    ("next", C.POINTER(_Core)), ("seqIdx", C.c_int), ("leftSeqPos", C.c_uint64), ("rightSeqPos", C.c_uint64),
    ("leftExtendable", C.c_char), ("rightExtendable", C.c_char), ("lowerSeqBound", C.c_uint64),
    ("upperSeqBound", C.c_uint64), ("lowerSeqBoundFlag", C.c_int), ("upperSeqBoundFlag", C.c_int),
    ("leftExtensionLen", C.c_int), ("rightExtensionLen", C.c_int), ("score", C.c_int), ("orient", C.c_char)]


class _SeqLib(C.Structure):  # sequence.h:36-46
    _fields_ = [("sequence", C.POINTER(C.c_int8)), ("identifiers", C.POINTER(C.c_char_p)),
                ("boundaries", C.POINTER(C.c_uint64)), ("offsets", C.POINTER(C.c_uint64)), ("length", C.c_uint64),
                ("count", C.c_int), ("markov_chain_order", C.c_int), ("markov_chain_prob_tables", C.c_void_p)]


def cores_from_list(head, n: int, reader=lambda c: c) -> CoreSet:
    f = {k: [] for k in ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext", "right_ext", "seq_idx",
                         "left_len", "right_len", "score", "lower_flag", "upper_flag")}
    p = head
    for _ in range(n):
        c = p.contents
        f["left_pos"].append(c.leftSeqPos); f["right_pos"].append(c.rightSeqPos)
        f["lower"].append(c.lowerSeqBound); f["upper"].append(c.upperSeqBound)
        f["orient"].append(c.orient[0]); f["left_ext"].append(c.leftExtendable[0]); f["right_ext"].append(c.rightExtendable[0])
        f["seq_idx"].append(c.seqIdx); f["left_len"].append(c.leftExtensionLen); f["right_len"].append(c.rightExtensionLen)
        f["score"].append(c.score); f["lower_flag"].append(c.lowerSeqBoundFlag); f["upper_flag"].append(c.upperSeqBoundFlag)
        p = c.next
    return CoreSet(**f)


def flankset_from_c(lib_p, head, n: int) -> FlankSet:
    sl = lib_p.contents
    seq = np.ctypeslib.as_array(sl.sequence, shape=(int(sl.length),)).copy() if sl.length else np.zeros(0, np.int8)
    bounds = np.array([sl.boundaries[i] for i in range(sl.count + 1)], np.uint64)
    offs = np.array([sl.offsets[i] for i in range(sl.count + 1)], np.uint64)
    ids = [sl.identifiers[i].decode() for i in range(sl.count)]
    return FlankSet(sequence=seq, boundaries=bounds, cores=cores_from_list(head, n), offsets=offs, identifiers=ids)


def load_sequence_subset_minimal(twobit: str, ranges: str, max_flanking_bp: int) -> FlankSet:
    """loadSequenceSubsetMinimal (sequence.c:505): BED-6 + 2bit -> library + cores."""
    L = _lib.lib()
    L.ramx_load_sequence_subset_minimal.restype = C.POINTER(_SeqLib)
    L.ramx_load_sequence_subset_minimal.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(_Core)),
                                                    C.POINTER(C.c_int), C.c_int]
    L.ramx_free_library.argtypes = [C.POINTER(_SeqLib), C.POINTER(_Core)]
    head = C.POINTER(_Core)()
    n = C.c_int()
    lp = L.ramx_load_sequence_subset_minimal(twobit.encode(), ranges.encode(), C.byref(head), C.byref(n), max_flanking_bp)
    fs = flankset_from_c(lp, head, n.value)
    L.ramx_free_library(lp, head)
    return fs


class _Packed(C.Structure):  # include/ramx.h ramx_packed_library
    _fields_ = [("length", C.c_uint64), ("n_windows", C.c_int32), ("win_start", C.POINTER(C.c_uint64)),
                ("win_byte", C.POINTER(C.c_uint64)), ("win_phase", C.POINTER(C.c_uint8)), ("bytes", C.POINTER(C.c_uint8)),
                ("n_bytes", C.c_uint64), ("n_start", C.POINTER(C.c_uint64)), ("n_len", C.POINTER(C.c_uint32)),
                ("n_blocks", C.c_int32)]


def load_sequence_subset_packed(twobit: str, ranges: str, max_flanking_bp: int):
    """ramx_load_sequence_subset_packed: the same windows kept as the .2bit file stores them.  Returns
    (FlankSet whose sequence was decoded through ramx_packed_decode, dict with the packed tables) -- test mirror."""
    L = _lib.lib()
    L.ramx_load_sequence_subset_packed.restype = C.POINTER(_SeqLib)
    L.ramx_load_sequence_subset_packed.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(_Core)), C.POINTER(C.c_int),
                                                   C.c_int, C.POINTER(C.POINTER(_Packed))]
    L.ramx_packed_decode.argtypes = [C.POINTER(_Packed), C.c_uint64, C.c_uint64, C.c_void_p]
    L.ramx_free_library.argtypes = [C.POINTER(_SeqLib), C.POINTER(_Core)]
    head = C.POINTER(_Core)()
    n = C.c_int()
    pk = C.POINTER(_Packed)()
    lp = L.ramx_load_sequence_subset_packed(twobit.encode(), ranges.encode(), C.byref(head), C.byref(n), max_flanking_bp, C.byref(pk))
    sl = lp.contents
    assert not sl.sequence, "a packed library has no one-byte-per-base sequence"
    p = pk.contents
    nw, nb = p.n_windows, p.n_blocks
    tables = dict(length=int(p.length), n_windows=nw, n_bytes=int(p.n_bytes), n_blocks=nb,
                  win_start=np.array([p.win_start[i] for i in range(nw + 1)], np.uint64),
                  win_byte=np.array([p.win_byte[i] for i in range(nw + 1)], np.uint64),
                  win_phase=np.array([p.win_phase[i] for i in range(nw)], np.uint8),
                  n_start=np.array([p.n_start[i] for i in range(nb)], np.uint64),
                  n_len=np.array([p.n_len[i] for i in range(nb)], np.uint32))
    seq = np.zeros(int(sl.length), np.int8)
    if sl.length:
        _lib.check(L.ramx_packed_decode(pk, 0, sl.length, seq.ctypes.data), "ramx_packed_decode")

    def decode(frm: int, count: int) -> np.ndarray:
        out = np.zeros(count, np.int8)
        _lib.check(L.ramx_packed_decode(pk, frm, count, out.ctypes.data), "ramx_packed_decode")
        return out
    tables["pieces"] = [(f, c, decode(f, c)) for f, c in _sample_ranges(int(sl.length))]
    bounds = np.array([sl.boundaries[i] for i in range(sl.count + 1)], np.uint64)
    offs = np.array([sl.offsets[i] for i in range(sl.count + 1)], np.uint64)
    ids = [sl.identifiers[i].decode() for i in range(sl.count)]
    fs = FlankSet(sequence=seq, boundaries=bounds, cores=cores_from_list(head, n.value), offsets=offs, identifiers=ids)
    L.ramx_free_library(lp, head)
    return fs, tables


def _sample_ranges(length: int):
    """(from, count) pieces for spot checks of ramx_packed_decode: single bases, window-straddling stretches, the ends."""
    rng = np.random.default_rng(length)
    out = [(0, min(length, 1)), (max(length - 1, 0), min(length, 1))]
    for _ in range(40):
        if length == 0:
            break
        f = int(rng.integers(0, length))
        out.append((f, int(min(length - f, rng.integers(1, 700)))))
    return out


# ----------------------------------------------------------------------------- writers (formats: SURVEY.md App. A)

_CODE_TO_2BIT = np.array([2, 1, 3, 0], np.uint8)   # A,C,G,T -> 2bit T=0 C=1 A=2 G=3


def write_twobit(path: str, records: Sequence[Tuple[str, np.ndarray]]) -> None:
    """Version-0 little-endian .2bit; codes 0..3 = ACGT, anything else becomes an N block."""
    names = [n.encode() for n, _ in records]
    header = struct.pack("<IIII", 0x1A412743, 0, len(records), 0)
    index_size = sum(1 + len(n) + 4 for n in names)
    bodies = []
    for _, codes in records:
        codes = np.asarray(codes, np.int8)
        isn = (codes < 0) | (codes > 3)
        d = np.diff(np.concatenate(([0], isn.view(np.int8), [0])))
        starts = np.nonzero(d == 1)[0].astype(np.uint32)
        sizes = (np.nonzero(d == -1)[0] - starts).astype(np.uint32)
        vals = _CODE_TO_2BIT[np.where(isn, 3, codes).astype(np.int64)]   # N positions are packed as T=0
        pad = (-len(vals)) % 4
        v = np.concatenate((vals, np.zeros(pad, np.uint8))).reshape(-1, 4)
        packed = ((v[:, 0] << 6) | (v[:, 1] << 4) | (v[:, 2] << 2) | v[:, 3]).astype(np.uint8)
        body = struct.pack("<II", len(codes), len(starts)) + starts.tobytes() + sizes.tobytes() + \
            struct.pack("<II", 0, 0) + packed.tobytes()
        bodies.append(body)
    with open(path, "wb") as f:
        f.write(header)
        off = len(header) + index_size
        for n, b in zip(names, bodies):
            f.write(struct.pack("<B", len(n)) + n + struct.pack("<I", off))
            off += len(b)
        for b in bodies:
            f.write(b)


def write_ranges(path: str, rows: Sequence[Tuple[str, int, int, int, int, str]]) -> None:
    with open(path, "w") as f:
        for name, start, end, le, re_, strand in rows:
            f.write(f"{name}\t{start}\t{end}\t{le}\t{re_}\t{strand}\n")
