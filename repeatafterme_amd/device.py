"""Thin Python handle on seam 2 of include/ramx.h (ramx_dev_*): init / upload / run_direction /
download / destroy.  Used by bench.py and the parity tests that want to look below
``extend_alignment`` (state peeks, repeated runs on resident data, the multi-GPU communicator)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .datamodel import CoreSet, ExtendParams
from .extend import RunInfo, _info, _params


def resolve_flanks(direction: int, cores: CoreSet, bandwidth: int, L: int):
    """-> (flank descriptor array, core index of every flank); C-ABI ramx_resolve_flanks."""
    L_ = _lib.lib()
    fc = _lib.FlatCores(cores.n, *[getattr(cores, k).ctypes.data for k in
                                   ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext",
                                    "right_ext", "left_len", "right_len", "score")])
    flanks = (_lib.Flank * max(cores.n, 1))()
    idx = np.zeros(max(cores.n, 1), np.int32)
    nx = L_.ramx_resolve_flanks(int(direction), C.byref(fc), bandwidth, L, flanks, idx.ctypes.data)
    return (flanks, nx), idx[:nx].copy()


class Device:
    def __init__(self, ordinal: int = 0):
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.ramx_dev_create(ordinal, C.byref(h)), "ramx_dev_create")
        self._h = h
        self._keep = None
        self.nx = 0
        self.p = None

    def close(self):
        if self._h:
            self._L.ramx_dev_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_library(self, sequence: np.ndarray):
        assert sequence.dtype == np.int8 and sequence.flags.c_contiguous
        _lib.check(self._L.ramx_dev_load_library(self._h, sequence.ctypes.data, len(sequence)), "ramx_dev_load_library")

    def begin_direction(self, flanks, p: ExtendParams):
        arr, nx = flanks
        cp, keep = _params(p)
        self._keep = keep
        _lib.check(self._L.ramx_dev_begin_direction(self._h, arr, nx, C.byref(cp)), "ramx_dev_begin_direction")
        self.nx, self.p = nx, p

    def run_direction(self) -> RunInfo:
        ci = _lib.RunInfo()
        _lib.check(self._L.ramx_dev_run_direction(self._h, C.byref(ci)), "ramx_dev_run_direction")
        self.last = _info(ci)
        return self.last

    def download(self):
        rows = self.last.rows_executed
        cons = np.zeros(max(rows, 1), np.int8)
        th = np.zeros(max(self.nx, 1), np.int32)
        tp = np.zeros(max(self.nx, 1), np.int32)
        _lib.check(self._L.ramx_dev_download(self._h, cons.ctypes.data, len(cons), th.ctypes.data, tp.ctypes.data),
                   "ramx_dev_download")
        return cons[:rows], th[:self.nx], tp[:self.nx]

    def peek_state(self, flank: int):
        B = 2 * self.p.bandwidth + 1
        cells = np.zeros(4 * (self.p.bandwidth + 1), np.int32)
        hi, po = C.c_int32(), C.c_int32()
        _lib.check(self._L.ramx_dev_peek_state(self._h, flank, cells.ctypes.data, C.byref(hi), C.byref(po)),
                   "ramx_dev_peek_state")
        return cells[:2 * B].reshape(B, 2).copy(), hi.value, po.value

    def unique_id(self) -> np.ndarray:
        uid = np.zeros(128, np.uint8)
        _lib.check(self._L.ramx_comm_unique_id(uid.ctypes.data), "ramx_comm_unique_id")
        return uid

    def comm_init(self, uid: np.ndarray, rank: int, nranks: int):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        _lib.check(self._L.ramx_dev_comm_init(self._h, uid.ctypes.data, rank, nranks), "ramx_dev_comm_init")

    def comm_size(self) -> int:
        return _lib.check(self._L.ramx_dev_comm_size(self._h), "ramx_dev_comm_size")

    def set_allreduce_callback(self, fn):
        """Test hook (ramx_dev_set_allreduce_cb): fn(list of 4 ints) -> list of 4 ints summed over ranks."""
        def _cb(ptr, _user):
            vals = fn([ptr[i] for i in range(4)])
            for i in range(4):
                ptr[i] = int(vals[i])
        self._cb = _lib.ALLREDUCE_CB(_cb)      # keep the trampoline alive
        _lib.check(self._L.ramx_dev_set_allreduce_cb(self._h, self._cb, None), "ramx_dev_set_allreduce_cb")

    def peer_setup(self, rank: int, world: int, all_gather_bytes, all_reduce_min, barrier) -> bool:
        """In-kernel vote exchange for the cross-device persistent path.  Two tiers, each enabled only if it works on
        every rank (self-test), tried in this order:
          "device": every rank's mailbox in fine-grained device memory, mapped into the others over hipIpc (xGMI);
          "host":   the mailboxes in one POSIX shared-memory segment registered with HIP (PCIe; ranks of one node).
        RAMX_PEER_KIND=device|host restricts the choice (tests).  all_gather_bytes(bytes64) -> list of world bytes
        objects; all_reduce_min(int) -> int; barrier() -> None.  Returns whether a tier is enabled; `peer_kind` names
        it ("device", "host" or None)."""
        import os
        only = os.environ.get("RAMX_PEER_KIND", "")
        token = 0x52414D58000000 + world
        self.peer_kind = None

        def selftest(ok):
            if all_reduce_min(ok) == 1:
                if self._L.ramx_dev_peer_selftest(self._h, 0, token) < 0:
                    ok = 0
                barrier()
                if ok and self._L.ramx_dev_peer_selftest(self._h, 1, token) != 1:
                    ok = 0
            return all_reduce_min(ok)

        if only in ("", "device"):
            ok = 1
            h = np.zeros(64, np.uint8)
            if self._L.ramx_dev_peer_export(self._h, h.ctypes.data) < 0:
                ok = 0
            handles = all_gather_bytes(h.tobytes())
            if all_reduce_min(ok) == 1:
                allh = np.frombuffer(b"".join(handles), np.uint8).copy()
                if self._L.ramx_dev_peer_import(self._h, allh.ctypes.data, rank, world) < 0:
                    ok = 0
            if selftest(ok) == 1:
                self._L.ramx_dev_peer_enable(self._h, 1)
                self.peer_kind = "device"
                return True
        if only in ("", "host"):
            # one name for the job: rank 0's pid + a random nonce, spread through the same all-gather; rank 0 creates the
            # segment exclusively, the others open it after a barrier; the name is unlinked in any case
            mine = np.zeros(64, np.uint8)
            tag = (b"/ramx_box_%d_%s" % (os.getpid(), os.urandom(8).hex().encode()))[:63]
            mine[:len(tag)] = np.frombuffer(tag, np.uint8)
            name = bytes(all_gather_bytes(mine.tobytes())[0]).split(b"\0", 1)[0]
            ok = 1
            try:
                if rank == 0:
                    ok = 1 if self._L.ramx_dev_hostbox_attach(self._h, name, rank, world) >= 0 else 0
                barrier()                   # the segment exists (or rank 0 failed: the others then fail to open it)
                if rank != 0:
                    ok = 1 if self._L.ramx_dev_hostbox_attach(self._h, name, rank, world) >= 0 else 0
                barrier()                   # every box has been cleared by its owner
                ok = selftest(ok)
            finally:
                if rank == 0:
                    self._L.ramx_hostbox_unlink(name)
            if ok == 1:
                self._L.ramx_dev_peer_enable(self._h, 1)
                self.peer_kind = "host"
                return True
        self._L.ramx_dev_peer_enable(self._h, 0)
        return False
