"""Multi-GPU host logic: flanks sharded over ranks, one 4 x int64 all-reduce per column.

One process per GPU.  Every rank holds the (small) core list and works on a contiguous block of it;
the consensus, the stop-rule state and the return value are replicated because every rank sees the
same all-reduced column sums; per-core results are exchanged once per direction with an all-gather so
that the caller (e.g. the overlap-avoidance step between the two directions) sees all of them.

The engine that runs one shard is pluggable so the sharding / merging logic can be exercised on CPU
with gloo (tests/test_sharded_gloo.py); the product engine is the HIP path with RCCL inside libramx.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

from .datamodel import CoreSet, ExtendParams


def partition(n: int, world: int, rank: int) -> slice:
    """Contiguous, balanced blocks: the first n % world ranks get one extra core."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return slice(lo, lo + q + (1 if rank < r else 0))


def gpu_engine(device):
    """Shard engine on the HIP path; `device` must already hold the library and the communicator."""
    from .device import resolve_flanks

    def run(direction: int, shard: CoreSet, sequence: np.ndarray, p: ExtendParams):
        flanks, idx = resolve_flanks(direction, shard, p.bandwidth, p.L)
        device.begin_direction(flanks, p)
        info = device.run_direction()
        cons, th, tp = device.download()
        return info.ret, info.rows_executed, cons, idx, th, tp
    return run


def extend_alignment_sharded(direction: int, cores: CoreSet, sequence: np.ndarray, master: np.ndarray,
                             p: ExtendParams, rank: int, world: int, engine: Callable,
                             all_gather: Callable[[np.ndarray], np.ndarray]) -> Tuple[int, int]:
    """Same contract as extend_alignment, with the work split over `world` ranks.
    all_gather(x: int32[k]) must return the concatenation over ranks, in rank order."""
    sl = partition(cores.n, world, rank)
    shard = cores.subset(sl)
    ret, rows, cons, idx, th, tp = engine(direction, shard, sequence, p)
    # write-back on the shard (ram_extend.c:1234-1247), then exchange
    ok = (th > 0) & (tp >= 0)
    target = shard.right_len if direction else shard.left_len
    target[idx[ok]] = tp[ok] + 1
    shard.score[idx[ok]] += th[ok]
    cores.left_len[:] = all_gather(shard.left_len)
    cores.right_len[:] = all_gather(shard.right_len)
    cores.score[:] = all_gather(shard.score)
    for r in range(rows):                                     # ram_extend.c:1092-1095
        if direction:
            master[p.L + p.l + r] = cons[r]
        else:
            master[p.L - r - 1] = cons[r]
    return ret, rows
