"""CPU stand-in shard engine for the gloo tests: the reference's column loop (ram_extend.c:970-1223)
restated around the oracle's row function, with the four candidate sums passed through an all-reduce."""
import numpy as np

from oracle import pyoracle as po


def make_engine(allreduce4):
    def run(direction, shard, sequence, p):
        n = shard.n
        W, L = p.bandwidth, p.L
        B = 2 * W + 1
        idx = np.array([i for i in range(n) if (shard.right_ext[i] if direction else shard.left_ext[i])], np.int64)
        score = np.zeros((2, max(n, 1), B, 2), np.int32)
        for o in range(-W, W + 1):
            score[1, :, o + W, :] = 0 if o == 0 else abs(o) * p.gapextn + p.gapopen
        high = np.zeros(n, np.int64); pos = np.zeros(n, np.int64)
        th = np.zeros(n, np.int64); tp = np.zeros(n, np.int64)
        max_ext, max_row, rows = 0, -1, 0
        cons = []

        def row(r, i, a):
            return po.oracle_nw_row(direction, r, int(i), max(n, 1), a, int(shard.left_pos[i]), int(shard.right_pos[i]),
                                    int(shard.orient[i]), score, int(shard.lower[i]), int(shard.upper[i]), sequence,
                                    p.matrix, p.gapopen, p.gapextn, W)
        for r in range(L):
            sums = np.zeros(4, np.int64)
            for a in range(4):
                for i in idx:
                    b, _ = row(r, i, a)
                    b = max(b, 0)
                    sums[a] += b if b >= high[i] + p.cappenalty else high[i] + p.cappenalty
            sums = allreduce4(sums)                       # <- the only exchange of the column
            curr, besta = 0, 0
            for a in range(4):
                if sums[a] > curr:
                    curr, besta = int(sums[a]), a
            cons.append(besta)
            for i in idx:
                b, bi = row(r, i, besta)
                if b > high[i]:
                    high[i], pos[i] = b, bi
            rows = r + 1
            if curr >= max_ext + abs(max_row - r) * p.minimprovement:
                max_row, max_ext = r, curr
                th[:], tp[:] = high, pos
            if abs(r - max_row) >= p.when_to_stop:
                break
        return max_row + 1, rows, np.array(cons, np.int8), idx, th[idx], tp[idx]
    return run
