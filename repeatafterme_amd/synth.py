"""Seeded synthetic flank sets: the N x L workloads of BASELINE.json.

``synth_family``      -- uniform windows, fully vectorised; used for the N=1,000 x 2,000 and
                         N=100,000 x 10,000 configurations (parity at size + bench).
``synth_adversarial`` -- small ragged sets (several cores per window, truncated flanks, N runs,
                         both strands, random extendable flags, cores at array position 0,
                         tightened bounds) used to pin edge semantics.

Every copy is ``core + flank`` where the flank starts with a mutated copy of a shared ancestor
(``div`` per-base divergence: 80% substitutions, 10% deletions, 10% insertions) followed by
i.i.d. uniform bases, so the extension recovers about ``K`` columns and then stops.
"""
from __future__ import annotations

import numpy as np

from .datamodel import CoreSet, FlankSet, SYM_N


def _revcomp_rows(rows: np.ndarray) -> np.ndarray:
    r = rows[:, ::-1].copy()
    acgt = r < 4
    r[acgt] = 3 - r[acgt]
    return r


def _mutate_batch(rng: np.random.Generator, anc: np.ndarray, m: int, div: float):
    """m mutated copies of ``anc``; returns (flat values, per-copy lengths)."""
    K = len(anc)
    u = rng.random((m, K))
    vals = np.broadcast_to(anc, (m, K)).copy()
    sub = u < 0.8 * div
    dele = (u >= 0.8 * div) & (u < 0.9 * div)
    ins = (u >= 0.9 * div) & (u < div)
    shift = rng.integers(1, 4, size=(m, K), dtype=np.int8)
    vals[sub] = (vals[sub] + shift[sub]) & 3
    counts = np.ones((m, K), np.int64)
    counts[dele] = 0
    counts[ins] = 2
    flat_counts = counts.ravel()
    out = np.repeat(vals.ravel(), flat_counts)
    ends = np.cumsum(flat_counts)
    ins_pos = ends[ins.ravel()] - 1            # second element of every doubled position
    out[ins_pos] = rng.integers(0, 4, size=len(ins_pos), dtype=np.int8)
    return out, counts.sum(axis=1)


def _flank_block(rng, anc, m, F, div):
    """(m, F) int8: mutated ancestor copy then uniform random fill."""
    blk = rng.integers(0, 4, size=(m, F), dtype=np.int8)
    if len(anc) == 0 or F == 0:
        return blk
    flat, lens = _mutate_batch(rng, anc, m, div)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    row = np.repeat(np.arange(m), lens)
    col = np.arange(len(flat)) - np.repeat(starts, lens)
    keep = col < F
    blk[row[keep], col[keep]] = flat[keep]
    return blk


def synth_family(n: int, L: int, W: int, K: int = 1500, seed: int = 1, div: float = 0.14,
                 both_sides: bool = False, minus_frac: float = 0.0, n_run_frac: float = 0.0,
                 core_len: int = 10, pad: int = 20, batch: int = 8192, shard: int = 0) -> FlankSet:
    """``shard`` > 0: same ancestor and core as shard 0 (they come from ``seed`` alone), independent copies -- the
    ranks of a sharded run each generate their own part of ONE family."""
    rng = np.random.default_rng(seed)
    anc_r = rng.integers(0, 4, size=K, dtype=np.int8)
    anc_l = rng.integers(0, 4, size=K, dtype=np.int8)
    core = rng.integers(0, 4, size=core_len, dtype=np.int8)
    if shard:
        rng = np.random.default_rng([seed, shard])
    F = L + W + pad
    FL = F if both_sides else 0
    win = FL + core_len + F
    seq = np.empty((n, win), np.int8)
    minus = rng.random(n) < minus_frac
    for b0 in range(0, n, batch):
        m = min(batch, n - b0)
        rows = seq[b0:b0 + m]
        rows[:, FL:FL + core_len] = core
        rows[:, FL + core_len:] = _flank_block(rng, anc_r, m, F, div)
        if both_sides:
            rows[:, :FL] = _flank_block(rng, anc_l, m, FL, div)[:, ::-1]
        if n_run_frac > 0:
            has = np.nonzero(rng.random(m) < n_run_frac)[0]
            for i in has:
                s = int(rng.integers(FL + core_len, FL + core_len + max(K, 1)))
                e = min(win, s + int(rng.integers(3, 30)))
                rows[i, s:e] = SYM_N
        mb = minus[b0:b0 + m]
        if mb.any():
            rows[mb] = _revcomp_rows(rows[mb])
    off = np.arange(n, dtype=np.int64) * win
    # + strand: core at [FL, FL+core_len); - strand (row reverse-complemented): core at [F, F+core_len)
    lo_core = np.where(minus, off + F, off + FL)
    hi_core = lo_core + core_len - 1
    left_pos = np.where(minus, hi_core, lo_core)     # '-' strand: left = higher coordinate (sequence.c:886-897)
    right_pos = np.where(minus, lo_core, hi_core)
    cores = CoreSet(left_pos=left_pos, right_pos=right_pos, lower=off, upper=off + win - 1,
                    orient=minus.astype(np.int8),
                    left_ext=np.full(n, 1 if both_sides else 0, np.int8), right_ext=np.ones(n, np.int8))
    boundaries = np.concatenate((off[1:], [n * win], [0])).astype(np.uint64)
    return FlankSet(sequence=seq.reshape(-1), boundaries=boundaries, cores=cores)


def synth_adversarial(seed: int, n_windows: int = 8, L: int = 120, W: int = 14, K: int = 90,
                      div: float = 0.15, lowercase: bool = False) -> FlankSet:
    """Ragged, hostile set: 1-4 cores per window at 0-60 bp spacing, flanks truncated to 0..K+L,
    N runs, 40% '-' strand, random extendable flags, 1-4 bp indels, bounds sometimes tightened
    inside the window (as the overlap-avoidance step does, ram_extend.c:445-499)."""
    rng = np.random.default_rng(seed)
    anc_r = rng.integers(0, 4, size=K, dtype=np.int8)
    anc_l = rng.integers(0, 4, size=K, dtype=np.int8)
    core = rng.integers(0, 4, size=int(rng.integers(1, 12)), dtype=np.int8)

    def mutated(anc):
        out = []
        i = 0
        while i < len(anc):
            u = rng.random()
            if u < 0.8 * div:
                out.append((anc[i] + rng.integers(1, 4)) & 3)
            elif u < 0.9 * div:
                i += int(rng.integers(0, 4))              # 1-4 bp deletion
            elif u < div:
                out.extend(rng.integers(0, 4, size=int(rng.integers(1, 5))).tolist())
                out.append(anc[i])
            else:
                out.append(anc[i])
            i += 1
        return np.array(out, np.int8)

    chunks, lp, rp, lo, up, ori, le, re_, sidx = [], [], [], [], [], [], [], [], []
    pos = 0
    for w in range(n_windows):
        ncore = int(rng.integers(1, 5))
        wstart = pos
        parts = []
        core_spans = []
        for c in range(ncore):
            minus = rng.random() < 0.4
            fl = int(rng.integers(0, K + L)) if rng.random() < 0.8 else 0
            fr = int(rng.integers(0, K + L)) if rng.random() < 0.8 else 0
            left = np.concatenate((mutated(anc_l), rng.integers(0, 4, size=L + W, dtype=np.int8)))[:fl][::-1]
            right = np.concatenate((mutated(anc_r), rng.integers(0, 4, size=L + W, dtype=np.int8)))[:fr]
            piece = np.concatenate((left, core, right)).astype(np.int8)
            if rng.random() < 0.3 and len(piece) > 8:
                s = int(rng.integers(0, len(piece) - 4))
                piece[s:s + int(rng.integers(1, 12))] = SYM_N
            cs = len(left)
            if minus:
                r = piece[::-1].copy()
                m4 = r < 4
                r[m4] = 3 - r[m4]
                piece = r
                cs = len(right)
            base = sum(len(p) for p in parts)
            core_spans.append((base + cs, base + cs + len(core) - 1, minus, base, base + len(piece) - 1))
            parts.append(piece)
        window = np.concatenate(parts)
        if lowercase:
            lc = rng.random(len(window)) < 0.1
            window = np.where(lc & (window < 4), window + 4, window).astype(np.int8)
        chunks.append(window)
        for (a, b, minus, plo, phi) in core_spans:
            if minus:
                lp.append(wstart + b); rp.append(wstart + a)
            else:
                lp.append(wstart + a); rp.append(wstart + b)
            # bounds: whole window, this core's own piece, or tightened a little further
            mode = rng.random()
            if mode < 0.4:
                blo, bhi = wstart, wstart + len(window) - 1
            elif mode < 0.8:
                blo, bhi = wstart + plo, wstart + phi
            else:
                blo = wstart + int(rng.integers(plo, a + 1))
                bhi = wstart + int(rng.integers(b, phi + 1))
            lo.append(blo); up.append(bhi)
            ori.append(1 if minus else 0)
            le.append(int(rng.random() < 0.75)); re_.append(int(rng.random() < 0.75))
            sidx.append(w)
        pos += len(window)
    seq = np.concatenate(chunks)
    ends = np.cumsum([len(c) for c in chunks])
    cores = CoreSet(left_pos=lp, right_pos=rp, lower=lo, upper=up, orient=ori, left_ext=le, right_ext=re_,
                    seq_idx=np.array(sidx, np.int32))
    return FlankSet(sequence=seq, boundaries=np.concatenate((ends, [0])).astype(np.uint64), cores=cores)


def result_digest(ret: int, cons: np.ndarray, ext_len: np.ndarray, score: np.ndarray) -> str:
    """sha1 over what ONE direction of the extension loop hands back (ram_extend.c:1092-1095, 1234-1257): the return
    value, the consensus byte of every executed column, and per core the extension length and the score increment.
    bench.py and tests/test_gpu_fullsize.py compare it with the digests of the compiled reference in
    tests/golden/fullsize_digests.json (written by tests/golden/make_fullsize_digest.py)."""
    import hashlib
    h = hashlib.sha1()
    h.update(np.int32(ret).tobytes())
    h.update(np.ascontiguousarray(cons, dtype=np.int8).tobytes())
    h.update(np.ascontiguousarray(ext_len, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(score, dtype=np.int32).tobytes())
    return h.hexdigest()


def writeback_from_trim(trim_high: np.ndarray, trim_pos: np.ndarray):
    """(extension length, score increment) per flank from the trimmed records, the rule of ram_extend.c:1234-1247."""
    ok = (trim_high > 0) & (trim_pos >= 0)
    return np.where(ok, trim_pos + 1, 0).astype(np.int32), np.where(ok, trim_high, 0).astype(np.int32)
