"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle and the committed
golden vectors.  Bit-exact: integer scores, identical consensus strings and extension lengths."""
import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd.synth import synth_adversarial, synth_family

from helpers import (assert_same_result, gpu_extend, load_api_vectors, oracle_extend, run_both_directions,
                     to_extend_params)

pytestmark = pytest.mark.gpu


def test_gpu_reproduces_golden_vectors():
    """tests/golden/api_vectors.npz: results of the compiled reference on 50 in-memory sets."""
    for k, mat, seq, cores, p, exp in load_api_vectors():
        c, m, rr, rl = run_both_directions(gpu_extend, cores, seq, p)
        assert [rr.ret, rl.ret] == exp["ret"].tolist(), f"case {k} ({mat})"
        assert np.array_equal(m, exp["master"]), f"case {k}"
        assert np.array_equal(c.left_len, exp["left_len"]) and np.array_equal(c.right_len, exp["right_len"]), f"case {k}"
        assert np.array_equal(c.score, exp["score"]), f"case {k}"


@pytest.mark.parametrize("W", [0, 1, 3, 14, 40, 80])
def test_gpu_adversarial_sets(W):
    """Ragged hostile sets: truncated flanks, N runs, both strands, lower-case codes, tightened bounds,
    mixed extendable flags, cores at array position 0 -- every matrix family."""
    for seed in range(200, 216):
        fs = synth_adversarial(seed, lowercase=(seed % 4 == 0))
        for mat in ("14p43g", "18p43g", "20p43g", "25p43g", "repeatscout"):
            p = po.Params.named(mat, bandwidth=W, L=120 if seed % 2 else 50, when_to_stop=30 if seed % 3 else 100)
            a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
            b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
            assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"seed={seed} W={W} {mat}")
            assert a[2].rows_executed == b[2].rows_executed and a[3].rows_executed == b[3].rows_executed
            assert a[2].limit_warning == b[2].limit_warning and a[3].limit_warning == b[3].limit_warning


def test_gpu_stop_rule_edges():
    """WHEN_TO_STOP / L / MINIMPROVEMENT / CAPPENALTY corners: stopafter 0 and 1, L = 1, loop running out
    exactly at L-1 (limit warning), negative minimprovement, zero cap."""
    fs = synth_family(96, 80, 8, K=40, seed=11, both_sides=True, minus_frac=0.3)
    for kw in (dict(when_to_stop=0), dict(when_to_stop=1), dict(L=1), dict(L=2, when_to_stop=1), dict(when_to_stop=40),
               dict(minimprovement=-5), dict(minimprovement=0, when_to_stop=7), dict(cappenalty=0), dict(cappenalty=-100000),
               dict(L=41, when_to_stop=1), dict(L=60, when_to_stop=20)):
        args = dict(bandwidth=8, L=80, when_to_stop=100)
        args.update(kw)
        p = po.Params.named("20p43g", **args)
        a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], str(kw))
        assert (a[2].rows_executed, a[2].limit_warning, a[3].rows_executed, a[3].limit_warning) == \
               (b[2].rows_executed, b[2].limit_warning, b[3].rows_executed, b[3].limit_warning), str(kw)


def test_gpu_empty_and_degenerate_inputs():
    from repeatafterme_amd.datamodel import CoreSet
    # no cores at all; cores present but none extendable; a single flank; flank of length zero
    seq = np.array([0, 1, 2, 3] * 20, np.int8)
    p = po.Params.named("14p43g", bandwidth=5, L=30, when_to_stop=10)
    sets = [
        CoreSet(left_pos=[], right_pos=[], lower=[], upper=[], orient=[], left_ext=[], right_ext=[]),
        CoreSet(left_pos=[10, 30], right_pos=[12, 33], lower=[0, 20], upper=[19, 79], orient=[0, 0], left_ext=[0, 0], right_ext=[0, 0]),
        CoreSet(left_pos=[10], right_pos=[12], lower=[0], upper=[79], orient=[0], left_ext=[1], right_ext=[1]),
        CoreSet(left_pos=[0, 79], right_pos=[79, 0], lower=[0, 0], upper=[79, 79], orient=[0, 1], left_ext=[1, 1], right_ext=[1, 1]),
    ]
    for i, c in enumerate(sets):
        a = run_both_directions(oracle_extend, c, seq, p)
        b = run_both_directions(gpu_extend, c, seq, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"degenerate {i}")
        assert a[2].rows_executed == b[2].rows_executed


def test_gpu_wide_band_generic_path():
    """bandwidth 500 (util/davidExtendConsRAM.pl:248 uses it): no on-chip band limit in this design."""
    fs = synth_family(70, 150, 500, K=100, seed=4, both_sides=False, minus_frac=0.3)
    p = po.Params.named("25p43g", bandwidth=500, L=150, when_to_stop=30)
    a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], "W=500")


def test_gpu_config2_n1000_l2000_bit_exact():
    """BASELINE.json configs[1]: synthetic N=1,000 x L=2,000, bandwidth 40, 14p43g -- full run vs oracle."""
    fs = synth_family(1000, 2000, 40, K=1500, seed=1)
    p = po.Params.named("14p43g", bandwidth=40, L=2000)
    a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], "cfg2")
    assert a[2].ret == 1500 and b[2].rows_executed == 1600
    assert b[2].overflow32 == 0


def test_gpu_both_sides_mixed_strands_20k():
    fs = synth_family(20000, 500, 40, K=260, seed=2, both_sides=True, minus_frac=0.3, n_run_frac=0.05)
    p = po.Params.named("14p43g", bandwidth=40, L=500)
    a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], "20k")


def test_gpu_band_state_bit_exact_per_cell():
    """Seam 2: after r columns the DP row in HBM equals the oracle's row, cell by cell, both states
    (what bnw_extend.c:1617-1648 asserts for the reference), plus high/pos."""
    from repeatafterme_amd.device import Device, resolve_flanks
    fs = synth_family(130, 60, 9, K=40, seed=21, both_sides=True, minus_frac=0.4, n_run_frac=0.2)
    W, L = 9, 37
    p = po.Params.named("18p43g", bandwidth=W, L=L, when_to_stop=1000)
    for direction in (1, 0):
        dev = Device(0)
        dev.load_library(fs.sequence)
        flanks, idx = resolve_flanks(direction, fs.cores, W, L)
        dev.begin_direction(flanks, to_extend_params(p))
        info = dev.run_direction()
        cons, th, tp = dev.download()
        assert info.rows_executed == L
        # drive the oracle row function with the device's consensus and compare the final rows
        B = 2 * W + 1
        n_align = fs.cores.n
        score = np.zeros((2, n_align, B, 2), np.int32)
        for o in range(-W, W + 1):
            score[1, :, o + W, :] = 0 if o == 0 else abs(o) * p.gapextn + p.gapopen
        c = fs.cores
        high = np.zeros(n_align, np.int64); pos = np.zeros(n_align, np.int64)
        for row in range(L):
            for n in idx:
                best, bi = po.oracle_nw_row(direction, row, int(n), n_align, int(cons[row]), int(c.left_pos[n]), int(c.right_pos[n]),
                                            int(c.orient[n]), score, int(c.lower[n]), int(c.upper[n]), fs.sequence,
                                            p.matrix, p.gapopen, p.gapextn, W)
                if best > high[n]:
                    high[n], pos[n] = best, bi
        for i in (0, 1, len(idx) // 2, len(idx) - 1):
            cells, hi, po_ = dev.peek_state(i)
            n = idx[i]
            sub, gap = score[(L - 1) % 2, n, :, 0].astype(np.int64), score[(L - 1) % 2, n, :, 1].astype(np.int64)
            want = np.stack([np.maximum(sub, gap), np.maximum(sub + p.gapopen, gap) + p.gapextn], axis=1)   # stored as (m, e)
            assert np.array_equal(cells, want), f"dir {direction} flank {i}"
            assert (hi, po_) == (high[n], pos[n])
        dev.close()


def test_gpu_full_recurrence_path_and_positive_penalties(monkeypatch):
    """The chain-free candidate evaluation is used only when gapopen <= 0 and gapextn <= 0; positive penalties
    (possible through -gapopen/-gapext) and RAMX_FORCE_CHAIN=1 run the full recurrence.  Both must match."""
    from repeatafterme_amd.device import Device, resolve_flanks
    fs = synth_adversarial(77)
    for go, ge in ((4, -3), (-10, 2), (3, 1), (0, 0), (-28, -5)):
        p = po.Params.named("20p43g", bandwidth=9, L=60, when_to_stop=20, gapopen=go, gapextn=ge)
        a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"go={go} ge={ge}")
    # forced full recurrence with ordinary penalties, through the device API (the flag is read at device creation)
    monkeypatch.setenv("RAMX_FORCE_CHAIN", "1")
    p = po.Params.named("14p43g", bandwidth=14, L=100, when_to_stop=30)
    for seed in (5, 6):
        fs = synth_adversarial(seed)
        c = fs.cores.copy()
        from repeatafterme_amd.datamodel import new_master
        m = new_master(p.L)
        o = po.oracle_extend(1, c, fs.sequence, m, p)
        dev = Device(0)
        dev.load_library(fs.sequence)
        flanks, idx = resolve_flanks(1, fs.cores, p.bandwidth, p.L)
        dev.begin_direction(flanks, to_extend_params(p))
        info = dev.run_direction()
        cons, th, tp = dev.download()
        dev.close()
        assert info.ret == o.ret and info.rows_executed == o.rows_executed
        assert np.array_equal(cons, m[p.L + 1: p.L + 1 + info.rows_executed])
        ok = (th > 0) & (tp >= 0)
        assert np.array_equal((tp[ok] + 1), c.right_len[idx[ok]]) and np.array_equal(th[ok], c.score[idx[ok]])


def test_seam1_library_cache_sees_in_place_rewrite_and_recycled_buffers():
    """Seam 1 keeps the library on the device between calls.  A buffer rewritten in place, or a new library that lands at
    the same address with the same length, must be uploaded again (the reference reads the caller's buffer on every
    call): results after the rewrite must equal the oracle's on the NEW content."""
    from repeatafterme_amd.synth import synth_family
    p = po.Params.named("14p43g", bandwidth=14, L=120, when_to_stop=30)
    a = synth_family(150, 120, 14, K=80, seed=91, both_sides=True, minus_frac=0.3)
    b = synth_family(150, 120, 14, K=80, seed=92, both_sides=True, minus_frac=0.3)
    assert len(a.sequence) == len(b.sequence)
    buf = a.sequence.copy()
    x = run_both_directions(oracle_extend, a.cores, a.sequence, p)
    y = run_both_directions(gpu_extend, a.cores, buf, p)
    assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], "first library")
    buf[:] = b.sequence                                   # same pointer, same length, new content
    x = run_both_directions(oracle_extend, b.cores, b.sequence, p)
    y = run_both_directions(gpu_extend, b.cores, buf, p)
    assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], "rewritten in place")
    assert not np.array_equal(x[1], run_both_directions(oracle_extend, a.cores, a.sequence, p)[1])
    # the same above 64 MiB, where the fingerprint is taken in 16 MiB chunks by worker threads (csrc/ramx_extend.c
    # fingerprint_large): the family sits at the END of an 80 MB buffer; a rewrite there must be seen
    pad = 80 * (1 << 20)
    big = np.full(pad + len(a.sequence), 99, np.int8)
    big[pad:] = a.sequence
    ca = a.cores.copy()
    for f in ("left_pos", "right_pos", "lower", "upper"):
        getattr(ca, f)[:] += pad
    x = run_both_directions(oracle_extend, ca, big, p)
    for rep in range(2):                                    # second pass: served from the device copy
        y = run_both_directions(gpu_extend, ca, big, p)
        assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], "large library, pass %d" % rep)
    cb = b.cores.copy()
    for f in ("left_pos", "right_pos", "lower", "upper"):
        getattr(cb, f)[:] += pad
    big[pad:] = b.sequence
    x = run_both_directions(oracle_extend, cb, big, p)
    y = run_both_directions(gpu_extend, cb, big, p)
    assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], "large library rewritten in place")
    # batch mode: one family of the batch rewritten in place between two calls
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.extend import extend_batch
    fams = [synth_family(40 + 10 * i, 120, 14, K=70, seed=200 + i) for i in range(4)]
    bufs = [f.sequence.copy() for f in fams]
    ep = to_extend_params(p)
    for rnd in range(2):
        if rnd == 1:
            alt = synth_family(40, 120, 14, K=70, seed=999)
            assert len(alt.sequence) == len(bufs[0])
            bufs[0][:] = alt.sequence
            fams[0] = alt
        cs = [f.cores.copy() for f in fams]; ms = [new_master(p.L) for _ in fams]
        extend_batch(1, [(c, s_, m) for c, s_, m in zip(cs, bufs, ms)], ep)
        for f, c, m in zip(fams, cs, ms):
            c0 = f.cores.copy(); m0 = new_master(p.L)
            po.oracle_extend(1, c0, f.sequence, m0, p)
            assert np.array_equal(m, m0) and np.array_equal(c.right_len, c0.right_len) and np.array_equal(c.score, c0.score), rnd
