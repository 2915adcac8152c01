"""The persistent register-resident kernel (one launch per direction, rows on chip, vote fused with a
device-wide barrier) against the streaming per-column kernel and the oracle.  Band widths 14, 20, 40 and 80 (one wave per
SIMD only) have a persistent instantiation; everything else, positive gap penalties, flank sets too large to be co-resident and
multi-rank runs use the streaming kernel."""
import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd.datamodel import CoreSet, new_master
from repeatafterme_amd.synth import synth_adversarial, synth_family

from helpers import assert_same_result, gpu_extend, oracle_extend, run_both_directions, to_extend_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["one-workgroup route", "device-wide cell-parallel", "device-wide lane-per-flank"])
def _small_family_route(request, monkeypatch):
    """Seam 1 runs a family of up to 512 extendable cores as a batch of one (block-local vote); every test of this
    file runs again with that route switched off, once through the device-wide cell-parallel kernel (K lanes per
    flank) and once through the device-wide lane-per-flank persistent kernel, so that all three keep their coverage at
    small sizes."""
    monkeypatch.delenv("RAMX_NO_FAMILY_ROUTE", raising=False)
    monkeypatch.delenv("RAMX_NO_CP_DEVICE", raising=False)
    if request.param != "one-workgroup route":
        monkeypatch.setenv("RAMX_NO_FAMILY_ROUTE", "1")
    if request.param == "device-wide lane-per-flank":
        monkeypatch.setenv("RAMX_NO_CP_DEVICE", "1")
    yield


def _run_device(fs, p, direction, monkeypatch, persistent):
    from repeatafterme_amd.device import Device, resolve_flanks
    if persistent:
        monkeypatch.delenv("RAMX_NO_PERSISTENT", raising=False)
    else:
        monkeypatch.setenv("RAMX_NO_PERSISTENT", "1")
    dev = Device(0)
    dev.load_library(fs.sequence)
    flanks, idx = resolve_flanks(direction, fs.cores, p.bandwidth, p.L)
    dev.begin_direction(flanks, to_extend_params(p))
    info = dev.run_direction()
    cons, th, tp = dev.download()
    state = [dev.peek_state(i) for i in (0, max(len(idx) - 1, 0))] if len(idx) else []
    dev.close()
    return info, cons, th, tp, state


@pytest.mark.parametrize("W", [14, 20, 40, 80])
def test_persistent_equals_streaming_and_oracle(W, monkeypatch):
    for seed in (300, 301, 302, 303):
        fs = synth_adversarial(seed, lowercase=(seed % 2 == 0))
        p = po.Params.named("14p43g" if seed % 2 else "repeatscout", bandwidth=W, L=110, when_to_stop=25)
        for direction in (1, 0):
            a = _run_device(fs, p, direction, monkeypatch, True)
            b = _run_device(fs, p, direction, monkeypatch, False)
            assert a[0].persistent == 1 and b[0].persistent == 0
            assert (a[0].ret, a[0].rows_executed, a[0].limit_warning) == (b[0].ret, b[0].rows_executed, b[0].limit_warning)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
            for (ca, ha, pa), (cb, hb, pb) in zip(a[4], b[4]):      # final DP rows, cell by cell
                assert np.array_equal(ca, cb) and (ha, pa) == (hb, pb)
        x = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        y = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], f"seed={seed} W={W}")


def test_persistent_stop_rule_and_degenerate_inputs():
    fs = synth_family(200, 90, 14, K=50, seed=17, both_sides=True, minus_frac=0.3)
    for kw in (dict(when_to_stop=0), dict(when_to_stop=1), dict(L=1), dict(L=2, when_to_stop=1), dict(minimprovement=-3),
               dict(cappenalty=0), dict(L=51, when_to_stop=1), dict(L=70, when_to_stop=20), dict(when_to_stop=1000)):
        args = dict(bandwidth=14, L=90, when_to_stop=100)
        args.update(kw)
        p = po.Params.named("20p43g", **args)
        a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], str(kw))
        assert b[2].persistent == 1
        assert (a[2].rows_executed, a[2].limit_warning, a[3].rows_executed, a[3].limit_warning) == \
               (b[2].rows_executed, b[2].limit_warning, b[3].rows_executed, b[3].limit_warning), str(kw)
    seq = np.array([0, 1, 2, 3] * 40, np.int8)
    p = po.Params.named("14p43g", bandwidth=14, L=40, when_to_stop=10)
    for c in (CoreSet(left_pos=[], right_pos=[], lower=[], upper=[], orient=[], left_ext=[], right_ext=[]),
              CoreSet(left_pos=[10, 30], right_pos=[12, 33], lower=[0, 20], upper=[19, 159], orient=[0, 0], left_ext=[0, 0], right_ext=[0, 0]),
              CoreSet(left_pos=[50], right_pos=[52], lower=[0], upper=[159], orient=[0], left_ext=[1], right_ext=[1])):
        a = run_both_directions(oracle_extend, c, seq, p)
        b = run_both_directions(gpu_extend, c, seq, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], "degenerate")
        assert a[2].rows_executed == b[2].rows_executed and b[2].persistent == 1


def test_mixed_fast_and_masked_waves_uneven_load():
    """Waves on the in-bounds fast path and ragged waves on the masked path meet at the same barrier."""
    a = synth_family(2000, 160, 40, K=120, seed=31, both_sides=True)
    b = synth_adversarial(41, n_windows=40, L=160, W=40, K=100)
    off = len(a.sequence)
    seq = np.concatenate((a.sequence, b.sequence))
    cores = CoreSet(**{k: np.concatenate((getattr(a.cores, k), getattr(b.cores, k) + (off if k in ("left_pos", "right_pos", "lower", "upper") else 0)))
                       for k in ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext", "right_ext")})
    p = po.Params.named("14p43g", bandwidth=40, L=160, when_to_stop=40)
    x = run_both_directions(oracle_extend, cores, seq, p)
    y = run_both_directions(gpu_extend, cores, seq, p)
    assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], "mixed")
    assert y[2].persistent == 1


def test_too_many_flanks_for_residency_fall_back_to_streaming():
    """More flanks than resident lanes (2 blocks x 256 CUs x 256 lanes = 131,072): streaming kernel, same results."""
    n = 131072 + 6400
    fs = synth_family(n, 12, 40, K=12, seed=9)
    p = po.Params.named("14p43g", bandwidth=40, L=12, when_to_stop=12)
    c1, c2 = fs.cores.copy(), fs.cores.copy()
    m1, m2 = new_master(12), new_master(12)
    a = po.oracle_extend(1, c1, fs.sequence, m1, p)
    b = gpu_extend(1, c2, fs.sequence, m2, p)
    assert b.persistent == 0
    assert a.ret == b.ret and np.array_equal(m1, m2) and np.array_equal(c1.right_len, c2.right_len) and np.array_equal(c1.score, c2.score)


def _scaled(p, k):
    """The same scoring system with every score multiplied by k (a valid user-supplied matrix for the C-ABI)."""
    q = po.Params(**{f: getattr(p, f) for f in ("bandwidth", "cappenalty", "minimprovement", "L", "when_to_stop", "l",
                                                  "gapopen", "gapextn", "matrix")})
    m = q.matrix.astype(np.int64).copy().reshape(100, 100)
    for a in range(4):
        for b in list(range(8)) + [99]:
            m[a, b] *= k
    q.matrix = m.reshape(-1).astype(np.int32)
    q.gapopen *= k; q.gapextn *= k; q.cappenalty *= k; q.minimprovement *= k
    return q


@pytest.mark.parametrize("W", [14, 40, 80])
def test_fast_band_bounds_and_fallback(W, monkeypatch):
    """The in-bounds fast band packs (score << 4 | cell) keys and reads int8 scores; scoring systems outside those
    bounds must take the general band for every wave (host check), and the switch RAMX_NO_FASTPACK forces it."""
    fs = synth_family(700, 200, W, K=150, seed=77, both_sides=True, minus_frac=0.3, n_run_frac=0.2)
    base = po.Params.named("14p43g", bandwidth=W, L=200, when_to_stop=40)
    ref = run_both_directions(oracle_extend, fs.cores, fs.sequence, base)
    for env in (None, "1"):
        if env is None:
            monkeypatch.delenv("RAMX_NO_FASTPACK", raising=False)
        else:
            monkeypatch.setenv("RAMX_NO_FASTPACK", env)
        got = run_both_directions(gpu_extend, fs.cores, fs.sequence, base)
        assert_same_result(ref[0], ref[1], ref[2:], got[0], got[1], got[2:], f"NO_FASTPACK={env}")
        assert got[2].persistent == 1
    monkeypatch.delenv("RAMX_NO_FASTPACK", raising=False)
    for k in (5, 20, 900):           # 5: still int8; 20: scores leave int8; 900: go + ge leaves int16 (streaming kernel)
        p = _scaled(base, k)
        a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"scale={k}")
        assert b[2].persistent == (1 if k < 900 else 0)
        # scaling every score scales nothing else: same consensus and lengths as the unscaled run
        assert np.array_equal(a[1], ref[1]) and np.array_equal(a[0].left_len, ref[0].left_len)


def _two_copy_family(n, L, W, gap, seed, short_frac=0.0):
    """Flanks = 300 aligned columns, `gap` random columns, then a SECOND shared copy (300 columns, 8 % substitutions, same
    offset in every flank): behind the first copy every flank sinks to its cap (the lean band's territory), the second copy
    makes the row scores climb again -- the band must leave the lean variant exactly when a lane could matter again."""
    fs = synth_family(n, L, W, K=300, seed=seed, core_len=12)
    win = len(fs.sequence) // n
    seq = fs.sequence.reshape(n, win).copy()
    rng = np.random.default_rng(seed + 1)
    anc2 = rng.integers(0, 4, size=300, dtype=np.int8)
    s0 = 12 + 300 + gap
    block = np.broadcast_to(anc2, (n, 300)).copy()
    sub = rng.random((n, 300)) < 0.08
    block[sub] = (block[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
    seq[:, s0:s0 + 300] = block
    fs.sequence = np.ascontiguousarray(seq.reshape(-1))
    if short_frac > 0:            # some flanks end early: their waves run the far-end-masked variants
        short = np.nonzero(rng.random(n) < short_frac)[0]
        fs.cores.upper[short] = fs.cores.right_pos[short] + rng.integers(200, 900, size=len(short))
    return fs


@pytest.mark.parametrize("W,matrix,gap,short_frac", [(40, "14p43g", 400, 0.0), (14, "20p43g", 250, 0.0), (20, "25p43g", 300, 0.05),
                                                       (40, "repeatscout", 350, 0.03), (80, "20p43g", 300, 0.03)])
def test_lean_band_switches_on_and_off_exactly(W, matrix, gap, short_frac, monkeypatch):
    """The LEAN band (prk_band_fast<.., LEAN>: no candidate rows, no best-cell index, taken while no lane of a wave can
    contribute more than its cap or set a record) across a run that leaves the alignment, sits at the cap for hundreds of
    columns and then meets a second shared copy: consensus, stop row, lengths and scores equal the oracle's; the final DP rows
    equal a run with RAMX_NO_LEAN=1 cell by cell.  All three routes of the autouse fixture (the cell-parallel one has no lean
    variant and serves as a second reference)."""
    L = 300 + gap + 300 + 150
    fs = _two_copy_family(700, L, W, gap, seed=4100 + W, short_frac=short_frac)
    p = po.Params.named(matrix, bandwidth=W, L=L, when_to_stop=L)
    a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"W={W} {matrix}")
    assert (a[2].rows_executed, a[2].ret) == (b[2].rows_executed, b[2].ret) and a[2].rows_executed == L
    # the second copy really is found again (the consensus behind the gap is the second ancestor, the best row lies in it)
    assert a[2].ret > 300 + gap
    import os
    if os.environ.get("RAMX_NO_CP_DEVICE"):
        lean = _run_device(fs, p, 1, monkeypatch, True)
        monkeypatch.setenv("RAMX_NO_LEAN", "1")
        full = _run_device(fs, p, 1, monkeypatch, True)
        monkeypatch.delenv("RAMX_NO_LEAN")
        assert lean[0].persistent == 1 and lean[0].lanes_per_flank == 1
        runs = [lean]
        # the packed-row kernel (csrc/ramx_kernels_packed.h: two cells per register in relative int16) takes every column from
        # the first one in which no flank has a low out-of-bounds cell (the 12-base cores here: row W - 12), LEAN rows included;
        # RAMX_NO_PK=1 keeps the whole direction on the int32 rows.  The rows it writes back are compared cell by cell below
        # like every other run's
        assert lean[0].packed_rows >= L - W and lean[0].lean_rows > 50, (W, lean[0].packed_rows, lean[0].lean_rows)
        assert full[0].packed_rows >= L - W and full[0].lean_rows == 0
        monkeypatch.setenv("RAMX_NO_PK", "1")
        runs.append(_run_device(fs, p, 1, monkeypatch, True))
        monkeypatch.delenv("RAMX_NO_PK")
        assert runs[-1][0].packed_rows == 0
        monkeypatch.setenv("RAMX_NO_PK", "1")
        monkeypatch.setenv("RAMX_NO_LEAN", "1")
        runs.append(_run_device(fs, p, 1, monkeypatch, True))         # the int32 rows without the LEAN band: the plainest reference
        monkeypatch.delenv("RAMX_NO_PK")
        monkeypatch.delenv("RAMX_NO_LEAN")
        # the leader path (prk_leader_rows: a wave with a few lanes that fail the LEAN test runs LEAN and computes those
        # lanes' candidate rows and best cell with all its lanes): off, and for any number of such lanes per wave
        for lm in ("0", "64"):
            monkeypatch.setenv("RAMX_LEADER_MAX", lm)
            runs.append(_run_device(fs, p, 1, monkeypatch, True))
        monkeypatch.delenv("RAMX_LEADER_MAX")
        for x in runs:
            assert np.array_equal(x[1], full[1]) and np.array_equal(x[2], full[2]) and np.array_equal(x[3], full[3])
            for (ca, ha, pa), (cb, hb, pb) in zip(x[4], full[4]):
                assert np.array_equal(ca, cb) and (ha, pa) == (hb, pb)


@pytest.mark.parametrize("W,matrix", [(40, "14p43g"), (14, "20p43g"), (20, "25p43g")])
def test_packed_rows_speculation_rolls_back_exactly(W, matrix, monkeypatch):
    """The packed-row kernel computes a row on the workgroup's own argmax before the device's vote is known
    (csrc/ramx_kernels_packed.h).  RAMX_TEST_PK_WRONG_EVERY=n replaces every n-th guess by another base: the saved row and records
    must come back and the row be computed again with the winner -- consensus, stop row, lengths, scores and the final DP rows
    equal the oracle's / a run without speculation (RAMX_NO_PK_SPEC=1), including a wrong guess at the stop row, at the last
    row, in rows that set a new maximum, and in rebase rows (every 16th)."""
    import os
    if not os.environ.get("RAMX_NO_CP_DEVICE"):
        pytest.skip("the lane-per-flank route is selected by the fixture's RAMX_NO_CP_DEVICE leg")
    L = 720
    fs = _two_copy_family(1500, L, W, 60, seed=5200 + W, short_frac=0.04)
    for kw in (dict(when_to_stop=L), dict(when_to_stop=30), dict(when_to_stop=1)):
        p = po.Params.named(matrix, bandwidth=W, L=L, **kw)
        monkeypatch.setenv("RAMX_NO_PK_SPEC", "1")
        ref = _run_device(fs, p, 1, monkeypatch, True)
        monkeypatch.delenv("RAMX_NO_PK_SPEC")
        assert ref[0].packed_rows > 0 and ref[0].respeculated_rows == 0
        x = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        y = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], f"W={W} {kw}")
        for every in ("0", "1", "2", "3", "7", "16"):
            monkeypatch.setenv("RAMX_TEST_PK_WRONG_EVERY", every)
            got = _run_device(fs, p, 1, monkeypatch, True)
            monkeypatch.delenv("RAMX_TEST_PK_WRONG_EVERY")
            assert (got[0].ret, got[0].rows_executed, got[0].limit_warning) == (ref[0].ret, ref[0].rows_executed, ref[0].limit_warning), (every, kw)
            assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), (every, kw)
            for (ca, ha, pa), (cb, hb, pb) in zip(got[4], ref[4]):
                assert np.array_equal(ca, cb) and (ha, pa) == (hb, pb), (every, kw)
            if every in ("1", "2", "3"):
                assert got[0].respeculated_rows > 0, (every, kw)          # the hook really makes rows run twice


@pytest.mark.parametrize("seg", ["16", "50", "128"])
def test_packed_rows_direction_in_pieces(seg, monkeypatch):
    """A long direction runs as several launches of the packed-row kernel (RAMX_PK_SEGMENT columns each; 2,048 by default): rows go
    back to HBM, the sums of the next row are handed over as plain words, the control block is passed on, and only the base words
    a piece reads are packed (the next piece's beside the running one).  With pieces of 16 / 50 / 128 columns -- boundaries in
    the aligned phase, in the LEAN phase, at the stop row, at rebase rows -- results and final DP rows equal a one-piece run and
    the oracle."""
    import os
    if not os.environ.get("RAMX_NO_CP_DEVICE"):
        pytest.skip("the lane-per-flank route is selected by the fixture's RAMX_NO_CP_DEVICE leg")
    W, L = 40, 760
    fs = _two_copy_family(900, L, W, 100, seed=6100, short_frac=0.05)
    for kw in (dict(when_to_stop=L), dict(when_to_stop=40)):
        p = po.Params.named("14p43g", bandwidth=W, L=L, **kw)
        monkeypatch.setenv("RAMX_PK_SEGMENT", "0")
        ref = _run_device(fs, p, 1, monkeypatch, True)
        monkeypatch.setenv("RAMX_PK_SEGMENT", seg)
        got = _run_device(fs, p, 1, monkeypatch, True)
        x = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        y = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        monkeypatch.delenv("RAMX_PK_SEGMENT")
        assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], f"seg={seg} {kw}")
        assert ref[0].packed_rows > 0 and got[0].packed_rows == ref[0].packed_rows
        assert (got[0].ret, got[0].rows_executed, got[0].limit_warning) == (ref[0].ret, ref[0].rows_executed, ref[0].limit_warning)
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
        for (ca, ha, pa), (cb, hb, pb) in zip(got[4], ref[4]):
            assert np.array_equal(ca, cb) and (ha, pa) == (hb, pb)


@pytest.mark.parametrize("W,matrix", [(40, "14p43g"), (20, "25p43g")])
def test_packed_rows_vote_wave_shape_equals_band_waves_alone(W, matrix, monkeypatch):
    """Up to 65,536 flanks a workgroup of the packed-row kernel is four band waves and a vote wave WITHOUT flanks (320 threads:
    the fifth wave polls the vote and runs the stop rule while the others compute); RAMX_PK_NO_VW=1 gives the four band waves
    alone (256 threads, wave 0 votes after its row).  Same results, same final DP rows, and the oracle's -- with rows ahead of
    the vote, forced wrong guesses and pieces in both shapes."""
    import os
    if not os.environ.get("RAMX_NO_CP_DEVICE"):
        pytest.skip("the lane-per-flank route is selected by the fixture's RAMX_NO_CP_DEVICE leg")
    L = 720
    fs = _two_copy_family(1100, L, W, 60, seed=7300 + W, short_frac=0.05)
    p = po.Params.named(matrix, bandwidth=W, L=L, when_to_stop=35)
    x = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    y = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], f"W={W}")
    ref = _run_device(fs, p, 1, monkeypatch, True)
    assert ref[0].packed_rows > 0
    for env in ({"RAMX_PK_NO_VW": "1"}, {"RAMX_PK_NO_VW": "1", "RAMX_TEST_PK_WRONG_EVERY": "3"}, {"RAMX_TEST_PK_WRONG_EVERY": "2"},
                {"RAMX_PK_SEGMENT": "64"}, {"RAMX_PK_NO_VW": "1", "RAMX_PK_SEGMENT": "64"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = _run_device(fs, p, 1, monkeypatch, True)
        for k in env:
            monkeypatch.delenv(k)
        assert (got[0].ret, got[0].rows_executed, got[0].limit_warning) == (ref[0].ret, ref[0].rows_executed, ref[0].limit_warning), env
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), env
        for (ca, ha, pa), (cb, hb, pb) in zip(got[4], ref[4]):
            assert np.array_equal(ca, cb) and (ha, pa) == (hb, pb), env


@pytest.mark.parametrize("which", ["RAMX_TEST_PK_SPREAD", "RAMX_TEST_PK_SPREAD_ROWS"], ids=["at entry", "a computed row"])
def test_packed_rows_refuse_a_row_outside_the_span(which, monkeypatch, capfd):
    """The packed rows are exact because every in-bounds cell lies within a span of its row's best cell that the host computes
    from the scoring system (csrc/ramx_packed.hip ramx_pk_plan).  The kernel does not only assume that: the rows it is handed
    and every 64th row it computes are checked, a row outside the span raises an error word, and the host repeats the direction
    on the per-column route.  The two hooks pass a span of 5 to one check or the other: results still equal the oracle's, the
    message names the check that refused, and `packed_rows` says the packed rows were dropped."""
    import os
    if not os.environ.get("RAMX_NO_CP_DEVICE"):
        pytest.skip("the lane-per-flank route is selected by the fixture's RAMX_NO_CP_DEVICE leg")
    W, L = 40, 720
    fs = _two_copy_family(700, L, W, 60, seed=8100, short_frac=0.05)
    p = po.Params.named("14p43g", bandwidth=W, L=L, when_to_stop=40)
    ok = _run_device(fs, p, 1, monkeypatch, True)
    assert ok[0].packed_rows > 0 and ok[0].persistent == 1
    monkeypatch.setenv(which, "5")
    capfd.readouterr()
    got = _run_device(fs, p, 1, monkeypatch, True)
    said = capfd.readouterr().err
    assert ("refused the rows it was handed" if which == "RAMX_TEST_PK_SPREAD" else "refused a row it computed") in said, said
    x = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    y = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    monkeypatch.delenv(which)
    assert_same_result(x[0], x[1], x[2:], y[0], y[1], y[2:], which)
    assert got[0].packed_rows == 0 and got[0].persistent == 0, (got[0].packed_rows, got[0].persistent)
    assert (got[0].ret, got[0].rows_executed) == (ok[0].ret, ok[0].rows_executed)
    assert np.array_equal(got[1], ok[1]) and np.array_equal(got[2], ok[2]) and np.array_equal(got[3], ok[3])
