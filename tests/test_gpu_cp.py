"""The cell-parallel band (csrc/ramx_kernels_cp.h): K = 2, 4, 8 or 16 lanes of a wavefront share one flank, the
insertion chain of a row becomes a DPP prefix-max scan.  Every family must come out exactly as the oracle's serial
loop does -- consensus, return values, extension lengths, scores, and the stored DP row cell by cell."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd import _lib
from repeatafterme_amd.datamodel import CoreSet, new_master
from repeatafterme_amd.extend import extend_batch
from repeatafterme_amd.synth import synth_adversarial, synth_family

from helpers import assert_same_result, gpu_extend, oracle_extend, run_both_directions, to_extend_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _cp_on(monkeypatch):
    for k in ("RAMX_NO_CP", "RAMX_CP_K", "RAMX_NO_CP_DEVICE", "RAMX_NO_FAMILY_ROUTE", "RAMX_NO_PERSISTENT"):
        monkeypatch.delenv(k, raising=False)
    yield


def _families(k, L, W, sizes=(5, 16, 17, 31, 40, 64, 65, 100, 128, 129, 200, 256, 300, 512)):
    fams = []
    for i in range(k):
        n = sizes[i % len(sizes)]
        if i % 4 == 3:
            fs = synth_adversarial(900 + i, n_windows=3 + i % 9, L=L, W=W, K=40 + 9 * (i % 7), lowercase=(i % 5 == 0))
        else:
            fs = synth_family(n, L, W, K=50 + 17 * (i % 9), seed=700 + i, both_sides=True, minus_frac=0.3,
                              n_run_frac=0.15, core_len=(10 if i % 3 else 2 * W + 5))
        fams.append(fs)
    return fams


def _check_batch(fams, p, min_cp=1):
    want = []
    for fs in fams:
        c = fs.cores.copy(); m = new_master(p.L)
        r1 = po.oracle_extend(1, c, fs.sequence, m, p)
        r0 = po.oracle_extend(0, c, fs.sequence, m, p)
        want.append((r1.ret, r0.ret, r1.rows_executed, r0.rows_executed, m, c))
    got_c = [fs.cores.copy() for fs in fams]
    got_m = [new_master(p.L) for _ in fams]
    ep = to_extend_params(p)
    ir = extend_batch(1, [(c, fs.sequence, m) for c, fs, m in zip(got_c, fams, got_m)], ep)
    il = extend_batch(0, [(c, fs.sequence, m) for c, fs, m in zip(got_c, fams, got_m)], ep)
    for i, (w, c, m) in enumerate(zip(want, got_c, got_m)):
        tag = f"family {i} ({fams[i].cores.n} cores, lanes/flank {ir[i].lanes_per_flank}/{il[i].lanes_per_flank})"
        assert (ir[i].ret, il[i].ret, ir[i].rows_executed, il[i].rows_executed) == w[:4], tag
        assert np.array_equal(m, w[4]), tag + ": consensus"
        assert np.array_equal(c.left_len, w[5].left_len) and np.array_equal(c.right_len, w[5].right_len), tag
        assert np.array_equal(c.score, w[5].score), tag
    lanes = [x.lanes_per_flank for x in ir + il]
    assert sum(1 for x in lanes if x > 1) >= min_cp, lanes
    return lanes


@pytest.mark.parametrize("W,matrix", [(40, "14p43g"), (14, "20p43g"), (20, "repeatscout"), (80, "20p43g"), (40, "25p43g")])
def test_cp_batch_equals_oracle(W, matrix):
    L = 180
    p = po.Params.named(matrix, bandwidth=W, L=L, when_to_stop=30)
    lanes = _check_batch(_families(28, L, W), p, min_cp=20)
    assert 16 in lanes and 8 in lanes and 4 in lanes


@pytest.mark.parametrize("K", [2, 4, 8])
@pytest.mark.parametrize("W", [14, 20, 40])
def test_cp_forced_lanes_per_flank(W, K, monkeypatch):
    """RAMX_CP_K caps the lanes per flank, so that every scan / reduction shape (quad, half row, row) runs on small
    families too; shapes whose block would exceed the register budget fall back to more lanes or to lane-per-flank."""
    monkeypatch.setenv("RAMX_CP_K", str(K))
    L = 120
    p = po.Params.named("18p43g", bandwidth=W, L=L, when_to_stop=25)
    _check_batch(_families(14, L, W, sizes=(3, 16, 33, 64, 100, 130, 250)), p, min_cp=0)


@pytest.mark.parametrize("route", ["one workgroup", "device-wide"])
def test_cp_stop_rule_and_degenerate_inputs(route, monkeypatch):
    """A family of 100 flanks alone runs device-wide (16 lanes per flank, several workgroups); RAMX_CP_SINGLE_MAX keeps it
    in one workgroup (4 lanes per flank), which is how it runs inside a batch."""
    if route == "one workgroup":
        monkeypatch.setenv("RAMX_CP_SINGLE_MAX", "100000")
    want_lanes = 4 if route == "one workgroup" else 16
    fs = synth_family(100, 90, 14, K=50, seed=17, both_sides=True, minus_frac=0.3)
    for kw in (dict(when_to_stop=0), dict(when_to_stop=1), dict(L=1), dict(L=2, when_to_stop=1), dict(minimprovement=-3),
               dict(cappenalty=0), dict(L=51, when_to_stop=1), dict(L=70, when_to_stop=20), dict(when_to_stop=1000)):
        args = dict(bandwidth=14, L=90, when_to_stop=100)
        args.update(kw)
        p = po.Params.named("20p43g", **args)
        a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], str(kw))
        assert b[2].lanes_per_flank == want_lanes and b[3].lanes_per_flank == want_lanes
        assert (a[2].rows_executed, a[2].limit_warning, a[3].rows_executed, a[3].limit_warning) == \
               (b[2].rows_executed, b[2].limit_warning, b[3].rows_executed, b[3].limit_warning), str(kw)
    seq = np.array([0, 1, 2, 3] * 40, np.int8)
    p = po.Params.named("14p43g", bandwidth=14, L=40, when_to_stop=10)
    for c in (CoreSet(left_pos=[50], right_pos=[52], lower=[0], upper=[159], orient=[0], left_ext=[1], right_ext=[1]),
              CoreSet(left_pos=[50, 90], right_pos=[52, 99], lower=[0, 60], upper=[159, 110], orient=[0, 1], left_ext=[1, 1], right_ext=[1, 1])):
        a = run_both_directions(oracle_extend, c, seq, p)
        b = run_both_directions(gpu_extend, c, seq, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], "degenerate")
        assert a[2].rows_executed == b[2].rows_executed and b[2].lanes_per_flank == 16


@pytest.mark.parametrize("W,n,K", [(14, 30, 16), (20, 60, 8), (20, 100, 4), (14, 200, 2), (80, 30, 16), (80, 60, 8), (40, 30, 16), (40, 128, 4)])
def test_cp_state_bit_exact_per_cell(W, n, K, monkeypatch):
    """After L columns the DP row kept by the cell-parallel kernel equals the oracle's row, cell by cell, both states
    (what bnw_extend.c:1617-1648 asserts for the reference): the scan re-associates the insertion chain exactly.
    RAMX_CP_SINGLE_MAX keeps the family in ONE workgroup (every lanes-per-flank shape of the family kernel); alone it
    would run device-wide above 32 flanks."""
    monkeypatch.setenv("RAMX_CP_PEEK", "1")
    monkeypatch.setenv("RAMX_CP_SINGLE_MAX", "100000")
    L = 45
    fs = synth_family(n, 70, W, K=40, seed=21 + W, both_sides=True, minus_frac=0.4, n_run_frac=0.2,
                      core_len=(2 * W + 3 if W != 20 else 7))
    p = po.Params.named("18p43g", bandwidth=W, L=L, when_to_stop=1000)
    lib = _lib.lib()
    B = 2 * W + 1
    from repeatafterme_amd.device import resolve_flanks
    for direction in (1, 0):
        c = fs.cores.copy(); m = new_master(L)
        info = gpu_extend(direction, c, fs.sequence, m, p)
        assert info.rows_executed == L and info.lanes_per_flank == K
        cons = m[L + 1: 2 * L + 1] if direction else m[L - 1::-1][:L]
        _, idx = resolve_flanks(direction, fs.cores, W, L)
        n_align = fs.cores.n
        score = np.zeros((2, n_align, B, 2), np.int32)
        for o in range(-W, W + 1):
            score[1, :, o + W, :] = 0 if o == 0 else abs(o) * p.gapextn + p.gapopen
        cc = fs.cores
        for row in range(L):
            for k in idx:
                po.oracle_nw_row(direction, row, int(k), n_align, int(cons[row]), int(cc.left_pos[k]), int(cc.right_pos[k]),
                                 int(cc.orient[k]), score, int(cc.lower[k]), int(cc.upper[k]), fs.sequence,
                                 p.matrix, p.gapopen, p.gapextn, W)
        for i in (0, 1, len(idx) // 2, len(idx) - 1):
            cells = np.zeros((B, 2), np.int32)
            rc = lib.ramx_dev_peek_family_state(None, int(i), cells.ctypes.data_as(C.c_void_p))
            assert rc == 0, lib.ramx_last_error()
            k = idx[i]
            sub, gap = score[(L - 1) % 2, k, :, 0].astype(np.int64), score[(L - 1) % 2, k, :, 1].astype(np.int64)
            want = np.stack([np.maximum(sub, gap), np.maximum(sub + p.gapopen, gap) + p.gapextn], axis=1)
            assert np.array_equal(cells, want), f"W {W} dir {direction} flank {i}: cells {np.nonzero((cells != want).any(axis=1))[0][:8]}"


def test_cp_scoring_bounds_fall_back(monkeypatch):
    """Scores outside int8 or large enough to leave the 23-bit key range must not take the cell-parallel kernel."""
    fs = synth_family(60, 120, 14, K=80, seed=5, both_sides=True)
    base = po.Params.named("14p43g", bandwidth=14, L=120, when_to_stop=30)
    ref = run_both_directions(oracle_extend, fs.cores, fs.sequence, base)
    got = run_both_directions(gpu_extend, fs.cores, fs.sequence, base)
    assert_same_result(ref[0], ref[1], ref[2:], got[0], got[1], got[2:], "base")
    assert got[2].lanes_per_flank == 8
    q = po.Params(**{f: getattr(base, f) for f in ("bandwidth", "cappenalty", "minimprovement", "L", "when_to_stop", "l",
                                                   "gapopen", "gapextn", "matrix")})
    mm = q.matrix.astype(np.int64).copy().reshape(100, 100)
    for a in range(4):
        for b in list(range(8)) + [99]:
            mm[a, b] *= 20
    q.matrix = mm.reshape(-1).astype(np.int32)
    q.gapopen *= 20; q.gapextn *= 20; q.cappenalty *= 20; q.minimprovement *= 20
    a = run_both_directions(oracle_extend, fs.cores, fs.sequence, q)
    b = run_both_directions(gpu_extend, fs.cores, fs.sequence, q)
    assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], "x20")
    assert b[2].lanes_per_flank == 1
    monkeypatch.setenv("RAMX_NO_CP", "1")
    got = run_both_directions(gpu_extend, fs.cores, fs.sequence, base)
    assert_same_result(ref[0], ref[1], ref[2:], got[0], got[1], got[2:], "NO_CP")
    assert got[2].lanes_per_flank == 1


# ---- device-wide mode: one flank set over many workgroups, vote through the ticketed shard words ----------------

@pytest.mark.parametrize("W,n,K,matrix", [(40, 1000, 16, "14p43g"), (40, 5000, 16, "14p43g"), (14, 3000, 16, "20p43g"),
                                          (80, 2500, 16, "20p43g"), (20, 20000, 4, "repeatscout"), (40, 700, 16, "25p43g"),
                                          (40, 12000, 8, "18p43g"),     # seven band waves + vote wave, saved row in registers
                                          (40, 15000, 8, "14p43g"),     # eight band waves + vote wave (576 threads)
                                          (40, 20000, 4, "14p43g"),     # 21 cells per lane: seven band waves + vote wave, saved row in LDS
                                          (40, 30000, 4, "18p43g"),     # 21 cells per lane at the capacity edge: vote, then band (no vote wave)
                                          (80, 10000, 8, "20p43g")])    # W = 80, 21 cells per lane, vote wave
def test_cp_device_wide_equals_oracle(W, n, K, matrix):
    """Flank sets above one workgroup (BASELINE config 2's N = 1,000 among them): the cell-parallel kernel in device-wide
    mode, both directions, mixed strands and N runs, against the oracle."""
    L = 150 if n <= 5000 else 60
    fs = synth_family(n, L, W, K=100 if n <= 5000 else 40, seed=50 + W, both_sides=True, minus_frac=0.3, n_run_frac=0.1,
                      core_len=(2 * W + 2 if n != 5000 else 9))
    p = po.Params.named(matrix, bandwidth=W, L=L, when_to_stop=30)
    a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"W={W} n={n}")
    assert (a[2].rows_executed, a[3].rows_executed) == (b[2].rows_executed, b[3].rows_executed)
    assert b[2].persistent == 1 and b[2].lanes_per_flank == K and b[3].lanes_per_flank == K


@pytest.mark.parametrize("W,n", [(40, 700), (40, 20000)])
def test_cp_device_wide_lost_ticket_times_out_and_falls_back(W, n, monkeypatch):
    """One workgroup withholds its words for row 7 (test hook): every vote wave runs into its bounded spin, raises the
    error word, every workgroup leaves, and the host repeats the direction on the lane-per-flank route -- same results,
    no hang.  n = 700: vote-wave kernel (blocks of 6 cells); n = 20,000: the simple order (21 cells per lane)."""
    monkeypatch.setenv("RAMX_TEST_CP_DROP_TICKET", "7")
    L = 60
    fs = synth_family(n, L, W, K=40, seed=77, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=W, L=L, when_to_stop=20)
    a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"W={W} n={n}")
    assert (a[2].rows_executed, a[3].rows_executed) == (b[2].rows_executed, b[3].rows_executed)
    assert b[2].lanes_per_flank == 1 and b[3].lanes_per_flank == 1      # the cell-parallel launch gave up


@pytest.mark.parametrize("K", [2, 4, 8, 16])
def test_cp_device_wide_small_sets_every_shape(K, monkeypatch):
    """With the one-workgroup route off even a tiny set runs device-wide: adversarial ragged sets (masked path, flanks
    that end early, empty flanks) through every lanes-per-flank shape."""
    monkeypatch.setenv("RAMX_NO_FAMILY_ROUTE", "1")
    monkeypatch.setenv("RAMX_CP_K", str(K))
    for seed in (300, 301, 302):
        fs = synth_adversarial(seed, n_windows=30, L=110, W=20, K=70, lowercase=(seed % 2 == 0))
        p = po.Params.named("14p43g" if seed % 2 else "repeatscout", bandwidth=20, L=110, when_to_stop=25)
        a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"seed={seed} K={K}")
        assert (a[2].rows_executed, a[3].rows_executed) == (b[2].rows_executed, b[3].rows_executed)


def test_cp_device_wide_config2_full():
    """BASELINE config 2 in full: N = 1,000 flanks x L = 2,000 bp, bandwidth 40, 14p43g, right extension, 1,600 columns."""
    fs = synth_family(1000, 2000, 40, K=1500, seed=1)
    p = po.Params.named("14p43g", bandwidth=40, L=2000)
    c1, c2 = fs.cores.copy(), fs.cores.copy()
    m1, m2 = new_master(2000), new_master(2000)
    a = po.oracle_extend(1, c1, fs.sequence, m1, p)
    b = gpu_extend(1, c2, fs.sequence, m2, p)
    assert (a.ret, a.rows_executed) == (b.ret, b.rows_executed) == (1500, 1600)
    assert np.array_equal(m1, m2) and np.array_equal(c1.right_len, c2.right_len) and np.array_equal(c1.score, c2.score)
    assert b.lanes_per_flank == 16 and b.persistent == 1


@pytest.mark.parametrize("W,matrix", [(80, "20p43g"), (40, "14p43g"), (14, "18p43g")])
def test_cp_batch_with_families_above_one_workgroup(W, matrix):
    """Families too large for one cell-parallel workgroup (W = 80: 129..512 flanks; W = 40: 257..512) run several
    workgroups each, all of them in ONE device-wide launch next to the one-workgroup families of the same batch; each
    family votes through its own ticket words.  Every family against its own oracle run."""
    L = 140
    sizes = (300, 40, 512, 129, 200, 90, 257, 140, 16, 400)
    fams = [synth_family(n, L, W, K=60 + 11 * i, seed=1200 + i, both_sides=True, minus_frac=0.3, n_run_frac=0.1,
                         core_len=(2 * W + 3 if i % 2 else 12)) for i, n in enumerate(sizes)]
    p = po.Params.named(matrix, bandwidth=W, L=L, when_to_stop=30)
    lanes = _check_batch(fams, p, min_cp=2 * len(sizes))
    assert all(x > 1 for x in lanes), lanes


@pytest.mark.parametrize("drop_family", [-1, 2])
def test_cp_batch_degrades_when_a_device_wide_vote_times_out(drop_family, monkeypatch):
    """Batch mode must degrade, not fail: one workgroup of a multi-workgroup family withholds its words for row 5 (test
    hook), that family's vote waves give up at their bounded spin, and ramx_dev_run_families repeats ONLY the affected
    families on the lane-per-flank route; every family still equals its own oracle run and the others keep their results
    (and their cell-parallel route).  drop_family = -1: every multi-workgroup family of the batch loses a ticket."""
    monkeypatch.setenv("RAMX_TEST_CP_DROP_TICKET", "5")
    if drop_family >= 0:
        monkeypatch.setenv("RAMX_TEST_CP_DROP_FAMILY", str(drop_family))
    W, L = 40, 90
    sizes = (300, 40, 512, 129, 400, 90, 16)          # multi-workgroup at W = 40: more than 256 flanks -> 300, 512, 400
    fams = [synth_family(n, L, W, K=50 + 7 * i, seed=2200 + i, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
            for i, n in enumerate(sizes)]
    p = po.Params.named("14p43g", bandwidth=W, L=L, when_to_stop=25)
    lanes = _check_batch(fams, p, min_cp=0)
    nf = len(sizes)
    for d in (0, 1):                                   # right pass, left pass
        got = lanes[d * nf:(d + 1) * nf]
        for i, n in enumerate(sizes):
            multi = n > 256
            fell_back = multi and (drop_family < 0 or drop_family == i)
            assert (got[i] == 1) == fell_back, (d, i, n, got)


# ---- vote-wave mode: band waves run DEPTH rows ahead of the confirmed vote; wrong guesses roll back ---------------------

@pytest.mark.parametrize("W,n,K,every,delay", [
    (40, 1000, 16, 1, 0),      # every row on a wrong guess: rollback at every row, incl. new-maximum and stop rows
    (40, 1000, 16, 3, 2),      # vote wave held back: the band waves are DEPTH rows ahead when the decision says "wrong"
    (40, 1000, 16, 5, 2),
    (40, 12000, 8, 2, 1),      # 11 cells per lane, eight band waves
    (40, 12000, 8, 7, 0),
    (14, 3000, 16, 4, 1),      # 2 cells per lane
    (80, 2500, 16, 3, 1),      # W = 80, 11 cells per lane
    (40, 20000, 4, 3, 1),      # 21 cells per lane: depth 1 (one row ahead)
    (20, 900, 16, 2, 3)])
def test_cp_vote_wave_forced_wrong_guesses_roll_back(W, n, K, every, delay, monkeypatch):
    """RAMX_TEST_CP_WRONG_EVERY=n replaces the guess of every n-th row by another base (identically in the band waves and
    the vote wave); RAMX_TEST_CP_VOTE_DELAY holds the vote wave back so that the band waves have run their full depth
    ahead when the decision arrives.  The rows computed on the wrong guess -- and everything computed after them -- must
    be rolled back and recomputed: results equal the oracle's, and the kernel reports the recomputed rows.  (The vote-wave mode
    runs one row on the guess between two workgroup barriers; round 3's barrier-free variant that ran several rows ahead was
    removed in round 4: it measured slower.)"""
    monkeypatch.setenv("RAMX_TEST_CP_WRONG_EVERY", str(every))
    if delay:
        monkeypatch.setenv("RAMX_TEST_CP_VOTE_DELAY", str(delay))
    L = 90 if n <= 5000 else 50
    fs = synth_family(n, L, W, K=50 if n <= 5000 else 30, seed=900 + W + every, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g" if W != 80 else "20p43g", bandwidth=W, L=L, when_to_stop=20)
    a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
    b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
    assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], f"W={W} n={n} every={every}")
    assert (a[2].rows_executed, a[3].rows_executed) == (b[2].rows_executed, b[3].rows_executed)
    assert b[2].persistent == 1 and b[2].lanes_per_flank == K
    # at least half of the perturbed rows were really wrong (a perturbed guess can hit the true winner by accident)
    assert b[2].respeculated_rows >= (b[2].rows_executed // every) // 2, (b[2].respeculated_rows, b[2].rows_executed)


def test_cp_vote_wave_rollback_at_stop_and_limit_rows(monkeypatch):
    """Wrong guesses exactly where the loop ends: the stop row (when_to_stop reached), the last row L-1, L = 1, and a run
    whose every row is a new maximum -- with the band waves ahead of the vote."""
    monkeypatch.setenv("RAMX_TEST_CP_WRONG_EVERY", "1")
    monkeypatch.setenv("RAMX_TEST_CP_VOTE_DELAY", "2")
    fs = synth_family(700, 80, 40, K=45, seed=31, both_sides=True, minus_frac=0.3)
    for kw in (dict(when_to_stop=0), dict(when_to_stop=1), dict(L=1), dict(L=2, when_to_stop=1), dict(L=46, when_to_stop=100),
               dict(L=45), dict(L=80, when_to_stop=5), dict(minimprovement=-3), dict(cappenalty=0)):
        args = dict(bandwidth=40, L=80, when_to_stop=30)
        args.update(kw)
        p = po.Params.named("14p43g", **args)
        a = run_both_directions(oracle_extend, fs.cores, fs.sequence, p)
        b = run_both_directions(gpu_extend, fs.cores, fs.sequence, p)
        assert_same_result(a[0], a[1], a[2:], b[0], b[1], b[2:], str(kw))
        assert (a[2].rows_executed, a[2].limit_warning, a[3].rows_executed, a[3].limit_warning) == \
               (b[2].rows_executed, b[2].limit_warning, b[3].rows_executed, b[3].limit_warning), str(kw)
        assert b[2].lanes_per_flank == 16
