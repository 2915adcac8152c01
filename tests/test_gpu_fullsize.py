"""BASELINE.json configs[2] size (N = 100,000 flanks x L = 10,000 bp, bandwidth 40): parity against the
oracle on a column prefix of the FULL flank set, and size-independent properties of the full run."""
import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.synth import synth_family

from helpers import gpu_extend, to_extend_params

pytestmark = pytest.mark.gpu

N, L, W, K = 100_000, 10_000, 40, 1500


@pytest.fixture(scope="module")
def family():
    return synth_family(N, L, W, K=K, seed=1)


PREFIX = 128      # > 2W + 1: rows >= W run the steady-state band (sentinel fills, far-end-masked variant) at the bench's own shape


def _prefix_vs_oracle(fs, cores, tag):
    p = po.Params.named("14p43g", bandwidth=W, L=PREFIX, when_to_stop=PREFIX)
    c1, c2 = cores.copy(), cores.copy()
    m1, m2 = new_master(PREFIX), new_master(PREFIX)
    a = po.oracle_extend(1, c1, fs.sequence, m1, p, trace=True)
    b = gpu_extend(1, c2, fs.sequence, m2, p)
    assert a.ret == b.ret and a.rows_executed == b.rows_executed == PREFIX, tag
    assert np.array_equal(m1, m2), tag
    assert np.array_equal(c1.right_len, c2.right_len) and np.array_equal(c1.score, c2.score), tag
    assert int(np.abs(a.col_sums).max()) < 2 ** 31 and b.overflow32 == 0
    assert b.persistent == 1 and b.lanes_per_flank == 1      # 196 workgroups x 512 lanes: the benchmarked kernel and shape
    return b


def test_full_width_prefix_bit_exact_vs_oracle(family):
    """All 100,000 flanks, first 128 columns (the band is 81 cells wide: rows 40.. are the steady state of the bench):
    int32-exact column sums drive the same consensus, lengths, scores."""
    _prefix_vs_oracle(family, family.cores, "full-length flanks")


def test_full_width_prefix_with_early_ending_flanks(family):
    """The same 100,000 flanks with 2 % of them ending 0..160 bp behind the core: their waves leave the in-bounds fast band
    for the far-end-masked variant (and the general band while r < W) while the other 1,500 waves stay on the fast
    band -- mixed variants at the bench's shape, every output against the oracle."""
    fs = family
    cores = fs.cores.copy()
    rng = np.random.default_rng(11)
    short = np.nonzero(rng.random(N) < 0.02)[0]
    cores.upper[short] = cores.right_pos[short] + rng.integers(0, 160, size=len(short))
    _prefix_vs_oracle(fs, cores, "2 % early-ending flanks")


def test_full_run_properties(family):
    """Full 10,000-column run: (1) recovers the planted ancestor, (2) idempotent, (3) invariant under a
    permutation of the flanks (integer sums are order independent), (4) the int32 guard holds."""
    fs = family
    p = po.Params.named("14p43g", bandwidth=W, L=L)           # default stopafter = 100
    c = fs.cores.copy(); m = new_master(L)
    r = gpu_extend(1, c, fs.sequence, m, p)
    assert r.ret == K and r.rows_executed == K + 100 and r.overflow32 == 0
    anc = np.random.default_rng(1).integers(0, 4, size=K, dtype=np.int8)   # first draw of synth_family(seed=1)
    cons = m[L + 1: L + 1 + K]
    assert (cons == anc).mean() > 0.995
    assert (c.right_len > 0).mean() > 0.99 and abs(np.median(c.right_len) - K) < 0.05 * K
    c2 = fs.cores.copy(); m2 = new_master(L)
    r2 = gpu_extend(1, c2, fs.sequence, m2, p)
    assert r2.ret == r.ret and np.array_equal(m, m2) and np.array_equal(c.score, c2.score)
    perm = np.random.default_rng(5).permutation(N)
    c3 = fs.cores.subset(perm); m3 = new_master(L)
    r3 = gpu_extend(1, c3, fs.sequence, m3, p)
    assert r3.ret == r.ret and np.array_equal(m, m3)
    assert np.array_equal(c3.right_len, c.right_len[perm]) and np.array_equal(c3.score, c.score[perm])


def _reference_digest(name):
    import json, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_digests.json")
    return json.load(open(path))[name]


def test_forced_full_length_run_equals_the_reference_digest(family):
    """stopafter = L forces all 10,000 columns (the bench workload).  The digest of (return value, all 10,000 consensus
    bytes, 100,000 extension lengths, 100,000 scores) equals the one the COMPILED REFERENCE produced on this exact workload
    (tests/golden/fullsize_digests.json, written by tests/golden/make_fullsize_digest.py from oracle/_ref/libramref.so:
    4,586 s of ram_extend.c:859-1258 on one core): reference parity over the whole run, not a prefix.  Two runs agree."""
    from repeatafterme_amd.synth import result_digest
    fs = family
    g = _reference_digest("cfg3")
    assert (g["workload"]["n"], g["workload"]["L"], g["workload"]["W"], g["workload"]["K"], g["workload"]["seed"]) == (N, L, W, K, 1)
    p = po.Params.named("14p43g", bandwidth=W, L=L, when_to_stop=L)
    outs = []
    for _ in range(2):
        c = fs.cores.copy(); m = new_master(L)
        r = gpu_extend(1, c, fs.sequence, m, p)
        assert r.rows_executed == L and r.ret == K == g["ret"]
        assert int(c.score.astype(np.int64).sum()) == g["sum_score"] and int(c.right_len.astype(np.int64).sum()) == g["sum_right_len"]
        assert result_digest(r.ret, m[L + 1:L + 1 + L], c.right_len, c.score) == g["sha1"]
        outs.append((m.copy(), c.right_len.copy(), c.score.copy()))
    assert all(np.array_equal(x, y) for x, y in zip(*outs))


def test_config2_full_run_equals_the_reference_digest():
    """BASELINE configs[1] (N = 1,000 x L = 2,000, all 2,000 columns) against the compiled reference's digest -- the
    cell-parallel device-wide kernel over the aligned phase, the tail and its mispredicted columns."""
    from repeatafterme_amd.synth import result_digest
    g = _reference_digest("cfg2")
    w = g["workload"]
    fs = synth_family(w["n"], w["L"], w["W"], K=w["K"], seed=w["seed"])
    p = po.Params.named("14p43g", bandwidth=w["W"], L=w["L"], when_to_stop=w["L"])
    c = fs.cores.copy(); m = new_master(w["L"])
    r = gpu_extend(1, c, fs.sequence, m, p)
    assert r.rows_executed == w["L"] and r.ret == g["ret"] and r.lanes_per_flank == 16
    assert result_digest(r.ret, m[w["L"] + 1:2 * w["L"] + 1], c.right_len, c.score) == g["sha1"]


@pytest.mark.parametrize("n,route", [(150, "one workgroup"), (1500, "device-wide")])
def test_wrapper_configuration_long_L(n, route):
    """The configuration the reference's wrapper runs (util/extend-stk.pl:352): -L 20000 -bandwidth 40, matrix by
    divergence, -minimprovement 30; a few thousand columns execute before the stop rule ends the run.  Both routes of
    seam 1 (batch of one / persistent kernel) against the oracle, both directions."""
    Lw = 20000
    fs = synth_family(n, Lw, 40, K=2600, seed=321 + n, both_sides=True, minus_frac=0.35, n_run_frac=0.05)
    p = po.Params.named("18p43g", bandwidth=40, L=Lw, when_to_stop=100, minimprovement=30)
    c1, c2 = fs.cores.copy(), fs.cores.copy()
    m1, m2 = new_master(Lw), new_master(Lw)
    want = [po.oracle_extend(d, c1, fs.sequence, m1, p) for d in (1, 0)]
    got = [gpu_extend(d, c2, fs.sequence, m2, p) for d in (1, 0)]
    for a, b in zip(want, got):
        assert (a.ret, a.rows_executed, a.limit_warning) == (b.ret, b.rows_executed, b.limit_warning), route
        assert a.rows_executed > 2500
    assert np.array_equal(m1, m2), route
    assert np.array_equal(c1.left_len, c2.left_len) and np.array_equal(c1.right_len, c2.right_len), route
    assert np.array_equal(c1.score, c2.score), route


def test_w80_above_the_cell_parallel_capacity_stays_register_resident():
    """W = 80 (161 cells per flank; the wrapper's widest band, util/extend-stk.pl:365) with more flanks than the cell-parallel
    kernel holds at 4 lanes per flank (32,768): the lane-per-flank kernel with ONE wave per SIMD keeps the row on chip (half
    of it in accumulation registers), up to 65,536 flanks.  44,000 flanks, 40 columns (rows < W: general band; a few flanks
    end early: masked waves) against the oracle."""
    n, Lx = 44_000, 40
    fs = synth_family(n, Lx, 80, K=400, seed=808, core_len=170)
    rng = np.random.default_rng(5)
    short = np.nonzero(rng.random(n) < 0.01)[0]
    fs.cores.upper[short] = fs.cores.right_pos[short] + rng.integers(0, 30, size=len(short))
    p = po.Params.named("20p43g", bandwidth=80, L=Lx, when_to_stop=Lx)
    c1, c2 = fs.cores.copy(), fs.cores.copy()
    m1, m2 = new_master(Lx), new_master(Lx)
    a = po.oracle_extend(1, c1, fs.sequence, m1, p)
    b = gpu_extend(1, c2, fs.sequence, m2, p)
    assert b.persistent == 1 and b.lanes_per_flank == 1
    assert (a.ret, a.rows_executed) == (b.ret, b.rows_executed)
    assert np.array_equal(m1, m2) and np.array_equal(c1.right_len, c2.right_len) and np.array_equal(c1.score, c2.score)
