"""Batch mode: many families in one launch, one workgroup per family, each family with its own
consensus, vote and stop rule.  Every family must come out exactly as if it had been run alone (oracle)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_batch
from repeatafterme_amd.synth import synth_adversarial, synth_family

from helpers import to_extend_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["cell-parallel where it applies", "lane per flank only"])
def _route(request, monkeypatch):
    """Families the cell-parallel kernel can take (W 14/20/40/80, non-positive penalties, up to 256 flanks) run K lanes
    per flank; every test of this file runs a second time with that kernel switched off so that the lane-per-flank
    family kernels (register-resident and streaming) keep their coverage."""
    if request.param == "lane per flank only":
        monkeypatch.setenv("RAMX_NO_CP", "1")
    else:
        monkeypatch.delenv("RAMX_NO_CP", raising=False)
    yield request.param


def _families(k):
    fams = []
    for i in range(k):
        if i % 3 == 0:
            fs = synth_family(int(40 + 37 * (i % 11)), 200, 40, K=60 + 13 * (i % 9), seed=500 + i, both_sides=True,
                              minus_frac=0.3, n_run_frac=0.1)
        else:
            fs = synth_adversarial(600 + i, n_windows=4 + i % 7, L=200, W=40, K=50 + 11 * (i % 8), lowercase=(i % 5 == 0))
        fams.append(fs)
    return fams


@pytest.mark.parametrize("W,matrix", [(40, "14p43g"), (14, "20p43g"), (20, "repeatscout")])
def test_batch_of_families_equals_one_by_one_oracle(W, matrix):
    fams = _families(60)
    p = po.Params.named(matrix, bandwidth=W, L=200, when_to_stop=30)
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
        assert (ir[i].ret, il[i].ret, ir[i].rows_executed, il[i].rows_executed) == w[:4], f"family {i}"
        assert np.array_equal(m, w[4]), f"family {i}: consensus"
        assert np.array_equal(c.left_len, w[5].left_len) and np.array_equal(c.right_len, w[5].right_len), f"family {i}"
        assert np.array_equal(c.score, w[5].score), f"family {i}"
        assert ir[i].persistent == 1


def test_batch_edge_cases_and_fallback():
    """Empty batch, families with no extendable core, a 600-flank family (too large for a workgroup: falls back to
    the single-family path), and an unsupported band width (whole batch falls back)."""
    ep = to_extend_params(po.Params.named("14p43g", bandwidth=40, L=80, when_to_stop=20))
    assert extend_batch(1, [], ep) == []
    big = synth_family(600, 80, 40, K=50, seed=71)
    small = synth_family(30, 80, 40, K=40, seed=72)
    none = synth_family(10, 80, 40, K=40, seed=73)
    none.cores.right_ext[:] = 0
    fams = [small, big, none]
    for W in (40, 9):
        p = po.Params.named("14p43g", bandwidth=W, L=80, when_to_stop=20)
        cs = [f.cores.copy() for f in fams]; ms = [new_master(80) for _ in fams]
        infos = extend_batch(1, [(c, f.sequence, m) for c, f, m in zip(cs, fams, ms)], to_extend_params(p))
        for i, f in enumerate(fams):
            c = f.cores.copy(); m = new_master(80)
            o = po.oracle_extend(1, c, f.sequence, m, p)
            assert infos[i].ret == o.ret and infos[i].rows_executed == o.rows_executed, (W, i)
            assert np.array_equal(ms[i], m) and np.array_equal(cs[i].right_len, c.right_len) and np.array_equal(cs[i].score, c.score)


@pytest.mark.parametrize("W,matrix,kw", [
    (80, "20p43g", {}),                                  # BASELINE config 5: wide band
    (9, "14p43g", {}),                                   # a width with no register-resident instantiation
    (33, "repeatscout", dict(gap=-2)),
    (40, "14p43g", dict(gapopen=3, gapextn=-4)),         # positive penalty: full candidate recurrence (CHAIN)
    (14, "25p43g", dict(gapopen=-20, gapextn=2)),
])
def test_streaming_family_kernel_any_width_and_gap_sign(W, matrix, kw, _route):
    """Families no register-resident kernel can take run in the streaming family kernel (persistent == 2): still one
    launch for all of them, each equal to its own oracle run.  (W = 80 with the built-in penalties is register-resident
    in the cell-parallel kernel: lanes_per_flank > 1, persistent == 1.)"""
    fams = _families(24)
    p = po.Params.named(matrix, bandwidth=W, L=160, when_to_stop=25, **kw)
    got_c = [fs.cores.copy() for fs in fams]
    got_m = [new_master(p.L) for _ in fams]
    ep = to_extend_params(p)
    ir = extend_batch(1, [(c, fs.sequence, m) for c, fs, m in zip(got_c, fams, got_m)], ep)
    il = extend_batch(0, [(c, fs.sequence, m) for c, fs, m in zip(got_c, fams, got_m)], ep)
    for i, fs in enumerate(fams):
        c = fs.cores.copy(); m = new_master(p.L)
        r1 = po.oracle_extend(1, c, fs.sequence, m, p)
        r0 = po.oracle_extend(0, c, fs.sequence, m, p)
        assert (ir[i].ret, il[i].ret, ir[i].rows_executed, il[i].rows_executed) == (r1.ret, r0.ret, r1.rows_executed, r0.rows_executed), (W, i)
        assert np.array_equal(got_m[i], m), (W, i)
        assert np.array_equal(got_c[i].left_len, c.left_len) and np.array_equal(got_c[i].right_len, c.right_len), (W, i)
        assert np.array_equal(got_c[i].score, c.score), (W, i)
        if W == 80 and not kw and _route != "lane per flank only" and ir[i].lanes_per_flank > 1:
            assert ir[i].persistent == 1
        else:
            assert ir[i].persistent == 2 and il[i].persistent == 2


@pytest.mark.parametrize("maxn", [20, 64, 100, 200, 400])
def test_streaming_family_kernel_block_sizes(maxn):
    """The streaming family kernel is launched with 64, 128, 256 or 512 threads depending on the largest family of
    the batch; every shape must fill its tables and reduce its vote correctly."""
    rng = np.random.default_rng(maxn)
    fams = [synth_family(int(rng.integers(max(1, maxn // 2), maxn + 1)) if i == 0 else int(rng.integers(1, maxn + 1)), 120, 9,
                         K=int(rng.integers(20, 120)), seed=4000 + maxn + i, both_sides=True, minus_frac=0.3, n_run_frac=0.3)
            for i in range(10)]
    p = po.Params.named("18p43g", bandwidth=9, L=120, when_to_stop=30)
    cs = [fs.cores.copy() for fs in fams]; ms = [new_master(p.L) for _ in fams]
    ep = to_extend_params(p)
    ir = extend_batch(1, [(c, fs.sequence, m) for c, fs, m in zip(cs, fams, ms)], ep)
    il = extend_batch(0, [(c, fs.sequence, m) for c, fs, m in zip(cs, fams, ms)], ep)
    for i, fs in enumerate(fams):
        c = fs.cores.copy(); m = new_master(p.L)
        r1 = po.oracle_extend(1, c, fs.sequence, m, p)
        r0 = po.oracle_extend(0, c, fs.sequence, m, p)
        assert (ir[i].ret, il[i].ret, ir[i].rows_executed, il[i].rows_executed) == (r1.ret, r0.ret, r1.rows_executed, r0.rows_executed), (maxn, i)
        assert np.array_equal(ms[i], m) and np.array_equal(cs[i].score, c.score), (maxn, i)
        assert np.array_equal(cs[i].left_len, c.left_len) and np.array_equal(cs[i].right_len, c.right_len), (maxn, i)
        assert ir[i].persistent == 2
