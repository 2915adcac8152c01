"""Randomised parity: many small, hostile families with random scoring systems, band widths and stop parameters,
through both seams (single-family extend_alignment and the batch kernel), each against its own oracle run.
Integer path: everything must be bit-exact (return values, executed columns, consensus, lengths, scores)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment, extend_batch
from repeatafterme_amd.synth import synth_adversarial, synth_family

from helpers import to_extend_params

pytestmark = pytest.mark.gpu


def _random_params(rng, W, L):
    name = ["14p43g", "18p43g", "20p43g", "25p43g", "repeatscout"][int(rng.integers(0, 5))]
    kw = dict(bandwidth=W, L=L, when_to_stop=int(rng.integers(1, 60)))
    if name == "repeatscout":
        kw.update(match=int(rng.integers(1, 6)), mismatch=-int(rng.integers(0, 6)), gap=-int(rng.integers(0, 9)))
    p = po.Params.named(name, **kw)
    u = rng.random()
    if u < 0.35:                                   # user gap penalties, including zero and equal
        p.gapopen = -int(rng.integers(0, 45)); p.gapextn = -int(rng.integers(0, 12))
    if rng.random() < 0.4:
        p.cappenalty = -int(rng.integers(0, 150))
    if rng.random() < 0.4:
        p.minimprovement = int(rng.integers(-5, 60))
    return p


def _random_family(rng, seed, L, W):
    if rng.random() < 0.5:
        return synth_family(int(rng.integers(1, 180)), L, W, K=int(rng.integers(0, L + 30)), seed=seed, div=float(rng.uniform(0.0, 0.35)),
                            both_sides=bool(rng.random() < 0.7), minus_frac=float(rng.uniform(0, 0.6)),
                            n_run_frac=float(rng.uniform(0, 0.4)))
    return synth_adversarial(seed, n_windows=int(rng.integers(1, 14)), L=L, W=W, K=int(rng.integers(5, L + 30)),
                             div=float(rng.uniform(0.02, 0.3)), lowercase=bool(rng.random() < 0.3))


def _oracle(fs, p):
    c = fs.cores.copy(); m = new_master(p.L)
    r1 = po.oracle_extend(1, c, fs.sequence, m, p)
    r0 = po.oracle_extend(0, c, fs.sequence, m, p)
    return (r1.ret, r0.ret, r1.rows_executed, r0.rows_executed, r1.limit_warning, r0.limit_warning), m, c


def _check(tag, want, infos, m, c):
    w, wm, wc = want
    ir, il = infos
    assert (ir.ret, il.ret, ir.rows_executed, il.rows_executed, ir.limit_warning, il.limit_warning) == w, tag
    assert np.array_equal(m, wm), tag + ": consensus"
    assert np.array_equal(c.left_len, wc.left_len) and np.array_equal(c.right_len, wc.right_len), tag + ": lengths"
    assert np.array_equal(c.score, wc.score), tag + ": scores"


@pytest.mark.parametrize("round_", range(12))
def test_random_families_and_scoring_systems(round_):
    rng = np.random.default_rng(9000 + round_)
    W = [14, 20, 40, int(rng.integers(1, 13)), int(rng.integers(21, 40)), int(rng.integers(41, 90))][round_ % 6]
    L = int(rng.integers(20, 170))
    p = _random_params(rng, W, L)
    ep = to_extend_params(p)
    fams = [_random_family(rng, 9100 + 40 * round_ + i, L, W) for i in range(40)]
    want = [_oracle(fs, p) for fs in fams]
    tag = f"round {round_} W={W} L={L} go={p.gapopen} ge={p.gapextn} cap={p.cappenalty} minimp={p.minimprovement} stop={p.when_to_stop}"
    # seam: batch (one workgroup per family where the batch kernel applies, one by one otherwise)
    cs = [fs.cores.copy() for fs in fams]; ms = [new_master(L) for _ in fams]
    ir = extend_batch(1, [(c, fs.sequence, m) for c, fs, m in zip(cs, fams, ms)], ep)
    il = extend_batch(0, [(c, fs.sequence, m) for c, fs, m in zip(cs, fams, ms)], ep)
    for i in range(len(fams)):
        _check(f"{tag} batch family {i}", want[i], (ir[i], il[i]), ms[i], cs[i])
    # seam: one family per call (persistent kernel for W in 14/20/40 with non-positive penalties, streaming otherwise)
    for i, fs in enumerate(fams[:12]):
        c = fs.cores.copy(); m = new_master(L)
        a = extend_alignment(1, c, fs.sequence, m, ep)
        b = extend_alignment(0, c, fs.sequence, m, ep)
        _check(f"{tag} single family {i}", want[i], (a, b), m, c)
