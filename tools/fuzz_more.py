"""Extended randomised parity (not part of the test suite): larger families than tests/test_gpu_fuzz.py -- up to a few
thousand flanks, so that the device-wide mode runs with many workgroups, four or eight band waves, every lanes-per-flank
shape -- random scoring systems and stop parameters, each run against the oracle.  Usage: fuzz_more.py [first] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as po
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment
from repeatafterme_amd.synth import synth_adversarial, synth_family
from helpers import to_extend_params
from test_gpu_fuzz import _random_params, _oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for rd in range(first, first + rounds):
    rng = np.random.default_rng(77000 + rd)
    W = [14, 20, 40, 80][rd % 4]
    L = int(rng.integers(20, 140))
    p = _random_params(rng, W, L)
    ep = to_extend_params(p)
    n = int([rng.integers(33, 130), rng.integers(130, 700), rng.integers(700, 3000), rng.integers(3000, 9000)][int(rng.integers(0, 4))])
    if rng.random() < 0.7:
        fs = synth_family(n, L, W, K=int(rng.integers(0, L + 30)), seed=88000 + rd, div=float(rng.uniform(0.0, 0.3)),
                          both_sides=bool(rng.random() < 0.7), minus_frac=float(rng.uniform(0, 0.6)), n_run_frac=float(rng.uniform(0, 0.3)))
    else:
        fs = synth_adversarial(88000 + rd, n_windows=max(1, n // 12), L=L, W=W, K=int(rng.integers(5, L + 30)),
                               div=float(rng.uniform(0.02, 0.3)), lowercase=bool(rng.random() < 0.3))
    w, wm, wc = _oracle(fs, p)
    c = fs.cores.copy(); m = new_master(L)
    a = extend_alignment(1, c, fs.sequence, m, ep)
    b = extend_alignment(0, c, fs.sequence, m, ep)
    got = (a.ret, b.ret, a.rows_executed, b.rows_executed, a.limit_warning, b.limit_warning)
    ok = got == w and np.array_equal(m, wm) and np.array_equal(c.left_len, wc.left_len) and np.array_equal(c.right_len, wc.right_len) \
        and np.array_equal(c.score, wc.score)
    print(f"round {rd}: W={W} L={L} cores={fs.cores.n} go={p.gapopen} ge={p.gapextn} stop={p.when_to_stop} lanes={a.lanes_per_flank}/{b.lanes_per_flank} "
          f"persistent={a.persistent}/{b.persistent} rows={a.rows_executed}/{b.rows_executed} {'ok' if ok else 'MISMATCH ' + str((got, w))}", flush=True)
    bad += 0 if ok else 1
print(f"{rounds} rounds, {bad} mismatches")
sys.exit(1 if bad else 0)
