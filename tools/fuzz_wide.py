"""Randomised parity at flank counts that take the two-waves-per-SIMD shape of the packed rows (66,000-125,000 flanks; W = 14 / 20 / 40),
short directions so that the single-threaded oracle keeps up.  Usage: RAMX_NO_CP_DEVICE=1 python tools/fuzz_wide.py [first] [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment
from repeatafterme_amd.synth import synth_family
from helpers import to_extend_params
from test_gpu_fuzz import _random_params, _oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bad = 0
for rd in range(first, first + rounds):
    rng = np.random.default_rng(55000 + rd)
    W = [40, 14, 20][rd % 3]
    L = int(rng.integers(60, 110))
    p = _random_params(rng, W, L)
    p.when_to_stop = int(rng.integers(10, 40))
    ep = to_extend_params(p)
    n = int(rng.integers(66000, 125000))
    fs = synth_family(n, L, W, K=int(rng.integers(20, L)), seed=54000 + rd, div=float(rng.uniform(0.0, 0.2)),
                      both_sides=False, minus_frac=float(rng.uniform(0, 0.5)), n_run_frac=float(rng.uniform(0, 0.1)))
    t0 = time.time()
    w, wm, wc = _oracle(fs, p)
    t1 = time.time()
    c = fs.cores.copy(); m = new_master(L)
    a = extend_alignment(1, c, fs.sequence, m, ep)
    b = extend_alignment(0, c, fs.sequence, m, ep)
    got = (a.ret, b.ret, a.rows_executed, b.rows_executed, a.limit_warning, b.limit_warning)
    ok = got == w and np.array_equal(m, wm) and np.array_equal(c.left_len, wc.left_len) and np.array_equal(c.right_len, wc.right_len) \
        and np.array_equal(c.score, wc.score)
    print(f"round {rd}: W={W} L={L} cores={fs.cores.n} go={p.gapopen} ge={p.gapextn} stop={p.when_to_stop} packed={a.packed_rows}/{b.packed_rows} "
          f"rows={a.rows_executed}/{b.rows_executed} oracle {t1 - t0:.0f} s {'ok' if ok else 'MISMATCH ' + str((got, w))}", flush=True)
    bad += 0 if ok else 1
print(f"{rounds} rounds, {bad} mismatches")
sys.exit(1 if bad else 0)
