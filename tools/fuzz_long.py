"""Randomised parity over LONG directions on the lane-per-flank route (the packed rows: rebase rows, the span check of every 16th row,
pieces, rows ahead of the vote): random scoring systems and stop parameters, 300-900 columns, against the oracle.
Usage: RAMX_NO_CP_DEVICE=1 RAMX_NO_FAMILY_ROUTE=1 python tools/fuzz_long.py [first] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment
from repeatafterme_amd.synth import synth_family
from helpers import to_extend_params
from test_gpu_fuzz import _random_params, _oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for rd in range(first, first + rounds):
    rng = np.random.default_rng(99000 + rd)
    W = [14, 20, 40, 80][rd % 4]
    L = int(rng.integers(300, 900))
    p = _random_params(rng, W, L)
    if rng.random() < 0.5:
        p.when_to_stop = int(rng.integers(20, 120))
    ep = to_extend_params(p)
    n = int(rng.integers(200, 2500))
    os.environ["RAMX_PK_SEGMENT"] = str(int(rng.choice([0, 64, 200, 2048])))
    fs = synth_family(n, L, W, K=int(rng.integers(50, L)), seed=98000 + rd, div=float(rng.uniform(0.0, 0.25)),
                      both_sides=bool(rng.random() < 0.5), minus_frac=float(rng.uniform(0, 0.5)), n_run_frac=float(rng.uniform(0, 0.2)),
                      core_len=int(rng.choice([12, 2 * W + 4])))
    w, wm, wc = _oracle(fs, p)
    c = fs.cores.copy(); m = new_master(L)
    a = extend_alignment(1, c, fs.sequence, m, ep)
    b = extend_alignment(0, c, fs.sequence, m, ep)
    got = (a.ret, b.ret, a.rows_executed, b.rows_executed, a.limit_warning, b.limit_warning)
    ok = got == w and np.array_equal(m, wm) and np.array_equal(c.left_len, wc.left_len) and np.array_equal(c.right_len, wc.right_len) \
        and np.array_equal(c.score, wc.score)
    print(f"round {rd}: W={W} L={L} cores={fs.cores.n} go={p.gapopen} ge={p.gapextn} stop={p.when_to_stop} seg={os.environ['RAMX_PK_SEGMENT']} "
          f"packed={a.packed_rows}/{b.packed_rows} twice={a.respeculated_rows}/{b.respeculated_rows} rows={a.rows_executed}/{b.rows_executed} "
          f"{'ok' if ok else 'MISMATCH ' + str((got, w))}", flush=True)
    bad += 0 if ok else 1
print(f"{rounds} rounds, {bad} mismatches")
sys.exit(1 if bad else 0)
