"""Dev script: GPU path vs oracle on adversarial + synthetic sets (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from repeatafterme_amd.synth import synth_family, synth_adversarial
from repeatafterme_amd.datamodel import new_master, ExtendParams
from repeatafterme_amd.extend import extend_alignment

def to_ep(p):
    return ExtendParams(bandwidth=p.bandwidth, cappenalty=p.cappenalty, minimprovement=p.minimprovement, L=p.L,
                        when_to_stop=p.when_to_stop, l=p.l, gapopen=p.gapopen, gapextn=p.gapextn, matrix=p.matrix)

def compare(fs, p, tag, verbose=False):
    c1, c2 = fs.cores.copy(), fs.cores.copy()
    m1, m2 = new_master(p.L, p.l), new_master(p.L, p.l)
    ok_all = True
    for d in (1, 0):
        r1 = po.oracle_extend(d, c1, fs.sequence, m1, p, trace=True)
        r2 = extend_alignment(d, c2, fs.sequence, m2, to_ep(p))
        ok = (r1.ret == r2.ret and r1.rows_executed == r2.rows_executed and r1.limit_warning == r2.limit_warning
              and np.array_equal(m1, m2) and np.array_equal(c1.left_len, c2.left_len)
              and np.array_equal(c1.right_len, c2.right_len) and np.array_equal(c1.score, c2.score))
        if not ok or verbose:
            print(tag, "dir", d, "OK" if ok else "MISMATCH", "ret", r1.ret, r2.ret, "rows", r1.rows_executed, r2.rows_executed,
                  "warn", r1.limit_warning, r2.limit_warning, "master_eq", np.array_equal(m1, m2),
                  "len_eq", np.array_equal(c1.right_len, c2.right_len), np.array_equal(c1.left_len, c2.left_len),
                  "score_eq", np.array_equal(c1.score, c2.score))
            if not ok:
                diff = np.nonzero(m1 != m2)[0]
                print("   first master diffs at", diff[:10], "oracle", m1[diff[:10]], "gpu", m2[diff[:10]])
        ok_all &= ok
    return ok_all

bad = tot = 0
t0 = time.time()
for seed in range(24):
    fs = synth_adversarial(seed, lowercase=(seed % 4 == 0))
    for W in (0, 1, 3, 14, 40):
        for mat in ("14p43g", "25p43g", "repeatscout"):
            p = po.Params.named(mat, bandwidth=W, L=120 if seed % 2 else 40, when_to_stop=30)
            tot += 1
            if not compare(fs, p, f"adv seed={seed} W={W} {mat}"):
                bad += 1
                if bad > 5: sys.exit(1)
print("adversarial", tot, "cases", bad, "bad", round(time.time() - t0, 1), "s", flush=True)

fs = synth_family(1000, 2000, 40, K=1500, seed=1)
p = po.Params.named("14p43g", bandwidth=40, L=2000)
t0 = time.time()
c = fs.cores.copy(); m = new_master(p.L)
r = extend_alignment(1, c, fs.sequence, m, to_ep(p))
print("cfg2 gpu:", r, round(time.time() - t0, 2), "s", flush=True)
t0 = time.time()
print("cfg2 parity:", compare(fs, p, "cfg2", verbose=True), round(time.time() - t0, 1), "s", flush=True)

fs = synth_family(20000, 600, 40, K=300, seed=2, both_sides=True, minus_frac=0.3, n_run_frac=0.05)
p = po.Params.named("14p43g", bandwidth=40, L=600)
print("20k parity:", compare(fs, p, "20k", verbose=True), flush=True)
