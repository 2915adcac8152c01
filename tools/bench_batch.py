"""Batch mode timing: F families of ~100 flanks (the cfg5-like workload), one launch vs one family at a time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment, extend_batch
from repeatafterme_amd.scoring import named_params
from repeatafterme_amd.synth import synth_family

F, W, L = int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 40, 1200
p = named_params("14p43g" if W != 80 else "20p43g", bandwidth=W, L=L)
fams = [synth_family(int(60 + (i * 37) % 90), L, W, K=300 + (i * 53) % 500, seed=1000 + i) for i in range(F)]
cols = None
for rep in range(2):
    cs = [f.cores.copy() for f in fams]; ms = [new_master(L) for _ in fams]
    t0 = time.perf_counter()
    infos = extend_batch(1, [(c, f.sequence, m) for c, f, m in zip(cs, fams, ms)], p)
    tb = time.perf_counter() - t0
cols = sum(i.rows_executed for i in infos)
fbp = sum(i.rows_executed * i.n_extendable for i in infos)
print(f"batch: {F} families, {sum(f.cores.n for f in fams)} flanks, {cols} columns in {tb*1e3:.1f} ms wall "
      f"(kernel {infos[0].loop_ms:.1f} ms) -> {fbp/tb/1e6:.1f} M flank-bp/s, {cols/tb/1e3:.1f} k family-columns/s")
sub = fams[:50]
cs2 = [f.cores.copy() for f in sub]; ms2 = [new_master(L) for _ in sub]
t0 = time.perf_counter()
r = [extend_alignment(1, c, f.sequence, m, p) for c, f, m in zip(cs2, sub, ms2)]
t1 = time.perf_counter() - t0
print(f"one by one (first 50 families): {t1*1e3:.1f} ms wall -> {t1/50*1e3:.2f} ms per family; batch {tb/F*1e3:.3f} ms per family")
assert all(np.array_equal(a, b) for a, b in zip(ms[:50], ms2))
