"""us per column and re-speculated rows of the device-wide cell-parallel kernel (vote-wave mode), by flank count:
python3 tools/cp_spec_timing.py [W] N1 N2 ...   (BASELINE config 2 = 1000; one eighth of configs 3/4 = 12500)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment
from repeatafterme_amd.scoring import named_params
from repeatafterme_amd.synth import synth_family

args = [int(x) for x in sys.argv[1:]]
W = 40
if args and args[0] in (14, 20, 40, 80):
    W = args.pop(0)
L = int(os.environ.get("SPEC_L", "2000"))
for n in (args or [1000, 12500]):
    fs = synth_family(n, L, W, K=1500, seed=3)
    p = named_params("14p43g" if W != 80 else "20p43g", bandwidth=W, L=L, when_to_stop=L)
    best = None
    for rep in range(4):
        c = fs.cores.copy(); m = new_master(L)
        r = extend_alignment(1, c, fs.sequence, m, p)
        us = 1e3 * r.loop_ms / max(r.rows_executed, 1)
        best = us if best is None else min(best, us)
    print(f"W {W} N {n:6d}: K={r.lanes_per_flank:2d} persistent={r.persistent} {best:6.3f} us/col  rows {r.rows_executed}  "
          f"re-speculated rows {r.respeculated_rows} ({100.0 * r.respeculated_rows / max(r.rows_executed, 1):.2f} %)", flush=True)
