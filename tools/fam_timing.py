"""Phase timing of the register-resident family kernel (needs a -DRAMX_PRK_TIMING build as repeatafterme_amd/libramx.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment
from repeatafterme_amd.scoring import named_params
from repeatafterme_amd.synth import synth_family
W = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
L = 1200
fs = synth_family(n, L, W, K=800, seed=5)
p = named_params("14p43g", bandwidth=W, L=L)
for rep in range(2):
    c = fs.cores.copy(); m = new_master(L)
    r = extend_alignment(1, c, fs.sequence, m, p)
print("rows", r.rows_executed, "loop_ms", r.loop_ms, "us/col", 1e3 * r.loop_ms / max(r.rows_executed, 1))
