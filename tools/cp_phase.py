"""Phase timing of the cell-parallel family kernel: run with RAMX_LIB=ab_tmp/libramx_T.so (tools/build_variant.sh T -DRAMX_CP_TIMING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment
from repeatafterme_amd.scoring import named_params
from repeatafterme_amd.synth import synth_family
L = 1500
for W, n in [(40, 16), (40, 100), (80, 100), (40, 250)]:
    fs = synth_family(n, L, W, K=1000, seed=5, core_len=2 * W + 4)
    p = named_params("14p43g" if W != 80 else "20p43g", bandwidth=W, L=L)
    for rep in range(2):
        c = fs.cores.copy(); m = new_master(L)
        sys.stderr.write(f"== W {W} n {n} rep {rep}\n"); sys.stderr.flush()
        r = extend_alignment(1, c, fs.sequence, m, p)
    print("W", W, "n", n, "K", r.lanes_per_flank, "us/col", 1e3 * r.loop_ms / max(r.rows_executed, 1), flush=True)
