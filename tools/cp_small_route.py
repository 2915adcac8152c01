"""One family through seam 1: one workgroup (family kernel) or the device-wide mode, by family size (RAMX_CP_SINGLE_MAX)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment
from repeatafterme_amd.scoring import named_params
from repeatafterme_amd.synth import synth_family
L = 1500
for W in (40, 80, 14, 20):
    for n in (8, 16, 24, 32, 48, 64, 100, 128):
        fs = synth_family(n, L, W, K=1000, seed=5, core_len=2 * W + 4)
        p = named_params("14p43g" if W != 80 else "20p43g", bandwidth=W, L=L)
        out = []
        for smax in ("100000", "1"):
            os.environ["RAMX_CP_SINGLE_MAX"] = smax
            best = None
            for rep in range(3):
                c = fs.cores.copy(); m = new_master(L)
                r = extend_alignment(1, c, fs.sequence, m, p)
                us = 1e3 * r.loop_ms / max(r.rows_executed, 1)
                best = us if best is None else min(best, us)
            out.append((r.lanes_per_flank, r.persistent, best))
        print(f"W {W:3d} n {n:4d}: one workgroup K={out[0][0]:2d} (persistent={out[0][1]}) {out[0][2]:6.2f} us/col | device-wide K={out[1][0]:2d} (persistent={out[1][1]}) {out[1][2]:6.2f} us/col", flush=True)
