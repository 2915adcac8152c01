"""us per column of one flank set through seam 1, device-wide: cell-parallel vs lane-per-flank persistent kernel, by N and W."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_alignment
from repeatafterme_amd.scoring import named_params
from repeatafterme_amd.synth import synth_family

L = 1000
sizes = [int(x) for x in sys.argv[1:]] or [1000, 2000, 4000, 8000, 16000, 32000, 65000]
for W in [int(x) for x in os.environ.get("TIMING_W", "40,80").split(",")]:
    for n in sizes:
        fs = synth_family(n, L, W, K=600, seed=5, core_len=2 * W + 4)
        p = named_params("14p43g" if W != 80 else "20p43g", bandwidth=W, L=L)
        out = []
        for nocp in (False, True):
            if nocp:
                os.environ["RAMX_NO_CP_DEVICE"] = "1"
            else:
                os.environ.pop("RAMX_NO_CP_DEVICE", None)
            best = None
            for rep in range(3):
                c = fs.cores.copy(); m = new_master(L)
                r = extend_alignment(1, c, fs.sequence, m, p)
                us = 1e3 * r.loop_ms / max(r.rows_executed, 1)
                best = us if best is None else min(best, us)
            out.append((r.lanes_per_flank, r.persistent, best, r.rows_executed))
        print(f"W {W:3d} N {n:6d}: cp K={out[0][0]:2d} {out[0][2]:6.2f} us/col | lane-per-flank (persistent={out[1][1]}) {out[1][2]:6.2f} us/col"
              f" | rows {out[0][3]} | x{out[1][2] / out[0][2]:.2f}", flush=True)
