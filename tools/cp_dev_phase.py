"""Phase timing of the device-wide cell-parallel mode on ONE family through the batch seam (multi-workgroup set):
run with RAMX_LIB=ab_tmp/libramx_T.so (tools/build_variant.sh T -DRAMX_CP_TIMING).  Args: W n [W n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.extend import extend_batch
from repeatafterme_amd.scoring import named_params
from repeatafterme_amd.synth import synth_family
L = 1500
args = [int(x) for x in sys.argv[1:]] or [40, 250, 40, 1000, 80, 250]
for W, n in zip(args[0::2], args[1::2]):
    fs = synth_family(n, L, W, K=1000, seed=5, core_len=2 * W + 4)
    p = named_params("14p43g" if W != 80 else "20p43g", bandwidth=W, L=L)
    for rep in range(2):
        sys.stderr.write(f"== W {W} n {n} rep {rep}\n"); sys.stderr.flush()
        res = extend_batch(1, [(fs.cores.copy(), fs.sequence, new_master(L))], p)
    print("W", W, "n", n, flush=True)
