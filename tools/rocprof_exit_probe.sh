#!/bin/bash
# Which teardown makes a profiled process die inside exit()?  (run ON the GPU box)  Each case: rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exit_probe; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/probe_case.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
case = sys.argv[1]
if case == "torch_only":
    import torch
    x = torch.ones(1024, device="cuda"); print(float(x.sum()))
elif case == "ramx_create_destroy":
    from repeatafterme_amd.device import Device
    d = Device(0); d.close(); print("ok")
elif case == "ramx_no_torch_streaming" or case == "ramx_no_torch_persistent" or case == "ramx_no_torch_noclose":
    if case == "ramx_no_torch_streaming":
        os.environ["RAMX_NO_PERSISTENT"] = "1"
    import numpy as np
    from repeatafterme_amd.datamodel import ExtendParams, new_master
    from repeatafterme_amd.device import Device, resolve_flanks
    from repeatafterme_amd.scoring import get_matrix
    from repeatafterme_amd.synth import synth_family
    mat, go, ge = get_matrix("14p43g")
    p = ExtendParams(bandwidth=40, cappenalty=-90, minimprovement=27, L=200, when_to_stop=200, l=1, gapopen=go, gapextn=ge, matrix=mat)
    fs = synth_family(70000, 200, 40, K=100, seed=1)
    d = Device(0); d.load_library(fs.sequence)
    fl, idx = resolve_flanks(1, fs.cores, 40, 200)
    d.begin_direction(fl, p); i = d.run_direction(); print(i.persistent, i.rows_executed)
    if case != "ramx_no_torch_noclose":
        d.close()
PY
for c in torch_only ramx_create_destroy ramx_no_torch_streaming ramx_no_torch_persistent ramx_no_torch_noclose; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -- python3 /tmp/probe_case.py $c > $OUT/$c.out 2> $OUT/$c.err
  echo "$c: exit $? $(grep -c SIGSEGV $OUT/$c.err) sigsegv; $(tail -1 $OUT/$c.out)"
  rm -rf $OUT/$c
done
