#!/bin/bash
# A/B of the packed kernel's launch shape at the bench size: two 256-thread workgroups per CU (default) against one 512-thread
# workgroup per CU (RAMX_PK_NO_256=1), aligned phase alone (L 1500) and the whole bench launch (L 10000); run ON the GPU box
cd $GRAFT_REPO_ROOT
env RAMX_TIMING=1 python bench.py --steps 1 --warmup 0 --no-cpu --no-seam1 --L 1500 2>&1 | grep -m1 'piece 0 (rows'
env RAMX_PK_NO_256=1 RAMX_TIMING=1 python bench.py --steps 1 --warmup 0 --no-cpu --no-seam1 --L 1500 2>&1 | grep -m1 'piece 0 (rows'
for rep in 1 2; do
for v in "RAMX_DUMMY=1" "RAMX_PK_NO_256=1"; do
  for L in 1500 10000; do
    env $v python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --L $L 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$v] L', $L, 'ms/step %.3f us/col %.3f digest %s' % (d['ms_per_step'], d['ms_per_step'] * 1e3 / d['config']['columns_per_step'], d['checks'].get('equals_reference_digest')))
"
  done
done; done
