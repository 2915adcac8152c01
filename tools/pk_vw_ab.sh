#!/bin/bash
# A/B of the 320-thread shape (four band waves + a vote wave without flanks) against four band waves alone (RAMX_PK_NO_VW=1), packed rows
# forced (RAMX_NO_CP_DEVICE=1), by flank count: aligned phase (L 1500) and whole bench launch.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT
export RAMX_NO_CP_DEVICE=1
for n in ${NS:-25000 50000 65000}; do
for v in "RAMX_DUMMY=1" "RAMX_PK_NO_VW=1"; do
for L in 1500 10000; do
  env $v python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --no-phases --flanks $n --L $L 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N', $n, 'L', $L, '[$v] us/col %.3f digest %s' % (d['ms_per_step'] * 1e3 / d['config']['columns_per_step'], d['checks'].get('equals_reference_digest')))"
done; done; done
W80=1
env RAMX_DUMMY=1 TIMING_W=80 python tools/cp_dev_timing.py 50000 2>/dev/null | tail -1
