#!/bin/bash
# the whole bench launch (L = 10,000: aligned phase + capped tail) on the cell-parallel route and on the packed-row route, by flank
# count (what a rank of a strong-scaling run holds): run ON the GPU box
cd $GRAFT_REPO_ROOT
for n in ${NS:-12500 25000 50000 65000}; do
for v in "RAMX_DUMMY=1" "RAMX_NO_CP_DEVICE=1"; do
  env $v python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --flanks $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N', $n, '[$v] ms/step %.3f us/col %.3f' % (d['ms_per_step'], d['ms_per_step'] * 1e3 / d['config']['columns_per_step']), (d.get('phases') or {}).get('aligned_us_per_column'), (d.get('phases') or {}).get('tail_us_per_column'))"
done; done
