#!/bin/bash
# A/B of an environment switch: tools/ab_env.sh VAR
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" "$1=1"; do
  for n in 65536 100000 131072; do
    env $v python bench.py --steps 1 --warmup 1 --no-cpu --flanks $n --L 1500 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('[$v] N', $n, 'us/col', round(r['us_per_column'],2))
"
  done
done; done
