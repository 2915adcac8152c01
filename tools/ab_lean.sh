#!/bin/bash
# the LEAN band and the early ticket of the lane-per-flank persistent kernel: bench workload with and without them (one box)
for rep in 1 2; do
  for env in "" "RAMX_NO_EARLY_TICKET=1" "RAMX_NO_LEAN=1"; do
    echo -n "${env:-lean + early ticket}: "
    env $env python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('us/col %.3f  value %.3f G flank-bp/s  digest ok %s' % (d['roofline']['us_per_column'], d['value'] / 1e9, d['checks'].get('equals_reference_digest')))"
  done
done
