#!/bin/bash
# usage: tools/sweep.sh  -- quick sensitivity sweep of the column kernel (block size, N)
for blk in 64 128 256; do
  for n in 25000 50000 100000 200000; do
    RAMX_BLOCK=$blk python bench.py --steps 1 --warmup 1 --no-cpu --flanks $n --L 2000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('block', $blk, 'N', $n, 'us/col', round(r['us_per_column'],2), 'GB/s', round(r['achieved']), 'frac', round(r['frac'],3))
"
  done
done
