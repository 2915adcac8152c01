#!/bin/bash
for blk in 64 128 256; do
  for n in 100000 65536 131072; do
    RAMX_BLOCK=$blk python bench.py --steps 1 --warmup 1 --no-cpu --flanks $n --L 1500 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('block', $blk, 'N', $n, 'us/col', round(r['us_per_column'],2), 'GB/s', round($n*1312.25/r['us_per_column']/1e3))
"
  done
done
