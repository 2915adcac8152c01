#!/bin/bash
for n in 1000 100000; do
  for w in 0 8 16 40 80 160; do
    python bench.py --steps 1 --warmup 1 --no-cpu --flanks $n --L 1200 --bandwidth $w 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('N', $n, 'W', $w, 'us/col', round(r['us_per_column'],2), 'GB/s', round(r['achieved']))
"
  done
done
