#!/bin/bash
# persistent-kernel model: us/column over N and W  (fixed part + per-band-step part)
for n in ${NS:-1000 16384 65536 100000 131072}; do
  for w in 14 20 40; do
    python bench.py --steps 2 --warmup 1 --no-cpu --flanks $n --L 3000 --bandwidth $w 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('N', $n, 'W', $w, r['kernel'][:24], 'us/col', round(r['us_per_column'],2), 'alg GB/s', round(r['achieved']))
"
  done
done
