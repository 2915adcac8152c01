#!/bin/bash
# A/B on ONE box: every ab_tmp/libramx_*.so is copied over the package's library and timed on the same points.
# usage: NS="65536 100000" WS="40" [BATCH=1] bash tools/ab_libs.sh
cp repeatafterme_amd/libramx.so /tmp/libramx_keep.so
for rep in 1 2; do
for lib in ab_tmp/libramx_*.so; do
  cp $lib repeatafterme_amd/libramx.so
  for n in ${NS:-100000}; do
    for w in ${WS:-40}; do
      python bench.py --steps 2 --warmup 1 --no-cpu --flanks $n --L 3000 --bandwidth $w 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$lib', 'N', $n, 'W', $w, 'us/col', round(r['us_per_column'],2))
"
    done
  done
  if [ -n "$BATCH" ]; then echo "$lib $(python tools/bench_batch.py 500 2>&1 | head -1)"; fi
done
done
cp /tmp/libramx_keep.so repeatafterme_amd/libramx.so
