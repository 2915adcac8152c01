#!/bin/bash
# A/B with ragged flanks (a fraction ends early): every ab_tmp/libramx_*.so on the same box
cp repeatafterme_amd/libramx.so /tmp/libramx_keep.so
for lib in ab_tmp/libramx_*.so; do
  cp $lib repeatafterme_amd/libramx.so
  for n in ${NS:-100000}; do for rg in ${RG:-0 0.001 0.05}; do
      python bench.py --steps 2 --warmup 1 --no-cpu --flanks $n --L 3000 --bandwidth 40 --ragged $rg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'N', $n, 'ragged', $rg, 'us/col', round(d['roofline']['us_per_column'],2))
"
  done; done
done
cp /tmp/libramx_keep.so repeatafterme_amd/libramx.so
