#!/bin/bash
# phase breakdown of the persistent kernel (needs ab_tmp/libramx_T.so = build with EXTRA=-DRAMX_PRK_TIMING)
cp repeatafterme_amd/libramx.so /tmp/libramx_keep.so
cp ab_tmp/libramx_T.so repeatafterme_amd/libramx.so
for n in ${NS:-100000}; do
  for w in ${WS:-40}; do
    echo "== N $n W $w"
    python bench.py --steps 1 --warmup 0 --no-cpu --flanks $n --L ${LCOLS:-3000} --bandwidth $w 2>&1 | grep -E "PRK_TIMING|PRK_LEANSTAT|us_per_column" | sed -e 's/.*"us_per_column": \([0-9.]*\).*/us_per_column \1/'
  done
done
cp /tmp/libramx_keep.so repeatafterme_amd/libramx.so
