#!/bin/bash
# A/B of the batch path on one box: every ab_tmp/libramx_*.so, tools/bench_batch.py at W = 40, 14 and 80
cp repeatafterme_amd/libramx.so /tmp/libramx_keep.so
for rep in 1 2; do
for lib in ab_tmp/libramx_*.so; do
  cp $lib repeatafterme_amd/libramx.so
  for w in ${WS:-40 14 80}; do echo "$lib W $w $(python tools/bench_batch.py 500 $w 2>&1 | head -2 | tr '\n' ' ' | cut -c1-260)"; done
done
done
cp /tmp/libramx_keep.so repeatafterme_amd/libramx.so
