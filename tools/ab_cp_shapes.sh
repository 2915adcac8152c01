#!/bin/bash
# cell-parallel shapes on the default library and every ab_tmp/libramx_*.so (one box): batch of 500 families at W = 40 and 80,
# device-wide at N = 1,000 and 12,500
for lib in repeatafterme_amd/libramx.so ab_tmp/libramx_*.so; do
  echo "== $lib"
  RAMX_LIB=$lib python tools/bench_batch.py 500 40 2>&1 | head -1
  RAMX_LIB=$lib python tools/bench_batch.py 500 80 2>&1 | head -1
  RAMX_LIB=$lib python tools/cp_spec_timing.py 1000 12500 2>&1 | grep -v "^$" | tail -4
done
