#!/bin/bash
# A/B of ab_tmp/libramx_*.so on the cell-parallel workloads (one box): single families, batches
for rep in 1 2; do
for lib in ab_tmp/libramx_*.so; do
  echo "== $lib"
  RAMX_LIB=$lib python tools/cp_timing.py 2>&1 | grep -E "W  40 n  (100|250)|W  80 n  100"
  RAMX_LIB=$lib python tools/bench_batch.py 500 40 2>&1 | head -1
  RAMX_LIB=$lib python tools/bench_batch.py 500 80 2>&1 | head -1
done
done
