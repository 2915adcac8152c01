#!/bin/bash
# phase breakdown of the persistent kernel for every ab_tmp/libramx_*T.so (builds with EXTRA=-DRAMX_PRK_TIMING), bench workload
for lib in ab_tmp/libramx_*T.so; do
  echo "== $lib"
  RAMX_LIB=$lib python bench.py --steps 1 --warmup 0 --no-cpu --no-seam1 --L ${LCOLS:-10000} 2>&1 | grep -E "PRK_TIMING|PRK_LEANSTAT|us_per_column" | sed -e 's/.*"us_per_column": \([0-9.]*\).*/us_per_column \1/' | head -12
done
