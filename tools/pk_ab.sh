#!/bin/bash
# aligned phase (L = 1,500) and whole run on the product library and on every ab_tmp/libramx_*.so (one box; probes give wrong results)
for L in ${LS:-1500 10000}; do
for lib in repeatafterme_amd/libramx.so ab_tmp/libramx_*.so; do
  echo -n "L $L $lib: "
  RAMX_LIB=$lib python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --L $L 2>/dev/null | python -c "
import sys, json
try:
  d = json.loads(sys.stdin.read().strip().splitlines()[-1])
  print('us/col %.3f  digest %s' % (d['ms_per_step'] * 1e3 / d['config']['columns_per_step'], d['checks'].get('equals_reference_digest')))
except Exception as e: print('failed', e)"
done; done
