#!/bin/bash
# bench workload on the default library and every ab_tmp/libramx_*.so (one box)
for rep in 1 2; do
  for lib in repeatafterme_amd/libramx.so ab_tmp/libramx_*.so; do
    echo -n "$lib: "
    RAMX_LIB=$lib python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 ${BENCH_ARGS} 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('us/col %.3f  value %.3f G flank-bp/s  digest ok %s' % (d['roofline']['us_per_column'], d['value'] / 1e9, d['checks'].get('equals_reference_digest')))"
  done
done
