#!/bin/bash
# phase breakdown of the persistent launch (run ON the GPU box; needs ab_tmp/libramx_T.so = tools/build_variant.sh T -DRAMX_PRK_TIMING):
# the aligned phase alone (L = 1,500: every wave FULL) and the whole bench run (L = 10,000), then the untimed library on both
for L in ${LS:-1500 10000}; do
  echo "== timing build, L $L"
  RAMX_LIB=ab_tmp/libramx_T.so python bench.py --steps 1 --warmup 0 --no-cpu --no-seam1 --L $L 2>&1 | grep -E "PRK_TIMING|PRK_LEANSTAT"
  echo "== product build, L $L"
  python bench.py --steps 3 --warmup 1 --no-cpu --no-seam1 --L $L 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.3f  us/col %.3f  digest %s' % (d['ms_per_step'], d['ms_per_step'] * 1e3 / d['config']['columns_per_step'], d['checks'].get('equals_reference_digest')))"
done
