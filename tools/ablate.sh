#!/bin/bash
# timing ablations of the column kernel (results are wrong by construction; never shipped)
cd $GRAFT_REPO_ROOT
for v in "" "-DRAMX_DBG_NOSTORE" "-DRAMX_DBG_NOMEM" "-DRAMX_DBG_NOLDS" "-DRAMX_DBG_NOMEM -DRAMX_DBG_NOLDS"; do
  rm -f repeatafterme_amd/csrc/build/ramx_device.o
  make -s -C repeatafterme_amd/csrc ../libramx.so EXTRA="$v" 2>&1 | grep -E "error" 
  for n in 1000 65536 131072; do
    python bench.py --steps 1 --warmup 1 --no-cpu --flanks $n --L 800 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('variant [$v] N', $n, 'us/col', round(r['us_per_column'],2))
"
  done
done
rm -f repeatafterme_amd/csrc/build/ramx_device.o
