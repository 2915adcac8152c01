#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" "-DRAMX_PLAIN_LDST"; do
  rm -f repeatafterme_amd/csrc/build/ramx_device.o
  make -s -C repeatafterme_amd/csrc ../libramx.so EXTRA="$v" 2>&1 | grep -E "error"
  for rep in 1 2; do
  for n in 1000 65536 100000 131072; do
    python bench.py --steps 1 --warmup 1 --no-cpu --flanks $n --L 1500 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('variant [$v] N', $n, 'us/col', round(r['us_per_column'],2))
"
  done; done
done
rm -f repeatafterme_amd/csrc/build/ramx_device.o; make -s -C repeatafterme_amd/csrc all
