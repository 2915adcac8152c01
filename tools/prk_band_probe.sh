#!/bin/bash
# column time of the persistent kernel with parts taken out (results wrong by construction), bench shape, L = 3000:
#   ab_tmp/libramx_nowait*.so (EXTRA=-DPRK_PROBE_NO_WAIT: nobody waits for the vote), *noband* (-DPRK_PROBE_NO_BAND: no band),
# all-LEAN columns (default) and all-full columns (RAMX_NO_LEAN=1); the normal library gives the real column time next to it
for lib in repeatafterme_amd/libramx.so ab_tmp/libramx_*.so; do
  for nl in 0 1; do
    echo -n "$lib lean_off=$nl: "
    if [ $nl = 1 ]; then export RAMX_NO_LEAN=1; else unset RAMX_NO_LEAN; fi
    RAMX_LIB=$lib python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --L ${LCOLS:-3000} 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('us/col %.3f' % d['roofline']['us_per_column'])"
  done
done
