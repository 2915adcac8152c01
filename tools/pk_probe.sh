#!/bin/bash
# ablation: the packed-row kernel when nobody waits for the vote (ab_tmp/libramx_NW.so = tools/build_variant.sh NW -DPRK_PROBE_NO_WAIT;
# results are wrong by construction, only the time is read): every row FULL (RAMX_NO_LEAN=1) and every row LEAN
run() { python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --L ${L:-1500} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('us/col %.3f' % (d['ms_per_step'] * 1e3 / d['config']['columns_per_step']))"; }
echo -n "product, L 1500: "; run
echo -n "product, RAMX_NO_LEAN=1, L 1500: "; RAMX_NO_LEAN=1 run
echo -n "no wait, all FULL: "; RAMX_LIB=ab_tmp/libramx_NW.so RAMX_NO_LEAN=1 RAMX_BENCH_NOCHECK=1 run
echo -n "no wait, LEAN where possible: "; RAMX_LIB=ab_tmp/libramx_NW.so RAMX_BENCH_NOCHECK=1 run
