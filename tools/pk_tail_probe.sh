#!/bin/bash
# what gates the capped tail of the bench launch (run ON the GPU box; results of the probe builds are wrong by construction):
#   product | RAMX_LEADER_MAX=0 (the leader's wave runs the FULL band) | ab_tmp/libramx_NL.so (-DPKB_PROBE_NO_LEADER: no leader rows)
#   | ab_tmp/libramx_TT.so (-DRAMX_PRK_TIMING -DRAMX_PRK_TIMING_TAIL: phases of the second half of the columns)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --L 10000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.3f us/col %.3f' % (d['ms_per_step'], d['ms_per_step'] * 1e3 / d['config']['columns_per_step']))"; }
echo -n "product: "; run
echo -n "RAMX_LEADER_MAX=0: "; RAMX_LEADER_MAX=0 run
echo -n "no leader rows (probe): "; RAMX_LIB=ab_tmp/libramx_NL.so RAMX_BENCH_NOCHECK=1 run
echo -n "RAMX_NO_PK_SPEC=1: "; RAMX_NO_PK_SPEC=1 run
RAMX_LIB=ab_tmp/libramx_TT.so python bench.py --steps 1 --warmup 0 --no-cpu --no-seam1 --L 10000 2>&1 | grep -E "PRK_TIMING"
