#!/bin/bash
# device-wide cell-parallel kernel: lanes per flank x workgroup size at one flank count (one box)
n=${1:-12500}
for cfg in "16 576" "8 576" "8 512" "8 320" "4 320" "4 512" "4 256" "2 512"; do
  set -- $cfg
  echo -n "K<=$1 threads $2: "
  RAMX_CP_K=$1 RAMX_CP_DEV_THREADS=$2 SPEC_L=1400 timeout -k 5 120 python tools/cp_spec_timing.py $n 2>&1 | grep "us/col" || echo "-"
done
