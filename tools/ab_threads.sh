#!/bin/bash
# device-wide cell-parallel kernel: workgroup size of the vote-wave mode by flank count (one box)
for n in ${SIZES:-500 1000 2000 3000 4000}; do
  for th in 192 256 320 576; do
    echo -n "threads $th: "
    RAMX_CP_DEV_THREADS=$th SPEC_L=1400 timeout -k 5 120 python tools/cp_spec_timing.py $n 2>&1 | grep "us/col"
  done
done
