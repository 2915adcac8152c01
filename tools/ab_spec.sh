#!/bin/bash
# A/B of the vote-wave mode: default library and ab_tmp/libramx_*.so on one box; aligned phase (L = 1400 < K) and with the tail (L = 2000)
for rep in 1 2; do
for lib in repeatafterme_amd/libramx.so ab_tmp/libramx_*.so; do
  for L in ${SPEC_LS:-1400 2000}; do
    echo "== $lib L=$L $SPEC_ENV"
    env $SPEC_ENV RAMX_LIB=$lib SPEC_L=$L timeout -k 5 120 python tools/cp_spec_timing.py ${SPEC_SIZES:-1000 12500} 2>&1 | grep "us/col"
  done
done
done
