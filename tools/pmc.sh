#!/bin/bash
# PMC counter passes for the column kernel (run ON the GPU box via gpurun).  Each --pmc set is its own
# run with no other tracing (the pool refuses pmc + sys-trace combinations).  Output: gpurun_out/pmc_<tag>/
TAG=${1:-r01}
ARGS=${2:-"--steps 1 --warmup 0 --no-cpu --L 256"}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" \
  "TCC_HIT_sum TCC_MISS_sum" \
  "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 bench.py $ARGS > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done: $set"
done
python3 tools/pmc_summary.py $OUT $TAG ${3:-256}
