#!/bin/bash
# Round-4 profile (run ON the GPU box): the bench line; rocprofv3 kernel-trace stats of the same command; PMC passes (one counter
# set per run, no tracing) over the bench's own launch and over the aligned phase alone (L = 1,500); the cell-parallel kernel at
# BASELINE config 2's N = 1,000.  Summaries go to gpurun_out/$TAG; tools/pmc_bench_summary.py then writes profiles/pmc_summary.json
# WITH the sha of the device sources (bench.py ignores a summary from other sources).
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
stats() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- "$@" > $O/trace_$name.out 2> $O/trace_$name.err
  echo "$name: exit $?"
  f=$(find $O/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$name.csv && head -5 $O/kernel_stats_$name.csv
  rm -rf $O/trace_$name
}
stats bench python3 bench.py --steps 3 --warmup 1 --no-cpu --no-seam1 --no-phases
stats cp_n1000 python3 tools/cp_spec_timing.py 1000
pmc() {  # name, command...
  local name=$1; shift
  local i=0
  for set in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
    "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/pmc_$name/pass$i -- "$@" > $O/pmc_$name.pass$i.log 2>&1 || echo "pmc $name pass $i: non-zero exit"
  done
  python3 tools/pmc_summary2.py $O/pmc_$name > $O/pmc_$name.json; rm -rf $O/pmc_$name $O/pmc_$name.pass*.log
  echo "pmc $name done"
}
pmc bench python3 bench.py --steps 1 --warmup 0 --no-cpu --no-seam1 --no-phases
pmc aligned python3 bench.py --steps 1 --warmup 0 --no-cpu --no-seam1 --no-phases --L 1500
ls $O
