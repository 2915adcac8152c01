#!/bin/bash
# Round profile (run ON the GPU box): bench line, rocprofv3 kernel-trace stats of the same command, PMC passes.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err && tail -c 2500 gpurun_out/$TAG/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/$TAG/trace_bench.json 2> gpurun_out/$TAG/trace.err
f=$(find gpurun_out/$TAG/trace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/$TAG/kernel_stats.csv; cat gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/trace
tools/pmc.sh $TAG "--steps 1 --warmup 0 --no-cpu --L 256" > gpurun_out/$TAG/pmc.log 2>&1
cp gpurun_out/pmc_$TAG/summary.json gpurun_out/$TAG/pmc_summary.json; rm -rf gpurun_out/pmc_$TAG
tail -12 gpurun_out/$TAG/pmc.log
