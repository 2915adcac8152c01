#!/bin/bash
# rocprofv3 kernel statistics of ONE real run through the executable (100,000 ranges, default -stopafter 100): which kernels a
# real run spends its device time in.  Run ON the GPU box.  Output: gpurun_out/$TAG/kernel_stats_e2e.csv
TAG=${1:-e2e}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
python3 tools/e2e_scale.py --n 100000 --keep /tmp/ramx_e2e_keep > $O/e2e_plain.log 2>&1
cd /tmp/ramx_e2e_keep
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_e2e -- $GRAFT_REPO_ROOT/repeatafterme_amd/RAMExtend -twobit g.2bit -ranges g.tsv -L 10000 -bandwidth 40 -matrix 14p43g -maxoccurrences 100000 > $O/e2e_stdout.log 2> $O/e2e_stderr.log
echo "exit $?"
f=$(find $O/trace_e2e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_e2e.csv && cat $O/kernel_stats_e2e.csv
rm -rf $O/trace_e2e /tmp/ramx_e2e_keep
