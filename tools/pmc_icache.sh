#!/bin/bash
# instruction-cache view of the persistent kernel (run ON the GPU box): separate --pmc passes, no tracing
OUT=gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu --L 256 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_icache/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "persistent" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
