#!/bin/bash
# instruction counts of the packed-row kernel per column (run ON the GPU box): two PMC passes over bench.py --L $L (default 1500: the aligned phase)
L=${L:-1500}; TAG=${1:-pkpmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O/pmc; mkdir -p $O/pmc
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc/pass$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-seam1 --L $L > $O/pmc.pass$i.log 2>&1 || echo "pass $i: non-zero exit"
done
python3 tools/pmc_summary2.py $O/pmc > $O/pmc_L$L.json; rm -rf $O/pmc $O/pmc.pass*.log
python3 - <<EOT
import json
d = json.load(open("$O/pmc_L$L.json"))
for k, e in d.items():
    if "packed" in k or "persistent" in k:
        a = e["avg_per_dispatch"]; n = e["dispatches"]
        print(k, "dispatches", n)
        for c in sorted(a): print("  %-24s %14.0f  per column %12.1f" % (c, a[c], a[c] / $L))
EOT
