#!/bin/bash
# Two ranks on ONE GPU (gloo for the host collectives, the vote through the mailboxes) against one rank with the same flanks per
# rank: what the cross-device step costs a column, for the cell-parallel route and for the lane-per-flank packed-row kernel.
# Run ON the GPU box.  25,000 flanks per rank: both ranks' launches are co-resident on the box's 256 CUs.
cd $GRAFT_REPO_ROOT
L=${L:-3000}; PER=${PER:-25000}
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s: n_gpus %d  flanks/rank %d  us/col %.3f  transport %s' % ('$1', d['n_gpus'], d['config']['flanks_this_rank'], d['ms_per_step'] * 1e3 / d['config']['columns_per_step'], d.get('transport')))"; }
for route in "cell-parallel" "packed rows"; do
  if [ "$route" = "packed rows" ]; then export RAMX_NO_CP_DEVICE=1; else unset RAMX_NO_CP_DEVICE; fi
  python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --L $L --flanks $PER 2>/dev/null | line "$route, one rank"
  RAMX_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu --L $L --flanks $((2 * PER)) 2>/dev/null | line "$route, two ranks on one GPU"
done
