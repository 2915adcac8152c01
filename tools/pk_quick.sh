#!/bin/bash
# quick check of a packed-kernel change ON the GPU box: the exactness tests of the persistent kernels, then the bench launch
# (aligned phase and whole run, digests)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py -m gpu -x -q 2>&1 | tail -3
for L in 1500 10000; do
python bench.py --steps 3 --warmup 1 --no-cpu --no-seam1 --L $L 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('L', $L, 'ms/step %.3f us/col %.3f digest %s' % (d['ms_per_step'], d['ms_per_step'] * 1e3 / d['config']['columns_per_step'], d['checks'].get('equals_reference_digest')))"
done
echo -n "RAMX_NO_PK_LOOK=1: "; RAMX_NO_PK_LOOK=1 python bench.py --steps 3 --warmup 1 --no-cpu --no-seam1 --L 10000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.3f us/col %.3f digest %s' % (d['ms_per_step'], d['ms_per_step'] * 1e3 / d['config']['columns_per_step'], d['checks'].get('equals_reference_digest')))"
