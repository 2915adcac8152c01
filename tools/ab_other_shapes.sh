for lib in repeatafterme_amd/libramx.so ab_tmp/libramx_g8l2.so; do
  echo "== $lib"
  RAMX_LIB=$lib python tools/bench_batch.py 500 40 2>&1 | head -1
  RAMX_LIB=$lib RAMX_NO_CP=1 python tools/bench_batch.py 500 40 2>&1 | head -1
  RAMX_LIB=$lib TIMING_W=80 python tools/cp_dev_timing.py 65000 2>&1 | tail -1
  RAMX_LIB=$lib TIMING_W=40 python tools/cp_dev_timing.py 65000 2>&1 | tail -1
  RAMX_LIB=$lib python bench.py --steps 2 --warmup 1 --no-cpu --no-seam1 --bandwidth 20 --L 3000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W=20 N=100k us/col %.3f' % d['roofline']['us_per_column'])"
done
