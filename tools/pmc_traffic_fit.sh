#!/bin/bash
# HBM traffic of the persistent kernel at two launch lengths -> fixed part (rows in and out once) + bytes per column.
# Separate --pmc passes, no tracing (run ON the GPU box).
OUT=gpurun_out/pmc_fit
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 256 1024; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/L${L}_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu --L $L > $OUT/L${L}_$c.log 2>&1 || echo "L=$L $c: profiler exit status non-zero (CSV may still be complete)"
  done
done
python3 - <<'PY'
import csv, glob, json
def total(L, c):
    v = 0.0; n = 0
    for f in glob.glob(f"gpurun_out/pmc_fit/L{L}_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "persistent" in row.get("Kernel_Name", "") and row["Counter_Name"] == c:
                v += float(row["Counter_Value"]); n += 1
    return v / max(n, 1)
b = {}
for L in (256, 1024):
    b[L] = total(L, "FETCH_SIZE") * 1024 * 2 + total(L, "WRITE_SIZE") * 1024      # gfx950 correction as in pmc_summary.py
slope = (b[1024] - b[256]) / (1024 - 256)
fixed = b[256] - slope * 256
out = {"bytes_L256": b[256], "bytes_L1024": b[1024], "bytes_per_column": slope, "fixed_bytes_per_launch": fixed}
json.dump(out, open("gpurun_out/pmc_fit/fit.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
