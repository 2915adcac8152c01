#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per dispatch of the two loop kernels -> profiles-ready JSON.
usage: pmc_summary.py <dir> <tag> [columns_per_persistent_launch]"""
import csv, glob, json, os, sys
out_dir, tag = sys.argv[1], sys.argv[2]
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 256
res = {"tag": tag}
for key, pat in (("column", "ramx_column_kernel<false"), ("persistent", "ramx_persistent_kernel")):
    acc, cnt = {}, {}
    for f in glob.glob(os.path.join(out_dir, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat not in row.get("Kernel_Name", ""):
                    continue
                name, val = row["Counter_Name"], float(row["Counter_Value"])
                acc[name] = acc.get(name, 0.0) + val
                cnt[name] = cnt.get(name, 0) + 1
    if not acc:
        continue
    avg = {k: acc[k] / cnt[k] for k in acc}
    ent = {"dispatches": max(cnt.values()), "avg_per_dispatch": avg}
    # HBM traffic, corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE (KiB, tallied at
    # 64 B per 128 B request) reads 1/2 of wide coalesced loads -> x2; WRITE_SIZE (KiB) is exact.
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        ent["fetch_bytes_corrected"] = avg["FETCH_SIZE"] * 1024 * 2
        ent["write_bytes"] = avg["WRITE_SIZE"] * 1024
        ent["hbm_bytes_per_launch"] = ent["fetch_bytes_corrected"] + ent["write_bytes"]
        ent["hbm_bytes_per_column"] = ent["hbm_bytes_per_launch"] / (cols if key == "persistent" else 1)
    res[key] = ent
json.dump(res, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
