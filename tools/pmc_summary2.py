#!/usr/bin/env python3
"""Per-kernel totals and per-dispatch averages of rocprofv3 --pmc counter CSVs (all passes under one directory) -> JSON."""
import csv, glob, json, os, sys
out_dir = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out_dir, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "ramx" not in k:
                continue
            k = k.split("(")[0]
            name, val = row["Counter_Name"], float(row["Counter_Value"])
            e = acc.setdefault(k, {}).setdefault(name, [0.0, 0])
            e[0] += val; e[1] += 1
res = {}
for k, d in acc.items():
    ent = {"dispatches": max(v[1] for v in d.values()), "avg_per_dispatch": {n: v[0] / v[1] for n, v in d.items()}}
    a = ent["avg_per_dispatch"]
    # HBM traffic as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE (KiB) reads 1/2 of wide coalesced
    # loads -> x2; WRITE_SIZE (KiB) is exact
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        ent["hbm_bytes_per_dispatch"] = a["FETCH_SIZE"] * 2048 + a["WRITE_SIZE"] * 1024
    res[k] = ent
print(json.dumps(res, indent=1))
