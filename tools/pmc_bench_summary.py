#!/usr/bin/env python3
"""profiles/pmc_summary.json (what bench.py's roofline block reads) from the PMC passes of the WHOLE bench launch:
profiles/TAG_pmc_bench.json (tools/pmc_summary2.py format) -> the `persistent` entry, per-column figures = per-launch / COLS.
Since round 3 the kernel's work per column depends on the phase (full band while the flanks align, LEAN band behind the end
of the alignment), so the counters are taken over the bench's own 10,000-column launch, not over a 256-column one.
usage: pmc_bench_summary.py TAG COLS FLANKS BANDWIDTH"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cols, flanks, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
doc = json.load(open(os.path.join(root, "profiles", f"{tag}_pmc_bench.json")))
ent = None
for want in ("ramx_packed_kernel", "ramx_persistent_kernel"):      # the packed rows are the bench's kernel since round 4
    for k, v in doc.items():
        if want in k and ent is None:
            ent = v
assert ent is not None, list(doc)
a = ent["avg_per_dispatch"]
out = {"tag": tag, "persistent": {"dispatches": ent["dispatches"], "avg_per_dispatch": a, "columns_per_launch": cols}}
if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
    # gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE (KiB) x 2 for wide coalesced loads, WRITE_SIZE (KiB) exact
    p = out["persistent"]
    p["fetch_bytes_corrected"] = a["FETCH_SIZE"] * 1024 * 2
    p["write_bytes"] = a["WRITE_SIZE"] * 1024
    p["hbm_bytes_per_launch"] = p["fetch_bytes_corrected"] + p["write_bytes"]
    p["hbm_bytes_per_column"] = p["hbm_bytes_per_launch"] / cols
sys.path.insert(0, root)
from bench import device_source_sha16
out["device_source_sha16"] = device_source_sha16()      # bench.py uses these counters only for the sources they were collected on
out["config"] = {"flanks": flanks, "bandwidth": W, "L": cols,
                 "command": "python3 bench.py --steps 1 --warmup 0 --no-cpu --no-seam1   (one rocprofv3 --pmc pass per counter set)"}
dst = os.path.join(root, "profiles", "pmc_summary.json")
# what a column costs when every row is FULL: the same counters over the aligned phase alone (a launch over the first 1,500
# columns of the bench set, profiles/TAG_pmc_aligned.json)
al = os.path.join(root, "profiles", f"{tag}_pmc_aligned.json")
if os.path.exists(al):
    acols = int(sys.argv[5]) if len(sys.argv) > 5 else 1500
    for k, v in json.load(open(al)).items():
        if "ramx_packed_kernel" in k and "SQ_INSTS_VALU" in v["avg_per_dispatch"]:
            out["persistent"]["full_band_insts_per_column"] = v["avg_per_dispatch"]["SQ_INSTS_VALU"] / acols
            out["persistent"]["full_band_source"] = f"profiles/{tag}_pmc_aligned.json (SQ_INSTS_VALU of a {acols}-column launch over the aligned phase of the bench set: every row FULL)"
elif os.path.exists(dst):
    old = json.load(open(dst)).get("persistent", {})
    for k in ("full_band_insts_per_column", "full_band_source"):
        if k in old:
            out["persistent"][k] = old[k]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out["persistent"].get(k) for k in ("columns_per_launch", "hbm_bytes_per_launch", "hbm_bytes_per_column")}),
      "SQ_INSTS_VALU per column:", a.get("SQ_INSTS_VALU", 0) / cols)
