#!/usr/bin/env python3
"""profiles/pmc_configs.json from the per-kernel PMC summaries of tools/profile_r03.sh (tools/pmc_summary2.py output):
one entry per (kernel, flank count) with the wave64 VALU instructions per column that bench.py's roofline block needs.
usage: pmc_configs.py TAG NAME:FLANKS:COLUMNS_PER_DISPATCH ...   (reads profiles/TAG_pmc_NAME.json)"""
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
entries = []
for spec in sys.argv[2:]:
    name, flanks, cols = spec.split(":")
    doc = json.load(open(os.path.join(root, "profiles", f"{tag}_pmc_{name}.json")))
    for kernel, ent in doc.items():
        m = re.search(r"ramx_cp_kernelILi(\d+)ELi(\d+)ELb1", kernel) or re.search(r"ramx_cp_kernel<(\d+), (\d+), true", kernel)
        if not m:
            continue
        a = ent["avg_per_dispatch"]
        e = {"tag": f"{tag}_pmc_{name}", "kernel": kernel, "flanks_per_rank": int(flanks), "bandwidth": int(m.group(1)),
             "lanes_per_flank": int(m.group(2)), "persistent": 1, "ranks": 1, "columns_per_dispatch": int(cols),
             "dispatches": ent["dispatches"], "SQ_INSTS_VALU_per_column": a["SQ_INSTS_VALU"] / int(cols)}
        for k in ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES"):
            if k in a:
                e[k + "_per_dispatch"] = a[k]
        if "hbm_bytes_per_dispatch" in ent:
            e["hbm_bytes_per_column"] = ent["hbm_bytes_per_dispatch"] / int(cols)
        entries.append(e)
json.dump({"entries": entries}, open(os.path.join(root, "profiles", "pmc_configs.json"), "w"), indent=1)
print(json.dumps(entries, indent=1))
