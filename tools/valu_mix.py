#!/usr/bin/env python3
"""Instruction mix of a kernel's hot basic blocks, weighted with the issue costs measured by tools/microbench/valu_rate.hip
(gfx950, wave64, two or more waves per SIMD): 2 cycles per wave-instruction for the double-rate set, 4 for every other
VALU instruction.  usage: valu_mix.py <device .s file> <kernel symbol substring> [min block size]   -> JSON on stdout

Get the .s with:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Irepeatafterme_amd/csrc -S --cuda-device-only \
                        repeatafterme_amd/csrc/ramx_device.hip -o /tmp/ramx_device.s"""
import json, re, sys

# measured (profiles/r02_valu_rate.log): these VOP1/VOP2 encodings with VGPR / inline-constant / literal operands issue in 2 cycles
DOUBLE_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_ashrrev_i32",
               "v_lshrrev_b32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fma_f32", "v_fmac_f32", "v_max_i16", "v_add_u16", "v_max_f16",
               "v_not_b32"}


def cost(op, operands):
    base = op[:-4] if op.endswith("_e32") else op
    if base.endswith("_e64") or "_sdwa" in base or "_dpp" in base:
        return 4
    if base in DOUBLE_RATE:
        # an SGPR source makes the add a 4-cycle instruction (measured for v_add_u32 / v_max_i32)
        if re.search(r"\bs\d+\b|\bs\[\d+", operands):
            return 4
        return 2
    return 4


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_block = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(":") is False and ":" in l)
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur)); cur = []; name = m.group(1); continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t)
    blocks.append((name, cur))
    out = {"kernel": lines[start].split(":")[0], "blocks": []}
    for name, ins in blocks:
        valu = [x for x in ins if x.startswith("v_")]
        if len(valu) < min_block:
            continue
        cyc = 0
        n2 = 0
        hist = {}
        for x in valu:
            op, _, rest = x.partition(" ")
            c = cost(op, rest.split(";")[0])
            cyc += c
            n2 += c == 2
            key = op.replace("_e32", "")
            hist[key] = hist.get(key, 0) + 1
        out["blocks"].append({"label": name, "valu": len(valu), "salu": sum(1 for x in ins if x.startswith("s_") and not x.startswith("s_nop")),
                              "lds": sum(1 for x in ins if x.startswith("ds_")), "double_rate": n2, "full_rate": len(valu) - n2, "issue_cycles": cyc,
                              "cycles_per_valu": cyc / len(valu),
                              "top": sorted(hist.items(), key=lambda kv: -kv[1])[:12]})
    tot_v = sum(b["valu"] for b in out["blocks"]); tot_c = sum(b["issue_cycles"] for b in out["blocks"])
    out["hot_blocks_valu"] = tot_v
    out["hot_blocks_double_rate"] = sum(b["double_rate"] for b in out["blocks"])
    # measured ticks (s_memtime) per wave-instruction per SIMD with >= 2 waves per SIMD: profiles/r02_valu_rate.log
    out["measured_ticks"] = {"double_rate": 2.38, "full_rate": 4.3}
    out["hot_blocks_ticks_per_valu"] = (out["hot_blocks_double_rate"] * 2.38 + (tot_v - out["hot_blocks_double_rate"]) * 4.3) / max(tot_v, 1)
    out["hot_blocks_cycles_per_valu"] = tot_c / max(tot_v, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
