#!/usr/bin/env python3
"""End-to-end RAMExtend at scale: write a synthetic genome (one 2bit record per flank) and a ranges file,
run the executable with RAMX_TIMING=1 and print the phase timings (load / extension / outputs).
    python tools/e2e_scale.py --n 100000 --L 10000 --bandwidth 40 [--both] [--keep DIR]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from repeatafterme_amd import _lib                                   # noqa: E402
from repeatafterme_amd.loader import write_ranges, write_twobit      # noqa: E402
from repeatafterme_amd.synth import synth_family                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--L", type=int, default=10000)
    ap.add_argument("--bandwidth", type=int, default=40)
    ap.add_argument("--K", type=int, default=1500)
    ap.add_argument("--both", action="store_true")
    ap.add_argument("--exe", default=_lib.CLI_PATH)
    ap.add_argument("--outputs", action="store_true", help="also write -cons/-outtsv/-outfa")
    ap.add_argument("--keep", default=None)
    a = ap.parse_args()
    d = a.keep or tempfile.mkdtemp(prefix="ramx_e2e_")
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    fs = synth_family(a.n, a.L, a.bandwidth, K=a.K, seed=1, both_sides=a.both, minus_frac=0.3)
    win = len(fs.sequence) // a.n
    seq = fs.sequence.reshape(a.n, win)
    c = fs.cores
    recs = [(f"s{i:06d}", seq[i]) for i in range(a.n)]
    rows = []
    for i in range(a.n):
        lo = int(min(c.left_pos[i], c.right_pos[i]) - c.lower[i])
        hi = int(max(c.left_pos[i], c.right_pos[i]) - c.lower[i]) + 1
        rows.append((f"s{i:06d}", lo, hi, int(c.left_ext[i]), int(c.right_ext[i]), "-" if c.orient[i] else "+"))
    write_twobit(os.path.join(d, "g.2bit"), recs)
    write_ranges(os.path.join(d, "g.tsv"), rows)
    print(f"generated {a.n} records x {win} bp in {time.time() - t0:.1f} s ({os.path.getsize(os.path.join(d, 'g.2bit')) / 1e6:.0f} MB 2bit)",
          flush=True)
    cmd = [a.exe, "-twobit", "g.2bit", "-ranges", "g.tsv", "-L", str(a.L), "-bandwidth", str(a.bandwidth), "-matrix", "14p43g",
           "-maxoccurrences", str(a.n)]
    if a.outputs:
        cmd += ["-cons", "cons.fa", "-outtsv", "out.tsv", "-outfa", "out.fa"]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=d, env=dict(os.environ, RAMX_TIMING="1"), stdout=open(os.path.join(d, "stdout.log"), "w"),
                       stderr=subprocess.PIPE, text=True)
    wall = time.time() - t0
    print(r.stderr, end="")
    print(f"exit {r.returncode}; wall {wall:.2f} s")
    for line in open(os.path.join(d, "stdout.log")):
        if line.startswith(("Extended", "Read in", "Program duration")):
            print(line, end="")
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
