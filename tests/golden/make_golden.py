#!/usr/bin/env python3
"""Generates the committed golden fixtures from the COMPILED REFERENCE (oracle/_ref, built by
`make -C oracle ref` from /root/reference).  Run in the build container only:

    python tests/golden/make_golden.py

Outputs (all data, no reference source text):
  inputs/extension-test2.{2bit,tsv}   the reference's own test data files (test/extension-test2.*)
  inputs/genome_<k>.{2bit,tsv}        synthetic genomes written by repeatafterme_amd.loader writers
  cli/<case>/{stdout,cons,tsv,fa}     what RAMExtend_ref printed / wrote for each case in CLI_CASES
  api_vectors.npz                     in-memory flank sets + the reference's extend_alignment results
"""
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
from repeatafterme_amd.datamodel import new_master  # noqa: E402
from repeatafterme_amd.loader import write_ranges, write_twobit  # noqa: E402
from repeatafterme_amd.synth import synth_adversarial, synth_family  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_genome  # noqa: E402

REF = "/root/reference"

# (case name, input stem, extra argv)
CLI_CASES = [
    ("t2_default", "extension-test2", []),
    ("t2_w40_14p", "extension-test2", ["-bandwidth", "40", "-matrix", "14p43g", "-vvv"]),
    ("t2_w80_20p", "extension-test2", ["-bandwidth", "80", "-matrix", "20p43g"]),
    ("t2_w40_25p", "extension-test2", ["-bandwidth", "40", "-matrix", "25p43g", "-v"]),
    ("t2_w40_rs", "extension-test2", ["-bandwidth", "40", "-matrix", "repeatscout"]),
    ("t2_w40_18p_gap", "extension-test2", ["-bandwidth", "40", "-matrix", "18p43g", "-gapopen", "-20", "-gapext", "-3",
                                           "-minimprovement", "30", "-addflanking", "7"]),
    ("t2_L100", "extension-test2", ["-L", "100", "-bandwidth", "20", "-matrix", "14p43g", "-vvv"]),
    ("g0_default", "genome_0", []),
    ("g0_w40_14p", "genome_0", ["-bandwidth", "40", "-matrix", "14p43g", "-L", "400", "-vvv", "-minimprovement", "30"]),
    ("g1_w14_20p", "genome_1", ["-L", "300", "-stopafter", "40"]),
    ("g1_rs", "genome_1", ["-matrix", "repeatscout", "-match", "2", "-mismatch", "-2", "-gap", "-6", "-L", "300",
                           "-bandwidth", "10", "-v"]),
    ("g2_w40_25p", "genome_2", ["-bandwidth", "40", "-matrix", "25p43g", "-L", "500", "-addflanking", "25", "-vv"]),
    ("g2_w3", "genome_2", ["-bandwidth", "3", "-matrix", "18p43g", "-L", "250", "-cappenalty", "-40"]),
    ("g3_w20_14p", "genome_3", ["-bandwidth", "20", "-matrix", "14p43g", "-L", "350", "-vvv"]),
    ("ov_default", "genome_ov", ["-L", "200", "-vvv"]),
    ("ov_w40", "genome_ov", ["-L", "150", "-bandwidth", "40", "-matrix", "14p43g"]),
]


def run_cli(case, stem, argv):
    out = os.path.join(HERE, "cli", case)
    os.makedirs(out, exist_ok=True)
    cmd = [po.REF_CLI, "-twobit", f"inputs/{stem}.2bit", "-ranges", f"inputs/{stem}.tsv",
           "-cons", os.path.join(out, "cons"), "-outtsv", os.path.join(out, "tsv"), "-outfa", os.path.join(out, "fa")] + argv
    for f in ("cons", "tsv", "fa"):
        if os.path.exists(os.path.join(out, f)):
            os.remove(os.path.join(out, f))
    r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
    assert r.returncode == 0, (cmd, r.stderr)
    # make the recorded command line location-independent
    txt = r.stdout.replace(out + "/", "")
    open(os.path.join(out, "stdout"), "w").write(txt)
    open(os.path.join(out, "argv"), "w").write(" ".join(argv) + "\n")


# cases whose -outmat trace (one line per extendable core and executed row: path string, best score, offset, base;
# ram_extend.c:1122-1132) is kept as well, gzip-compressed
OUTMAT_CASES = ["t2_default", "t2_w40_14p", "t2_L100", "g0_w40_14p", "g1_rs", "g2_w3", "g3_w20_14p", "ov_w40"]


def outmat_traces():
    import gzip
    import tempfile
    for case, stem, argv in CLI_CASES:
        if case not in OUTMAT_CASES:
            continue
        with tempfile.TemporaryDirectory() as td:
            mat = os.path.join(td, "mat")
            cmd = [po.REF_CLI, "-twobit", f"inputs/{stem}.2bit", "-ranges", f"inputs/{stem}.tsv", "-outmat", mat] + argv
            r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
            assert r.returncode == 0, (cmd, r.stderr)
            data = open(mat, "rb").read()
        with gzip.GzipFile(os.path.join(HERE, "cli", case, "outmat.gz"), "wb", mtime=0) as fh:
            fh.write(data)
        print("outmat", case, data.count(b"\n"), "lines")


# -vvvv: the per-row lines of ram_extend.c:992-1090 and 1134-1214 (candidate passes, **OUT_OF_SEQ**, **CAPPED**, totals, the
# "Alignment Extension" block); the reference's whole stdout, gzip-compressed.  (case name, input stem, argv)
VERBOSE_CASES = [
    ("t2_vvvv", "extension-test2", ["-bandwidth", "5", "-L", "30", "-vvvv"]),
    ("g3_vvvv", "genome_3", ["-bandwidth", "20", "-matrix", "14p43g", "-L", "60", "-vvvv"]),
    ("ov_vvvv", "genome_ov", ["-L", "40", "-bandwidth", "14", "-vvvv", "-cappenalty", "-10"]),
    ("g1_vvvv_rs", "genome_1", ["-matrix", "repeatscout", "-match", "2", "-mismatch", "-2", "-gap", "-6", "-L", "50", "-bandwidth", "10",
                                "-vvvv"]),
    ("g2_vvvv_w3", "genome_2", ["-bandwidth", "3", "-matrix", "18p43g", "-L", "80", "-cappenalty", "-40", "-vvvv"]),
    # -vvvvv (VERBOSE 12): + the boundary rows (ram_extend.c:949-959), every cell of every candidate row (:1013-1024) and the
    # sequence around every core's edge (:1066, report.c printExtensionRegion)
    ("t2_v5", "extension-test2", ["-bandwidth", "10", "-L", "30", "-vvvvv"]),
    ("g3_v5", "genome_3", ["-bandwidth", "20", "-matrix", "14p43g", "-L", "40", "-vvvvv"]),
    ("ov_v5", "genome_ov", ["-L", "30", "-bandwidth", "14", "-vvvvv", "-cappenalty", "-10"]),
    # (bandwidth >= 10 in all of them: with a narrower band the ten bases printExtensionRegion shows ahead of a core of the FIRST
    # record reach in front of the library in the last rows, where the reference computes `lastAligned + j` in uint64 and reads
    # the byte in front of the sequence array -- undefined, not reproducible; ramx prints a blank there)
    ("g2_v5_w10", "genome_2", ["-bandwidth", "10", "-matrix", "18p43g", "-L", "30", "-cappenalty", "-40", "-vvvvv"]),
]


def verbose_traces():
    import gzip
    for case, stem, argv in VERBOSE_CASES:
        out = os.path.join(HERE, "cli", case)
        os.makedirs(out, exist_ok=True)
        cmd = [po.REF_CLI, "-twobit", f"inputs/{stem}.2bit", "-ranges", f"inputs/{stem}.tsv"] + argv
        r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
        assert r.returncode == 0, (cmd, r.stderr)
        with gzip.GzipFile(os.path.join(out, "vvvv.gz"), "wb", mtime=0) as fh:
            fh.write(r.stdout.encode())
        open(os.path.join(out, "argv"), "w").write(" ".join(argv) + "\n")
        open(os.path.join(out, "stem"), "w").write(stem + "\n")
        print("vvvv", case, r.stdout.count("\n"), "lines")


def api_vectors():
    """Reference results for in-memory sets (adversarial + two uniform families)."""
    data = {}
    cases = []
    k = 0
    for seed in range(12):
        fs = synth_adversarial(seed, lowercase=(seed % 4 == 0))
        for (W, mat, L, wts) in ((14, "20p43g", 120, 30), (40, "14p43g", 60, 100), (3, "repeatscout", 120, 25), (0, "25p43g", 50, 10)):
            cases.append((fs, W, mat, L, wts))
    fs = synth_family(300, 260, 20, K=180, seed=7, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    cases.append((fs, 20, "18p43g", 260, 100))
    fs = synth_family(64, 150, 40, K=120, seed=8, both_sides=True, minus_frac=0.5)
    cases.append((fs, 40, "14p43g", 150, 100))
    for (fs, W, mat, L, wts) in cases:
        p = po.Params.named(mat, bandwidth=W, L=L, when_to_stop=wts)
        c = fs.cores.copy()
        m = new_master(L)
        rr = po.ref_extend(1, c, fs.sequence, m, p, boundaries=fs.boundaries)
        rl = po.ref_extend(0, c, fs.sequence, m, p, boundaries=fs.boundaries)
        pre = f"c{k}_"
        data[pre + "sequence"] = fs.sequence
        for f in ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext", "right_ext", "seq_idx"):
            data[pre + f] = getattr(fs.cores, f)
        data[pre + "params"] = np.array([W, p.cappenalty, p.minimprovement, L, wts, p.gapopen, p.gapextn], np.int32)
        data[pre + "matrix_name"] = np.array(mat)
        data[pre + "matrix"] = p.matrix
        data[pre + "ret"] = np.array([rr.ret, rl.ret], np.int32)
        data[pre + "master"] = m
        data[pre + "left_len"] = c.left_len
        data[pre + "right_len"] = c.right_len
        data[pre + "score"] = c.score
        k += 1
    data["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "api_vectors.npz"), **data)
    print("api vectors:", k, "cases")


def main():
    assert po.have_ref() and os.path.exists(po.REF_CLI), "build oracle/_ref first (make -C oracle ref)"
    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    for ext in ("2bit", "tsv"):
        shutil.copyfile(f"{REF}/test/extension-test2.{ext}", os.path.join(HERE, "inputs", f"extension-test2.{ext}"))
        os.chmod(os.path.join(HERE, "inputs", f"extension-test2.{ext}"), 0o644)
    for k in range(4):
        recs, rows = make_genome(k)
        write_twobit(os.path.join(HERE, "inputs", f"genome_{k}.2bit"), recs)
        write_ranges(os.path.join(HERE, "inputs", f"genome_{k}.tsv"), rows)
    # nested / overlapping cores: the only layout where the overlap-avoidance pass (ram_extend.c:445-499) fires
    rng = np.random.default_rng(77)
    fam = rng.integers(0, 4, 400)
    ctg_a = rng.integers(0, 4, 1500)
    ctg_a[250:650] = fam
    ctg_a[900:1300] = np.where(rng.random(400) < 0.1, rng.integers(0, 4, 400), fam)
    ctg_b = rng.integers(0, 4, 900)
    ctg_b[200:600] = (3 - fam)[::-1]
    write_twobit(os.path.join(HERE, "inputs", "genome_ov.2bit"), [("ctgA", ctg_a), ("ctgB", ctg_b)])
    write_ranges(os.path.join(HERE, "inputs", "genome_ov.tsv"), [
        ("ctgA", 400, 520, 1, 1, "+"), ("ctgA", 450, 462, 1, 1, "-"), ("ctgA", 1050, 1170, 1, 1, "+"),
        ("ctgB", 280, 400, 1, 1, "-"), ("ctgA", 1100, 1112, 1, 0, "-"), ("ctgB", 300, 310, 0, 1, "+")])
    for case, stem, argv in CLI_CASES:
        run_cli(case, stem, argv)
        print("cli case", case)
    outmat_traces()
    verbose_traces()
    api_vectors()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "verbose":     # only the -vvvv traces (the other fixtures stay as they are)
        assert po.have_ref() and os.path.exists(po.REF_CLI), "build oracle/_ref first (make -C oracle ref)"
        verbose_traces()
    else:
        main()
