#!/usr/bin/env python3
"""Extend every family of a Stockholm file with ONE RAMExtend process per scoring group.

The reference's wrapper (util/extend-stk.pl) starts one RAMExtend per family (:349-364).  This driver performs the
same per-family preparation (repeatafterme_amd/stockholm.py), then hands all families that share a matrix /
-minimprovement to `RAMExtend -batch`, which extends them in one launch per direction.  Per family it leaves what the
wrapper's -onlyextend mode leaves (extend-stk.pl:391-427):

    <id>-linup.tsv  <id>-repam.log  <id>-repam-ranges.tsv  <id>-ext-cons.fa  <id>-repam-repseq.fa  <id>-combined-cons.fa

(the re-alignment and Stockholm rewriting that follow in the wrapper belong to RepeatModeler and are out of scope).
"""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from repeatafterme_amd import stockholm as stk          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_EXE = os.path.join(HERE, "..", "repeatafterme_amd", "RAMExtend")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-assembly", required=True, help="genome .2bit")
    ap.add_argument("-input", required=True, help="Stockholm file with one or more seed alignments")
    ap.add_argument("-outdir", default=".", help="where the per-family files go")
    ap.add_argument("-bandwidth", type=int, default=40)                 # extend-stk.pl:197
    ap.add_argument("-min_aligning_seqs", type=int, default=3)          # extend-stk.pl:192
    ap.add_argument("-L", type=int, default=20000)                      # extend-stk.pl:352
    ap.add_argument("-ramextend", default=DEFAULT_EXE)
    ap.add_argument("-one_by_one", action="store_true", help="start one RAMExtend per family, as the wrapper does")
    a = ap.parse_args(argv)

    seeds = stk.read_stockholm(a.input)
    if not seeds:
        sys.exit(f"No seed alignments found in {a.input}!")
    os.makedirs(a.outdir, exist_ok=True)
    groups = {}
    for seed in seeds:
        fid = seed.name
        print(f"Working on {fid}..")
        rows, extendable = stk.ranges_for(seed)
        tdiv = stk.family_divergence(seed)
        matrix, minimp = stk.choose_scoring(tdiv, a.min_aligning_seqs)
        base = os.path.join(a.outdir, fid)
        with open(base + "-linup.tsv", "w") as fh:
            for r in rows:
                fh.write("%s\t%d\t%d\t%d\t%d\t%s\n" % r)
        print(f"  - Consensus length [recalculated]: {len(stk.reference_sequence(seed))}")
        print(f"  - Divergence: {tdiv:.2f} %")
        print(f"  - Instances: {len(seed.rows)}")
        if extendable > 3:                                              # extend-stk.pl:351
            groups.setdefault((matrix, minimp), []).append((seed, base))
        else:
            print(f"  **Too few extendable sequences ({extendable}) for RAMExtend**")

    for (matrix, minimp), fams in groups.items():
        common = ["-twobit", a.assembly, "-L", str(a.L), "-bandwidth", str(a.bandwidth), "-matrix", matrix, "-vvv",
                  "-minimprovement", str(minimp)]
        print(f"  - Running RAMExtend [bandwidth={a.bandwidth}, matrix={matrix}, minimprovement={minimp}] on "
              f"{len(fams)} families..")
        if a.one_by_one:
            for seed, base in fams:
                with open(base + "-repam.log", "w") as log:
                    rc = subprocess.run([a.ramextend] + common + ["-ranges", base + "-linup.tsv", "-outtsv",
                                        base + "-repam-ranges.tsv", "-outfa", base + "-repam-repseq.fa", "-cons",
                                        base + "-ext-cons.fa"], stdout=log, stderr=subprocess.STDOUT).returncode
                if rc:
                    sys.exit(f"  RAMExtend failed! [{rc}] see {base}-repam.log")
        else:
            lst = os.path.join(a.outdir, f"batch-{matrix}-{minimp}.list")
            with open(lst, "w") as fh:
                for seed, base in fams:
                    fh.write("\t".join([base + "-linup.tsv", base + "-repam.log", base + "-ext-cons.fa",
                                        base + "-repam-ranges.tsv", base + "-repam-repseq.fa"]) + "\n")
            rc = subprocess.run([a.ramextend] + common + ["-batch", lst]).returncode
            if rc:
                sys.exit(f"  RAMExtend -batch failed! [{rc}]")
        for seed, base in fams:                                         # extend-stk.pl:397-417
            reference = stk.reference_sequence(seed)
            found_right = False
            out = [">combined\n"]
            if os.path.exists(base + "-ext-cons.fa"):
                for line in open(base + "-ext-cons.fa"):
                    if line.startswith(">left-extension"):
                        continue
                    if line.startswith(">right-extension"):
                        found_right = True
                        out.append(reference + "\n")
                        continue
                    out.append(line)
            if not found_right:
                out.append(reference + "\n")
            with open(base + "-combined-cons.fa", "w") as fh:
                fh.writelines(out)
            t = [l.strip() for l in open(base + "-repam.log") if l.startswith("Extended ")]
            print(f"RAMExtend Results [{seed.name}]: " + "  ".join(t))
    return 0


if __name__ == "__main__":
    sys.exit(main())
