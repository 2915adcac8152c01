"""`RAMExtend -batch` and tools/extend_stk.py: many families, one process, one launch per direction.  Every family's
log / -cons / -outtsv / -outfa must be byte-identical to what the reference binary (oracle/_ref/RAMExtend_ref) writes
when it is started for that family alone, which is how the reference's wrapper runs it (util/extend-stk.pl:349-364)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd import _lib
from repeatafterme_amd.loader import write_ranges, write_twobit

from helpers import make_genome

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(po.REF_CLI), reason="oracle/_ref/RAMExtend_ref not present")]


def _norm(txt):
    return ["<v>" if l.startswith("RAMExtend Version") else "<t>" if l.startswith("Program duration is") else l
            for l in txt.splitlines()]


def _families(tmp_path, seeds):
    """One 2bit holding the contigs of every family (prefixed f<k>_), one ranges file per family."""
    records, per_family = [], []
    for k, seed in enumerate(seeds):
        recs, rows = make_genome(seed)
        records += [(f"f{k}_{name}", seq) for name, seq in recs]
        per_family.append([(f"f{k}_{r[0]}",) + tuple(r[1:]) for r in rows])
    write_twobit(str(tmp_path / "all.2bit"), records)
    for k, rows in enumerate(per_family):
        write_ranges(str(tmp_path / f"fam{k}.tsv"), rows)
    return per_family


def _read(path):
    return open(path).read() if os.path.exists(path) else None


@pytest.mark.parametrize("bandwidth,matrix,extra", [
    (40, "14p43g", ["-vvv", "-minimprovement", "30"]),        # the wrapper's configuration: batch kernel
    (14, "25p43g", ["-stopafter", "20"]),                     # batch kernel, quiet
    (7, "20p43g", ["-vvv"]),                                  # width without a register-resident kernel: streaming family kernel
    (20, "repeatscout", ["-vvv", "-addflanking", "5"]),
])
def test_batch_cli_equals_reference_per_family(bandwidth, matrix, extra, tmp_path):
    seeds = list(range(40, 40 + 9))
    fams = _families(tmp_path, seeds)
    common = ["-twobit", "all.2bit", "-bandwidth", str(bandwidth), "-matrix", matrix, "-L", "300"] + extra
    with open(tmp_path / "batch.list", "w") as fh:
        fh.write("# ranges\tlog\tcons\ttsv\tfa\n\n")
        for k in range(len(fams)):
            cols = [f"fam{k}.tsv", f"ours{k}.log", f"ours{k}.cons", f"ours{k}.tsv", f"ours{k}.fa"]
            if k == 3:
                cols[2] = "-"                      # "-" = this output is not wanted
            if k == 4:
                cols = cols[:2]                    # trailing fields may be left out
            fh.write("\t".join(cols) + "\n")
    r = subprocess.run([_lib.CLI_PATH] + common + ["-batch", "batch.list"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"{len(fams)} families done" in r.stdout
    for k in range(len(fams)):
        ref = subprocess.run([po.REF_CLI] + common + ["-ranges", f"fam{k}.tsv", "-cons", f"ref{k}.cons", "-outtsv",
                             f"ref{k}.tsv", "-outfa", f"ref{k}.fa"], cwd=tmp_path, capture_output=True, text=True)
        assert ref.returncode == 0, ref.stderr
        assert _norm(ref.stdout) == _norm(open(tmp_path / f"ours{k}.log").read()), (k, common)
        for ext in ("cons", "tsv", "fa"):
            ours = _read(tmp_path / f"ours{k}.{ext}")
            if (k == 3 and ext == "cons") or k == 4:
                assert ours is None
            else:
                assert ours == _read(tmp_path / f"ref{k}.{ext}"), (k, ext, common)


def test_extend_stk_driver_equals_wrapper_commands(tmp_path):
    """Stockholm in -> per-family files out; each equals what the wrapper's own RAMExtend command line produces."""
    seeds = list(range(60, 66))
    fams = _families(tmp_path, seeds)
    mdiv = [12.5, 17.0, 20.25, 23.22, 12.5, 17.0]
    with open(tmp_path / "in.stk", "w") as fh:
        for k, rows in enumerate(fams):
            fh.write(f"# STOCKHOLM 1.0\n#=GF ID    fam{k}\n#=GF DE    Source:gsa, mDiv={mdiv[k]:.2f}, all.2bit:1\n")
            for (name, s, e, lf, rf, o) in rows:
                body = "ACGT" * 3
                fh.write(f"{name}:{s + 1}-{e}_{o} {'' if lf else '.' * 12}{body}{'' if rf else '.' * 12}\n")
            fh.write("//\n")
    r = subprocess.run([sys.executable, os.path.join(HERE, "..", "tools", "extend_stk.py"), "-assembly", "all.2bit",
                        "-input", "in.stk", "-outdir", "out", "-L", "400", "-bandwidth", "40"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    checked = 0
    for k, rows in enumerate(fams):
        base = f"out/fam{k}"
        assert _read(tmp_path / f"{base}-linup.tsv") == "".join("%s\t%d\t%d\t%d\t%d\t%s\n" % row for row in rows)
        if sum(1 for row in rows if row[3] or row[4]) <= 3:
            assert not os.path.exists(tmp_path / f"{base}-repam.log")
            continue
        matrix, minimp = {12.5: ("14p43g", 30), 17.0: ("18p43g", 30), 20.25: ("20p43g", 30), 23.22: ("25p43g", 27)}[mdiv[k]]
        # util/extend-stk.pl:352
        ref = subprocess.run([po.REF_CLI, "-twobit", "all.2bit", "-L", "400", "-bandwidth", "40", "-matrix", matrix,
                              "-ranges", f"{base}-linup.tsv", "-outtsv", f"ref{k}.tsv", "-outfa", f"ref{k}.fa", "-cons",
                              f"ref{k}.cons", "-vvv", "-minimprovement", str(minimp)], cwd=tmp_path, capture_output=True, text=True)
        assert ref.returncode == 0, ref.stderr
        assert _norm(ref.stdout) == _norm(_read(tmp_path / f"{base}-repam.log")), k
        assert _read(tmp_path / f"ref{k}.cons") == _read(tmp_path / f"{base}-ext-cons.fa")
        assert _read(tmp_path / f"ref{k}.tsv") == _read(tmp_path / f"{base}-repam-ranges.tsv")
        assert _read(tmp_path / f"ref{k}.fa") == _read(tmp_path / f"{base}-repam-repseq.fa")
        comb = _read(tmp_path / f"{base}-combined-cons.fa")
        assert comb.startswith(">combined\n") and "ACGTACGTACGT\n" in comb
        checked += 1
    assert checked >= 3
