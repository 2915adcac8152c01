"""Live CLI parity: the reference binary compiled by oracle/Makefile (oracle/_ref/RAMExtend_ref, shipped to the
GPU box as a built artefact) and our RAMExtend run on the same freshly generated genomes with randomised options;
stdout, -cons, -outtsv and -outfa must be byte-identical (version / duration lines aside)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd import _lib
from repeatafterme_amd.loader import write_ranges, write_twobit

from helpers import make_genome

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(po.REF_CLI), reason="oracle/_ref/RAMExtend_ref not present")]


def _norm(txt):
    return ["<v>" if l.startswith("RAMExtend Version") else "<t>" if l.startswith("Program duration is") else l
            for l in txt.splitlines()]


@pytest.mark.parametrize("seed", list(range(20, 32)))
def test_cli_live_against_reference_binary(seed, tmp_path):
    rng = np.random.default_rng(seed)
    recs, rows = make_genome(seed)
    write_twobit(str(tmp_path / "g.2bit"), recs)
    write_ranges(str(tmp_path / "g.tsv"), rows)
    matrix = ["14p43g", "18p43g", "20p43g", "25p43g", "repeatscout"][seed % 5]
    argv = ["-matrix", matrix, "-bandwidth", str([14, 40, 20, 7, 3, 60][seed % 6]), "-L", str(int(rng.integers(60, 400))),
            "-stopafter", str(int(rng.integers(5, 120)))]
    if seed % 3 == 0:
        argv += ["-vvv"]
    if seed % 4 == 1:
        argv += ["-addflanking", str(int(rng.integers(1, 30)))]
    if seed % 5 == 2 and matrix != "repeatscout":
        argv += ["-gapopen", "-22", "-gapext", "-4", "-minimprovement", "20", "-cappenalty", "-60"]
    outs = {}
    for tag, exe in (("ref", po.REF_CLI), ("ours", _lib.CLI_PATH)):
        d = tmp_path / tag
        d.mkdir()
        cmd = [exe, "-twobit", "../g.2bit", "-ranges", "../g.tsv", "-cons", "cons", "-outtsv", "tsv", "-outfa", "fa"] + argv
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, (tag, r.stderr)
        outs[tag] = (_norm(r.stdout), {f: (open(d / f).read() if os.path.exists(d / f) else None) for f in ("cons", "tsv", "fa")})
    assert outs["ref"][0] == outs["ours"][0], argv
    assert outs["ref"][1] == outs["ours"][1], argv
