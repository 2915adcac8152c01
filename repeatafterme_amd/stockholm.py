"""Seed alignments (Stockholm) -> RAMExtend ranges: the data format on the caller's side of the hot path.

Mirrors what the reference's wrapper does before it starts RAMExtend for a family (util/extend-stk.pl:242-371):
row names ``[assembly:]sequence:start-end[_orient]`` become BED-6 rows with the two "extendable" flags derived from
how close the aligned row comes to the alignment's edges, the substitution matrix is chosen from the family's
divergence and ``-minimprovement`` from ``min_aligning_seqs``.

The wrapper gets its Stockholm parser, consensus caller and Kimura divergence from RepeatModeler's Perl modules
(SeedAlignmentCollection / MultAln), which are NOT part of the reference repository.  Those three pieces are therefore
re-stated here from the file format and the published K2P formula; everything that the reference repository itself
holds (flag rule, start-1, matrix thresholds, the RAMExtend command line) follows extend-stk.pl line by line.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field

_NAME = re.compile(r"^(?:(\S+):)?([^:\s]+):(\d+)-(\d+)(?:_([+-]))?$")
# extend-stk.pl:330-342 -- "Tolerate up to 10bp from edge"
_LEFT_OK = re.compile(r"^[.]{0,10}[^.]")
_RIGHT_OK = re.compile(r"[^.][.]{0,10}$")


@dataclass
class SeedRow:
    assembly: str
    sequence_name: str
    start: int        # 1-based, as written in the file
    end: int
    orient: str
    aligned: str


@dataclass
class SeedAlignment:
    id: str = ""
    accession: str = ""
    description: str = ""
    rf: str = ""
    rows: list = field(default_factory=list)

    @property
    def name(self) -> str:
        """extend-stk.pl:247-253"""
        return self.id or self.accession or "Unnamed_Family"


def read_stockholm(path: str) -> list[SeedAlignment]:
    """All ``# STOCKHOLM 1.0 ... //`` records of a file.  Rows repeated in later blocks are concatenated."""
    out, cur, index = [], None, {}
    with open(path) as fh:
        for raw in fh:
            line = raw.rstrip("\r\n")
            if line.startswith("# STOCKHOLM"):
                cur, index = SeedAlignment(), {}
                continue
            if cur is None or not line.strip():
                continue
            if line.startswith("//"):
                out.append(cur)
                cur = None
                continue
            if line.startswith("#=GF"):
                parts = line.split(None, 2)
                tag, val = parts[1], (parts[2] if len(parts) > 2 else "")
                if tag == "ID":
                    cur.id = val.strip()
                elif tag == "AC":
                    cur.accession = val.strip()
                elif tag == "DE":
                    cur.description = (cur.description + " " + val.strip()).strip()
                continue
            if line.startswith("#=GC"):
                parts = line.split(None, 2)
                if len(parts) == 3 and parts[1] == "RF":
                    cur.rf += parts[2].strip()
                continue
            if line.startswith("#"):
                continue
            parts = line.split()
            if len(parts) != 2:
                continue
            name, aligned = parts
            m = _NAME.match(name)
            if not m:
                raise ValueError(f"{path}: cannot parse row name '{name}' ([assembly:]sequence:start-end[_orient])")
            if name in index:
                index[name].aligned += aligned
                continue
            start, end, orient = int(m.group(3)), int(m.group(4)), m.group(5)
            if orient is None:
                orient = "+"
                if start > end:
                    start, end, orient = end, start, "-"
            row = SeedRow(m.group(1) or "", m.group(2), start, end, orient, aligned)
            index[name] = row
            cur.rows.append(row)
    return out


def ranges_for(seed: SeedAlignment):
    """BED-6 rows of extend-stk.pl:303-347 and the number of rows extendable on at least one side."""
    rows, extendable = [], 0
    for r in seed.rows:
        if re.match(r"^gi\|\d+$", r.sequence_name):       # :308-311
            continue
        left = bool(_LEFT_OK.search(r.aligned))
        right = bool(_RIGHT_OK.search(r.aligned))
        # :330-346 -- NB the wrapper writes "1 0" when only the LEFT edge is reached and "0 1" when only the right one
        if left and right:
            flags = (1, 1)
        elif left:
            flags = (1, 0)
        elif right:
            flags = (0, 1)
        else:
            flags = (0, 0)
        extendable += 1 if (left or right) else 0
        rows.append((r.sequence_name, r.start - 1, r.end, flags[0], flags[1], r.orient))
    return rows, extendable


def majority_consensus(seed: SeedAlignment) -> str:
    """Plurality base per reference column (stand-in for MultAln::consensus, which the reference repository does not
    hold).  With an RF line the reference columns are the ones it marks (x); without one, every column in which bases
    outnumber the gaps of the rows spanning it."""
    width = max((len(r.aligned) for r in seed.rows), default=0)
    spans = []
    for r in seed.rows:
        body = r.aligned.rstrip(".")
        spans.append((len(r.aligned) - len(r.aligned.lstrip(".")), len(body)))
    marked = seed.rf if re.search("x", seed.rf or "", re.I) else None
    out = []
    for c in range(width):
        if marked is not None and (c >= len(marked) or marked[c] not in "xX"):
            continue
        cnt, gaps = {}, 0
        for r, (lo, hi) in zip(seed.rows, spans):
            if c < lo or c >= hi:
                continue
            ch = r.aligned[c].upper()
            if ch in ".-":
                gaps += 1
            else:
                cnt[ch] = cnt.get(ch, 0) + 1
        if not cnt or (marked is None and gaps > sum(cnt.values())):
            continue
        best = max(sorted(cnt), key=lambda k: cnt[k])
        out.append(best if best in "ACGT" else "N")
    return "".join(out)


def reference_sequence(seed: SeedAlignment) -> str:
    """extend-stk.pl:262-277: the RF line without gaps, or a called consensus when RF only marks columns with x."""
    rf = seed.rf
    if not rf or re.search("x", rf, re.I):
        return majority_consensus(seed)
    return re.sub(r"[-.]", "", rf)


def kimura_divergence(seed: SeedAlignment) -> float:
    """Average Kimura two-parameter distance (percent) of the rows to the column consensus, no CpG adjustment.
    K = -1/2 ln((1 - 2p - q) sqrt(1 - 2q)), p = transitions / sites, q = transversions / sites."""
    width = max((len(r.aligned) for r in seed.rows), default=0)
    marked = seed.rf if re.search("x", seed.rf or "", re.I) else None
    cons = []
    for c in range(width):
        cnt = {}
        if marked is not None and (c >= len(marked) or marked[c] not in "xX"):
            cons.append(None)          # insertion columns carry no reference base
            continue
        for r in seed.rows:
            ch = r.aligned[c].upper() if c < len(r.aligned) else "."
            if ch in "ACGT":
                cnt[ch] = cnt.get(ch, 0) + 1
        cons.append(max(sorted(cnt), key=lambda k: cnt[k]) if cnt else None)
    purine = set("AG")
    ks = []
    for r in seed.rows:
        ti = tv = sites = 0
        for c, ch in enumerate(r.aligned.upper()):
            if ch not in "ACGT" or cons[c] is None:
                continue
            sites += 1
            if ch != cons[c]:
                if (ch in purine) == (cons[c] in purine):
                    ti += 1
                else:
                    tv += 1
        if sites == 0:
            continue
        p, q = ti / sites, tv / sites
        a, b = 1 - 2 * p - q, 1 - 2 * q
        if a <= 0 or b <= 0:
            continue
        ks.append(-0.5 * math.log(a * math.sqrt(b)) * 100.0)
    return sum(ks) / len(ks) if ks else 0.0


def family_divergence(seed: SeedAlignment) -> float:
    """extend-stk.pl:279-289: 'mDiv=NN.NN' in the description wins over the computed divergence."""
    m = re.search(r"mDiv=(\d+\.\d+)", seed.description)
    return float(m.group(1)) if m else kimura_divergence(seed)


def choose_scoring(tdiv: float, min_aligning_seqs: int = 3):
    """(matrix name, -minimprovement) of extend-stk.pl:291-304."""
    div, diag = 14, 10
    if tdiv >= 16:
        div = 18
        if tdiv >= 19:
            div = 20
            if tdiv >= 22.5:
                div, diag = 25, 9
    return f"{div}p43g", min_aligning_seqs * diag
