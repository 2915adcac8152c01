"""Input surface on CPU: BED-6 + .2bit loader through the C-ABI (csrc/ramx_loader.c) against
(i) the window table the reference printed for the golden CLI cases and (ii) the reference loader
itself when oracle/_ref is present; plus format edge cases of our own writers/readers."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd import _lib
from repeatafterme_amd.loader import (_Core, _SeqLib, cores_from_list, flankset_from_c, load_sequence_subset_minimal,
                                      load_sequence_subset_packed, write_ranges, write_twobit)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
FLAGS = {"L_BOUNDARY": 0, "SEQ_BOUNDARY": 1, "CORE_BOUNDARY": 2, "EXT_BOUNDARY": 3}


def _golden_table(case):
    """Rows of the verbose core table the reference printed: n, ident, orient, l/r, seq[]-core, hard, soft, flags."""
    rows = []
    for line in open(os.path.join(G, "cli", case, "stdout")):
        m = re.match(r"^(\d+)\s+(\S+)\s+(\d+)-(\d+) ([+-])\s+(\d)/(\d) .*\]\s.*?\s(\d+)-(\d+) (\d+)-(\d+) (\d+)-(\d+) (\w+)/(\w+)$", line.rstrip("\n"))
        if m:
            rows.append(m.groups())
    return rows


@pytest.mark.parametrize("case,stem,maxflank", [("t2_w40_14p", "extension-test2", 10040), ("g0_w40_14p", "genome_0", 440),
                                                ("g3_w20_14p", "genome_3", 370), ("ov_default", "genome_ov", 214)])
def test_loader_matches_reference_table(case, stem, maxflank):
    fs = load_sequence_subset_minimal(os.path.join(G, "inputs", stem + ".2bit"), os.path.join(G, "inputs", stem + ".tsv"), maxflank)
    rows = _golden_table(case)
    assert len(rows) == fs.cores.n and len(rows) > 0
    c = fs.cores
    for i, (n, ident, bs, be, orient, le, re_, lp, rp, hlo, hhi, slo, shi, flo, fhi) in enumerate(rows):
        assert int(n) == i and fs.identifiers[c.seq_idx[i]].startswith(ident.rstrip("."))
        assert (int(lp), int(rp)) == (c.left_pos[i], c.right_pos[i])
        assert (orient == "-") == bool(c.orient[i]) and (int(le), int(re_)) == (c.left_ext[i], c.right_ext[i])
        lo = 0 if c.seq_idx[i] == 0 else int(fs.boundaries[c.seq_idx[i] - 1])
        assert (int(hlo), int(hhi)) == (lo, int(fs.boundaries[c.seq_idx[i]]) - 1)
        assert (int(slo), int(shi)) == (c.lower[i], c.upper[i])       # before the overlap pass
        assert (FLAGS[flo], FLAGS[fhi]) == (c.lower_flag[i], c.upper_flag[i])
        # BED range printed = offsets + core - window start
        core_lo = min(c.left_pos[i], c.right_pos[i]) - lo + int(fs.offsets[c.seq_idx[i]])
        assert int(bs) == core_lo


@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built here")
@pytest.mark.parametrize("stem,maxflank", [("extension-test2", 10014), ("genome_0", 60), ("genome_1", 314), ("genome_2", 2000),
                                           ("genome_3", 5), ("genome_ov", 150)])
@pytest.mark.parametrize("threads", ["1", "5"])
def test_loader_vs_reference_loader_live(stem, maxflank, threads, monkeypatch):
    monkeypatch.setenv("RAMX_LOADER_THREADS", threads)       # the decoding threads split the windows by bases
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libramref.so"))
    ref.loadSequenceSubsetMinimal.restype = C.POINTER(_SeqLib)
    ref.loadSequenceSubsetMinimal.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(_Core)), C.POINTER(C.c_int), C.c_int]
    ref.dnaUtilOpen()
    head = C.POINTER(_Core)()
    n = C.c_int()
    tb, bed = os.path.join(G, "inputs", stem + ".2bit"), os.path.join(G, "inputs", stem + ".tsv")
    lp = ref.loadSequenceSubsetMinimal(tb.encode(), bed.encode(), C.byref(head), C.byref(n), maxflank)
    want = flankset_from_c(lp, head, n.value)
    got = load_sequence_subset_minimal(tb, bed, maxflank)
    assert np.array_equal(want.sequence, got.sequence)
    assert np.array_equal(want.boundaries, got.boundaries) and np.array_equal(want.offsets, got.offsets)
    assert want.identifiers == got.identifiers
    for f in ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext", "right_ext", "seq_idx", "lower_flag", "upper_flag"):
        assert np.array_equal(getattr(want.cores, f), getattr(got.cores, f)), f


def test_twobit_roundtrip_n_blocks_and_versions(tmp_path):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 4, 1003).astype(np.int8)
    a[0:3] = 99; a[500:517] = 99; a[1000:1003] = 99
    b = rng.integers(0, 4, 77).astype(np.int8)
    tb = str(tmp_path / "x.2bit")
    write_twobit(tb, [("chrA", a), ("b", b)])
    bed = str(tmp_path / "x.tsv")
    write_ranges(bed, [("chrA", 400, 410, 1, 1, "+"), ("b", 30, 40, 1, 1, "-"), ("chrA", 2, 5, 1, 0, "+")])
    fs = load_sequence_subset_minimal(tb, bed, 1000)
    # sorted: chrA:2-5, chrA:400-410, b ... (strcmp order: "b" > "chrA"? 'b' < 'c' so b first)
    assert fs.identifiers == ["b", "chrA", "chrA"]
    w_b = fs.sequence[: int(fs.boundaries[0])]
    assert np.array_equal(w_b, b)                                   # both sides, whole record
    lo1, hi1 = int(fs.boundaries[0]), int(fs.boundaries[1])
    o1 = int(fs.offsets[1])
    assert np.array_equal(fs.sequence[lo1:hi1], a[o1:o1 + hi1 - lo1])
    # same content from a big-endian, version-1 file
    raw = open(tb, "rb").read()
    sig, ver, cnt, _ = struct.unpack("<IIII", raw[:16])
    out = struct.pack(">IIII", sig, 1, cnt, 0)
    pos = 16
    entries = []
    for _ in range(cnt):
        ln = raw[pos]; name = raw[pos + 1:pos + 1 + ln]; off = struct.unpack("<I", raw[pos + 1 + ln:pos + 5 + ln])[0]
        entries.append((name, off)); pos += 5 + ln
    delta = 4 * cnt
    for name, off in entries:
        out += bytes([len(name)]) + name + struct.pack(">Q", off + delta)
    body = bytearray()
    p = pos
    for _ in range(cnt):
        size, nb = struct.unpack("<II", raw[p:p + 8]); p += 8
        rec = struct.pack(">II", size, nb)
        for _k in range(2 * nb):
            rec += struct.pack(">I", struct.unpack("<I", raw[p:p + 4])[0]); p += 4
        mb = struct.unpack("<I", raw[p:p + 4])[0]; p += 4
        rec += struct.pack(">I", mb)
        rec += struct.pack(">I", 0); p += 4
        nbytes = (size + 3) // 4
        rec += raw[p:p + nbytes]; p += nbytes
        body += rec
    tb2 = str(tmp_path / "y.2bit")
    open(tb2, "wb").write(out + bytes(body))
    fs2 = load_sequence_subset_minimal(tb2, bed, 1000)
    assert np.array_equal(fs.sequence, fs2.sequence) and np.array_equal(fs.cores.left_pos, fs2.cores.left_pos)


def test_loader_error_paths(tmp_path):
    """Error behaviour: exit codes of the reference (1 for its own checks, 255 from kent's errAbort)."""
    tb = str(tmp_path / "x.2bit")
    write_twobit(tb, [("s", np.zeros(50, np.int8))])
    code = ("import sys; sys.path.insert(0, %r); from repeatafterme_amd.loader import load_sequence_subset_minimal as f; "
            "f(sys.argv[1], sys.argv[2], 10)" % ROOT)
    bad = str(tmp_path / "bad.tsv"); open(bad, "w").write("s\t1\t5\t1\t1\tx\n")
    r = subprocess.run(["python3", "-c", code, tb, bad], capture_output=True, text=True)
    assert r.returncode == 1 and "ranges file does not appear to be in the correct format" in r.stdout
    short = str(tmp_path / "short.tsv"); open(short, "w").write("s\t1\t5\t+\n")
    r = subprocess.run(["python3", "-c", code, tb, short], capture_output=True, text=True)
    assert r.returncode == 1
    missing = str(tmp_path / "m.tsv"); open(missing, "w").write("nope\t1\t5\t1\t1\t+\n")
    r = subprocess.run(["python3", "-c", code, tb, missing], capture_output=True, text=True)
    assert r.returncode == 255 and "nope is not in" in r.stderr
    notb = str(tmp_path / "n.2bit"); open(notb, "wb").write(b"hello world, not a 2bit")
    ok = str(tmp_path / "ok.tsv"); open(ok, "w").write("#comment\n\ns\t1\t5\t1\t1\t+\n")
    r = subprocess.run(["python3", "-c", code, notb, ok], capture_output=True, text=True)
    assert r.returncode == 255 and "valid twoBitSig" in r.stderr
    r = subprocess.run(["python3", "-c", code, tb, ok], capture_output=True, text=True)
    assert r.returncode == 0
    # a record header whose N-block count the file cannot hold (corrupt / truncated): an error message, not a crash
    raw = bytearray(open(tb, "rb").read())
    off = struct.unpack("<I", raw[16 + 1 + 1:16 + 1 + 1 + 4])[0]          # index entry: length byte, "s", offset
    raw[off + 4:off + 8] = struct.pack("<I", 0x7fffffff)
    corrupt = str(tmp_path / "c.2bit"); open(corrupt, "wb").write(bytes(raw))
    r = subprocess.run(["python3", "-c", code, corrupt, ok], capture_output=True, text=True)
    assert r.returncode == 255 and "truncated" in r.stderr, (r.returncode, r.stderr)
    cut = str(tmp_path / "cut.2bit"); open(cut, "wb").write(open(tb, "rb").read()[:off + 6])
    r = subprocess.run(["python3", "-c", code, cut, ok], capture_output=True, text=True)
    assert r.returncode == 255 and "truncated" in r.stderr, (r.returncode, r.stderr)


def test_many_n_blocks_per_record(tmp_path):
    """A record with hundreds of N blocks and windows all along it: the blocks that touch a window are found by a binary search
    (csrc/ramx_loader.c first_nblock), byte loader and packed loader agree with the plain array."""
    rng = np.random.default_rng(11)
    a = rng.integers(0, 4, 40000).astype(np.int8)
    starts = np.sort(rng.choice(np.arange(10, 39900, 50), 400, replace=False))
    for st in starts:
        a[st:st + int(rng.integers(1, 30))] = 99
    tb = str(tmp_path / "n.2bit"); write_twobit(tb, [("rec", a)])
    ranges = [("rec", int(x), int(x) + 20, 1, 1, "+" if i % 2 else "-") for i, x in enumerate(range(100, 39000, 700))]
    bed = str(tmp_path / "n.tsv"); write_ranges(bed, ranges)
    fs = load_sequence_subset_minimal(tb, bed, 150)
    fp, t = load_sequence_subset_packed(tb, bed, 150)
    assert np.array_equal(fs.sequence, fp.sequence) and fs.cores.n == len(ranges)
    isn = np.zeros(len(fs.sequence), bool)
    for s0, ln in zip(t["n_start"], t["n_len"]):
        isn[int(s0):int(s0) + int(ln)] = True
    assert np.array_equal(isn, fs.sequence == 99) and isn.sum() > 100
    for i in range(fs.cores.n):
        lo = 0 if i == 0 else int(fs.boundaries[i - 1])
        hi = int(fs.boundaries[i]); o = int(fs.offsets[i])
        assert np.array_equal(fs.sequence[lo:hi], a[o:o + hi - lo]), i


def test_overlap_avoidance_vs_quadratic_restatement():
    """ramx_overlap_avoidance (bucketed) == the reference's O(N^2) double loop (ram_extend.c:445-499)."""
    L = _lib.lib()
    L.ramx_overlap_avoidance.argtypes = [C.POINTER(_Core), C.POINTER(_SeqLib)]
    rng = np.random.default_rng(9)
    for trial in range(30):
        n = int(rng.integers(1, 14))
        nseq = n
        idents = [b"chr%d" % int(rng.integers(0, 3)) for _ in range(nseq)]
        sizes = rng.integers(50, 120, nseq)
        bounds = np.concatenate((np.cumsum(sizes), [0])).astype(np.uint64)
        offs = np.concatenate((rng.integers(0, 300, nseq), [0])).astype(np.uint64)
        arr = (_Core * n)()
        for i in range(n):
            lo = 0 if i == 0 else int(bounds[i - 1])
            a = lo + int(rng.integers(10, 30)); b = a + int(rng.integers(2, 12))
            c = arr[i]
            c.next = C.pointer(arr[i + 1]) if i + 1 < n else None
            c.seqIdx = i
            c.orient = bytes([int(rng.random() < 0.5)])
            if c.orient[0]:
                c.leftSeqPos, c.rightSeqPos = b, a
            else:
                c.leftSeqPos, c.rightSeqPos = a, b
            c.lowerSeqBound, c.upperSeqBound = lo, int(bounds[i]) - 1
            c.rightExtensionLen = int(rng.integers(0, 25))
            c.leftExtendable = c.rightExtendable = b"\x01"
        model = [(int(c.leftSeqPos), int(c.rightSeqPos), int(c.lowerSeqBound), int(c.upperSeqBound), c.orient[0],
                  c.rightExtensionLen) for c in arr]
        lowers = [m[2] for m in model]; uppers = [m[3] for m in model]; flags = [[0, 0] for _ in model]
        for s in range(n):                                   # restatement of the reference double loop
            lps, rps, _, _, os_, rl = model[s]
            slo = 0 if s == 0 else int(bounds[s - 1])
            ext = rps - rl if os_ else rps + rl
            g = int(offs[s]) + (ext - slo + 1)
            for r in range(n):
                if idents[s] == idents[r] and g > int(offs[r]):
                    rlo = 0 if r == 0 else int(bounds[r - 1])
                    p = rlo + (g - int(offs[r]))
                    if model[r][4]:
                        if model[r][0] <= p <= uppers[r]:
                            uppers[r] = p; flags[r][1] = 3
                    else:
                        if lowers[r] <= p <= model[r][0]:
                            lowers[r] = p; flags[r][0] = 3
        id_arr = (C.c_char_p * (nseq + 1))(*(idents + [None]))
        sl = _SeqLib(None, id_arr, bounds.ctypes.data_as(C.POINTER(C.c_uint64)), offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                     int(bounds[nseq - 1]), nseq, 0, None)
        devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
        try:
            L.ramx_overlap_avoidance(arr, C.byref(sl))
            C.CDLL(None).fflush(None)
        finally:
            os.dup2(saved, 1); os.close(devnull); os.close(saved)
        assert [int(c.lowerSeqBound) for c in arr] == lowers and [int(c.upperSeqBound) for c in arr] == uppers, trial
        assert [[c.lowerSeqBoundFlag, c.upperSeqBoundFlag] for c in arr] == flags


# ---- the windows kept packed (4 bases per byte, as in the .2bit file), never one byte per base ----------

@pytest.mark.parametrize("stem,maxflank", [("extension-test2", 10014), ("genome_0", 60), ("genome_1", 314), ("genome_2", 2000),
                                           ("genome_3", 5), ("genome_ov", 150)])
@pytest.mark.parametrize("threads", ["1", "5"])
def test_packed_loader_equals_byte_loader(stem, maxflank, threads, monkeypatch):
    """ramx_load_sequence_subset_packed against ramx_load_sequence_subset_minimal on the golden inputs (cores on both strands,
    several per record, N runs, windows that start at every phase of a packed byte): same cores / boundaries / offsets /
    identifiers, ramx_packed_decode gives back exactly the one-byte-per-base library, whole and in pieces; the packed
    payload is a quarter of it."""
    monkeypatch.setenv("RAMX_LOADER_THREADS", threads)
    tb, tsv = os.path.join(G, "inputs", stem + ".2bit"), os.path.join(G, "inputs", stem + ".tsv")
    a = load_sequence_subset_minimal(tb, tsv, maxflank)
    b, t = load_sequence_subset_packed(tb, tsv, maxflank)
    assert np.array_equal(a.sequence, b.sequence)
    assert np.array_equal(a.boundaries, b.boundaries) and np.array_equal(a.offsets, b.offsets) and a.identifiers == b.identifiers
    for f in ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext", "right_ext", "seq_idx", "lower_flag", "upper_flag"):
        assert np.array_equal(getattr(a.cores, f), getattr(b.cores, f)), f
    assert t["length"] == len(a.sequence) and t["n_windows"] == a.cores.n
    assert np.array_equal(t["win_start"][1:], a.boundaries[:a.cores.n])
    assert t["n_bytes"] <= len(a.sequence) // 4 + 2 * a.cores.n
    for frm, cnt, piece in t["pieces"]:
        assert np.array_equal(piece, a.sequence[frm:frm + cnt]), (frm, cnt)
    # every N of the library lies in exactly one run of the table, and nothing else does
    isn = np.zeros(len(a.sequence), bool)
    for s0, ln in zip(t["n_start"], t["n_len"]):
        assert not isn[int(s0):int(s0) + int(ln)].any()
        isn[int(s0):int(s0) + int(ln)] = True
    assert np.array_equal(isn, a.sequence == 99)


def test_packed_loader_synthetic_phases_and_n_runs(tmp_path):
    """Windows starting at every phase 0..3 of a packed byte, N runs that straddle window edges and byte edges, a record
    whose last byte is partial: packed == byte loader."""
    rng = np.random.default_rng(4)
    recs, rows = [], []
    for r in range(6):
        n = 997 + r
        codes = rng.integers(0, 4, size=n).astype(np.int8)
        for _ in range(5):
            s0 = int(rng.integers(0, n - 40)); codes[s0:s0 + int(rng.integers(1, 37))] = 99
        recs.append((f"rec{r}", codes))
        pos = 50 + r
        for k in range(4):
            rows.append((f"rec{r}", pos, pos + 9 + k, int(rng.integers(0, 2)), 1, "+-"[(r + k) & 1]))
            pos += 150 + k
    tb, tsv = str(tmp_path / "p.2bit"), str(tmp_path / "p.tsv")
    write_twobit(tb, recs); write_ranges(tsv, rows)
    for maxflank in (7, 61, 5000):
        a = load_sequence_subset_minimal(tb, tsv, maxflank)
        b, t = load_sequence_subset_packed(tb, tsv, maxflank)
        assert np.array_equal(a.sequence, b.sequence) and set(t["win_phase"].tolist()) == {0, 1, 2, 3}
        assert (a.sequence == 99).any()
        for frm, cnt, piece in t["pieces"]:
            assert np.array_equal(piece, a.sequence[frm:frm + cnt])
