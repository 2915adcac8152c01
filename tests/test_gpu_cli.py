"""The RAMExtend executable (drop-in CLI seam) against what the reference binary printed and wrote:
stdout, -cons, -outtsv, -outfa, for the reference's own fixture and the synthetic genomes."""
import os
import subprocess

import pytest

from repeatafterme_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CASES = sorted(c for c in os.listdir(os.path.join(G, "cli")) if os.path.exists(os.path.join(G, "cli", c, "stdout")))
VERBOSE = sorted(c for c in os.listdir(os.path.join(G, "cli")) if os.path.exists(os.path.join(G, "cli", c, "vvvv.gz")))
STEM = {"t2": "extension-test2", "g0": "genome_0", "g1": "genome_1", "g2": "genome_2", "g3": "genome_3", "ov": "genome_ov"}


def _norm(txt):
    out = []
    for line in txt.splitlines():
        if line.startswith("RAMExtend Version"):
            line = "RAMExtend Version <v>"       # build id differs by design
        if line.startswith("Program duration is"):
            line = "Program duration is <t>"
        out.append(line)
    return out


@pytest.mark.parametrize("case", CASES)
def test_cli_matches_reference_outputs(case, tmp_path):
    argv = open(os.path.join(G, "cli", case, "argv")).read().split()
    stem = STEM[case.split("_")[0]]
    cmd = [_lib.CLI_PATH, "-twobit", f"inputs/{stem}.2bit", "-ranges", f"inputs/{stem}.tsv", "-cons", str(tmp_path / "cons"),
           "-outtsv", str(tmp_path / "tsv"), "-outfa", str(tmp_path / "fa")] + argv
    r = subprocess.run(cmd, cwd=G, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = r.stdout.replace(str(tmp_path) + "/", "")
    want = open(os.path.join(G, "cli", case, "stdout")).read()
    assert _norm(got) == _norm(want)
    for f in ("cons", "tsv", "fa"):
        ref_f = os.path.join(G, "cli", case, f)
        if os.path.exists(ref_f):
            assert open(tmp_path / f).read() == open(ref_f).read(), f
        else:
            assert not os.path.exists(tmp_path / f) or f != "cons"


OUTMAT = [c for c in CASES if os.path.exists(os.path.join(G, "cli", c, "outmat.gz"))]


@pytest.mark.parametrize("case", OUTMAT)
def test_cli_outmat_trace_matches_reference(case, tmp_path):
    """-outmat: one line per extendable core and executed row -- the path string of the band (which state holds each
    cell's score), the row's best score, its offset and the consensus base (ram_extend.c:1122-1132,
    bnw_extend.c:1027-1044) -- byte-identical to the reference binary's file, with the ordinary outputs unchanged."""
    import gzip
    argv = open(os.path.join(G, "cli", case, "argv")).read().split()
    stem = STEM[case.split("_")[0]]
    cmd = [_lib.CLI_PATH, "-twobit", f"inputs/{stem}.2bit", "-ranges", f"inputs/{stem}.tsv", "-cons", str(tmp_path / "cons"),
           "-outtsv", str(tmp_path / "tsv"), "-outfa", str(tmp_path / "fa"), "-outmat", str(tmp_path / "mat")] + argv
    r = subprocess.run(cmd, cwd=G, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = gzip.open(os.path.join(G, "cli", case, "outmat.gz"), "rb").read().decode().splitlines()
    got = open(tmp_path / "mat").read().splitlines()
    assert len(got) == len(want), (len(got), len(want))
    for k, (a, b) in enumerate(zip(got, want)):
        assert a == b, f"line {k}: {a!r} != {b!r}"
    for f in ("tsv", "fa"):
        ref_f = os.path.join(G, "cli", case, f)
        if os.path.exists(ref_f):
            assert open(tmp_path / f).read() == open(ref_f).read(), f


def test_cli_version_and_usage():
    r = subprocess.run([_lib.CLI_PATH, "-version"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("RAMExtend Version 0.0.7 - build ")
    r = subprocess.run([_lib.CLI_PATH, "-h"], capture_output=True, text=True)      # extend-stk.pl:202-208 probes this way
    assert r.returncode == 1 and "RAMExtend Version 0.0.7" in r.stdout and "Usage:" in r.stdout
    r = subprocess.run([_lib.CLI_PATH, "-ranges", "x", "-matrix", "nope", "-twobit", "y"], capture_output=True, text=True)
    assert r.returncode == 1


def test_cli_prefix_matching_quirks(tmp_path):
    """cmd_line_opts.c: the first argv entry having the option text as a prefix wins (-gap matches -gapopen)."""
    base = [_lib.CLI_PATH, "-twobit", "inputs/extension-test2.2bit", "-ranges", "inputs/extension-test2.tsv"]
    a = subprocess.run(base + ["-bandw", "40", "-matrix", "14p43g"], cwd=G, capture_output=True, text=True)
    assert a.returncode == 0 and "BANDWIDTH (bandwidth) 14" in a.stdout       # "-bandw" is NOT matched by "-bandwidth"
    b = subprocess.run(base + ["-bandwidthXYZ", "40", "-matrix", "14p43g"], cwd=G, capture_output=True, text=True)
    assert "BANDWIDTH (bandwidth) 40" in b.stdout
    c = subprocess.run(base + ["-matrix", "14p43g", "-gapopen", "-20", "-gapextn", "-3"], cwd=G, capture_output=True, text=True)
    assert "GAP_OPEN = -20" in c.stdout and "GAP_EXT = -3" in c.stdout
    d = subprocess.run(base + ["-matrix", "14p43g", "-gapopen", "-20"], cwd=G, capture_output=True, text=True)
    assert "GAP_OPEN = -33" in d.stdout                                       # both are required (ram_extend.c:292-293)


def test_extend_alignment_struct_entry():
    """Seam 1 with the reference's own struct layout: linked list, seqLib, scoringSystem (ram_extend.h:9-13)."""
    import ctypes as C
    import numpy as np
    from oracle import pyoracle as po
    from repeatafterme_amd.loader import _Core, _SeqLib
    from repeatafterme_amd.scoring import _Scoring
    from repeatafterme_amd.synth import synth_adversarial
    from repeatafterme_amd.datamodel import new_master
    L = _lib.lib()
    L.ramx_get_matrix.restype = C.POINTER(_Scoring); L.ramx_get_matrix.argtypes = [C.c_char_p]
    L.ramx_extend_alignment.restype = C.c_int
    L.ramx_extend_alignment.argtypes = [C.c_int, C.POINTER(_Core), C.c_void_p, C.POINTER(_SeqLib), C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Scoring), C.c_void_p]
    fs = synth_adversarial(33)
    c = fs.cores
    n = c.n
    arr = (_Core * n)()
    for i in range(n):
        a = arr[i]
        a.next = C.pointer(arr[i + 1]) if i + 1 < n else None
        a.seqIdx = int(c.seq_idx[i]); a.leftSeqPos = int(c.left_pos[i]); a.rightSeqPos = int(c.right_pos[i])
        a.leftExtendable = bytes([int(c.left_ext[i])]); a.rightExtendable = bytes([int(c.right_ext[i])])
        a.lowerSeqBound = int(c.lower[i]); a.upperSeqBound = int(c.upper[i]); a.orient = bytes([int(c.orient[i])])
    seq = np.ascontiguousarray(fs.sequence)
    sl = _SeqLib(seq.ctypes.data_as(C.POINTER(C.c_int8)), None, None, None, len(seq), 0, 0, None)
    sp = L.ramx_get_matrix(b"20p43g")
    Lx, W = 90, 14
    m = new_master(Lx)
    L.ramx_set_runtime(0, 25, 1)
    rr = L.ramx_extend_alignment(1, arr, None, C.byref(sl), m.ctypes.data, W, -90, 27, Lx, n, sp, None)
    rl = L.ramx_extend_alignment(0, arr, None, C.byref(sl), m.ctypes.data, W, -90, 27, Lx, n, sp, None)
    p = po.Params.named("20p43g", bandwidth=W, L=Lx, when_to_stop=25)
    c2 = c.copy(); m2 = new_master(Lx)
    o1 = po.oracle_extend(1, c2, fs.sequence, m2, p); o0 = po.oracle_extend(0, c2, fs.sequence, m2, p)
    assert (rr, rl) == (o1.ret, o0.ret) and np.array_equal(m, m2)
    assert [a.rightExtensionLen for a in arr] == c2.right_len.tolist()
    assert [a.leftExtensionLen for a in arr] == c2.left_len.tolist()
    assert [a.score for a in arr] == c2.score.tolist()


def test_packed_library_cache_forgets_a_freed_library(tmp_path):
    """the device copy of a packed library is keyed on the seqLib pointer.  Load, extend, free, then load ANOTHER genome
    (malloc hands out the same addresses again) and extend: the second result must be the second genome's -- equal to the
    oracle on the second library's own bases -- not an extension against the first library's bases still on the device."""
    import ctypes as C
    import numpy as np
    from oracle import pyoracle as po
    from repeatafterme_amd.loader import _Core, _SeqLib, _Packed, cores_from_list, write_twobit, write_ranges
    from repeatafterme_amd.scoring import _Scoring
    from repeatafterme_amd.datamodel import new_master
    L = _lib.lib()
    L.ramx_get_matrix.restype = C.POINTER(_Scoring); L.ramx_get_matrix.argtypes = [C.c_char_p]
    L.ramx_load_sequence_subset_packed.restype = C.POINTER(_SeqLib)
    L.ramx_load_sequence_subset_packed.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(_Core)), C.POINTER(C.c_int),
                                                   C.c_int, C.POINTER(C.POINTER(_Packed))]
    L.ramx_packed_decode.argtypes = [C.POINTER(_Packed), C.c_uint64, C.c_uint64, C.c_void_p]
    L.ramx_free_library.argtypes = [C.POINTER(_SeqLib), C.POINTER(_Core)]
    L.ramx_extend_alignment.restype = C.c_int
    L.ramx_extend_alignment.argtypes = [C.c_int, C.POINTER(_Core), C.c_void_p, C.POINTER(_SeqLib), C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Scoring), C.c_void_p]
    sp = L.ramx_get_matrix(b"14p43g")
    Lx, W = 120, 14
    L.ramx_set_runtime(0, 30, 1)
    rets = []
    for g in (0, 1):
        # two genomes of the SAME shape (same record names and lengths, same ranges, different bases): the second library's blocks are
        # of the same sizes as the first one's, which is what makes malloc reuse the addresses
        rng = np.random.default_rng(900 + g)
        anc = rng.integers(0, 4, size=260, dtype=np.int8)
        recs, rows = [], []
        for i in range(40):
            fl = anc.copy()
            sub = rng.random(260) < 0.12
            fl[sub] = (fl[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
            core = np.array([0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3], np.int8)
            recs.append((f"s{i:03d}", np.concatenate((core, fl, rng.integers(0, 4, size=60, dtype=np.int8)))))
            rows.append((f"s{i:03d}", 0, 12, 0, 1, "+"))
        tb, rg = str(tmp_path / f"g{g}.2bit"), str(tmp_path / f"g{g}.tsv")
        write_twobit(tb, recs); write_ranges(rg, rows)
        head = C.POINTER(_Core)(); n = C.c_int(); pk = C.POINTER(_Packed)()
        lp = L.ramx_load_sequence_subset_packed(tb.encode(), rg.encode(), C.byref(head), C.byref(n), Lx + W, C.byref(pk))
        sl = lp.contents
        seq = np.zeros(int(sl.length), np.int8)
        _lib.check(L.ramx_packed_decode(pk, 0, sl.length, seq.ctypes.data), "ramx_packed_decode")
        cores = cores_from_list(head, n.value)
        m = new_master(Lx)
        rr = L.ramx_extend_alignment(1, head, None, lp, m.ctypes.data, W, -90, 27, Lx, n.value, sp, None)
        got = cores_from_list(head, n.value)
        p = po.Params.named("14p43g", bandwidth=W, L=Lx, when_to_stop=30)
        c2 = cores.copy(); m2 = new_master(Lx)
        o1 = po.oracle_extend(1, c2, seq, m2, p)
        assert rr == o1.ret and np.array_equal(m, m2), f"genome {g}: consensus differs from the oracle on this library's bases"
        assert np.array_equal(got.right_len, c2.right_len) and np.array_equal(got.score, c2.score), f"genome {g}"
        rets.append((rr, m.copy()))
        L.ramx_free_library(lp, head)
    assert not np.array_equal(rets[0][1], rets[1][1])           # the two genomes really extend differently


SEAM1 = os.path.join(ROOT, "oracle", "_ref", "RAMExtend_seam1")


@pytest.mark.skipif(not os.path.exists(SEAM1), reason="oracle/_ref/RAMExtend_seam1 not built (needs /root/reference at build time)")
@pytest.mark.parametrize("case", CASES)
def test_reference_main_with_only_the_loop_replaced(case, tmp_path):
    """INTEGRATION.md section B, for real: the reference's own main(), loader, report and writers, linked against
    libramx with extend_alignment() replaced by the seam-1 binding (oracle/seam1_shim.c).  Its output must be
    byte-identical to the unmodified reference's, version line included."""
    argv = open(os.path.join(G, "cli", case, "argv")).read().split()
    stem = STEM[case.split("_")[0]]
    cmd = [SEAM1, "-twobit", f"inputs/{stem}.2bit", "-ranges", f"inputs/{stem}.tsv", "-cons", str(tmp_path / "cons"),
           "-outtsv", str(tmp_path / "tsv"), "-outfa", str(tmp_path / "fa")] + argv
    r = subprocess.run(cmd, cwd=G, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = r.stdout.replace(str(tmp_path) + "/", "").splitlines()
    want = open(os.path.join(G, "cli", case, "stdout")).read().splitlines()
    got = [l for l in got if not l.startswith("Program duration is")]
    want = [l for l in want if not l.startswith("Program duration is")]
    assert got == want
    for f in ("cons", "tsv", "fa"):
        ref_f = os.path.join(G, "cli", case, f)
        if os.path.exists(ref_f):
            assert open(tmp_path / f).read() == open(ref_f).read(), f


@pytest.mark.parametrize("case", VERBOSE)
def test_cli_vvvv_per_row_lines_match_reference(case):
    """-vvvv: the reference's per-row lines (ram_extend.c:992-1090, 1134-1214) -- for every row, candidate base and
    extendable core the candidate row's best score and column, `-- max(0,best_score) = 0!`, ` **OUT_OF_SEQ**`,
    ` **CAPPED** contributing = ...`, the totals per candidate, the chosen base, and the "Alignment Extension" block with
    its float columns and the new-maximum / extensions-since lines -- written from the device's candidate-row trace (full
    recurrence).  Whole stdout against the reference binary's, both directions; five cases incl. short flanks
    (OUT_OF_SEQ), a tight cap (CAPPED), repeatscout scoring and bandwidth 3.  The `*_v5` cases run `-vvvvv` (VERBOSE 12): in
    addition the boundary rows (ram_extend.c:949-959), EVERY cell (gap and substitution state) of every candidate row
    (:1013-1024) and the sequence around every core's edge (report.c printExtensionRegion) -- four cases, both strands."""
    import gzip
    argv = open(os.path.join(G, "cli", case, "argv")).read().split()
    stem = open(os.path.join(G, "cli", case, "stem")).read().strip()
    cmd = [_lib.CLI_PATH, "-twobit", f"inputs/{stem}.2bit", "-ranges", f"inputs/{stem}.tsv"] + argv
    r = subprocess.run(cmd, cwd=G, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = gzip.open(os.path.join(G, "cli", case, "vvvv.gz"), "rb").read().decode()
    g, w = _norm(r.stdout), _norm(want)
    assert len(g) == len(w), (len(g), len(w))
    for i, (a, b) in enumerate(zip(g, w)):
        assert a == b, f"line {i}: {a!r} != {b!r}"
