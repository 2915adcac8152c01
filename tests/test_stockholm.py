"""Stockholm seed alignments -> RAMExtend ranges (the caller-side data format, extend-stk.pl:303-347)."""
import os
import re

from repeatafterme_amd import stockholm as stk

HERE = os.path.dirname(os.path.abspath(__file__))


def _write(tmp_path, text):
    p = tmp_path / "f.stk"
    p.write_text(text)
    return str(p)


def test_flags_follow_the_ten_column_tolerance(tmp_path):
    dots = lambda n: "." * n
    text = "\n".join([
        "# STOCKHOLM 1.0", "#=GF ID    famA", "#=GF DE    Source:gsa, mDiv=23.22, x.2bit:108",
        "#=GC RF    " + "x" * 40, "//"[:0],
        "chr1:101-140_+ " + "ACGT" * 10,                                  # reaches both edges
        "chr1:201-229_- " + dots(10) + "A" * 29 + dots(1),                # 10 leading dots still count as the edge
        "chr2:301-328_+ " + dots(11) + "C" * 28 + dots(1),                # 11 do not: right only
        "hg38:chr3:401-420_- " + "G" * 20 + dots(20),                     # left only; assembly prefix is dropped
        "chr4:501-510_+ " + dots(15) + "T" * 10 + dots(15),               # neither
        "gi|12345:1-10_+ " + "A" * 40,                                    # unresolved RepeatModeler name: skipped
        "//", ""])
    seeds = stk.read_stockholm(_write(tmp_path, text))
    assert len(seeds) == 1 and seeds[0].name == "famA"
    rows, extendable = stk.ranges_for(seeds[0])
    assert rows == [("chr1", 100, 140, 1, 1, "+"), ("chr1", 200, 229, 1, 1, "-"), ("chr2", 300, 328, 0, 1, "+"),
                    ("chr3", 400, 420, 1, 0, "-"), ("chr4", 500, 510, 0, 0, "+")]
    assert extendable == 4
    assert stk.family_divergence(seeds[0]) == 23.22
    assert stk.choose_scoring(23.22) == ("25p43g", 27)


def test_matrix_thresholds():
    # extend-stk.pl:291-304
    assert stk.choose_scoring(0.0) == ("14p43g", 30)
    assert stk.choose_scoring(15.99) == ("14p43g", 30)
    assert stk.choose_scoring(16.0) == ("18p43g", 30)
    assert stk.choose_scoring(19.0) == ("20p43g", 30)
    assert stk.choose_scoring(22.49) == ("20p43g", 30)
    assert stk.choose_scoring(22.5, 4) == ("25p43g", 36)


def test_several_records_and_names(tmp_path):
    text = ("# STOCKHOLM 1.0\n#=GF AC    DF0000001\nchr1:5-8 ACGT\nchr1:20-11 ACGTACGTAC\n//\n"
            "# STOCKHOLM 1.0\nchrX:1-4_+ AC-T\n//\n")
    seeds = stk.read_stockholm(_write(tmp_path, text))
    assert [s.name for s in seeds] == ["DF0000001", "Unnamed_Family"]
    assert stk.ranges_for(seeds[0])[0] == [("chr1", 4, 8, 1, 1, "+"), ("chr1", 10, 20, 1, 1, "-")]
    assert stk.reference_sequence(seeds[1]) == "ACT"


def test_reference_fixture_family():
    """test/ce10-fam2.stk of the reference (100 instances; its CC line states the consensus length)."""
    path = os.path.join(HERE, "golden", "inputs", "ce10-fam2.stk")
    seed, = stk.read_stockholm(path)
    assert seed.name == "rnd-1_family-92" and len(seed.rows) == 100
    rows, extendable = stk.ranges_for(seed)
    assert extendable == 100 and len(rows) == 100
    assert rows[0] == ("chrIV", 1930657, 1930718, 1, 1, "-")
    for (name, s, e, lf, rf, o), r in zip(rows, seed.rows):
        assert (s, e, o) == (r.start - 1, r.end, r.orient) and 0 <= s < e
        assert lf == (len(r.aligned) - len(r.aligned.lstrip(".")) <= 10)
        assert rf == (len(r.aligned) - len(r.aligned.rstrip(".")) <= 10)
    width = int(re.search(r"refLength=(\d+)", open(path).read()).group(1))
    assert len(seed.rf) == width and all(len(r.aligned) == width for r in seed.rows)
    assert len(stk.reference_sequence(seed)) == seed.rf.lower().count("x") == 66
    assert 5.0 < stk.kimura_divergence(seed) < 30.0


def test_flag_rule_and_scoring_ladder_against_the_wrapper_vectors():
    """tests/golden/stk_vectors.json: made by tests/golden/make_stk_vectors.pl -- Perl, with the reference wrapper's own regular
    expressions (util/extend-stk.pl:330-346), its start - 1 (:314), its gi|NNN rule (:308-311), its 'more than 3 extendable rows'
    (:364) and its divergence -> matrix / -minimprovement ladder (:291-302) -- on the reference's three fixtures
    test/ce10-fam{1,2,3}.stk (copies under tests/golden/inputs/)."""
    import json
    doc = json.load(open(os.path.join(HERE, "golden", "stk_vectors.json")))
    assert [f["file"] for f in doc["families"]] == ["ce10-fam1.stk", "ce10-fam2.stk", "ce10-fam3.stk"]
    for fam in doc["families"]:
        seed, = stk.read_stockholm(os.path.join(HERE, "golden", "inputs", fam["file"]))
        rows, extendable = stk.ranges_for(seed)
        assert [list(r) for r in rows] == fam["rows"], fam["file"]
        assert extendable == fam["extendable_count"] and (extendable > 3) == fam["runs_ramextend"]
        # no mDiv in these descriptions: the divergence is the computed one (:279-283)
        assert fam["mDiv"] is None and stk.family_divergence(seed) == stk.kimura_divergence(seed)
    assert any(r[3:5] != [1, 1] for r in doc["families"][2]["rows"])      # fam3 has rows that miss an edge: the rule is exercised
    for tdiv, mas, matrix, minimp in doc["scoring_ladder"]:
        assert stk.choose_scoring(tdiv, mas) == (matrix, minimp), (tdiv, mas)
