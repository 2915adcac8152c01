"""Pins the oracle's banded row (oracle/ramx_oracle.c: ramx_oracle_nw_row) on the reference's own
unit-test vectors for compute_nw_row (reference bnw_extend.c:1384-1651):
  - reciprocity: left-extension rows on a library == right-extension rows on its reverse library,
    all 11 cells x 2 states x 19 rows x 3 cores (one reverse-strand core, one N);
  - the known-answer row n=0, row=6 (final_right_gap / final_right_sub, bnw_extend.c:1566-1567).
The arrays below are the DATA of that test (libraries, cores, consensus, expected rows)."""
import numpy as np

from oracle import pyoracle as po

W = 5   # MAXOFFSET, bnw_extend.c:1394
# bnw_extend.c:1506-1513
P_LEFT = np.array([2, 1, 0, 0, 3, 2, 1, 3, 2, 0, 0, 3, 1, 3, 0, 2,
                   1, 3, 0, 2, 0, 3, 3, 1, 0, 2, 1, 0, 3, 3, 2, 99,
                   2, 1, 3, 2, 0, 1, 1, 3, 1, 0, 0, 3, 2, 3, 0, 2], np.int8)
# bnw_extend.c:1529-1536
P_RIGHT = np.array([2, 0, 3, 1, 3, 0, 0, 2, 3, 1, 2, 3, 0, 0, 1, 2,
                    99, 2, 3, 3, 0, 1, 2, 0, 1, 3, 3, 0, 2, 0, 3, 1,
                    2, 0, 3, 2, 3, 0, 0, 1, 3, 1, 1, 0, 2, 3, 1, 2], np.int8)
CONS = [1, 3, 0, 0, 1, 3, 1, 1, 0, 2, 3, 1, 2, 3, 0, 0, 1, 2, 2]   # :1515
# (leftSeqPos, rightSeqPos, orient): :1521-1523 and :1537-1539
L_CORES = [(12, 14, 0), (16, 18, 1), (44, 46, 0)]
R_CORES = [(1, 3, 0), (29, 31, 1), (33, 35, 0)]
LOWER = [0, 16, 32]
UPPER = [16, 32, 48]
FINAL_RIGHT_GAP = [-35, -21, -26, -17, -22, -7, -12, -17, -22, -27, -32]   # :1566
FINAL_RIGHT_SUB = [-31, -45, -31, -25, 26, -32, -25, -37, -47, -27, -57]   # :1567


def _boundary(n_align, go, ge):
    B = 2 * W + 1
    s = np.zeros((2, n_align, B, 2), np.int32)
    for o in range(-W, W + 1):
        s[1, :, o + W, :] = 0 if o == 0 else abs(o) * ge + go     # :1443-1480
    return s


def test_reciprocity_and_known_row():
    mat, go, ge = po.get_matrix("20p43g")                         # :1414
    ls, rs = _boundary(3, go, ge), _boundary(3, go, ge)
    for n in range(3):
        for row in range(19):
            br, _ = po.oracle_nw_row(1, row, n, 3, CONS[row], R_CORES[n][0], R_CORES[n][1], R_CORES[n][2], rs,
                                     LOWER[n], UPPER[n] - 1, P_RIGHT, mat, go, ge, W)
            bl, _ = po.oracle_nw_row(0, row, n, 3, CONS[row], L_CORES[n][0], L_CORES[n][1], L_CORES[n][2], ls,
                                     LOWER[n], UPPER[n] - 1, P_LEFT, mat, go, ge, W)
            assert np.array_equal(ls[row % 2, n], rs[row % 2, n]), f"reciprocity n={n} row={row}"   # :1617-1631
            assert br == bl
            if n == 0 and row == 6:                                # :1633-1648
                assert rs[0, 0, :, 1].tolist() == FINAL_RIGHT_GAP
                assert rs[0, 0, :, 0].tolist() == FINAL_RIGHT_SUB
                assert br == 26


def test_matrices_match_reference_values():
    """score_system.c:207-376 spot values + gap penalties :210-211,252-253,295-296,338-339."""
    exp = {"14p43g": (-33, -7, 9, -21, 11), "18p43g": (-30, -6, 9, -18, 10), "20p43g": (-28, -5, 9, -17, 10),
           "25p43g": (-25, -5, 8, -15, 9)}
    for name, (go, ge, aa, at, cc) in exp.items():
        m, g1, g2 = po.get_matrix(name)
        m = m.reshape(100, 100)
        assert (g1, g2) == (go, ge)
        assert m[0, 0] == aa and m[0, 3] == at and m[3, 0] == at and m[1, 1] == cc and m[2, 2] == cc
        assert m[0, 99] == -1 and m[99, 2] == -1 and m[5, 1] == -1 and m[1, 5] == -1
        assert m[2, 0] != m[0, 2]          # non-symmetric: [cons][seq]
    m, go, ge = po.get_repeatscout_matrix(1, -1, -5)
    m = m.reshape(100, 100)
    assert (go, ge) == (0, -5) and m[0, 0] == 1 and m[0, 1] == -1 and m[0, 99] == -1 and m[4, 5] == -1 and m[4, 1] == -1
