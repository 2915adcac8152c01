"""Host logic of the product on CPU: scoring systems and flank resolution through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from repeatafterme_amd import _lib
from repeatafterme_amd.device import resolve_flanks
from repeatafterme_amd.scoring import get_matrix, get_repeatscout_matrix, matrix_lambda, named_params
from repeatafterme_amd.synth import synth_adversarial

IDX = [i * 100 + j for i in list(range(8)) + [99] for j in list(range(8)) + [99]]


@pytest.mark.parametrize("name", ["14p43g", "18p43g", "20p43g", "25p43g"])
def test_named_matrices_equal_oracle(name):
    a, b = get_matrix(name), po.get_matrix(name)
    assert np.array_equal(a[0][IDX], b[0][IDX]) and a[1:] == b[1:]


def test_repeatscout_matrix_equal_oracle():
    for args in ((1, -1, -5), (2, -3, -7), (5, -4, -1)):
        a, b = get_repeatscout_matrix(*args), po.get_repeatscout_matrix(*args)
        assert np.array_equal(a[0][IDX], b[0][IDX]) and a[1:] == b[1:] == (0, args[2])


def test_lambda_values():
    """calculateLambda (score_system.c:37-89) as run by the reference in the build container."""
    exp = {"14p43g": 0.12506866455078125, "18p43g": 0.12520599365234375, "20p43g": 0.12267303466796875,
           "25p43g": 0.13137054443359375}
    for k, v in exp.items():
        assert abs(matrix_lambda(k) - v) < 1e-12


def test_cli_defaults_per_matrix():
    """ram_extend.c:301-344"""
    assert (named_params("14p43g").minimprovement, named_params("14p43g").cappenalty) == (27, -90)
    assert (named_params("25p43g").minimprovement, named_params("25p43g").cappenalty) == (24, -90)
    assert (named_params("repeatscout").minimprovement, named_params("repeatscout").cappenalty) == (3, -20)


def _ref_rule(direction, lp, rp, lo, up, orient, W, row, offset):
    """The reference's unsigned index logic (bnw_extend.c:778-788,824-868) in Python ints mod 2^64."""
    M = 1 << 64
    if direction:
        start = (rp - 1) % M if orient else (rp + 1) % M
    else:
        start = (lp + 1) % M if orient else (lp - 1) % M
    so = -offset - row if direction == orient else offset + row
    idx = (start + so) % M
    if start < W and so < 0 and abs(so) > start:
        return None
    if idx > up or idx < lo:
        return None
    return idx


def test_flank_resolution_matches_reference_index_rule():
    """ramx_resolve_flanks (csrc/ramx_extend.c) vs the reference rule, cell by cell (SURVEY App. D rule 1)."""
    for seed in range(6):
        fs = synth_adversarial(seed)
        c = fs.cores
        # push a few cores to the very start of the array to hit the start < W underflow guard
        for W, L in ((14, 40), (3, 25), (40, 30)):
            for direction in (0, 1):
                (fl, nx), idx = resolve_flanks(direction, c, W, L)
                assert nx == len(idx)
                for i in range(nx):
                    n = idx[i]
                    f = fl[i]
                    for row in sorted({0, 1, min(max(W - 1, 0), L - 1), min(W, L - 1), L - 1}):   # rows the loop can reach
                        for off in (-W, -W // 2, -1, 0, 1, W):
                            t = off + row
                            want = _ref_rule(direction, int(c.left_pos[n]), int(c.right_pos[n]), int(c.lower[n]),
                                             int(c.upper[n]), int(c.orient[n]), W, row, off)
                            inb = f.t_lo <= t <= f.t_hi
                            assert inb == (want is not None), (seed, W, direction, n, row, off)
                            if inb:
                                assert f.start + f.step * t == want
                                assert f.compl_ == c.orient[n]


def test_flank_resolution_at_array_start():
    """core at position 0/1 of the array: start = -1 or 0 (uint64 wrap in the reference)."""
    from repeatafterme_amd.datamodel import CoreSet
    c = CoreSet(left_pos=[0, 2, 1], right_pos=[1, 0, 3], lower=[0, 0, 0], upper=[30, 30, 30], orient=[0, 1, 0],
                left_ext=[1, 1, 1], right_ext=[1, 1, 1])
    for W in (2, 5):
        for direction in (0, 1):
            (fl, nx), idx = resolve_flanks(direction, c, W, 20)
            for i in range(nx):
                n = idx[i]
                for row in range(0, 8):
                    for off in range(-W, W + 1):
                        want = _ref_rule(direction, int(c.left_pos[n]), int(c.right_pos[n]), 0, 30, int(c.orient[n]), W, row, off)
                        t = off + row
                        assert (fl[i].t_lo <= t <= fl[i].t_hi) == (want is not None), (W, direction, n, row, off)
