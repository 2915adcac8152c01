"""Shared test helpers: parameter conversion, golden vectors, result comparison."""
import os

import numpy as np

from oracle import pyoracle as po
from repeatafterme_amd.datamodel import CoreSet, ExtendParams, FlankSet, new_master

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def to_extend_params(p: po.Params) -> ExtendParams:
    return ExtendParams(bandwidth=p.bandwidth, cappenalty=p.cappenalty, minimprovement=p.minimprovement, L=p.L,
                        when_to_stop=p.when_to_stop, l=p.l, gapopen=p.gapopen, gapextn=p.gapextn, matrix=p.matrix)


def load_api_vectors():
    z = np.load(os.path.join(GOLDEN, "api_vectors.npz"))
    out = []
    for k in range(int(z["n_cases"])):
        pre = f"c{k}_"
        cores = CoreSet(**{f: z[pre + f] for f in ("left_pos", "right_pos", "lower", "upper", "orient", "left_ext",
                                                    "right_ext", "seq_idx")})
        W, cap, mini, L, wts, go, ge = [int(v) for v in z[pre + "params"]]
        p = po.Params(bandwidth=W, cappenalty=cap, minimprovement=mini, L=L, when_to_stop=wts, l=1, gapopen=go,
                      gapextn=ge, matrix=z[pre + "matrix"])
        exp = dict(ret=z[pre + "ret"], master=z[pre + "master"], left_len=z[pre + "left_len"],
                   right_len=z[pre + "right_len"], score=z[pre + "score"])
        out.append((k, str(z[pre + "matrix_name"]), z[pre + "sequence"], cores, p, exp))
    return out


def run_both_directions(extend_fn, cores, sequence, p):
    """extend_fn(direction, cores, sequence, master, p) -> object with .ret; right then left like main()."""
    c = cores.copy()
    m = new_master(p.L, p.l)
    rr = extend_fn(1, c, sequence, m, p)
    rl = extend_fn(0, c, sequence, m, p)
    return c, m, rr, rl


def assert_same_result(c1, m1, r1, c2, m2, r2, tag=""):
    assert [r.ret for r in r1] == [r.ret for r in r2], f"{tag}: return values differ"
    assert np.array_equal(m1, m2), f"{tag}: consensus (master) differs at {np.nonzero(m1 != m2)[0][:8]}"
    assert np.array_equal(c1.right_len, c2.right_len), f"{tag}: rightExtensionLen differs"
    assert np.array_equal(c1.left_len, c2.left_len), f"{tag}: leftExtensionLen differs"
    assert np.array_equal(c1.score, c2.score), f"{tag}: score differs"


def gpu_extend(direction, cores, sequence, master, p):
    from repeatafterme_amd.extend import extend_alignment
    return extend_alignment(direction, cores, np.ascontiguousarray(sequence, np.int8), master, to_extend_params(p))


def oracle_extend(direction, cores, sequence, master, p):
    return po.oracle_extend(direction, cores, sequence, master, p)
