"""N > 1 path on CPU: two gloo ranks, flanks sharded, one 4 x int64 all-reduce per column; the merged
result must equal the single-process oracle.  Exercises repeatafterme_amd.sharded (partition, write-back,
all-gather merge) with the CPU stand-in engine of tests/sharded_ref.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.sharded import extend_alignment_sharded
    from repeatafterme_amd.synth import synth_adversarial
    from helpers import to_extend_params
    from sharded_ref import make_engine

    def allreduce4(v):
        t = torch.from_numpy(np.asarray(v, np.int64).copy())
        dist.all_reduce(t)
        return t.numpy()

    def all_gather(x):
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([len(x)]))
        m = int(max(s.item() for s in sizes))
        buf = torch.zeros(m, dtype=torch.int32); buf[:len(x)] = torch.from_numpy(np.asarray(x, np.int32))
        outs = [torch.zeros(m, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(outs, buf)
        return np.concatenate([o[:int(s.item())].numpy() for o, s in zip(outs, sizes)])

    res = []
    for seed in (3, 8):
        fs = synth_adversarial(seed, n_windows=5, L=40, W=6, K=30)
        p = po.Params.named("20p43g", bandwidth=6, L=40, when_to_stop=12)
        c = fs.cores.copy(); m = new_master(p.L)
        rets = []
        for d in (1, 0):
            ret, rows = extend_alignment_sharded(d, c, fs.sequence, m, to_extend_params(p), rank, world,
                                                 make_engine(allreduce4), all_gather)
            rets.append((ret, rows))
        res.append((rets, m.copy(), c.left_len.copy(), c.right_len.copy(), c.score.copy()))
    out[rank] = res
    dist.destroy_process_group()


def test_two_rank_sharded_run_equals_single_process_oracle():
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.synth import synth_adversarial
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for k, seed in enumerate((3, 8)):
        fs = synth_adversarial(seed, n_windows=5, L=40, W=6, K=30)
        p = po.Params.named("20p43g", bandwidth=6, L=40, when_to_stop=12)
        c = fs.cores.copy(); m = new_master(p.L)
        r1 = po.oracle_extend(1, c, fs.sequence, m, p); r0 = po.oracle_extend(0, c, fs.sequence, m, p)
        for rank in range(world):
            rets, mm, ll, rl, sc = out[rank][k]
            assert rets == [(r1.ret, r1.rows_executed), (r0.ret, r0.rows_executed)], (seed, rank)
            assert np.array_equal(mm, m) and np.array_equal(ll, c.left_len) and np.array_equal(rl, c.right_len)
            assert np.array_equal(sc, c.score)


def test_partition_is_contiguous_and_balanced():
    from repeatafterme_amd.sharded import partition
    for n in (0, 1, 7, 64, 100001):
        for world in (1, 2, 3, 8):
            parts = [partition(n, world, r) for r in range(world)]
            assert parts[0].start == 0 and parts[-1].stop == n
            assert all(parts[i].stop == parts[i + 1].start for i in range(world - 1))
            sizes = [p.stop - p.start for p in parts]
            assert max(sizes) - min(sizes) <= 1
