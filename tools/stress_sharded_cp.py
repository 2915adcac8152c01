"""Repeat the two-rank in-kernel vote exchange (cell-parallel device-wide kernel, both ranks on one GPU) and compare every
repetition with the first: any difference is a race.  Usage: python tools/stress_sharded_cp.py [W] [repetitions] [kind]"""
import os, socket, sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, W, reps, kind):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.device import Device
    from repeatafterme_amd.sharded import extend_alignment_sharded, gpu_engine
    from repeatafterme_amd.synth import synth_family
    from helpers import to_extend_params

    def all_gather(x):
        outs = [None] * world
        dist.all_gather_object(outs, np.asarray(x, np.int32))
        return np.concatenate(outs)

    def ag_bytes(b):
        lst = [None] * world
        dist.all_gather_object(lst, b)
        return lst

    def ar_min(v):
        t = torch.tensor([v]); dist.all_reduce(t, op=dist.ReduceOp.MIN); return int(t.item())

    fs = synth_family(333, 150, W, K=100, seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=W, L=150, when_to_stop=25)
    dev = Device(0)
    os.environ["RAMX_PEER_KIND"] = kind
    assert dev.peer_setup(rank, world, ag_bytes, ar_min, dist.barrier)
    dev.load_library(fs.sequence)
    first = None
    bad = 0
    ovf = 0
    for it in range(reps):
        c = fs.cores.copy(); m = new_master(p.L)
        rets = []
        for d in (1, 0):
            rets.append(extend_alignment_sharded(d, c, fs.sequence, m, to_extend_params(p), rank, world, gpu_engine(dev), all_gather))
            ovf += int(dev.last.overflow32 != 0)
        assert dev.last.persistent == 1 and dev.last.lanes_per_flank > 1, (dev.last.persistent, dev.last.lanes_per_flank)
        cur = (rets, m.copy(), c.left_len.copy(), c.right_len.copy(), c.score.copy())
        if first is None:
            first = cur
            continue
        names = ["rets", "master", "left_len", "right_len", "score"]
        for nm, a, b in zip(names, first, cur):
            if nm == "rets":
                if a != b: print(f"rank {rank} rep {it}: rets {a} vs {b}", flush=True); bad += 1
            elif not np.array_equal(a, b):
                w = np.flatnonzero(a != b)
                print(f"rank {rank} rep {it}: {nm} differs at {w[:10].tolist()} ({len(w)} places): first {a[w[:10]].tolist()} now {b[w[:10]].tolist()}", flush=True)
                bad += 1
    print(f"rank {rank}: {reps} repetitions, {bad} differences, {ovf} directions reported a column sum outside int32", flush=True)
    dev.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    kind = sys.argv[3] if len(sys.argv) > 3 else "device"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, W, reps, kind), nprocs=2, join=True)
