"""The sharded control flow of libramx on the GPU: two ranks (processes) share the test box's single GPU, each
owning half of the flanks; the per-column 4 x int64 vote is all-reduced through the test hook (gloo) because
RCCL refuses two ranks on one device.  Everything but the ncclAllReduce call itself is the production path:
fold kernel, reduced vote read by the next column kernel, stop rule replicated on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out, peer=False, W=20, kind="", fail_rank=None, no_cp=False, delay=None, shape=(333, 150, 100)):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.device import Device
    from repeatafterme_amd.sharded import extend_alignment_sharded, gpu_engine
    from repeatafterme_amd.synth import synth_family
    from helpers import to_extend_params

    def allreduce4(v):
        t = torch.tensor(v, dtype=torch.int64)
        dist.all_reduce(t)
        return t.tolist()

    def all_gather(x):
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([len(x)]))
        m = int(max(s.item() for s in sizes))
        buf = torch.zeros(m, dtype=torch.int32); buf[:len(x)] = torch.from_numpy(np.asarray(x, np.int32))
        outs = [torch.zeros(m, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(outs, buf)
        return np.concatenate([o[:int(s.item())].numpy() for o, s in zip(outs, sizes)])

    fs = synth_family(shape[0], shape[1], W, K=shape[2], seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=W, L=shape[1], when_to_stop=25)
    if fail_rank is not None:
        os.environ["RAMX_TEST_FAIL_PRK_RANK"] = str(fail_rank)
    if delay is not None:
        os.environ["RAMX_TEST_DELAY_RANK"] = delay      # "rank:ms": that rank launches late, its peers' kernels wait for its words
    if no_cp:
        os.environ["RAMX_NO_CP_DEVICE"] = "1"           # the lane-per-flank persistent kernel instead of the cell-parallel one
    dev = Device(0)
    dev.set_allreduce_callback(allreduce4)
    enabled = False
    if peer:
        def ag_bytes(b):
            lst = [None] * world
            dist.all_gather_object(lst, b)
            return lst

        def ar_min(v):
            t = torch.tensor([v]); dist.all_reduce(t, op=dist.ReduceOp.MIN); return int(t.item())
        if kind:
            os.environ["RAMX_PEER_KIND"] = kind
        enabled = dev.peer_setup(rank, world, ag_bytes, ar_min, dist.barrier)
        enabled = enabled and (not kind or dev.peer_kind == kind)
    else:
        os.environ["RAMX_NO_PERSISTENT"] = "1"          # per-column launches + host collective
    dev.load_library(fs.sequence)
    c = fs.cores.copy(); m = new_master(p.L)
    rets = []
    for d in (1, 0):
        rets.append(extend_alignment_sharded(d, c, fs.sequence, m, to_extend_params(p), rank, world, gpu_engine(dev), all_gather))
    used_persistent = dev.last.persistent
    lanes = dev.last.lanes_per_flank
    packed = dev.last.packed_rows
    dev.close()
    out[rank] = (rets, m.copy(), c.left_len.copy(), c.right_len.copy(), c.score.copy(), enabled, used_persistent, lanes, packed)
    dist.destroy_process_group()


def test_two_ranks_one_gpu_equal_single_process_oracle():
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.synth import synth_family
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    fs = synth_family(333, 150, 20, K=100, seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=20, L=150, when_to_stop=25)
    c = fs.cores.copy(); m = new_master(p.L)
    r1 = po.oracle_extend(1, c, fs.sequence, m, p); r0 = po.oracle_extend(0, c, fs.sequence, m, p)
    for rank in range(world):
        rets, mm, ll, rl, sc, enabled, used, lanes, packed = out[rank]
        assert rets == [(r1.ret, r1.rows_executed), (r0.ret, r0.rows_executed)], rank
        assert np.array_equal(mm, m) and np.array_equal(ll, c.left_len) and np.array_equal(rl, c.right_len)
        assert np.array_equal(sc, c.score)
        assert used == 0


@pytest.mark.parametrize("no_cp", [False, True], ids=["cell-parallel", "lane-per-flank"])
@pytest.mark.parametrize("W,kind", [(14, "device"), (40, "device"), (20, "host"), (40, "host"), (40, ""), (80, "device")])
def test_two_ranks_cross_device_persistent_path(W, kind, no_cp):
    """Same two ranks, but with the mailboxes enabled: each rank runs ONE persistent launch per direction and the
    per-column vote is exchanged from inside the kernels (system-scope stores into every rank's box).  kind "device":
    fine-grained device memory mapped over hipIpc; "host": one POSIX shared-memory segment registered with HIP; "":
    the production order (device first).  Both kernels that carry the exchange: the cell-parallel kernel (K lanes per
    flank, speculation on the workgroup's vote) and the lane-per-flank persistent kernel.  The two launches share the
    test box's single GPU."""
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.synth import synth_family
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True, W, kind, None, no_cp), nprocs=world, join=True)
    fs = synth_family(333, 150, W, K=100, seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=W, L=150, when_to_stop=25)
    c = fs.cores.copy(); m = new_master(p.L)
    r1 = po.oracle_extend(1, c, fs.sequence, m, p); r0 = po.oracle_extend(0, c, fs.sequence, m, p)
    for rank in range(world):
        rets, mm, ll, rl, sc, enabled, used, lanes, packed = out[rank]
        assert enabled, "peer self-test failed"
        assert used == 1, "cross-device persistent launch was not taken (or fell back)"
        assert (lanes == 1) == no_cp, lanes
        assert rets == [(r1.ret, r1.rows_executed), (r0.ret, r0.rows_executed)], rank
        assert np.array_equal(mm, m) and np.array_equal(ll, c.left_len) and np.array_equal(rl, c.right_len)
        assert np.array_equal(sc, c.score)


@pytest.mark.parametrize("K", [4, 2])
def test_two_ranks_cell_parallel_long_blocks(K, monkeypatch):
    """RAMX_CP_K caps the lanes per flank: 21 and 41 cells per lane at W = 40 -- the device-wide kernel WITHOUT a vote wave
    (vote, then band), which is what 25,000-50,000 flanks per rank get.  Same exchange through the mailboxes."""
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.synth import synth_family
    monkeypatch.setenv("RAMX_CP_K", str(K))             # inherited by the spawned ranks
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True, 40, "device", None, False), nprocs=world, join=True)
    fs = synth_family(333, 150, 40, K=100, seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=40, L=150, when_to_stop=25)
    c = fs.cores.copy(); m = new_master(p.L)
    r1 = po.oracle_extend(1, c, fs.sequence, m, p); r0 = po.oracle_extend(0, c, fs.sequence, m, p)
    for rank in range(world):
        rets, mm, ll, rl, sc, enabled, used, lanes, packed = out[rank]
        assert enabled and used == 1 and lanes == K, (enabled, used, lanes)
        assert rets == [(r1.ret, r1.rows_executed), (r0.ret, r0.rows_executed)], rank
        assert np.array_equal(mm, m) and np.array_equal(ll, c.left_len) and np.array_equal(rl, c.right_len)
        assert np.array_equal(sc, c.score), np.flatnonzero(sc != c.score)[:10]


@pytest.mark.parametrize("no_cp", [False, True], ids=["cell-parallel", "lane-per-flank"])
@pytest.mark.parametrize("late", [0, 1])
def test_two_ranks_one_launches_late(late, no_cp):
    """One rank's single launch starts 30 ms after the other's: the early kernel reaches the exchange of row 0 long before
    its peer has written anything.  A cleared mailbox slot must not pass for row 0's word (the tag is row + 1), and the
    mailbox is cleared before -- not while -- the peer may write into it."""
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.synth import synth_family
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True, 14, "device", None, no_cp, f"{late}:30"), nprocs=world, join=True)
    fs = synth_family(333, 150, 14, K=100, seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=14, L=150, when_to_stop=25)
    c = fs.cores.copy(); m = new_master(p.L)
    r1 = po.oracle_extend(1, c, fs.sequence, m, p); r0 = po.oracle_extend(0, c, fs.sequence, m, p)
    for rank in range(world):
        rets, mm, ll, rl, sc, enabled, used, lanes, packed = out[rank]
        assert enabled and used == 1 and (lanes == 1) == no_cp
        assert rets == [(r1.ret, r1.rows_executed), (r0.ret, r0.rows_executed)], rank
        assert np.array_equal(mm, m) and np.array_equal(ll, c.left_len) and np.array_equal(rl, c.right_len)
        assert np.array_equal(sc, c.score), np.flatnonzero(sc != c.score)[:10]


def test_rank_local_failure_after_agreement_falls_back_on_all_ranks():
    """A rank whose persistent launch fails AFTER the ranks have agreed on the mailbox path (forced by a test hook) must
    still take part in the second agreement: every rank then repeats the direction with per-column launches -- nobody
    hangs in a collective the failed rank never joins, and the results are still the oracle's."""
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.synth import synth_family
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True, 20, "host", 1), nprocs=world, join=True)
    fs = synth_family(333, 150, 20, K=100, seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=20, L=150, when_to_stop=25)
    c = fs.cores.copy(); m = new_master(p.L)
    r1 = po.oracle_extend(1, c, fs.sequence, m, p); r0 = po.oracle_extend(0, c, fs.sequence, m, p)
    for rank in range(world):
        rets, mm, ll, rl, sc, enabled, used, lanes, packed = out[rank]
        assert enabled
        assert used == 0, "the direction must have been repeated with per-column launches on every rank"
        assert rets == [(r1.ret, r1.rows_executed), (r0.ret, r0.rows_executed)], rank
        assert np.array_equal(mm, m) and np.array_equal(ll, c.left_len) and np.array_equal(rl, c.right_len)
        assert np.array_equal(sc, c.score)


@pytest.mark.parametrize("no_cp", [False, True], ids=["cell-parallel", "lane-per-flank"])
def test_two_ranks_at_a_real_shard_size(no_cp, monkeypatch):
    """The mailbox exchange with shards of the size a real multi-GPU run has: 2 x 20,000 flanks (the other tests of this file run
    333 flanks in ~5 workgroups per rank).  Each rank's launch is 79-90 workgroups; the two launches are co-resident on the box's
    256 CUs and exchange every column's vote through the mailboxes.  Results = the single-process oracle on all 40,000 flanks."""
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.synth import synth_family
    if not no_cp:
        monkeypatch.setenv("RAMX_CP_K", "2")            # 224 flanks per workgroup: 90 workgroups per rank, both ranks resident
    world, shape, W = 2, (40000, 60, 40), 40
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True, W, "device", None, no_cp, None, shape), nprocs=world, join=True)
    fs = synth_family(shape[0], shape[1], W, K=shape[2], seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=W, L=shape[1], when_to_stop=25)
    c = fs.cores.copy(); m = new_master(p.L)
    r1 = po.oracle_extend(1, c, fs.sequence, m, p); r0 = po.oracle_extend(0, c, fs.sequence, m, p)
    for rank in range(world):
        rets, mm, ll, rl, sc, enabled, used, lanes, packed = out[rank]
        assert enabled and used == 1, (enabled, used)
        assert (lanes == 1) == no_cp, lanes
        assert rets == [(r1.ret, r1.rows_executed), (r0.ret, r0.rows_executed)], rank
        assert np.array_equal(mm, m) and np.array_equal(ll, c.left_len) and np.array_equal(rl, c.right_len)
        assert np.array_equal(sc, c.score)


@pytest.mark.parametrize("segment,exchanger", [("0", True), ("48", True), ("0", False)],
                         ids=["one launch", "pieces of 48 columns", "one launch, no exchanger workgroup"])
def test_two_ranks_packed_rows_cross_device(segment, exchanger, monkeypatch):
    """The packed-row kernel on the cross-device route: each rank runs its first rows on the int32 kernel, hands the sums of the
    next row over to the packed kernel (a launch that starts with this device's sums only: the exchange of that row follows), and
    -- second case -- continues in pieces of 48 columns, every piece starting the same way; the device's exchange duties are done
    by a workgroup of their own (third case: by workgroup 0).  2 x 20,000 flanks, results = the
    single-process oracle; `packed_rows` shows the path was taken on every rank."""
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    from repeatafterme_amd.synth import synth_family
    monkeypatch.setenv("RAMX_PK_SEGMENT", segment)
    if not exchanger:
        monkeypatch.setenv("RAMX_NO_PK_EXCHANGER", "1")     # workgroup 0 forwards the rank's totals (what a full device does)
    world, shape, W = 2, (40000, 200, 150), 40
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True, W, "device", None, True, None, shape), nprocs=world, join=True)
    fs = synth_family(shape[0], shape[1], W, K=shape[2], seed=12, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
    p = po.Params.named("14p43g", bandwidth=W, L=shape[1], when_to_stop=25)
    c = fs.cores.copy(); m = new_master(p.L)
    r1 = po.oracle_extend(1, c, fs.sequence, m, p); r0 = po.oracle_extend(0, c, fs.sequence, m, p)
    for rank in range(world):
        rets, mm, ll, rl, sc, enabled, used, lanes, packed = out[rank]
        assert enabled and used == 1 and lanes == 1, (enabled, used, lanes)
        assert packed > 50, packed
        assert rets == [(r1.ret, r1.rows_executed), (r0.ret, r0.rows_executed)], rank
        assert np.array_equal(mm, m) and np.array_equal(ll, c.left_len) and np.array_equal(rl, c.right_len)
        assert np.array_equal(sc, c.score)


_RCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
os.environ["RAMX_COMM_SINGLE"] = "1"          # a communicator of one rank
os.environ["RAMX_FORCE_COLLECTIVE"] = "1"     # ... whose collectives are used
os.environ["RAMX_NO_PEER"] = "1"              # the per-column route: one ncclAllReduce per column between the launches
if {torch_first}:
    import torch                                 # torch's HIP / RCCL first: libramx binds to them (DESIGN.md section 7)
import numpy as np
from oracle import pyoracle as po
from repeatafterme_amd.datamodel import new_master
from repeatafterme_amd.device import Device, resolve_flanks
from repeatafterme_amd.synth import synth_family, writeback_from_trim
from helpers import to_extend_params
fs = synth_family(700, 120, 20, K=80, seed=21, both_sides=True, minus_frac=0.3, n_run_frac=0.1)
p = po.Params.named("14p43g", bandwidth=20, L=120, when_to_stop=25)
dev = Device(0)
dev.comm_init(dev.unique_id(), 0, 1)
assert dev.comm_size() == 1
dev.load_library(fs.sequence)
c = fs.cores.copy(); m = new_master(p.L)
r = po.oracle_extend(1, c, fs.sequence, m, p)
flanks, idx = resolve_flanks(1, fs.cores, p.bandwidth, p.L)
dev.begin_direction(flanks, to_extend_params(p))
info = dev.run_direction()
cons, th, tp = dev.download()
assert info.persistent == 0 and info.launches >= info.rows_executed, (info.persistent, info.launches)    # per-column launches
assert (info.ret, info.rows_executed) == (r.ret, r.rows_executed), (info.ret, r.ret)
assert np.array_equal(cons[:r.rows_executed], m[p.L + 1:p.L + 1 + r.rows_executed])
ext, sc = writeback_from_trim(th, tp)
assert np.array_equal(ext, c.right_len[idx]) and np.array_equal(sc, (c.score - fs.cores.score)[idx])
loaded = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l or "libamdhip64" in l]
print("RCCL_OK", sorted(set(os.path.basename(x) for x in loaded)), sorted(set(os.path.dirname(x) for x in loaded)))
dev.close()
"""


@pytest.mark.parametrize("torch_first", [True, False], ids=["torch imported first", "executable order (no torch)"])
def test_rccl_single_rank_communicator_runs_the_collective_route(torch_first):
    """The collective north_star names -- one RCCL all-reduce of the column's sums (ram_extend.c:1052-1085 is what it sums) -- on the
    one GPU a test box has: ramx_comm_unique_id -> ramx_dev_comm_init(d, id, 0, 1) -> ramx_dev_comm_size == 1 -> a direction on
    the per-column route whose every column goes through ncclAllReduce (csrc/ramx_device.hip host_allreduce_shards), result =
    oracle.  Once with torch's runtime loaded first and once in the executable's order (no torch in the process: libramx brings
    /opt/rocm's HIP and RCCL), DESIGN.md section 7's two cases."""
    import subprocess
    env = dict(os.environ)
    for k in ("RAMX_NO_PERSISTENT", "RAMX_NO_CP_DEVICE", "RAMX_CP_K"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT.format(root=ROOT, torch_first=torch_first)], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.returncode, r.stdout[-600:], r.stderr[-1500:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_OK")][0]
    assert "librccl" in line
    if not torch_first:
        assert "torch" not in line, line                # no torch runtime in the process
