#!/usr/bin/env python3
"""bench.py -- RAMExtend extension loop on MI355X: flank-bp aligned / s (= extension columns/s x flanks).

Workload (BASELINE.json configs[2] / configs[3]): synthetic N = 100,000 flanks x L = 10,000 bp,
bandwidth 40, matrix 14p43g, shared 1,500 bp ancestor at 14 % divergence, -stopafter L so that all L
columns of the right extension are executed.  One "step" = one full pass of the extension loop
(ramx_dev_run_direction) over the flank set already resident in HBM.

N > 1: one process per GPU (torch.distributed launcher), flanks sharded over ranks.  Default is STRONG scaling --
BASELINE configs[3]: the same 100,000 flanks split over the ranks, one vote exchange per column; `--scaling weak`
gives every rank its own 100,000 flanks.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from repeatafterme_amd import _lib  # noqa: E402  (loads libramx before torch: see DESIGN.md "runtime")
from repeatafterme_amd.datamodel import ExtendParams  # noqa: E402
from repeatafterme_amd.device import Device, resolve_flanks  # noqa: E402
from repeatafterme_amd.scoring import get_matrix  # noqa: E402
from repeatafterme_amd.synth import result_digest as synth_digest, synth_family, writeback_from_trim  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue: 256 CUs x 4 SIMDs, 2.4 GHz, a wave64 VALU instruction every 2 cycles per SIMD (MI355X_MICROARCH.md,
# "Wave scheduling" / cycle constants).  tools/microbench/valu_rate.hip shows that only part of the integer VALU set
# reaches that rate on gfx950 (profiles/r02_valu_rate.log); the roofline keeps the guide's 2-cycle peak and reports the
# instruction-mix-weighted figure next to it.
N_SIMD, CLOCK_GHZ, GUIDE_ISSUE_CYCLES = 1024, 2.4, 2.0
VALU_PEAK_GINST = N_SIMD * CLOCK_GHZ / GUIDE_ISSUE_CYCLES


def device_source_sha16() -> str:
    """sha256 (first 16 hex digits) over the device sources of libramx: what ties a PMC summary under profiles/ to the kernels
    that are being timed (the library itself is rebuilt on every box; the sources travel)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "repeatafterme_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "ramx_kernels_*.h")) + glob.glob(os.path.join(d, "ramx_*_api.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes_per_flank_column(W: int) -> float:
    """read prev row + write curr row + 1/4 B base + 16 B (high,pos)."""
    B = 2 * W + 1
    return 16.0 * B + 16.25


def cpu_baseline(fs, p: ExtendParams, n_sample: int, cols: int):
    """Times the reference (oracle/_ref) or, failing that, the oracle port, single-threaded."""
    from oracle import pyoracle as po
    from repeatafterme_amd.datamodel import new_master
    sub = fs.cores.subset(slice(0, n_sample))
    hi = int(sub.upper.max()) + 1
    seq = fs.sequence[:hi]
    pp = po.Params(bandwidth=p.bandwidth, cappenalty=p.cappenalty, minimprovement=p.minimprovement, L=cols,
                   when_to_stop=cols, l=1, gapopen=p.gapopen, gapextn=p.gapextn, matrix=p.matrix)
    m = new_master(cols)
    kind = "reference" if po.have_ref() else "port"
    t0 = time.perf_counter()
    if kind == "reference":
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(1)
        os.dup2(devnull, 1)     # the reference prints its limit warning on stdout
        try:
            r = po.ref_extend(1, sub, seq, m, pp)
        finally:
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        rows = cols
    else:
        r = po.oracle_extend(1, sub, seq, m, pp)
        rows = r.rows_executed
    dt = time.perf_counter() - t0
    return {"value": rows * n_sample / dt, "unit": "flank-bp/s", "cores": 1, "host": host_cpu(), "kind": kind,
            "sample": f"first {n_sample} flanks of the same set, {rows} columns, right extension, "
                      f"W={p.bandwidth}, single thread, {dt:.1f} s",
            "columns_per_sec_at_sample_N": rows / dt}, m, sub


def self_launch(n: int) -> int:
    """Started as `python3 bench.py --gpus N` (N > 1) without a launcher: run the N ranks as children of a
    torch.distributed.run child (fresh processes: this parent never initialises HIP, and nothing is exec'ed after a
    HIP call) and return its exit code.  Fewer devices than ranks is an error, never a silent 1-GPU run; the only
    exception is the explicit over-subscribed rehearsal RAMX_BENCH_BACKEND=gloo (several ranks on one GPU, mailboxes
    through the same code path, host collectives through gloo because RCCL refuses two ranks per device)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()            # does not create a HIP context
    if os.environ.get("RAMX_BENCH_BACKEND", "nccl") == "nccl" and ndev < n:
        print(f"bench: --gpus {n} but only {ndev} device(s) visible on this node", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # The contract is ONE JSON line on stdout: whatever else the ranks' libraries write there (gloo announces its
    # connections on stdout) goes to stderr instead.
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def host_cpu():
    """CPU model and core count of the host the cpu_baseline runs on."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"model": model, "logical_cpus": os.cpu_count(), "usable_cpus": usable}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--flanks", type=int, default=100000, help="flanks (total under strong scaling, per GPU under weak scaling)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the --flanks set is split over the ranks (BASELINE configs[3]); weak = --flanks per rank")
    ap.add_argument("--no-seam1", action="store_true", help="skip the seam-1 (ramx_extend_flat, host buffers) timing")
    ap.add_argument("--no-phases", action="store_true", help="skip the extra launch that separates aligned phase and capped tail "
                    "(profiling runs: every dispatch of the kernel is then a timed step)")
    ap.add_argument("--L", type=int, default=10000)
    ap.add_argument("--bandwidth", type=int, default=40)
    ap.add_argument("--cpu-flanks", type=int, default=20000)
    ap.add_argument("--cpu-cols", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--ragged", type=float, default=0.0,
                    help="development only: this fraction of the flanks ends early (uniform in [0, L)), so that waves "
                         "take the far-end-masked band; the headline workload uses full-length flanks (0)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))        # `python3 bench.py --gpus N` with no launcher: start the N ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a different rank count")

    import torch
    if world > 1 and os.environ.get("RAMX_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench: {world} ranks asked for, {torch.cuda.device_count()} devices visible")
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        backend = os.environ.get("RAMX_BENCH_BACKEND", "nccl")   # gloo only for the over-subscribed 1-GPU rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank % torch.cuda.device_count()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    W, L = args.bandwidth, args.L
    strong = world > 1 and args.scaling == "strong"
    if strong:
        base, rem = divmod(args.flanks, world)
        N = base + (1 if rank < rem else 0)          # this rank's shard of the fixed flank set
        total_flanks = args.flanks
    else:
        N = args.flanks
        total_flanks = N * world
    mat, go, ge = get_matrix("14p43g")
    p = ExtendParams(bandwidth=W, cappenalty=-90, minimprovement=27, L=L, when_to_stop=L, l=1,
                     gapopen=go, gapextn=ge, matrix=mat, matrix_name="14p43g")
    t0 = time.time()
    # strong scaling: every rank generates its own part of ONE family (same ancestor); weak: one family per rank
    fs = synth_family(N, L, W, K=1500, seed=1, shard=rank) if strong else synth_family(N, L, W, K=1500, seed=1 + rank)
    t_gen = time.time() - t0

    ndev = max(_lib.lib().ramx_device_count(), 1)
    dev = Device(local_rank % ndev)     # (% ndev only matters when ranks are over-subscribed on purpose in tests)
    peer_path = False
    if world > 1:
        on = "cuda" if dist.get_backend() == "nccl" else "cpu"
        if dist.get_backend() == "nccl":
            uid = torch.zeros(128, dtype=torch.uint8, device=on)
            if rank == 0:
                uid = torch.from_numpy(dev.unique_id().copy()).to(on)
            dist.broadcast(uid, 0)
            dev.comm_init(uid.cpu().numpy(), rank, world)       # RCCL communicator inside libramx
        else:
            # 1-GPU rehearsal only (RCCL refuses two ranks per device): host collectives through gloo
            def _cb(v):
                t = torch.tensor(v, dtype=torch.int64)
                dist.all_reduce(t)
                return t.tolist()
            dev.set_allreduce_callback(_cb)

        def ag_bytes(b):
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(on)
            outs = [torch.zeros(64, dtype=torch.uint8, device=on) for _ in range(world)]
            dist.all_gather(outs, t)
            return [bytes(o.cpu().numpy().tobytes()) for o in outs]

        def ar_min(v):
            t = torch.tensor([v], dtype=torch.int64, device=on)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())
        if os.environ.get("RAMX_NO_PEER") is None:
            peer_path = dev.peer_setup(rank, world, ag_bytes, ar_min, dist.barrier)   # vote exchanged from inside the kernel
    t0 = time.time()
    dev.load_library(fs.sequence)
    flanks, idx = resolve_flanks(1, fs.cores, W, L)
    if args.ragged > 0:
        rng = np.random.default_rng(7)
        arr, nx = flanks
        for i in np.nonzero(rng.random(nx) < args.ragged)[0]:
            arr[int(i)].t_hi = min(arr[int(i)].t_hi, int(rng.integers(0, L)))
    dev.begin_direction(flanks, p)
    t_upload = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def result_digest():
        # the digest format of tests/golden/fullsize_digests.json (the compiled reference's results on this workload)
        cons, th, tp = dev.download()
        ext, sc = writeback_from_trim(th, tp)
        return synth_digest(dev.last.ret, cons, ext, sc)

    infos = []
    digest0 = None
    for _ in range(args.warmup):
        dev.run_direction()
        digest0 = result_digest()
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        infos.append(dev.run_direction())   # repeatable: K(-1) re-creates the boundary state
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # outside the timed region: every pass over the same resident input must give the same consensus and the same
    # per-flank trimmed scores / positions (a race in the vote or the barrier would show up here)
    digest1 = result_digest()
    repeatable = (digest0 is None) or (digest0 == digest1)
    if not repeatable:
        raise SystemExit(f"bench: results of the timed passes differ from the warm-up pass ({digest0} vs {digest1})")

    cols = sum(i.rows_executed for i in infos)
    value = cols * total_flanks / dt
    # average launch duration of the dominant kernel, measured live with HIP events on libramx's own stream over
    # the timed region (ev_begin .. ev_end bracket the column loop inside ramx_dev_run_direction) / launches.
    # It includes the inter-launch gaps, so it can only under-state `achieved`; the event-bracketed samples of
    # single launches are reported next to it.
    rows = sum(i.rows_executed for i in infos)
    loop_ms = float(sum(i.loop_ms for i in infos))
    persistent = all(i.persistent for i in infos)
    per_col_bytes = algorithmic_bytes_per_flank_column(W) * N          # one column over this GPU's flanks
    lanes = infos[0].lanes_per_flank if infos else 1
    if persistent:
        # ONE launch per step processes N flanks x L columns; its duration is the event-timed loop
        packed = bool(infos and infos[0].packed_rows > 0)
        kernel = (f"ramx_packed_kernel<{W}>" if packed else f"ramx_persistent_kernel<{W}>") if lanes == 1 else f"ramx_cp_kernel<{W},{lanes},device-wide>"
        n_launch = len(infos)
        kavg_ms = loop_ms / n_launch
        abytes = per_col_bytes * rows / n_launch
    else:
        kernel = "ramx_column_kernel<false,false,256>"
        n_launch = rows
        kavg_ms = loop_ms / max(rows, 1)
        abytes = per_col_bytes
    alg_gbps = abytes / (kavg_ms * 1e-3) / 1e9 if kavg_ms > 0 else 0.0
    us_col = loop_ms * 1e3 / max(rows, 1)
    # PMC figures are valid only for the configuration they were collected on: profiles/pmc_summary.json (the headline
    # lane-per-flank configuration) and the per-configuration list profiles/pmc_configs.json (cell-parallel shapes)
    traffic = None
    valu = None
    insts_per_col = None
    full_band_insts = None
    pmc_source = None
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc) and world == 1:
        try:
            doc = json.load(open(pmc))
            ent = doc.get("persistent" if persistent else "column", {})
            cfg = doc.get("config", {"flanks": 100000, "bandwidth": 40})
            # the counters describe ONE build: the summary names the device sources it was collected on
            if doc.get("device_source_sha16") != device_source_sha16():
                pmc_source = "profiles/pmc_summary.json is from other device sources (%s): not used" % doc.get("device_source_sha16")
            elif cfg.get("flanks") == N and cfg.get("bandwidth") == W and lanes == 1 and cfg.get("L", L) == L:
                per_col = ent.get("hbm_bytes_per_column")
                fit = doc.get("persistent_fit")
                if persistent and fit:
                    # two PMC launches of different length (tools/pmc_traffic_fit.sh): rows in and out once per launch +
                    # the base stream per column
                    traffic = fit["fixed_bytes_per_launch"] + fit["bytes_per_column"] * rows / n_launch
                elif per_col is not None:
                    traffic = per_col * rows / n_launch if persistent else per_col
                cnt = ent.get("avg_per_dispatch", {})
                cols_prof = ent.get("columns_per_launch") or (ent["hbm_bytes_per_launch"] / ent["hbm_bytes_per_column"] if ent.get("hbm_bytes_per_column") else None)
                if cnt.get("SQ_INSTS_VALU") and cols_prof:
                    insts_per_col = cnt["SQ_INSTS_VALU"] / (cols_prof if persistent else 1.0)
                    full_band_insts = ent.get("full_band_insts_per_column") if persistent else None
                    pmc_source = "profiles/pmc_summary.json"
        except Exception:
            traffic = None
    pmc2 = os.path.join(ROOT, "profiles", "pmc_configs.json")
    if insts_per_col is None and os.path.exists(pmc2):
        try:
            for ent in json.load(open(pmc2)).get("entries", []):
                if (ent.get("flanks_per_rank") == N and ent.get("bandwidth") == W and ent.get("lanes_per_flank") == lanes
                        and bool(ent.get("persistent")) == bool(persistent) and ent.get("ranks", 1) == world):
                    insts_per_col = ent["SQ_INSTS_VALU_per_column"]
                    if ent.get("hbm_bytes_per_column") is not None:
                        traffic = ent["hbm_bytes_per_column"] * rows / n_launch if persistent else ent["hbm_bytes_per_column"]
                    pmc_source = "profiles/pmc_configs.json:" + ent.get("tag", "")
                    break
        except Exception:
            pass
    if insts_per_col:
        ginst = insts_per_col / (us_col * 1e-6) / 1e9           # wave64 VALU instructions per second, whole chip
        mix = None
        try:
            mix = json.load(open(os.path.join(ROOT, "profiles", "r04_valu_mix_packed.json")))
        except Exception:
            pass
        valu = {"insts_per_column": insts_per_col, "G_wave_inst_per_sec": ginst, "pmc_source": pmc_source,
                "peak_guide_2cycle": VALU_PEAK_GINST, "frac_guide_2cycle": ginst / VALU_PEAK_GINST}
        if full_band_insts:
            # the LEAN band skips work that provably cannot matter: what the same columns would cost with the full band everywhere
            valu["full_band_insts_per_column"] = full_band_insts
            valu["frac_if_every_column_ran_the_full_band"] = full_band_insts / (us_col * 1e-6) / 1e9 / VALU_PEAK_GINST
        if mix and persistent and lanes == 1:
            # instruction-mix-weighted issue cost of the band (static mix of the hot blocks x measured ticks per opcode class)
            tpi = mix["hot_blocks_ticks_per_valu"]
            valu["measured_ticks_per_inst"] = tpi
            valu["frac_at_measured_issue_cost"] = insts_per_col * tpi / (N_SIMD * CLOCK_GHZ * 1e3 * us_col)
            valu["note"] = ("only v_add/sub/and/or/mov (VGPR operands) issue every 2 cycles on gfx950; v_max3, SDWA, DPP, v_lshl_or, "
                            "v_cndmask take 4 (profiles/r02_valu_rate.log): at the kernel's own mix the chip's issue slots are this "
                            "fraction busy over the whole column, barrier wait and the CUs without a workgroup included")
        roof = {"bound": "valu", "achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G wave64-inst/s", "frac": ginst / VALU_PEAK_GINST}
    elif persistent:
        # a register-resident launch with no instruction count for THIS configuration: the rows never move, so there is no
        # honest byte or instruction rate to quote -- say what was measured (time per column) and nothing else
        roof = {"bound": "latency" if lanes > 1 else "valu", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "G wave64-inst/s",
                "frac": None, "note": (pmc_source or "no PMC instruction count collected for this (flanks, bandwidth, lanes per flank, ranks)")
                                      + "; us_per_column is the measured quantity"}
    else:
        # streaming kernel: the rows really are read once and written once per column
        roof = {"bound": "hbm", "achieved": (traffic / (kavg_ms * 1e-3) / 1e9) if (traffic and kavg_ms > 0) else alg_gbps,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None}
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
    roof.update({"traffic": traffic, "kernel": kernel, "kernel_avg_us": kavg_ms * 1e3, "launches_timed": n_launch, "us_per_column": us_col,
                 "valu": valu,
                 "hbm": {"algorithmic_bytes_per_flank_column": algorithmic_bytes_per_flank_column(W),
                         "algorithmic_bytes_per_launch": abytes, "algorithmic_GBps": alg_gbps,
                         "algorithmic_over_peak": alg_gbps / HBM_PEAK_GBS,
                         "measured_traffic_GBps": (traffic / (kavg_ms * 1e-3) / 1e9) if (traffic and kavg_ms > 0) else None,
                         "note": ("SURVEY 8(d) bytes that are NOT moved: the rows stay in registers / LDS for the whole launch, so the "
                                  "algorithmic rate may exceed the 8 TB/s peak; `traffic` is what the PMC counters saw")
                                 if persistent else "streaming kernel: rows read once and written once per column"}})

    # ---- outside the timed region: which regime was measured.  With K = 1,500 shared columns and stopafter = L the timed launch
    # is two regimes: the ALIGNED phase (every flank still aligns: full rows -- the regime of every real run, which stops ~100
    # columns behind the end of the alignment) and the CAPPED TAIL (flanks at their caps: LEAN rows, exact and digest-checked, but
    # cheaper).  One more launch over the first K columns separates them.
    phases = None
    if world == 1 and persistent and lanes == 1 and L > 1500 and args.ragged == 0 and not args.no_phases:
        K = 1500
        pk = ExtendParams(bandwidth=W, cappenalty=p.cappenalty, minimprovement=p.minimprovement, L=K, when_to_stop=K, l=1,
                          gapopen=go, gapextn=ge, matrix=mat, matrix_name="14p43g")
        dev.begin_direction(flanks, pk)
        dev.run_direction()
        best_k = min(dev.run_direction().loop_ms for _ in range(3))
        dev.begin_direction(flanks, p)                      # (the timed configuration again, for whatever follows)
        full_ms = loop_ms / n_launch
        i0 = infos[0]
        phases = {"aligned_columns": K, "aligned_us_per_column": best_k * 1e3 / K,
                  "tail_columns": L - K, "tail_us_per_column": (full_ms - best_k) * 1e3 / (L - K),
                  "packed_rows": int(i0.packed_rows), "lean_rows_of_the_first_wave": int(i0.lean_rows),
                  "rows_computed_twice_after_a_wrong_guess": int(i0.respeculated_rows),
                  "note": "aligned = a launch over the first 1,500 columns alone (every row FULL); tail = the rest of the timed launch "
                          "(rows LEAN once the flanks sit at their caps); `value` is the whole launch"}
    # ---- outside the timed region: seam 1 (host buffers in, host results out) on the same set ----------------------
    seam1 = None
    if world == 1 and not args.no_seam1:
        from repeatafterme_amd.datamodel import new_master
        from repeatafterme_amd.extend import extend_alignment
        c1 = fs.cores.copy(); m1 = new_master(L)
        t0 = time.perf_counter(); r_right = extend_alignment(1, c1, fs.sequence, m1, p); t_r = time.perf_counter() - t0
        t0 = time.perf_counter(); r_left = extend_alignment(0, c1, fs.sequence, m1, p); t_l = time.perf_counter() - t0
        s_cols = r_right.rows_executed + r_left.rows_executed
        seam1 = {"what": "ramx_extend_flat right then left, host arrays in / results out: flatten + 1-byte library upload + pack + loop + download",
                 "ms_right": t_r * 1e3, "ms_left": t_l * 1e3, "columns": s_cols, "columns_per_sec": s_cols / (t_r + t_l),
                 "flank_bp_per_sec": (r_right.rows_executed * r_right.n_extendable + r_left.rows_executed * r_left.n_extendable) / (t_r + t_l),
                 "loop_ms_right": r_right.loop_ms, "prep_ms_right": r_right.prep_ms}
    # ---- ... and a REAL run: the same set with the reference's default -stopafter 100 (what util/extend-stk.pl:365 runs): the
    # loop stops 100 columns behind the end of the alignment, so prep / download are a large part of the call
    real_run = None
    if world == 1 and not args.no_seam1 and args.ragged == 0:
        from repeatafterme_amd.datamodel import new_master
        from repeatafterme_amd.extend import extend_alignment
        pr = ExtendParams(bandwidth=W, cappenalty=p.cappenalty, minimprovement=p.minimprovement, L=L, when_to_stop=100, l=1,
                          gapopen=go, gapextn=ge, matrix=mat, matrix_name="14p43g")
        best = None
        for _ in range(2):
            c1 = fs.cores.copy(); m1 = new_master(L)
            t0 = time.perf_counter(); rr = extend_alignment(1, c1, fs.sequence, m1, pr); t_r = time.perf_counter() - t0
            t0 = time.perf_counter(); rl = extend_alignment(0, c1, fs.sequence, m1, pr); t_l = time.perf_counter() - t0
            if best is None or t_r + t_l < best[0]:
                best = (t_r + t_l, t_r, t_l, rr, rl)
        _, t_r, t_l, rr, rl = best
        real_run = {"what": "ramx_extend_flat right then left with -stopafter 100 (the wrapper's call): flatten + library on the device + pack + "
                            "loop + download + write-back, best of 2",
                    "ms_right": t_r * 1e3, "ms_left": t_l * 1e3, "columns_right": rr.rows_executed, "columns_left": rl.rows_executed,
                    "ret_right": rr.ret, "loop_ms_right": rr.loop_ms, "prep_ms_right": rr.prep_ms,
                    "loop_us_per_column": rr.loop_ms * 1e3 / max(rr.rows_executed, 1),
                    "flank_bp_per_sec_incl_prep_and_download": rr.rows_executed * rr.n_extendable / t_r if t_r > 0 else None}
    # ---- outside the timed region: BASELINE configs[1] (N = 1,000 x L = 2,000: a parity configuration, not the bench
    # line) through seam 1 -- the size class where a column is a latency chain, served by the cell-parallel kernel ------
    cfg1 = None
    if world == 1 and not args.no_seam1 and args.flanks == 100000 and W == 40:
        from repeatafterme_amd.datamodel import new_master
        from repeatafterme_amd.extend import extend_alignment
        f1 = synth_family(1000, 2000, W, K=1500, seed=3)
        p1 = ExtendParams(bandwidth=W, cappenalty=p.cappenalty, minimprovement=p.minimprovement, L=2000, when_to_stop=2000, l=1,
                          gapopen=go, gapextn=ge, matrix=mat, matrix_name="14p43g")
        best = None
        for _ in range(3):
            c1 = f1.cores.copy(); m1 = new_master(2000)
            r1 = extend_alignment(1, c1, f1.sequence, m1, p1)
            us = 1e3 * r1.loop_ms / max(r1.rows_executed, 1)
            best = us if best is None else min(best, us)
        cfg1 = {"workload": "synthetic N=1000 flanks x L=2000 bp, bandwidth=40, right extension, all L columns (BASELINE configs[1])",
                "us_per_column": best, "columns": r1.rows_executed, "lanes_per_flank": r1.lanes_per_flank, "one_launch": bool(r1.persistent),
                "columns_computed_twice_after_a_wrong_guess": r1.respeculated_rows,
                "flank_bp_per_sec": 1000 * 1e6 / best if best else None}
        gold1 = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")
        if os.path.exists(gold1) and "cfg2" in json.load(open(gold1)):
            g1 = json.load(open(gold1))["cfg2"]
            d1 = synth_digest(r1.ret, m1[2001:4001], c1.right_len, c1.score)
            cfg1["result_sha1"] = d1
            cfg1["equals_reference_digest"] = bool(d1 == g1["sha1"])
            if d1 != g1["sha1"]:
                raise SystemExit(f"bench: configs[1] result digest {d1} differs from the compiled reference's {g1['sha1']}")
    ranks_seen = dist.get_world_size() if dist is not None else 1       # size of the communicator the vote crossed
    if dist is not None and dist.get_backend() == "nccl":
        ranks_seen = min(ranks_seen, dev.comm_size())                      # RCCL communicator inside libramx
    if world == 1:
        transport = "none (single GPU)"
    elif peer_path and persistent:
        transport = "in-kernel mailboxes, " + ("peer device memory (xGMI)" if getattr(dev, "peer_kind", None) == "device" else "registered host shared memory (PCIe)")
    else:
        transport = "RCCL all-reduce between column launches" if dist.get_backend() == "nccl" else "host all-reduce (gloo rehearsal hook) between column launches"
    checks = {"result_sha1": digest1, "same_as_warmup_pass": bool(digest0 is not None)}
    # reference parity at full size: the compiled reference's digest of this exact workload (tests/golden/make_fullsize_digest.py)
    gold = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")
    if world == 1 and os.path.exists(gold) and args.ragged == 0:
        for name, g in json.load(open(gold)).items():
            w = g["workload"]
            if (w["n"], w["L"], w["W"], w["K"], w["seed"]) == (N, L, W, 1500, 1) and g["when_to_stop"] == L:
                checks["reference_digest"] = g["sha1"]
                checks["equals_reference_digest"] = bool(digest1 == g["sha1"])
                if not checks["equals_reference_digest"]:
                    raise SystemExit(f"bench: result digest {digest1} differs from the compiled reference's {g['sha1']} ({name})")
    out = {
        "metric": "flank_bp_aligned_per_sec (extension columns/s x flanks)", "value": value, "unit": "flank-bp/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": (None if world == 1 else ("strong" if strong else "weak")), "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": (f"synthetic N={total_flanks} flanks" + (f" split over {world} GPUs" if strong else (" per GPU" if world > 1 else ""))
                                + f" x L={L} bp, bandwidth={W}, matrix 14p43g, "
                                f"K=1500 @14% divergence, right extension, stopafter=L (all L columns)"),
                   "flanks_this_rank": N,
                   "flanks_total": total_flanks, "columns_per_step": cols // max(args.steps, 1),
                   "parallelism": (f"flank-sharded x{world}, per-column vote "
                                   + (("exchanged inside the persistent kernels (mailboxes in "
                                       + ("peer device memory over xGMI)" if getattr(dev, "peer_kind", None) == "device"
                                          else "registered host shared memory over PCIe)")) if (peer_path and persistent)
                                      else "all-reduced with RCCL between column launches")) if world > 1 else "single GPU"},
        "columns_per_sec": cols / dt,
        "ranks_seen": ranks_seen, "transport": transport,
        "checks": checks,
        # ALGORITHMIC: 4 candidate rows per cell and column; LEAN rows evaluate none of them (see `phases`)
        "algorithmic_cell_updates_per_sec": cols / dt * total_flanks * (2 * W + 1) * 4,
        "phases": phases,
        "roofline": roof,
        "seam1": seam1,
        "real_run": real_run,
        "configs1_n1000": cfg1,
        "setup": {"synth_s": t_gen, "upload_pack_s": t_upload},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        cb, m_cpu, sub = cpu_baseline(fs, p, min(args.cpu_flanks, N), args.cpu_cols)
        out["cpu_baseline"] = cb
        out["gpu_over_cpu"] = value / cb["value"]
        # the same sample on the GPU (outside every timed region): consensus, lengths and scores must equal the CPU result
        from repeatafterme_amd.datamodel import new_master
        from repeatafterme_amd.extend import extend_alignment
        g = fs.cores.subset(slice(0, min(args.cpu_flanks, N)))
        pg = ExtendParams(bandwidth=W, cappenalty=p.cappenalty, minimprovement=p.minimprovement, L=args.cpu_cols,
                          when_to_stop=args.cpu_cols, l=1, gapopen=go, gapextn=ge, matrix=mat, matrix_name="14p43g")
        m_gpu = new_master(args.cpu_cols)
        hi = int(g.upper.max()) + 1
        extend_alignment(1, g, np.ascontiguousarray(fs.sequence[:hi]), m_gpu, pg)
        same = bool(np.array_equal(m_gpu, m_cpu) and np.array_equal(g.right_len, sub.right_len) and np.array_equal(g.score, sub.score))
        out["checks"]["gpu_equals_cpu_on_baseline_sample"] = same
        if not same:
            raise SystemExit("bench: the GPU result on the cpu_baseline sample differs from the CPU result")
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
