"""bench.py's launcher contract, the part that needs no GPU: `python3 bench.py --gpus N` with fewer than N devices is an error
(exit 2, nothing on stdout), never a silent one-GPU run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_more_ranks_than_devices_is_an_error():
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() > 0 else 2
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RAMX_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert r.stdout.strip() == ""
    assert "device(s) visible" in r.stderr
