"""world_size-2 gloo test (CPU) of the N>1 path of bench.py: rendezvous on 127.0.0.1, barrier, max-over-ranks
timing, one JSON line from rank 0, one independent witness seed per rank (weak scaling, no data-path collective)."""
import json
import os
import subprocess
import sys

from hglib import ROOT


def test_bench_two_ranks_gloo():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--selftest-dist"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # only rank 0 reports
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["distinct_witness_seeds"]
    # max over ranks: rank 1 sleeps 20 ms per step
    assert d["ms_per_step"] >= 19.0
    assert abs(d["value"] - d["ms_per_step"] / 2) < 1e-9
