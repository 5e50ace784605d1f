"""world_size-2 gloo tests of the N>1 path (one process per rank, rendezvous on 127.0.0.1).

CPU (-m "not gpu"): bench.py's rendezvous / barrier / max-over-ranks / rank-0 reporting, and the caller-side exchange of a
sharded proof - partial result buffers all-gathered over gloo and combined by hg_shard_combine_host - against Python integers.
GPU (-m gpu): the real two-process sharded prove (tests/dist_worker.py "prove"), both ranks on device 0, compared with the oracle."""
import json
import os
import subprocess
import sys

import pytest

from hglib import ROOT


def _launch(script_args, port, timeout=600):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_two_ranks_gloo():
    out = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--selftest-dist"], 29517, 300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # only rank 0 reports
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["distinct_witness_seeds"]
    # max over ranks: rank 1 sleeps 20 ms per step
    assert d["ms_per_step"] >= 19.0
    assert abs(d["value"] - d["ms_per_step"] / 2) < 1e-9
    assert d["exchange_selftest"] == "identical"   # partial buffers all-gathered over gloo, combined, compared with Python integers


def test_sharded_exchange_two_ranks_gloo():
    out = _launch([os.path.join(ROOT, "tests", "dist_worker.py"), "combine"], 29519, 300)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-2000:])
    assert out.stdout.count("COMBINE OK") == 2, out.stdout


def test_bench_collective_deadline_exits_nonzero():
    """bench.py's guard around the first collective: a block that outlives its deadline ends the process with exit code 3."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.deadline(0.2, 'selftest'):\n    time.sleep(5)\nprint('NOT REACHED')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "NOT REACHED" not in r.stdout and "did not finish within" in r.stderr
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.deadline(5, 'selftest'):\n    pass\ntime.sleep(0.3); print('FINE')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "FINE" in r.stdout


@pytest.mark.gpu
def test_sharded_prove_two_processes_gloo_matches_the_oracle():
    out = _launch([os.path.join(ROOT, "tests", "dist_worker.py"), "prove"], 29521, 900)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert out.stdout.count("PROVE OK") == 2, out.stdout


@pytest.mark.gpu
def test_sharded_prove_two_processes_per_rank_tables_at_the_headline_size():
    """BASELINE config 4's size with per-rank tables in two real processes (tests/dist_worker.py "prove_c3"): every process evaluates
    the cone of its own share only, the exchange goes over gloo, both ranks reproduce the oracle's transcript for two witnesses."""
    out = _launch([os.path.join(ROOT, "tests", "dist_worker.py"), "prove_c3"], 29523, 1500)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert out.stdout.count("PROVE_C3 OK") == 2, out.stdout


@pytest.mark.gpu
def test_round_by_round_prover_two_processes_one_allreduce_per_round():
    """SURVEY 8(e)'s per-round exchange across two real processes (tests/dist_worker.py "prove_seq"): mode 3, the round sums
    all-reduced over gloo once per sum-check round, both ranks reproduce the oracle's mode-3 transcript."""
    out = _launch([os.path.join(ROOT, "tests", "dist_worker.py"), "prove_seq"], 29525, 900)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert out.stdout.count("PROVE_SEQ OK") == 2, out.stdout


@pytest.mark.gpu
def test_round_by_round_prover_two_processes_node_ownership_at_the_headline_size():
    """The memory-sharded form of the per-round exchange (round 6; tests/dist_worker.py "prove_seq_own"): n=32768 k=16, mode 3, two real
    processes over gloo, per-rank tables and node ownership; both proofs are the CPU oracle's, resident_bytes < full on every rank."""
    out = _launch([os.path.join(ROOT, "tests", "dist_worker.py"), "prove_seq_own"], 29527, 1500)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert out.stdout.count("PROVE_SEQ_OWN OK") == 2, out.stdout


@pytest.mark.gpu
def test_sharded_prove_over_real_rccl_two_gpus():
    """hg_prove_sharded over the library's own RCCL communicator with TWO ranks on TWO GPUs (tests/dist_worker.py "prove_rccl"; the
    all-reduce of the result buffer crosses xGMI), n=4096 k=2 and n=32768 k=16 with per-rank tables, both ranks' proofs equal to the
    oracle's. Skipped on a one-GPU box: RCCL refuses two ranks on one device ("Duplicate GPU detected"), so there is nothing to fake.
    The device count is read in a child process - the test process itself never initialises a GPU ahead of the torchrun launch."""
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import __graft_entry__ as e; print(e.load_package().device_count())" % ROOT],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    if int(r.stdout.strip().splitlines()[-1]) < 2:
        pytest.skip("needs two GPUs (this box has %s)" % r.stdout.strip().splitlines()[-1])
    out = _launch([os.path.join(ROOT, "tests", "dist_worker.py"), "prove_rccl"], 29527, 1800)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert out.stdout.count("PROVE_RCCL OK") == 2, out.stdout
