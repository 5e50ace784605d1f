#!/usr/bin/env python3
"""CPU baseline of kind "reference": times the reference's own rayon prover on THIS host, when that is possible.

Needs what the authoring image does not have: `cargo` (nightly), network access for the reference's git dependencies and
a checkout of nulltea/hyper-greco at $HYPER_GRECO. bench.py calls `measure()` first and falls back to the oracle port
(kind "port") when it returns None. The workload is the same synthetic witness the GPU proves, written in the
reference's JSON format (scripts/witness_to_json.py) because the n=32768 fixture is a missing blob
[REF .MISSING_LARGE_BLOBS:5]. Timing = the "GKR prove" span [REF bfv-gkr/src/sk_encryption_circuit.rs:455-457] printed by
tracing-forest [REF bfv-gkr/src/test.rs:9-17], 1 warm + 3 timed runs, median."""
import os
import re
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BITS = {(1024, 1): 27, (2048, 1): 52, (4096, 2): 55, (8192, 4): 55, (16384, 8): 54, (32768, 16): 59}
_UNIT = {"s": 1e3, "ms": 1.0, "µs": 1e-3, "us": 1e-3, "ns": 1e-6}


def parse_span_ms(text, name="GKR prove"):
    """tracing-forest prints `GKR prove [ 1.88s | 37.12% / 99.31% ]`: first duration of the named span, in ms."""
    m = re.search(re.escape(name) + r"\s*\[\s*([0-9.]+)\s*(s|ms|µs|us|ns)\b", text)
    return float(m.group(1)) * _UNIT[m.group(2)] if m else None


def measure(n, k, seed, runs=3, timeout_s=1800):
    repo = os.environ.get("HYPER_GRECO")
    if not repo or not shutil.which("cargo") or (n, k) not in BITS:
        return None
    bits = BITS[(n, k)]
    fixture = os.path.join(repo, "bfv-gkr", "src", "data", "goldilocks", f"sk_enc_{n}_{k}x{bits}_65537.json")
    # The reference test reads its witness from this fixed path. A shipped fixture is never lost: it is moved aside first and put
    # back whatever happens (a half-written file included); a fixture that did not exist (the n = 16384 / 32768 blobs) is removed again.
    backup = fixture + ".hg-backup"
    if os.path.exists(backup):
        return None   # (an earlier run died between the two renames: do not touch anything, a human should look)
    had_fixture = os.path.exists(fixture)
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import witness_to_json
        sys.path.insert(0, ROOT)
        import __graft_entry__ as entry
        hg = entry.load_package()
        import json
        w = hg.Witness.synthetic(hg.params_builtin(n, k), seed)
        if had_fixture:
            os.replace(fixture, backup)
        with open(fixture, "w") as f:
            json.dump(witness_to_json.arrays_to_args(n, k, w.arrays()), f)
        test = f"test_sk_enc_valid_goldilocks_{n}_{k}x{bits}_65537"
        cmd = ["cargo", "test", "-r", "-p", "bfv-gkr", test, "--", "--nocapture"]
        times = []
        for i in range(runs + 1):
            out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=timeout_s)
            if out.returncode != 0:
                return None
            ms = parse_span_ms(out.stdout + out.stderr)
            if ms is None:
                return None
            if i:
                times.append(ms)
        cores = os.cpu_count() or 1
        return {"value": round(statistics.median(times), 3), "unit": "ms", "cores": cores, "kind": "reference",
                "sample": f"cargo test -r {test} (same synthetic witness as the GPU run, written by scripts/witness_to_json.py): 'GKR prove' span, "
                          f"1 warm + {runs} timed runs, median; rayon on all {cores} host cores",
                "runs_ms": [round(t, 3) for t in times]}
    except (OSError, subprocess.SubprocessError, ValueError, ImportError) as ex:
        sys.stderr.write(f"[reference_baseline] not measured: {ex}\n")
        return None
    finally:
        try:
            if had_fixture:
                if os.path.exists(backup):
                    os.replace(backup, fixture)
            elif os.path.exists(fixture):
                os.remove(fixture)
        except OSError as ex:
            sys.stderr.write(f"[reference_baseline] could not restore {fixture}: {ex}\n")


if __name__ == "__main__":
    print(measure(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x4752454330))
