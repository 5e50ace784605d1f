#!/usr/bin/env python3
"""Headline benchmark: GKR prove of the BFV sk-encryption circuit, n=32768 k=16 Goldilocks (BASELINE.json
configs[2]), on N MI355X GPUs of one node.

A "step" = one GKR prove (the reference's "GKR prove" span [REF bfv-gkr/src/sk_encryption_circuit.rs:455-457],
plus the output-claim evaluation :444-448) of a synthetic witness whose node tables are already resident in
HBM - a DIFFERENT witness every step (`--witnesses`, default 4 seeds in rotation: the reference proves each witness once,
test.rs:37-38), every proof of the timed region checked afterwards against the walked proof of the same seed and one of
them against the CPU oracle. N > 1, default `--mode shard`: ONE proof is sharded over the N GPUs (strong scaling): every rank holds the
same witness, runs its share (the Lasso node split by memory, the node reductions dealt out whole) and
ONE RCCL all-reduce of the scalar result buffer per proof, issued by the library on the prover stream, is the only exchange;
`value` = max-over-ranks step time = ms per proof.
`--mode dp`: every rank proves its own independent witness (weak scaling, no collective), `value` = step time / N.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP events on
the prover stream over K timed proves run back to back with the K proves behind `value`: see "timed regions" in main()) and
`cpu_baseline` (the CPU oracle = a port, timed on this host)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the CPU baseline shares this process with the product library's OpenMP runtime: idle teams must sleep, not spin (read when the
# runtimes load; measured on 8 cores: the Vanilla-node share of the oracle prove 630 ms spinning vs 142 ms passive)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
# profile class of the library -> kernel symbol in rocprofv3 output. The class with the largest share of the GPU time
# in a warm-up prove is the "dominant kernel" of the roofline object; its launches are then timed with HIP events
# inside the timed region.
CLASS_SYMBOL = {
    "sc_round2<grand_product,ext>": "k_st_step2(",
    "sc_round<grand_product,ext>": "k_st_step<1, hg::E2",
    "sc_round<grand_product,base>": "k_st_step<1, unsigned long",      # (two instantiations: slot form and memory form)
    "sc_round<grand_product,hash>": "k_gp_first_hash",                 # (k_gp_first_hash_slot at the headline size, k_gp_first_hash<..> otherwise)
    "sc_round2<collation,ext>": "k_col_step2(",
    "sc_round<collation,ext>": "k_st_step<0, hg::E2",
    "sc_round<collation,base>": "k_st_step<0, unsigned long",
    "sc_round<prodsum>": "k_ps_one(",
    "sc_round2<prodsum>": "k_ps_step2<",      # (two instantiations: eq-factored jobs and the others)
}
# The class `roofline` reports, FIXED per round so that `frac` is comparable between runs of one round: the class with the largest isolated
# time in this round's committed one-stream trace (profiles/r06_one_stream_kernel_trace_summary.txt: the four base-field first rounds of
# the grand-product layers, 0.32 ms per prove). Until round 5 the line named whichever of two tied classes was ahead in the run.
ROOFLINE_CLASS = "sc_round<grand_product,base>"
PMC_TAG = "r06"  # profiles/<tag>_pmc_hbm_traffic.json, <tag>_pmc_sq.json, <tag>_bn254_pmc_sq.json, <tag>_isa_mix.json: this round's committed passes
PMC_CMD = ("rocprofv3 --pmc FETCH_SIZE -- python3 scripts/prove_once.py 32768 16 2 ; rocprofv3 --pmc WRITE_SIZE -- (same): separate passes, "
           "HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md), "
           "scripts/pmc_summary.py; rocprofv3 --pmc SQ_INSTS_VALU ... -- (same), scripts/pmc_sq.py (scripts/measure_r06.sh; taken inside this run when rocprofv3 is on the box, see roofline.counters.measured)")
VALU_PEAK_G = 256 * 4 * 2.4e9 / 4 / 1e9   # CUs x SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction = 614 G wave-instr/s
VALU_MEASURED_G = 532.0                   # what v_mad_u64_u32 / VOP3 issues at on this chip (scripts/ub/ratebench.hip: 0.52 G/s per SIMD)
NSIMD = 256 * 4


class IssueModel:
    """The VALU issue roof of a kernel, priced per instruction class instead of one nominal rate (round 6). scripts/ub/ratebench.hip
    measures what every opcode the kernels are made of issues at on this chip (profiles/r06_ratebench.txt, wave-instructions per second
    and SIMD, 8 waves per SIMD, independent chains): the plain VOP1 / VOP2 integer forms (class A: v_mov_b32, v_add_u32, v_and_b32,
    v_lshrrev_b32 ...) at ~1.0 G/s, everything else (class B: v_mad_u64_u32, carries, 64-bit shifts and adds, compares, selects,
    v_fma_f64) at ~0.54 G/s - two and about four cycles of the clock the chip holds under that load. scripts/isa_census.py --json gives
    each kernel's static class mix (profiles/r06_isa_mix.json, hash-checked like the counter files). The issue time of N VALU
    wave-instructions of a kernel is N (f_A / R_A + f_B / R_B) / 1024 SIMDs; issue_frac = that over the measured duration. Static mix of the
    whole kernel body, dynamic count from SQ_INSTS_VALU: loops dominate both, prologues make it approximate."""
    CLASS_A_ROWS = ("v_mov_b32 ", "v_add_u32", "v_and_b32", "v_lshrrev_b32", "v_sub_u32", "v_xor_b32", "v_mov_b32 imm")
    SKIP_ROWS = ("(2 instr)", "(4 instr)", "v_cndmask_b32 ")   # multi-instruction rows; the VOP2 select row is a vcc-dependent chain, not an issue rate

    def __init__(self):
        self.rate_a = self.rate_b = None
        self.mix, self.note = None, None
        try:
            a, b = [], []
            for ln in open(os.path.join(ROOT, "profiles", f"{PMC_TAG}_ratebench.txt")):
                if "G wave-instr/s/SIMD" not in ln or any(x in ln for x in self.SKIP_ROWS):
                    continue
                rate = float(ln.split(" ms")[1].split("G wave-instr")[0])
                (a if any(ln.startswith(x) for x in self.CLASS_A_ROWS) else b).append(rate)
            self.rate_a, self.rate_b = sum(a) / len(a) * 1e9, sum(b) / len(b) * 1e9
        except Exception as e:  # noqa: BLE001
            self.note = f"profiles/{PMC_TAG}_ratebench.txt: {e}; "
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f"{PMC_TAG}_isa_mix.json")))
            if d.get("_meta", {}).get("code_hash") != code_hash():
                self.note = (self.note or "") + f"profiles/{PMC_TAG}_isa_mix.json: taken on code {d.get('_meta', {}).get('code_hash')}, this is {code_hash()} - refused; "
            else:
                d.pop("_meta", None)
                self.mix = d
        except Exception as e:  # noqa: BLE001
            self.note = (self.note or "") + f"profiles/{PMC_TAG}_isa_mix.json: {e}; "

    def ok(self):
        return bool(self.rate_a and self.rate_b and self.mix)

    def share_a(self, sym):
        """class-A share of the VALU instructions of every kernel whose demangled name contains `sym` (None: unknown)"""
        if not self.mix or not sym:
            return None
        key = sym.replace("hg::", "").rstrip("(")
        m = [v for k, v in self.mix.items() if key in k]
        tot = sum(v["valu"] for v in m)
        return sum(v["class_a"] for v in m) / tot if tot else None

    def seconds(self, n_valu, sym):
        """issue time of n_valu wave-instructions of kernel `sym` on the whole chip"""
        fa = self.share_a(sym)
        if not self.ok() or n_valu is None or fa is None:
            return None
        return n_valu * (fa / self.rate_a + (1.0 - fa) / self.rate_b) / NSIMD

    def describe(self):
        return {"class_a_G_per_simd": round(self.rate_a / 1e9, 3) if self.rate_a else None, "class_b_G_per_simd": round(self.rate_b / 1e9, 3) if self.rate_b else None,
                "files": [f"profiles/{PMC_TAG}_ratebench.txt", f"profiles/{PMC_TAG}_isa_mix.json"], **({"note": self.note} if self.note else {})}


def live_counter_passes(n, k, budget_s=200.0):
    """The three PMC passes behind roofline.traffic / wave_insts_per_launch, taken INSIDE this run when rocprofv3 is on the box (child
    processes, the profiled program directly after `--`; FETCH_SIZE and WRITE_SIZE in separate passes as the MI355X guide prescribes, no
    trace domain beside --pmc). Returns (dir with <tag>_pmc_hbm_traffic.json / <tag>_pmc_sq.json, note) or (None, reason): the caller then reads
    the committed files of profiles/."""
    import shutil
    import subprocess
    import tempfile
    if (n, k) != (32768, 16):
        return None, "counter passes exist for n=32768 k=16 only"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on this box"
    td = tempfile.mkdtemp(prefix="hg_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("HG_LIB", None) if False else None
    t_end = time.time() + budget_s
    prog = [sys.executable, os.path.join(ROOT, "scripts", "prove_once.py"), "32768", "16", "2"]
    sets = (("FETCH", ["FETCH_SIZE"]), ("WRITE", ["WRITE_SIZE"]),
            ("SQ", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES"]))
    try:
        for name, ctrs in sets:
            left = t_end - time.time()
            if left < 10:
                return None, "counter passes: time budget spent"
            subprocess.run([exe, "--pmc", *ctrs, "-d", os.path.join(td, name), "-o", "run", "--output-format", "csv", "--", *prog],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left, check=True)
        import glob
        f = lambda nm: glob.glob(os.path.join(td, nm, "**", "*counter_collection.csv"), recursive=True)[0]
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), f("FETCH"), f("WRITE"), os.path.join(td, f"{PMC_TAG}_pmc_hbm_traffic.json")],
                       stdout=subprocess.DEVNULL, check=True, timeout=60)
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_sq.py"), f("SQ"), "40", os.path.join(td, f"{PMC_TAG}_pmc_sq.json")],
                       stdout=subprocess.DEVNULL, check=True, timeout=60)
        # the same for the bn254 prove (config 5): VALU wave-instructions per kernel of three proves; a failure here leaves the bn254 roofline
        # to the committed file and does not take the Goldilocks passes with it
        try:
            left = t_end - time.time()
            if left > 20:
                subprocess.run([exe, "--pmc", "SQ_INSTS_VALU", "-d", os.path.join(td, "BN"), "-o", "run", "--output-format", "csv", "--",
                                sys.executable, os.path.join(ROOT, "scripts", "bn254_prove_bench.py")],
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left, check=True)
                import collections
                import csv
                per = collections.defaultdict(float)
                for r in csv.DictReader(open(f("BN"))):
                    if r["Counter_Name"] == "SQ_INSTS_VALU" and "hg::bn::" in r["Kernel_Name"]:
                        per[r["Kernel_Name"].split("(")[0]] += float(r["Counter_Value"])
                WG = ("k_bn_ntt_stage", "k_bn_ntt4_", "k_bn_gate_eval", "k_bn_lift_signed", "k_bn_lift_jobs", "k_bn_bitrev", "k_bn_scale", "k_bn_powers")  # witness generation: outside the timed prove
                wit = sum(v for kn, v in per.items() if any(w in kn for w in WG))
                json.dump({"code_hash": code_hash(), "command": "rocprofv3 --pmc SQ_INSTS_VALU -- python3 scripts/bn254_prove_bench.py (3 proves of n=32768 k=16), inside this bench run",
                           "valu_wave_insts_per_prove": sum(per.values()) / 3, "prove_valu_wave_insts_per_prove": (sum(per.values()) - wit) / 3,
                           "by_kernel_per_prove": {kn: v / 3 for kn, v in sorted(per.items(), key=lambda kv: -kv[1])}},
                          open(os.path.join(td, f"{PMC_TAG}_bn254_pmc_sq.json"), "w"))
        except Exception:  # noqa: BLE001
            pass
        return td, None
    except Exception as e:  # noqa: BLE001
        return None, f"counter passes failed ({type(e).__name__}: {e})"


# kernels of a prove_once run that are not part of a prove (witness generation, memsets)
NOT_PROVE = ("k_ntt4", "k_gate_eval", "k_lift", "__amd_rocclr", "k_output_mle")
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from code_hash import code_hash  # noqa: E402


class Counters:
    """This round's committed rocprofv3 PMC passes (PMC counters cannot be read from inside this process), valid for n=32768 k=16
    and ONLY for the code state they were taken on: a file whose `code_hash` is not the hash of the sources this process runs
    (scripts/code_hash.py) is refused, and every figure that would come from it is null with the reason in `note`."""

    def __init__(self, n, k, tag=PMC_TAG, live_dir=None, live_note=None):
        self.traffic, self.sq, self.note, self.proves = None, None, None, 2
        self.files = {}
        self.measured = "in this run (rocprofv3 child processes of bench.py)" if live_dir else f"read from profiles/ ({live_note or 'no live pass'})"
        self.dir = live_dir or os.path.join(ROOT, "profiles")
        if (n, k) != (32768, 16):
            self.note = "counter passes exist for n=32768 k=16 only"
            return
        want = code_hash()
        for attr, name in (("traffic", f"{tag}_pmc_hbm_traffic.json"), ("sq", f"{tag}_pmc_sq.json")):
            path = os.path.join(self.dir, name)
            try:
                d = json.load(open(path))
            except Exception:
                self.note = (self.note or "") + f"profiles/{name}: missing; "
                continue
            got = d.get("_meta", {}).get("code_hash")
            if got != want:
                self.note = (self.note or "") + f"profiles/{name}: taken on code {got}, this is {want} - refused; "
                continue
            self.proves = d["_meta"].get("proves") or self.proves   # counted by the pass itself (k_clear_words launches); 2 in files older than round 6
            d.pop("_meta", None)
            setattr(self, attr, d)
            self.files[attr] = f"profiles/{name}" if not live_dir else f"{name} (this run)"

    @staticmethod
    def _match(d, sym):
        return [(k, v) for k, v in d.items() if sym in k]

    def hbm_bytes(self, sym):
        """(PMC HBM bytes per launch, launches per prove) over every kernel whose name contains `sym`"""
        if not self.traffic or not sym:
            return None, None
        m = self._match(self.traffic, sym)
        ln = sum(v["launches"] for _, v in m)
        if not ln:
            return None, None
        return sum(v["launches"] * v["hbm_bytes_per_launch"] for _, v in m) / ln, ln / self.proves

    def valu(self, sym):
        """VALU wave-instructions per launch over every kernel whose name contains `sym`"""
        if not self.sq or not sym:
            return None
        m = self._match(self.sq, sym)
        ln = sum(v["launches"] for _, v in m)
        return sum(v.get("SQ_INSTS_VALU", 0.0) for _, v in m) / ln if ln else None

    def issue_seconds_of_a_prove(self, im):
        """sum over the prove's kernels of their VALU instructions priced with each kernel's own class mix (IssueModel); None if unknown"""
        if not self.sq or not im.ok():
            return None
        tot = 0.0
        for k, v in self.sq.items():
            if any(x in k for x in NOT_PROVE):
                continue
            name = k.split("(")[0].replace("hg::dev::", "").replace("hg::bn::", "bn::").replace("hg::", "").replace("void ", "")
            t = im.seconds(v.get("SQ_INSTS_VALU", 0.0), name)
            if t is None:   # a kernel the census does not know: all class B
                t = v.get("SQ_INSTS_VALU", 0.0) / im.rate_b / NSIMD
            tot += t
        return tot / self.proves

    def prove_totals(self):
        """(HBM bytes, VALU wave-instructions) of one prove: every kernel of the run but witness generation"""
        def tot(d, f):
            return sum(f(v) for k, v in d.items() if not any(x in k for x in NOT_PROVE)) / self.proves if d else None
        return tot(self.traffic, lambda v: v["launches"] * v["hbm_bytes_per_launch"]), tot(self.sq, lambda v: v.get("SQ_INSTS_VALU", 0.0))


def toolchain_probe():
    """What a pin against the real Rust prover needs, looked for on THIS host (recorded in cpu_baseline.reference_attempt)."""
    import shutil
    import socket
    import subprocess
    cargo = shutil.which("cargo") is not None
    nightly = False
    if shutil.which("rustc"):
        try:
            nightly = "nightly" in subprocess.run(["rustc", "--version"], capture_output=True, text=True, timeout=10).stdout
        except Exception:
            pass
    network = False
    try:
        socket.create_connection(("github.com", 443), timeout=2).close()
        network = True
    except Exception:
        pass
    return {"cargo": cargo, "rustc_nightly": nightly, "network": network, "reference_checkout": bool(os.environ.get("HYPER_GRECO")),
            "pin": "not attempted: " + ("no cargo on this host" if not cargo else "no network for the reference's git dependencies" if not network else
                                        "no reference checkout ($HYPER_GRECO)" if not os.environ.get("HYPER_GRECO") else "see rust/hg-shim/tests/proof_dump.rs")}


def cgroup_cpu_quota():
    """CPUs the container's cgroup grants this process (cpu.max: quota / period), None if unlimited or unknown. os.cpu_count()
    reports the machine's cores; the quota is what the baseline can actually use (exceeding it gets the process throttled)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, -(-int(q) // int(per)))
    except Exception:
        pass
    return None


def cpu_baseline(n, k, seed, budget_s=60.0, proofs_out=None):
    """CPU baseline on this host's cores, same witness as the GPU run, GKR-prove span only.
    kind "reference": the reference's own rayon prover (needs cargo + network + $HYPER_GRECO: scripts/reference_baseline.py);
    otherwise kind "port": the CPU oracle (this repo's restatement, OpenMP where the reference uses rayon). Method: the thread
    count is chosen among {16, 32, 64, 128, 256} on the small configuration, then 1 warm + 3 timed runs, median. Bounded: a small config is timed first and the largest config whose predicted time fits the
    budget is run (scaled by the ratio of Lasso rows, stated in `sample`)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    attempt = toolchain_probe()
    try:
        import reference_baseline
        ref = reference_baseline.measure(n, k, seed + n)
        if ref:
            ref["reference_attempt"] = attempt
            return ref
    except Exception:
        pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib  # CPU oracle: used here only as the thing being timed for the baseline line
    import statistics
    import __graft_entry__ as entry
    hg = entry.load_package()
    cores = os.cpu_count() or 1

    def run(nn, kk, threads):
        p = hg.params_builtin(nn, kk)
        w = hg.Witness.synthetic(p, seed + nn)
        inp = orclib.Inputs(w.arrays())
        op = orclib.params(nn, kk)
        proof, tm = orclib.prove(op, inp, threads=threads)
        if proofs_out is not None:
            proofs_out[(nn, kk)] = proof   # (same seed as witness 0 of the GPU run: main() compares the bytes)
        return tm[1]  # GKR prove span only (witness generation excluded, like the GPU number)

    def rows(nn, kk):
        return (kk + max(1, kk // 2) + 3) * 2 * nn

    # thread count: chosen on the small configuration (more threads than 128 thrash on this kind of host: a 256-thread run of
    # n=16384 was measured at 294 s against 2.2 s with 64)
    cands = sorted({min(cores, t) for t in (16, 32, 64, 128, 256)})
    run(4096, 2, cands[-1])  # warm the page cache / thread pool
    sweep = {th: run(4096, 2, th) for th in cands}
    best = min(sweep, key=sweep.get)
    t_small = sweep[best]
    ladder = [(32768, 16), (16384, 8), (8192, 4), (4096, 2)]
    for nn, kk in ladder:
        predicted = 4 * t_small * rows(nn, kk) / rows(4096, 2) / 1000.0   # four runs, linear in the Lasso rows (measured: sub-linear, so this over-predicts)
        if predicted <= budget_s or (nn, kk) == (4096, 2):
            runs = [run(nn, kk, best) for _ in range(4)][1:]   # 1 warm + 3 timed
            ms = statistics.median(runs)
            scale = rows(n, k) / rows(nn, kk)
            sample = (f"oracle GKR prove at n={nn} k={kk}, {best} OpenMP threads (best of {cands} on n=4096 k=2), 1 warm + 3 timed runs, median")
            if scale != 1:
                sample += f"; scaled x{scale:.2f} (Lasso rows ratio) to n={n} k={k}"
            return {"value": round(ms * scale, 3), "unit": "ms", "cores": best, "kind": "port", "sample": sample,
                    "measured_ms": round(ms, 3), "runs_ms": [round(r, 1) for r in runs],
                    "thread_sweep_ms_n4096": {str(t): round(v, 1) for t, v in sweep.items()}, "host_cores": cores,
                    "cgroup_cpu_quota": cgroup_cpu_quota(),   # CPUs the container may use: more threads than this only get the process throttled
                    "reference_attempt": attempt}
    return None


def max_over_ranks(elapsed, world, dist, torch, device):
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def witness_seed(base_seed, n, rank):
    return base_seed + n + 7919 * rank  # one independent witness per rank


def selftest_dist(args, rank, world, dist, torch):
    """gloo on CPU: the same rendezvous / barrier / max-over-ranks / rank-0 reporting as the GPU path, with a
    dummy timed body (each rank sleeps (rank+1) * 10 ms per step)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dist, torch, "cpu")
    seeds = [witness_seed(args.seed, args.n, r) for r in range(world)]
    # the caller-side exchange of `--mode shard` without its device part: seeded partial buffers, all-gather, hg_shard_combine_host
    import numpy as np
    import __graft_entry__ as entry
    hg = entry.load_package()
    P = 0xFFFFFFFF00000001
    part = np.random.default_rng(77 + rank).integers(0, P, size=1031, dtype=np.uint64)
    part[:8] = P - 1
    if world > 1:
        t = torch.from_numpy(part.view(np.int64).copy())
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        gathered = torch.stack(bufs).numpy().view(np.uint64)
    else:
        gathered = part[None, :]
    want = np.array([sum(int(gathered[r][i]) for r in range(world)) % P for i in range(part.size)], dtype=np.uint64)
    exchange = "identical" if np.array_equal(hg.shard_combine_host(gathered), want) else "DIFFERENT"
    if rank == 0:
        print(json.dumps({"selftest": "dist", "n_gpus": world, "steps": args.steps, "ms_per_step": elapsed / args.steps * 1e3,
                          "value": elapsed / args.steps * 1e3 / world, "distinct_witness_seeds": len(set(seeds)) == world,
                          "exchange_selftest": exchange}), flush=True)
    if world > 1:
        dist.destroy_process_group()


class deadline:
    """Bounds a block that contains a collective: when it has not finished after `seconds`, the process reports on stderr and EXITS
    with a non-zero code (os._exit from a watchdog thread: nothing is re-executed, no GPU call is made from the watchdog)."""

    def __init__(self, seconds, what):
        self.seconds, self.what, self.done = seconds, what, None

    def __enter__(self):
        if self.seconds and self.seconds > 0:
            import threading
            self.done = threading.Event()

            def watch():
                if not self.done.wait(self.seconds):
                    sys.stderr.write(f"[bench] rank {os.environ.get('RANK', '0')}: {self.what} did not finish within {self.seconds:.0f} s - exiting\n")
                    sys.stderr.flush()
                    os._exit(3)
            threading.Thread(target=watch, daemon=True).start()
        return self

    def __exit__(self, *exc):
        if self.done is not None:
            self.done.set()
        return False


def measure_end_to_end(hg, ctx, bfv, pk, witnesses, walked, args):
    """BfvEncrypt::prove as the reference's caller sees it [REF sk_encryption_circuit.rs:417-460, poly.rs:12-44]: JSON args ->
    get_inputs -> upload -> circuit.evaluate on the device -> GKR prove, every stage warmed, median of 5, a different witness
    per call. Reported in config.end_to_end, never part of `value`."""
    import statistics
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import witness_to_json
    NW = len(witnesses)
    res = {}
    # arrays -> proof: hg_prove (upload, witness generation into the context's resident tables, launch-graph replay)
    for i in range(3 * NW):   # walks, capture
        proof, tm = bfv.prove(ctx, pk, witnesses[i % NW])
        assert proof == walked[i % NW]
    runs = []
    for i in range(5):
        t0 = time.perf_counter()
        proof, tm = bfv.prove(ctx, pk, witnesses[i % NW])
        runs.append(((time.perf_counter() - t0) * 1e3, tm))
        assert proof == walked[i % NW], "hg_prove: proof differs from the resident prove of the same witness"
    runs.sort(key=lambda r: r[0])
    med = runs[len(runs) // 2]
    res["arrays_to_proof_ms"] = round(med[0], 3)
    res["stages_ms"] = {"upload": round(med[1]["upload_ms"], 3), "witness_gen_device": round(med[1]["witness_ms"], 3), "gkr_prove": round(med[1]["prove_ms"], 3)}
    # a run of witnesses through hg_prove_stream: witness i+1's upload + evaluate under witness i's prove
    run = [witnesses[i % NW] for i in range(4 * NW)]
    for i in range(2):                                                       # both table sets walk twice and record their graphs
        proofs, tm = bfv.prove_stream(ctx, pk, run)
        assert proofs == [walked[i % NW] for i in range(len(run))], "hg_prove_stream: a proof differs from the resident prove of the same witness"
    ts = []
    for i in range(5):
        proofs, tm = bfv.prove_stream(ctx, pk, run)
        ts.append(tm["total_ms"] / len(run))
    assert proofs == [walked[i % NW] for i in range(len(run))]
    res["pipelined_arrays_to_proof_ms"] = round(statistics.median(ts), 3)
    res["pipelined_run"] = len(run)
    # JSON -> arrays: the reference's fixture format, written from witness 0 (the n=32768 fixture itself is a missing blob)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "w.json")
        with open(path, "w") as f:
            json.dump(witness_to_json.arrays_to_args(args.n, args.k, witnesses[0].arrays()), f)
        res["json_bytes"] = os.path.getsize(path)
        ts = []
        for i in range(6):
            t0 = time.perf_counter()
            wj = hg.Witness.from_json(bfv.params, path)
            ts.append((time.perf_counter() - t0) * 1e3)
        proof, _ = bfv.prove(ctx, pk, wj)
        assert proof == walked[0], "the witness read back from JSON proves differently"
        res["json_parse_ms"] = round(statistics.median(ts[1:]), 3)
        res["json_parse_threads"] = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    res["json_to_proof_ms"] = round(res["json_parse_ms"] + res["arrays_to_proof_ms"], 3)
    res["note"] = "median of 5 after warm-up, a different witness per call; hg_witness_from_json + hg_prove (host arrays are pageable memory); pipelined_*: hg_prove_stream over a run of witnesses, per proof"
    return res


def measure_cold_start(hg, device, args, witness, expect):
    """The drop-in caller's sequence, COLD, as test_sk_enc_valid runs it [REF bfv-gkr/src/test.rs:31-44]: BfvEncrypt::new + setup, read the
    args file, prove ONCE, verify - in fresh contexts of this process (the HIP runtime itself is already up: hg_create of a second
    context). Two variants: with hg_warmup behind hg_setup (what a drop-in BfvEncrypt::setup calls: tables, staging and the launch graph
    recorded on a zero witness, so the first prove replays) and without (the first prove walks the protocol)."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import witness_to_json
    res = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "w.json")
        with open(path, "w") as f:
            json.dump(witness_to_json.arrays_to_args(args.n, args.k, witness.arrays()), f)
        for warm in (True, False):
            t0 = time.perf_counter()
            c = hg.Context(device)
            t1 = time.perf_counter()
            b = hg.BfvEncrypt.new(args.n, args.k)
            k = b.setup(c)
            t2 = time.perf_counter()
            wu = b.warmup(c, k) if warm else 0.0
            t3 = time.perf_counter()
            w = b.get_inputs(path)
            t4 = time.perf_counter()
            proof, tm = b.prove(c, k, w, cap=1 << 20)
            t5 = time.perf_counter()
            ok, err = hg.verify_device(c, k, w, proof)
            t6 = time.perf_counter()
            assert proof == expect, "cold start: the first proof of a fresh context differs from the resident prove of the same witness"
            assert ok, err
            r = {"hg_create_ms": round((t1 - t0) * 1e3, 2), "setup_ms": round((t2 - t1) * 1e3, 2), "witness_from_json_ms": round((t4 - t3) * 1e3, 2),
                 "first_prove_ms": round((t5 - t4) * 1e3, 3),
                 "first_prove_stages_ms": {"upload": round(tm["upload_ms"], 3), "witness_gen_device": round(tm["witness_ms"], 3), "gkr_prove": round(tm["prove_ms"], 3)},
                 "verify_device_ms": round((t6 - t5) * 1e3, 2)}
            if warm: r["warmup_ms"] = round((t3 - t2) * 1e3, 2)
            res["with_hg_warmup" if warm else "without_warmup"] = r
            k.free()
            c.close()
    res["first_prove_ms"] = res["with_hg_warmup"]["first_prove_ms"]
    res["setup_ms"] = round(res["with_hg_warmup"]["setup_ms"] + res["with_hg_warmup"]["warmup_ms"], 2)
    res["note"] = ("fresh hg_create + hg_setup (+ hg_warmup) + hg_witness_from_json + the FIRST hg_prove + hg_verify_device, one pass each (no warm-up, no median); "
                   "first_prove_ms = hg_prove from the parsed (pageable) arrays: staging, upload, circuit evaluation on the device, GKR prove; setup_ms = hg_setup + hg_warmup")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ring-degree", "--n", dest="n", type=int, default=32768)
    ap.add_argument("--crt-moduli", "--k", dest="k", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true", help="do not run the rocprofv3 --pmc passes inside this run (read profiles/ instead)")
    ap.add_argument("--witnesses", type=int, default=4, help="number of different synthetic witnesses proven in rotation (one per step)")
    ap.add_argument("--collective-timeout", type=float, default=300.0,
                    help="N>1: seconds the communicator set-up and the first sharded prove may take before the rank exits non-zero")
    ap.add_argument("--seed", type=int, default=0x4752454330)
    ap.add_argument("--mode", choices=["shard", "dp"], default="shard",
                    help="N>1: shard ONE proof over the GPUs (strong scaling, default) or one independent proof per GPU (weak)")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="CPU/gloo self-test of the N>1 plumbing (rendezvous, barrier, max-over-ranks, rank-0 JSON); proves nothing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if args.selftest_dist:
        return selftest_dist(args, rank, world, dist, torch)
    import __graft_entry__ as entry
    hg = entry.load_package()

    # test hooks (single-GPU box): HG_BENCH_BACKEND=gloo reduces through host tensors, HG_BENCH_SAME_DEVICE=1 puts
    # every rank on device 0. The driver's multi-GPU runs use neither: RCCL ("nccl"), one rank per GPU.
    backend = os.environ.get("HG_BENCH_BACKEND", "nccl")
    if os.environ.get("HG_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    shard = world > 1 and args.mode == "shard"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = hg.Context(local_rank)
    bfv = hg.BfvEncrypt.new(args.n, args.k)
    pk = bfv.setup(ctx)
    # NW different witnesses, each with its node tables resident in HBM before anything is timed. Sharded: every rank holds the same
    # NW witnesses; dp: every rank has its own NW.
    NW = max(1, args.witnesses)
    wrank = 0 if shard else rank
    seeds = [witness_seed(args.seed, args.n, wrank) + 104729 * j for j in range(NW)]
    witnesses = [hg.Witness.synthetic(bfv.params, sd) for sd in seeds]
    # node tables -> HBM (outside the timed region). Sharded: a rank keeps only the tables its share reads (hg_witness_gen_shard:
    # the witness is not replicated; HG_BENCH_REPLICATE=1 restores full copies)
    per_rank = shard and os.environ.get("HG_BENCH_REPLICATE") != "1"
    out = hg.ProofBuffer()
    unsharded = None
    if shard:  # the sharded proofs must equal the single-GPU proofs bit for bit: proven once from a full copy, which is then released
        unsharded = []
        for w in witnesses:
            fv = hg.witness_gen(ctx, pk, w)
            unsharded.append(hg.prove_resident(ctx, pk, fv, out).bytes())
            fv.free()
    vals = [hg.witness_gen_shard(ctx, pk, w, rank, world) if per_rank else hg.witness_gen(ctx, pk, w) for w in witnesses]
    resident = vals[0].info()

    lib_comm = shard and backend == "nccl"   # the library's own RCCL all-reduce (device buffers, no torch hop, no host staging)
    exchange_note = None
    rccl_ranks_seen = None
    if lib_comm:
        # every rank must take the same path: a failure anywhere (no librccl, init error) sends ALL ranks to the torch exchange
        err = None
        uid = [None]
        if rank == 0:
            try:
                uid = [hg.comm_unique_id()]
            except Exception as e:  # noqa: BLE001
                err = str(e)
        dist.broadcast_object_list(uid, src=0)     # 128 bytes, once: the only use of torch.distributed on this path besides barriers
        if uid[0] is None:
            err = err or "rank 0 could not create an RCCL id"
        else:
            try:
                with deadline(args.collective_timeout, "hg_comm_init (ncclCommInitRank over %d ranks)" % world):
                    hg.comm_init(ctx, uid[0], rank, world)
                rccl_ranks_seen = hg.comm_count(ctx)   # ncclCommCount of the library's communicator
            except Exception as e:  # noqa: BLE001
                err = str(e)
        flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            if not err:
                hg.comm_destroy(ctx)
            lib_comm = False
            exchange_note = "library RCCL communicator unavailable (%s): exchange through torch.distributed all_gather" % (err or "another rank failed")

    def step(j=0, out=out):
        v = vals[j % NW]
        if not shard:
            return hg.prove_resident(ctx, pk, v, out)
        if lib_comm:
            return hg.prove_sharded(ctx, pk, v, out)   # this rank's jobs -> one ncclAllReduce on the prover stream -> replay
        import numpy as np
        part = hg.prove_shard_begin(ctx, pk, v, rank, world)         # this rank's jobs, one stream sync
        t = torch.from_numpy(part.view(np.int64).copy()).to(red_dev)
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)                                     # the only exchange of the proof (RCCL over xGMI)
        gathered = torch.stack(bufs).cpu().numpy().view(np.uint64)
        hg.prove_shard_combine(ctx, gathered, world)                 # lane-wise sum mod p (GP#1 round sums are partial sums)
        return hg.prove_shard_finish(ctx, out)                       # transcript replay -> identical bytes on every rank

    if shard:
        # every rank must hand the collective a result buffer of the same length (they walk the same protocol): a mismatch would
        # hang the all-reduce, so check it once, up front, through the caller-side entry points (no collective inside)
        n_local = int(len(hg.prove_shard_begin(ctx, pk, vals[0], rank, world)))
        hg.prove_shard_finish(ctx, out)   # (clears the pending shard; its partial-sum "proof" is discarded)
        lens = [None] * world
        dist.all_gather_object(lens, n_local)
        assert len(set(lens)) == 1, f"ranks disagree on the result-buffer length: {lens}"
    # the first collective of the data path under a deadline: a hang (a rank that never arrives, a dead link) must end THIS process with
    # a non-zero exit code instead of holding the node until the driver's limit
    with deadline(args.collective_timeout if world > 1 else 0, "first sharded prove (the first RCCL all-reduce of the data path)"):
        walked = [step(j).bytes() for j in range(NW)]   # plain launches: the reference every later proof of witness j is compared with
    assert len(set(walked)) == NW or NW == 1, "different witnesses gave identical proofs"
    if shard:
        assert walked == unsharded, "sharded proof differs from the single-GPU proof"
    for _ in range(2):   # second walk, then the capture: from here on every step replays the launch graph of its values object
        for j in range(NW):
            assert step(j).bytes() == walked[j], "proof changed between runs"
    for i in range(args.warmup):
        step(i)

    # which kernel class dominates? untimed proves with events on every class, every launch on ONE stream (with the
    # second stream active a class's event time also contains whatever shared the GPU with it)
    ctx.set_option("one_stream", 1)
    ctx.profile(2)
    ctx.profile_reset()
    for i in range(3):
        step(i)
    ctx.profile(0)
    ctx.set_option("one_stream", 0)
    step()
    seen = [s for s in ctx.profile_get() if s["name"] in CLASS_SYMBOL and s["launches"]]
    # the round's fixed class (ROOFLINE_CLASS) where the parameter set launches it, else the class with the largest isolated time
    DOMINANT = ROOFLINE_CLASS if any(s["name"] == ROOFLINE_CLASS for s in seen) else max(seen, key=lambda s: s["total_ms"])["name"]
    ctx.profile_select(DOMINANT)
    import gc
    gc.collect()
    gc.disable()  # a cyclic-GC pass over the interpreter's (PyTorch-sized) heap costs 80-100 ms: keep it out of the timed steps
    outs = [hg.ProofBuffer(cap=max(1 << 16, 2 * len(walked[0]))) for _ in range(args.steps)]   # one per timed step: all of them checked afterwards

    def timed(k, keep=False):
        barrier()
        t0 = time.perf_counter()
        for i in range(k):
            step(i, outs[i] if keep else out)
        barrier()
        return time.perf_counter() - t0

    # Three timed regions of K steps each, back to back, same schedule (two streams), a different witness every step:
    #  A: HIP events around every launch of the dominant class. Events take the prover off its cached launch graphs (recorded into
    #     a graph as event nodes they serialise its branches: 4.9 ms per prove, scripts/ub/graph_events.hip), so these K steps walk
    #     the protocol and launch kernel by kernel. `roofline` comes from here.
    #  B: nothing but the proves: the product's steady state (the launch graph of each step's values object replayed). `value`.
    #  C: the same with hg_set_option("graph", 0): every prove walks the protocol (`walked_ms_per_step`).
    ctx.profile(1)
    step()
    ctx.profile_reset()
    elapsed_a = timed(args.steps)
    ctx.profile(0)
    gpu_ms_a = out.timings()["gpu_ms"]
    for i in range(NW):   # (back on the launch graphs)
        assert step(i).bytes() == walked[i], "proof changed between runs"
    elapsed = timed(args.steps, keep=True)
    gpu_ms = outs[-1].timings()["gpu_ms"]
    enqueue_ms = outs[-1].timings()["enqueue_ms"]
    ctx.set_option("graph", 0)
    step()
    elapsed_c = timed(args.steps)
    ctx.set_option("graph", 1)
    gc.enable()
    for i in range(args.steps):   # every proof of region B against the walked proof of the same witness
        assert outs[i].bytes() == walked[i % NW], f"timed step {i}: the replayed proof differs from the walked proof of witness {i % NW}"
    for i in range(NW):
        assert step(i).bytes() == walked[i]
    resident_all = None
    peak_all = None
    witness_ms_all = None
    dp_leg = None
    if shard:
        lst = [None] * world
        dist.all_gather_object(lst, round(resident["resident_bytes"] / 1e6, 1))
        resident_all = lst
        lst = [None] * world
        dist.all_gather_object(lst, round(resident.get("peak_bytes", resident["resident_bytes"]) / 1e6, 1))
        peak_all = lst
        lst = [None] * world
        dist.all_gather_object(lst, round(vals[0].timings["witness_ms"] + vals[0].timings["upload_ms"], 3))
        witness_ms_all = lst
        # The same N GPUs used the other way round, in the same run: one INDEPENDENT proof per GPU (no data-path collective, weak
        # scaling) - the aggregate proofs/s the node delivers when latency of one proof does not matter. A different witness per rank
        # (own full table set), the launch graph replayed, every proof compared with the walked proof of its witness.
        dp_w = hg.Witness.synthetic(bfv.params, witness_seed(args.seed, args.n, rank) + 7)
        dp_vals = hg.witness_gen(ctx, pk, dp_w)
        dp_out = hg.ProofBuffer()
        dp_ref = hg.prove_resident(ctx, pk, dp_vals, dp_out).bytes()
        for _ in range(3):
            assert hg.prove_resident(ctx, pk, dp_vals, dp_out).bytes() == dp_ref
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hg.prove_resident(ctx, pk, dp_vals, dp_out)
        barrier()
        dp_elapsed = max_over_ranks(time.perf_counter() - t0, world, dist, torch, red_dev)
        assert dp_out.bytes() == dp_ref
        refs = [None] * world
        import hashlib
        dist.all_gather_object(refs, hashlib.sha256(dp_ref).hexdigest())
        dp_leg = {"mode": f"dp{world}: one independent proof per GPU, no data-path collective (weak scaling)",
                  "ms_per_step": round(dp_elapsed / args.steps * 1e3, 4), "proofs_per_step": world,
                  "proofs_per_s": round(world * args.steps / dp_elapsed, 2), "distinct_proofs": len(set(refs)),
                  "note": "timed like `value` (barrier, K steps, barrier, max over ranks) right after the sharded regions; `value` stays the sharded proof's ms"}
        dp_vals.free()
    elapsed = max_over_ranks(elapsed, world, dist, torch, red_dev)
    elapsed_a = max_over_ranks(elapsed_a, world, dist, torch, red_dev)
    elapsed_c = max_over_ranks(elapsed_c, world, dist, torch, red_dev)
    ms_per_step = elapsed / args.steps * 1e3
    ms_per_step_a = elapsed_a / args.steps * 1e3
    ms_per_step_c = elapsed_c / args.steps * 1e3
    dom = [s for s in ctx.profile_get() if s["name"] == DOMINANT][0]
    first = walked[0]

    # one extra, untimed pass with events on every kernel class (one stream: isolated class times): the per-class breakdown
    ctx.set_option("one_stream", 1)
    step()
    ctx.profile(2)
    ctx.profile_reset()
    step()
    ctx.profile(0)
    ctx.set_option("one_stream", 0)
    # algo_GB: what the class streams by THIS implementation's algorithm (each live table read once, each folded table written once);
    # model_GB: the same launches in the traffic model of the REFERENCE's algorithm - larger where an algebraic shortcut
    # avoids tables (grand product #1's top layer runs on the read rows only, the collation sum-check on two tables)
    # hbm_GB: what the class moves to or from HBM by design (hg_kernel_stat::hbm_bytes: no credit for tables that are recomputed and
    # never stored, for the intermediate folds of a two-round launch or for rounds inside LDS) - the numerator of every HBM fraction
    # below; algo_GB: the per-round accounting of SURVEY 8(d) applied to this implementation's tables (a fused launch credited with
    # both rounds; comparable across rounds of this work, NOT a traffic figure); model_GB: the same launches in the reference's model
    classes = {s["name"]: {"launches": s["launches"], "ms": round(s["total_ms"], 4), "hbm_GB": round(s["hbm_bytes"] / 1e9, 4),
                           "algo_GB": round(s["algo_bytes"] / 1e9, 4), "model_GB": round(s["model_bytes"] / 1e9, 4)}
               for s in ctx.profile_get() if s["launches"]}

    # the dominant class once more with every launch on one stream (isolated kernel duration), untimed
    ctx.set_option("one_stream", 1)
    step()
    ctx.profile(1)
    ctx.profile_reset()
    for i in range(3):
        step(i)
    ctx.profile(0)
    iso = [s for s in ctx.profile_get() if s["name"] == DOMINANT][0]
    iso_ms = iso["total_ms"] / max(iso["launches"], 1)
    iso_achieved = iso["hbm_bytes"] / max(iso["launches"], 1) / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
    iso_model = iso["model_bytes"] / max(iso["launches"], 1) / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
    iso_gpu_ms = out.timings()["gpu_ms"]
    ctx.set_option("one_stream", 0)

    end_to_end = None
    cold_start = None
    verify_info = None
    if world == 1 and rank == 0 and not args.no_end_to_end:
        end_to_end = measure_end_to_end(hg, ctx, bfv, pk, witnesses, walked, args)
        cold_start = measure_cold_start(hg, local_rank, args, witnesses[0], walked[0])
        # BfvEncrypt::verify (host side, like the reference's; OpenMP over the table-sized loops) on the proof of witness 0
        vt = []
        for _ in range(4):
            t0 = time.perf_counter()
            ok, why = hg.verify(pk, witnesses[0], walked[0])
            vt.append((time.perf_counter() - t0) * 1e3)
            assert ok, why
        vd = []
        for _ in range(4):
            t0 = time.perf_counter()
            ok, why = hg.verify_device(ctx, pk, witnesses[0], walked[0])
            vd.append((time.perf_counter() - t0) * 1e3)
            assert ok, why
        verify_info = {"goldilocks_ms": round(sorted(vt[1:])[1], 2), "host_threads": min(64, os.cpu_count() or 1),
                       "device_ms": round(sorted(vd[1:])[1], 2),
                       "note": "hg_verify on the host / hg_verify_device with the table-sized checks as kernels (median of 3 after one warm-up, public "
                               "inputs uploaded per call); the reference reports 107.9 ms on an M1 Pro (README.md:44)"}

    if rank == 0:
        per_launch_bytes = dom["hbm_bytes"] / max(dom["launches"], 1)
        avg_ms = dom["total_ms"] / max(dom["launches"], 1)
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        live_dir, live_note = (None, "not requested (--no-live-counters)") if args.no_live_counters or world > 1 else live_counter_passes(args.n, args.k)
        pmc = Counters(args.n, args.k, live_dir=live_dir, live_note=live_note)
        im = IssueModel()
        sym = CLASS_SYMBOL.get(DOMINANT, "")
        traffic, _ = pmc.hbm_bytes(sym)
        valu_pl = pmc.valu(sym)
        valu_ach = valu_pl / (avg_ms * 1e-3) / 1e9 if valu_pl and avg_ms > 0 else None
        valu_frac = valu_ach / VALU_PEAK_G if valu_ach else None
        hbm_frac = achieved / HBM_PEAK_GBS
        # the issue roof of THIS kernel: its VALU instructions priced by class (IssueModel) over the same launch duration
        issue_s = im.seconds(valu_pl, sym)
        issue_frac = issue_s / (avg_ms * 1e-3) if issue_s is not None and avg_ms > 0 else None
        issue_frac_iso = issue_s / (iso_ms * 1e-3) if issue_s is not None and iso_ms > 0 else None
        vbest = issue_frac if issue_frac is not None else valu_frac
        valu_bound = vbest is not None and vbest > hbm_frac

        def class_row(name, c):
            """one class against both roofs, isolated duration (one stream): frac = the larger of the two"""
            ms = c["ms"]
            symc = CLASS_SYMBOL.get(name, "")
            pb, _ = pmc.hbm_bytes(symc)
            vi = pmc.valu(symc)
            ln = max(c["launches"], 1)
            hf = c["hbm_GB"] / (ms * 1e-3) / HBM_PEAK_GBS if ms > 0 else None
            pf = pb * ln / 1e9 / (ms * 1e-3) / HBM_PEAK_GBS if pb and ms > 0 else None
            vf = vi * ln / 1e9 / (ms * 1e-3) / VALU_PEAK_G if vi and ms > 0 else None
            it = im.seconds(vi * ln if vi else None, symc)
            isf = it / (ms * 1e-3) if it is not None and ms > 0 else None
            vbest = isf if isf is not None else vf
            fr = max(x for x in (hf, vbest) if x is not None) if (hf is not None or vbest is not None) else None
            return {"kernel": name, "symbol": symc, "launches_per_step": c["launches"], "isolated_avg_launch_us": round(ms / ln * 1e3, 2),
                    "hbm_frac": round(hf, 4) if hf is not None else None, "pmc_hbm_frac": round(pf, 4) if pf is not None else None,
                    "issue_frac": round(isf, 4) if isf is not None else None, "class_a_share": round(im.share_a(symc), 3) if im.share_a(symc) is not None else None,
                    "valu_frac_nominal_614G": round(vf, 4) if vf is not None else None, "frac": round(fr, 4) if fr is not None else None,
                    "bound": "issue" if (vbest is not None and hf is not None and vbest > hf) else "hbm"}
        pmc_bytes_prove, valu_prove = pmc.prove_totals()
        line = {
            "metric": f"GKR prove ms, n={args.n} k={args.k} Goldilocks; achieved HBM GB/s vs roofline",
            "value": round(ms_per_step if (shard or world == 1) else ms_per_step / world, 4),
            "unit": "ms",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": False,
            "scaling": "strong" if shard else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"GKR prove (sk_encryption_circuit.rs:444-457) of the BFV sk-enc circuit, n={args.n} k={args.k} "
                                   f"Goldilocks/GoldilocksExt2, a new witness every step ({NW} seeded synthetic witnesses in rotation), node tables resident in HBM",
                       "witnesses": NW, "witness_seeds": seeds,
                       "proof_checks": {"timed_steps_equal_to_the_walked_proof_of_their_witness": args.steps,
                                        "distinct_proofs": len(set(walked))},
                       "n": args.n, "k": args.k, "field": "goldilocks", "proofs_per_step": 1 if (shard or world == 1) else world,
                       "parallelism": ((f"shard{world}: one proof, Lasso node split by memory and node reductions dealt over {world} GPUs, "
                                        + ("one RCCL all-reduce of the result buffer per proof (inside the library)" if lib_comm
                                           else "one all-gather of the result buffers per proof through torch.distributed"))
                                       if shard else f"dp{world}: one independent proof per GPU, no data-path collective"),
                       **({"exchange_note": exchange_note} if exchange_note else {}),
                       "proof_bytes": len(first), "gpu_ms_events": round(gpu_ms, 4),
                       "host_enqueue_ms": round(enqueue_ms, 4),
                       "walked_ms_per_step": round(ms_per_step_c, 4),
                       "timed_regions": {"value": f"{args.steps} proves, a different witness each, the launch graph of each witness's tables replayed, no profiling events",
                                         "walked_ms_per_step": f"{args.steps} proves right after, hg_set_option(graph, 0): the protocol walked and launched kernel by kernel",
                                         "roofline": f"{args.steps} proves immediately before, plain launches with HIP events around the dominant class: "
                                                     f"{ms_per_step_a:.4f} ms per prove, {gpu_ms_a:.4f} ms of GPU time"},
                       **({"rccl_ranks_seen": rccl_ranks_seen} if rccl_ranks_seen is not None else {}),
                       **({"resident_node_tables": {"rank0_MB": round(resident["resident_bytes"] / 1e6, 1), "per_rank_MB": resident_all,
                                                    "per_rank_peak_MB": peak_all, "per_rank_witness_ms": witness_ms_all,
                                                    "full_set_MB": round(resident["full_bytes"] / 1e6, 1),
                                                    "note": "hg_witness_gen_shard: a rank keeps the Lasso input and the inputs of the node reductions it owns, and evaluates only "
                                                            "the cone of those tables (peak = its tables + the cone's subset tables; witness ms = upload + evaluation of the cone)"}} if shard else {}),
                       **({"dp_aggregate": dp_leg} if dp_leg else {}),
                       **({"end_to_end": end_to_end} if end_to_end else {}),
                       **({"cold_start": cold_start} if cold_start else {}),
                       **({"verify": verify_info} if verify_info else {}),
                       "witness_gen_ms_device_first_call": round(vals[0].timings["witness_ms"], 2), "upload_ms_first_call": round(vals[0].timings["upload_ms"], 2)},
            # `achieved` / `avg_launch_us`: HIP events around the dominant kernel class over the K proves of timed region A, where its launches
            # share the GPU with the second stream (Vanilla / FFT reductions, counter sorts, openings); `isolated`: the same
            # launches timed in an extra untimed prove with every launch on one stream.
            # The class with the largest isolated GPU time, against BOTH roofs; `frac` is the larger fraction and `bound` names its roof.
            # HBM: bytes the class moves to or from HBM by design (`hbm_bytes_per_launch`) over the average launch duration measured with
            # HIP events inside timed region A (its launches share the GPU with the other streams there); `traffic` = the same launches'
            # HBM bytes by the PMC counters. VALU: SQ_INSTS_VALU per launch over the same duration against 614 G wave-instr/s.
            # `bound` / `frac`: the larger of the HBM fraction and the ISSUE fraction (`issue_frac`: the kernel's VALU instructions priced by
            # instruction class - IssueModel - over the launch duration; `peak` then is the rate at which THIS kernel's mix issues).
            "roofline": {"bound": "issue" if valu_bound else "hbm", "kernel": DOMINANT, "symbol": sym,
                         "achieved": round(valu_ach if valu_bound else achieved, 2),
                         "peak": (round(valu_ach / issue_frac, 1) if issue_frac else round(VALU_PEAK_G, 1)) if valu_bound else HBM_PEAK_GBS,
                         "unit": "G wave-instr/s" if valu_bound else "GB/s",
                         "frac": round(vbest if valu_bound else hbm_frac, 4), "traffic": round(traffic) if traffic else None,
                         "issue_frac": round(issue_frac, 4) if issue_frac is not None else None,
                         "issue": {"frac": round(issue_frac, 4) if issue_frac is not None else None,
                                   "frac_isolated": round(issue_frac_iso, 4) if issue_frac_iso is not None else None,
                                   "class_a_share": round(im.share_a(sym), 3) if im.share_a(sym) is not None else None,
                                   "issue_us_per_launch": round(issue_s * 1e6, 2) if issue_s is not None else None,
                                   "rates": im.describe(),
                                   "note": "sum over the kernel's VALU instructions of 1 / (issue rate of their class on this chip); class mix from the assembly, count from SQ_INSTS_VALU"},
                         "hbm": {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4),
                                 "hbm_bytes_per_launch": round(per_launch_bytes),
                                 "pmc_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and avg_ms > 0 else None},
                         "valu": {"achieved": round(valu_ach, 1) if valu_ach else None, "peak": round(VALU_PEAK_G, 1), "unit": "G wave-instr/s",
                                  "frac": round(valu_frac, 4) if valu_frac else None,
                                  "frac_of_measured_issue_rate": round(valu_ach / VALU_MEASURED_G, 4) if valu_ach else None,
                                  "wave_insts_per_launch": round(valu_pl) if valu_pl else None},
                         "counters": {"measured": pmc.measured, "files": pmc.files, "code_hash": code_hash(), "command": PMC_CMD, **({"note": pmc.note} if pmc.note else {})},
                         "launches_per_step": dom["launches"] // max(args.steps, 1),
                         "avg_launch_us": round(avg_ms * 1e3, 3),
                         "isolated": {"avg_launch_us": round(iso_ms * 1e3, 3), "hbm_achieved": round(iso_achieved, 2),
                                      "hbm_frac": round(iso_achieved / HBM_PEAK_GBS, 4),
                                      "valu_frac": round(valu_pl / (iso_ms * 1e-3) / 1e9 / VALU_PEAK_G, 4) if valu_pl and iso_ms > 0 else None,
                                      "gpu_ms_one_stream": round(iso_gpu_ms, 4),
                                      "note": "hg_set_option(one_stream): no cross-stream overlap"},
                         # the same launches priced in the reference algorithm's traffic model (its 2 alpha table pairs per
                         # layer, hash rows stored): the speed in units of the reference's work - NOT a bandwidth and not bounded by 1
                         "reference_model": {"bytes_per_launch": round(dom["model_bytes"] / max(dom["launches"], 1)),
                                             "x_hbm_peak_isolated": round(iso_model / HBM_PEAK_GBS, 4)}},
            "kernel_classes": classes,
            # the four largest classes of the per-class pass (one stream: isolated durations) against both roofs
            "roofline_by_class": [class_row(name, c) for name, c in sorted(classes.items(), key=lambda kv: -kv[1]["ms"])[:5] if name != "aux"][:4],
            # whole prove: GPU time of one prove (HIP events around the whole enqueue) against the bytes it moves by design, the bytes
            # the counters saw, and the VALU instructions it issues; reference_model_GB / x_hbm_peak: SURVEY 8(d)'s model of the
            # REFERENCE's algorithm over the same time (comparable across rounds; above the HBM peak because most of that traffic is no
            # longer generated - it stopped being a roofline when the shortcuts went in)
            "whole_prove": {"gpu_ms": round(gpu_ms, 4),
                            "hbm_GB": round(sum(c["hbm_GB"] for c in classes.values()), 3),
                            "hbm_frac": round(sum(c["hbm_GB"] for c in classes.values()) / (gpu_ms * 1e-3) / HBM_PEAK_GBS, 4) if gpu_ms > 0 else None,
                            "pmc_GB": round(pmc_bytes_prove / 1e9, 3) if pmc_bytes_prove else None,
                            "pmc_hbm_frac": round(pmc_bytes_prove / 1e9 / (gpu_ms * 1e-3) / HBM_PEAK_GBS, 4) if pmc_bytes_prove and gpu_ms > 0 else None,
                            "valu_wave_insts": round(valu_prove) if valu_prove else None,
                            "valu_frac": round(valu_prove / 1e9 / (gpu_ms * 1e-3) / VALU_PEAK_G, 4) if valu_prove and gpu_ms > 0 else None,
                            "valu_frac_of_measured_issue_rate": round(valu_prove / 1e9 / (gpu_ms * 1e-3) / VALU_MEASURED_G, 4) if valu_prove and gpu_ms > 0 else None,
                            "issue_frac": round(pmc.issue_seconds_of_a_prove(im) / (gpu_ms * 1e-3), 4) if pmc.issue_seconds_of_a_prove(im) and gpu_ms > 0 else None,
                            "reference_model_GB": round(sum(c["model_GB"] for c in classes.values()), 3),
                            "reference_model_x_hbm_peak": round(sum(c["model_GB"] for c in classes.values()) / (gpu_ms * 1e-3) / HBM_PEAK_GBS, 4) if gpu_ms > 0 else None},
        }
        if world == 1:
            # the same proof over bn256::Fr (BASELINE config 5's field) on the same witness: reported next to the headline
            # number, never part of `value`; 5 runs after one warm-up, median of 5
            try:
                witness = witnesses[0]
                ctx.prove_bn254(pk, witness, cap=1 << 25)
                runs = sorted((ctx.prove_bn254(pk, witness, cap=1 << 25)[1:] for _ in range(5)), key=lambda t: t[1])
                best = runs[len(runs) // 2]
                pb = ctx.prove_bn254(pk, witness, cap=1 << 25)[0]
                t0 = time.perf_counter()
                okb, whyb = hg.verify_bn254(pk, witness, pb)
                bn_verify_ms = (time.perf_counter() - t0) * 1e3
                line["bn254"] = {"workload": f"BfvEncrypt::prove over bn256::Fr (hg_prove_bn254), n={args.n} k={args.k}, witness 0",
                                 "witness_gen_ms": round(best[0], 2), "prove_ms": round(best[1], 2), "dtype": "u256 (4x64 Montgomery)",
                                 "runs_ms": [round(r[1], 2) for r in runs], "statistic": "median of 5 after one warm-up",
                                 "verify": {"accepted": bool(okb), "host_verify_ms": round(bn_verify_ms, 1),
                                            "note": "hg_verify_bn254 on the host; the reference reports 529 ms on an M1 Pro (README.md:56)"}}
                # this path is integer-ALU bound, not HBM bound: VALU wave-instructions of one prove (committed SQ_INSTS_VALU pass of
                # this round, only valid for n=32768 k=16) over the measured time, against the VALU issue peak of the chip
                try:
                    bn_live = os.path.join(pmc.dir, f"{PMC_TAG}_bn254_pmc_sq.json")
                    sq = json.load(open(bn_live if os.path.exists(bn_live) else os.path.join(ROOT, "profiles", f"{PMC_TAG}_bn254_pmc_sq.json")))
                    if (args.n, args.k) == (32768, 16) and sq.get("code_hash") == code_hash():
                        insts = sq["prove_valu_wave_insts_per_prove"]   # WITHOUT witness generation: the timed span is the prove
                        ach = insts / (best[1] * 1e-3) / 1e9
                        # issue roof: the listed kernels priced with their own class mix, the remainder as class B
                        bn_issue = None
                        if im.ok():
                            WG = ("k_bn_ntt", "k_bn_gate_eval", "k_bn_lift", "k_bn_bitrev", "k_bn_scale", "k_bn_powers")
                            listed, bn_issue = 0.0, 0.0
                            for kn, nv in sq.get("by_kernel_per_prove", {}).items():
                                if any(w in kn for w in WG):
                                    continue
                                listed += nv
                                t = im.seconds(nv, kn.replace("hg::bn::", "bn::").replace("hg::dev::", "").replace("void ", ""))
                                bn_issue += t if t is not None else nv / im.rate_b / NSIMD
                            bn_issue += max(insts - listed, 0.0) / im.rate_b / NSIMD
                        line["bn254"]["roofline"] = {"bound": "valu", "achieved": round(ach, 1), "peak": round(VALU_PEAK_G, 1), "unit": "G wave-instr/s",
                                                     "frac": round(ach / VALU_PEAK_G, 4), "frac_of_measured_issue_rate": round(ach / VALU_MEASURED_G, 4),
                                                     "issue_frac": round(bn_issue / (best[1] * 1e-3), 4) if bn_issue else None,
                                                     "valu_wave_insts_per_prove": round(insts),
                                                     "source": {"file": f"{PMC_TAG}_bn254_pmc_sq.json ({'this run' if os.path.exists(bn_live) else 'profiles/'})", "command": sq["command"], "code_hash": sq["code_hash"]}}
                    elif (args.n, args.k) == (32768, 16):
                        line["bn254"]["roofline"] = {"note": f"profiles/{PMC_TAG}_bn254_pmc_sq.json was taken on code {sq.get('code_hash')}, this is {code_hash()}: refused"}
                except Exception:
                    pass
            except Exception as ex:
                line["bn254"] = {"error": str(ex)}
        if world == 1 and not args.no_end_to_end:
            # the protocol with both soundness fixes upstream is likely to make (absorbing transcript, extension-field
            # memory checking): every round's challenge depends on the round's message, so the rounds run one after the other
            try:
                outm = hg.ProofBuffer()
                runs = []
                for _ in range(4):
                    hg.prove_resident_mode(ctx, pk, vals[0], outm, 3)
                    runs.append((outm.timings()["prove_ms"], int(outm.timings()["sync_ms"]), int(outm.timings()["enqueue_ms"])))
                okm, whym = hg.verify(pk, witnesses[0], outm.bytes(), mode=3)
                med = sorted(runs[1:])[1]
                line["sound_mode"] = {"mode": 3, "prove_ms": round(med[0], 2), "stream_synchronisations": med[1], "mailbox_round_trips": med[2],
                                      "runs_ms": [round(r[0], 2) for r in runs[1:]], "accepted_by_hg_verify_mode_3": bool(okm),
                                      "note": "hg_prove_resident_mode(.., 3): median of 3 after one warm-up; the host transcript answers every round through a pinned mailbox"}
                # the same mode on two ranks with ONE all-reduce per sum-check round (the exchange an absorbing transcript
                # leaves; hg_prove_resident_mode_sharded). The ranks are two threads of this process, each with its own context, BOTH on
                # this GPU: what is shown is that the path runs here and gives the single-rank proof, not a multi-GPU time.
                try:
                    import threading
                    ref3 = outm.bytes()
                    ctx_b = hg.Context(local_rank)
                    pk_b = bfv.setup(ctx_b)
                    vals_b = hg.witness_gen(ctx_b, pk_b, witnesses[0])
                    group = hg.Group.local(2)
                    got, errs = [None, None], []

                    def seq_rank(r):
                        try:
                            o = hg.ProofBuffer()
                            hg.prove_resident_mode_sharded(ctx if r == 0 else ctx_b, pk if r == 0 else pk_b, vals[0] if r == 0 else vals_b, o, 3, r, group)
                            got[r] = (o.bytes(), o.timings())
                        except Exception as ex:
                            errs.append(str(ex))

                    hung = False
                    for _ in range(2):
                        ths = [threading.Thread(target=seq_rank, args=(r,), daemon=True) for r in range(2)]
                        for t in ths: t.start()
                        for t in ths: t.join(120)
                        if any(t.is_alive() for t in ths):   # a rank still inside the library: its context must outlive it
                            hung = True
                            errs.append("a rank thread did not finish within 120 s (its context, key and tables are leaked, not freed under it)")
                            break
                    line["sound_mode"]["sharded_two_ranks_one_gpu"] = (
                        {"error": "; ".join(errs)} if errs or None in got else
                        {"identical_to_single_rank": got[0][0] == ref3 and got[1][0] == ref3, "allreduces_per_proof": int(got[0][1]["replay_ms"]),
                         "prove_ms_per_rank": [round(g[1]["prove_ms"], 2) for g in got],
                         "note": "two ranks as threads on ONE GPU, in-process group (hg_group_local): every rank folds everything and evaluates half of each round's sums; one all-reduce of <= 6 words per round"})
                    # ... and WITHOUT replicating the witness (round 6): per-rank tables (hg_witness_gen_shard), a node reduction runs on its owner alone
                    if not hung and not errs:
                        sv = [hg.witness_gen_shard(ctx if r == 0 else ctx_b, pk if r == 0 else pk_b, witnesses[0], r, 2) for r in range(2)]
                        infos = [v.info() for v in sv]
                        group2 = hg.Group.local(2)
                        got2, errs2 = [None, None], []

                        def own_rank(r):
                            try:
                                o = hg.ProofBuffer()
                                hg.prove_resident_mode_sharded(ctx if r == 0 else ctx_b, pk if r == 0 else pk_b, sv[r], o, 3, r, group2)
                                got2[r] = (o.bytes(), o.timings())
                            except Exception as ex:
                                errs2.append(str(ex))

                        for _ in range(2):
                            ths = [threading.Thread(target=own_rank, args=(r,), daemon=True) for r in range(2)]
                            for t in ths: t.start()
                            for t in ths: t.join(120)
                            if any(t.is_alive() for t in ths):
                                hung = True
                                errs2.append("a rank thread did not finish within 120 s")
                                break
                        line["sound_mode"]["node_ownership_two_ranks_one_gpu"] = (
                            {"error": "; ".join(errs2)} if errs2 or None in got2 else
                            {"identical_to_single_rank": got2[0][0] == ref3 and got2[1][0] == ref3, "allreduces_per_proof": int(got2[0][1]["replay_ms"]),
                             "prove_ms_per_rank": [round(g[1]["prove_ms"], 2) for g in got2],
                             "resident_MB_per_rank": [round(i["resident_bytes"] / 1e6, 1) for i in infos], "full_set_MB": round(infos[0]["full_bytes"] / 1e6, 1),
                             "note": "two ranks as threads on ONE GPU: each holds its share of the node tables only; a Vanilla / FFT node's reduction runs on its owner, the "
                                     "other rank joins the same all-reduces with zeros; the Lasso node keeps the tile-split form"})
                        if not hung:
                            for v in sv: v.free()
                    if not hung:
                        vals_b.free(); pk_b.free(); ctx_b.close()
                except Exception as ex:
                    line["sound_mode"]["sharded_two_ranks_one_gpu"] = {"error": str(ex)}
            except Exception as ex:
                line["sound_mode"] = {"error": str(ex)}
        if world == 1 and not args.no_end_to_end:
            # BASELINE config 4 as far as one GPU can show it: the shares of 2 / 4 / 8 ranks run one after the other on this GPU, each
            # from its own resident tables (hg_witness_gen_shard), each through its own launch graph; the partial result buffers are
            # summed as the all-reduce would (hg_shard_combine_host) and replayed: the proof must be the single-GPU proof. Reported:
            # the slowest rank's share (what a node of that many GPUs would wait for before its one collective).
            try:
                import numpy as np
                proj = {}
                for wsz in (2, 4, 8):
                    svals = [hg.witness_gen_shard(ctx, pk, witnesses[0], r, wsz) for r in range(wsz)]
                    per_rank, parts = [], []
                    for r in range(wsz):
                        ts = []
                        for i in range(6):   # walk, walk, capture, three replays (a new begin replaces the pending share of the one before)
                            t0 = time.perf_counter()
                            part = hg.prove_shard_begin(ctx, pk, svals[r], r, wsz)
                            ts.append((time.perf_counter() - t0) * 1e3)
                        per_rank.append(sorted(ts[3:])[1])
                        parts.append(part.copy())
                    gathered = np.stack(parts)
                    t0 = time.perf_counter()
                    hg.prove_shard_combine(ctx, gathered, wsz)     # lane-wise sum mod p on the host (a node does it in the all-reduce)
                    t1 = time.perf_counter()
                    same = hg.prove_shard_finish(ctx, out).bytes() == walked[0]
                    t_fin = (time.perf_counter() - t0) * 1e3
                    proj[str(wsz)] = {"slowest_rank_ms": round(max(per_rank), 3), "fastest_rank_ms": round(min(per_rank), 3), "combine_and_replay_ms": round(t_fin, 3),
                                      "host_combine_ms": round((t1 - t0) * 1e3, 3), "replay_ms": round(out.timings()["replay_ms"], 3),
                                      "projected_total_ms": round(max(per_rank) + t_fin, 3),
                                      "per_rank_ms": [round(t, 3) for t in per_rank],
                                      "resident_MB_per_rank_max": round(max(v.info()["resident_bytes"] for v in svals) / 1e6, 1),
                                      "peak_MB_per_rank_max": round(max(v.info()["peak_bytes"] for v in svals) / 1e6, 1),
                                      "witness_ms_per_rank_max": round(max(v.timings["witness_ms"] + v.timings["upload_ms"] for v in svals), 3),
                                      "proof_equals_single_gpu": bool(same)}
                    assert same, "virtual ranks: the combined proof differs from the single-GPU proof"
                    for v in svals:
                        v.free()
                line["config"]["shard_projection"] = dict(proj, note="ONE GPU running every rank's share in turn (no collective, no xGMI): per-rank wall time of hg_prove_shard_begin "
                                                                     "from the rank's launch graph, median of 3; not a multi-GPU measurement")
            except Exception as ex:
                line["config"]["shard_projection"] = {"error": str(ex)}
        if world == 1 and not args.no_cpu_baseline:
            oracle_proofs = {}
            try:
                line["cpu_baseline"] = cpu_baseline(args.n, args.k, args.seed, proofs_out=oracle_proofs)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"error": str(ex)}
            # the baseline leg proved witness 0 with the CPU oracle when the headline size fitted its budget: same bytes?
            if (args.n, args.k) in oracle_proofs:
                same = oracle_proofs[(args.n, args.k)] == walked[0]
                line["config"]["proof_checks"]["witness_0_vs_cpu_oracle"] = "identical" if same else "DIFFERENT"
                assert same, "the HIP proof of witness 0 differs from the CPU oracle's"
                # ... and every other witness of the rotation (each timed proof was compared with the walked proof of its witness: with
                # this, every timed proof is tied to the oracle's bytes)
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    import orclib  # CPU oracle: the checker
                    th = int(line["cpu_baseline"].get("cores", 16)) if isinstance(line.get("cpu_baseline"), dict) else 16
                    op = orclib.params(args.n, args.k)
                    t0 = time.perf_counter()
                    verdicts = ["identical"]
                    for j in range(1, NW):
                        ref, _ = orclib.prove(op, orclib.Inputs(witnesses[j].arrays()), threads=th)
                        verdicts.append("identical" if ref == walked[j] else "DIFFERENT")
                    line["config"]["proof_checks"]["every_witness_vs_cpu_oracle"] = verdicts
                    line["config"]["proof_checks"]["oracle_check_s"] = round(time.perf_counter() - t0, 1)
                    assert all(v == "identical" for v in verdicts), "a HIP proof differs from the CPU oracle's"
                except AssertionError:
                    raise
                except Exception as ex:
                    line["config"]["proof_checks"]["every_witness_vs_cpu_oracle"] = "error: " + str(ex)
            else:
                line["config"]["proof_checks"]["witness_0_vs_cpu_oracle"] = "not run (the oracle was timed on a smaller configuration)"
        print(json.dumps(line), flush=True)

    if lib_comm:
        hg.comm_destroy(ctx)
    for v in vals:
        v.free()
    pk.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
