#!/usr/bin/env python3
"""Seed corpora for tests/fuzz/fuzz_host.cpp under build/fuzz/: the reference's n=1024 witness files (whole, truncated at a few points,
with single flipped bytes) and the CPU oracle's proofs of them (test infrastructure: the oracle is the checker that writes the known-good
proof the verifier is then fuzzed around)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
out = os.path.join(ROOT, "build", "fuzz")
rng = random.Random(6)
for d in ("json", "verify", "verifybn"): os.makedirs(os.path.join(out, d), exist_ok=True)
for name in ("sk_enc_1024_1x27_65537.json", "bn254_sk_enc_1024_1x27_65537.json"):
    raw = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    open(os.path.join(out, "json", name), "wb").write(raw)
    for i, cut in enumerate((0, 1, 17, len(raw) // 3, len(raw) - 2)): open(os.path.join(out, "json", f"{name}.cut{i}"), "wb").write(raw[:cut])
    for i in range(8):
        b = bytearray(raw); at = rng.randrange(len(b)); b[at] ^= 1 << rng.randrange(8)
        open(os.path.join(out, "json", f"{name}.flip{i}"), "wb").write(bytes(b))
open(os.path.join(out, "json", "tiny0"), "wb").write(b'{"s":["1"],"e":[],"k1":["-1"]}')
open(os.path.join(out, "json", "tiny1"), "wb").write(b'{"ais":[["1","2"],["3"]],"ct0is":[[]]}')
p = orclib.params(1024, 1)
import json
def arrays(path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("hyper_greco_amd", os.path.join(ROOT, "hyper-greco_amd", "__init__.py"))
    hg = importlib.util.module_from_spec(spec); sys.modules["hyper_greco_amd"] = hg; spec.loader.exec_module(hg)
    return hg
hg = arrays(None)
w = hg.BfvEncrypt.new(1024, 1).get_inputs(os.path.join(ROOT, "tests", "golden", "sk_enc_1024_1x27_65537.json"))
proof, _ = orclib.prove(p, orclib.Inputs(w.arrays()), threads=4)
open(os.path.join(out, "verify", "oracle_proof"), "wb").write(proof)
open(os.path.join(out, "verify", "oracle_proof.cut"), "wb").write(proof[: len(proof) // 2])
pb, _ = orclib.prove_f("bn254", p, orclib.bn254_fixture_inputs(1024, 1, 27), threads=4)
open(os.path.join(out, "verifybn", "oracle_proof"), "wb").write(pb)
open(os.path.join(out, "verifybn", "oracle_proof.cut"), "wb").write(pb[: len(pb) // 3])
print("seeds written under", out, len(proof), len(pb))
