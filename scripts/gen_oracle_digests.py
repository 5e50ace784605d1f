"""Regenerates tests/golden/oracle_proof_digests.json for the reference-held witnesses under tests/golden/.

The digests are REGRESSION values of the CPU oracle's own transcripts (oracle/, C++): they pin the oracle against itself across
rounds, they are not reference-produced bytes (the Rust prover cannot be built in this image). The witnesses
are the reference's own JSON files (bfv-gkr/src/data/{goldilocks,bn254}/), copied as data.
Existing entries are kept as they are (the bn254_1024_1 entry is the Python-integer oracle's proof, which the C++ Fr oracle
reproduces byte for byte: tests/test_oracle_kats.py); run with --check to compare instead of writing."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib  # noqa: E402

GL = [(1024, 1, 27), (2048, 1, 52), (4096, 2, 55), (8192, 4, 55)]
BN = [(1024, 1, 27), (2048, 1, 52), (4096, 2, 55)]


def main():
    path = os.path.join(orclib.GOLDEN, "oracle_proof_digests.json")
    gold = json.load(open(path))
    new = {}
    for n, k, bits in GL:
        proof, _ = orclib.prove(orclib.params(n, k), orclib.fixture_inputs(n, k, bits), threads=8)
        new[f"{n}_{k}"] = {"sha256": hashlib.sha256(proof).hexdigest(), "len": len(proof), "head": proof[:64].hex(), "tail": proof[-64:].hex()}
    for n, k, bits in BN:
        proof, _ = orclib.prove_f("bn254", orclib.params(n, k), orclib.bn254_fixture_inputs(n, k, bits), threads=8)
        new[f"bn254_{n}_{k}"] = {"bytes": len(proof), "sha256": hashlib.sha256(proof).hexdigest(),
                                 "note": "C++ Fr oracle (orcbn_prove) proof of tests/golden/bn254_sk_enc_%d_%dx%d_65537.json; a regression value of the oracle, not a reference-produced value" % (n, k, bits)}
    bad = 0
    for key, v in new.items():
        if key in gold:
            if any(gold[key][f] != v[f] for f in v if f != "note" and f in gold[key]):
                print("MISMATCH", key)
                bad += 1
        else:
            gold[key] = v
            print("new entry", key, v["sha256"])
    if "--check" not in sys.argv and not bad:
        json.dump(gold, open(path, "w"), indent=1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
