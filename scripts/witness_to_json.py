#!/usr/bin/env python3
"""Writes a witness of this library (hg_witness_synthetic, or any hg.Witness) as a JSON file in the reference's format
(BfvSkEncryptArgs: decimal strings, [REF bfv-gkr/src/sk_encryption_circuit.rs:64-73]) - the inverse of get_inputs'
layouts [REF :365-415, poly.rs:12-44]. With it the reference's own test (cargo test test_sk_enc_valid_goldilocks_<cfg>)
can prove the SAME witness as the HIP prover at sizes whose fixtures are missing blobs (n = 16384, 32768).
usage: witness_to_json.py <n> <k> <seed> <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def arrays_to_args(n, k, d):
    """d: tables as laid out by get_inputs (hg.Witness.arrays()) -> dict of the reference's JSON fields."""
    SZ = 2 * n
    s = [str(int(v)) for v in d["s"][:n]]                                 # Poly::new_padded: coefficients then zeros
    e = [str(int(v)) for v in d["e"][SZ - 1 - n:SZ - 1]]                  # Poly::new_shifted(.., 2^L - 1): zeros in front
    k1 = [str(int(v)) for v in d["k1"][SZ - 1 - n:SZ - 1]]
    ais = [[str(int(v)) for v in d["ais"][z * SZ:z * SZ + n]] for z in range(k)]
    r1is = [[str(int(v)) for v in d["r1is"][z * SZ:z * SZ + 2 * n - 1]] for z in range(k)]
    r2is = [[str(int(v)) for v in d["r2is"][z * n:z * n + n - 1]] for z in range(k)]      # Poly::new + one trailing zero
    ct0is = [[str(int(v)) for v in d["ct0is"][z * SZ + SZ - n - 1:z * SZ + SZ - 1]] for z in range(k)]  # shifted by 2^L, first dropped
    return {"s": s, "e": e, "k1": k1, "r2is": r2is, "r1is": r1is, "ais": ais, "ct0is": ct0is}


def main():
    n, k, seed, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3], 0), sys.argv[4]
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    hg = entry.load_package()
    w = hg.Witness.synthetic(hg.params_builtin(n, k), seed)
    json.dump(arrays_to_args(n, k, w.arrays()), open(out, "w"))
    print("wrote", out)


if __name__ == "__main__":
    main()
