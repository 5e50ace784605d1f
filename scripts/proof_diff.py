#!/usr/bin/env python3
"""Diffs two proof byte streams (e.g. one dumped by the Rust reference's `cargo test test_sk_enc_valid_*` and the one
`hg_prove` produced for the same witness) and names the first diverging protocol element.

  HG_PROOF_MAP=/tmp/map.tsv python scripts/prove_once.py 1024 1 1      # writes the byte-offset map of our proof
  python scripts/proof_diff.py ours.bin theirs.bin /tmp/map.tsv

The map labels every element with the convention of the un-vendored `gkr` crate that decides its bytes (DESIGN.md 2:
C1 message format, C2 power order, C3 variable order, C4 fold; G1 node order, G2 alpha per claim, G3 Libra/zkCNN
forms, G4 root of unity), so a mismatch points at the one function to flip. Field elements are 8-byte big-endian
canonical Goldilocks values (transcript.rs:183-195), extension elements two of them."""
import sys

def load_map(path):
    m = []
    for line in open(path):
        off, label = line.rstrip("\n").split("\t", 1)
        m.append((int(off), label))
    return m

def label_at(m, off):
    cur = "(before the first element)"
    start = 0
    for o, l in m:
        if o > off:
            break
        cur, start = l, o
    return cur, start

def main():
    a = open(sys.argv[1], "rb").read()
    b = open(sys.argv[2], "rb").read()
    m = load_map(sys.argv[3]) if len(sys.argv) > 3 else []
    n = min(len(a), len(b))
    first = next((i for i in range(n) if a[i] != b[i]), None)
    if first is None:
        print("identical" if len(a) == len(b) else "identical for the common %d bytes; lengths %d vs %d" % (n, len(a), len(b)))
        return 0 if len(a) == len(b) else 1
    lab, start = label_at(m, first)
    felt = (first - start) // 8
    print("first difference at byte %d (field element %d of its section, 0-based; extension element %d limb %d)" % (first, felt, felt // 2, felt % 2))
    print("section: %s (starts at byte %d)" % (lab, start))
    e = first - first % 8
    print("ours  : %s" % a[e:e + 16].hex())
    print("theirs: %s" % b[e:e + 16].hex())
    same_sections = [l for o, l in m if o + 8 <= first]
    print("sections before it that agree entirely: %d" % max(0, len(same_sections) - 1))
    if len(a) != len(b):
        print("lengths differ: %d vs %d bytes (C1: coefficients per round / which one is omitted)" % (len(a), len(b)))
    return 1

if __name__ == "__main__":
    sys.exit(main())
