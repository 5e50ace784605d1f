#!/usr/bin/env python3
"""Generates tests/golden/constants.json from the reference's constants/*.rs (data only: N, bounds,
q_i, k0_i). Run in the authoring container where /root/reference exists; the output is committed so
that tests never read /root/reference. [REF bfv-gkr/src/constants/*.rs]"""
import json, re, glob, os, sys
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/bfv-gkr/src/constants"
out = {}
for path in sorted(glob.glob(os.path.join(ref, "sk_enc_constants_*.rs"))):
    txt = open(path).read()
    def num(name):
        return int(re.search(name + r": (?:usize|u64) = (\d+);", txt).group(1))
    def arr(name):
        body = re.search(name + r": \[[^\]]*\] = \[([^\]]*)\];", txt).group(1)
        return [int(x.strip().strip('"')) for x in body.split(",") if x.strip()]
    n = num("const N")
    ent = dict(n=n, e_bound=num("E_BOUND"), s_bound=num("S_BOUND"), k1_bound=num("K1_BOUND"),
               r1_bounds=arr("R1_BOUNDS"), r2_bounds=arr("R2_BOUNDS"), qis=arr("QIS"), k0is=arr("K0IS"))
    ent["k"] = len(ent["qis"])
    out[f"{n}_{ent['k']}"] = ent
json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "constants.json"), "w"), indent=1)
print("wrote", list(out))
