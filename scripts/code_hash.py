#!/usr/bin/env python3
"""Content hash of the product sources (csrc/*.hip, *.hpp, *.inc, *.cpp, the Makefile, include/hg.h): the code state a counter file
belongs to. rocprofv3 PMC passes are separate runs whose numbers bench.py reads from profiles/: every such file carries this hash
and bench.py ignores a file whose hash is not the one of the sources it runs (git is not available on the GPU box)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def code_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hyper-greco_amd", "csrc")
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".hpp", ".inc", ".cpp")) or f == "Makefile"]
    files.append(os.path.join(ROOT, "include", "hg.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]

if __name__ == "__main__":
    print(code_hash())
