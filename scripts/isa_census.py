#!/usr/bin/env python3
"""What hipcc made of the kernels (no GPU needed): per kernel of csrc/kernels.hip (or the file given) the VGPR count, occupancy and
scratch bytes of the resource-usage remarks, and from the gfx950 assembly the counts that decide whether an inner loop keeps its
loads in flight: flat accesses, `s_waitcnt vmcnt(0) lgkmcnt(0)`, plain `vmcnt(0)`, scratch accesses, s_nop.
usage: scripts/isa_census.py [file.hip] [name-filter-regex]"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith((".hip", ".cpp")) else os.path.join(ROOT, "hyper-greco_amd", "csrc", "kernels.hip")
flt = re.compile(sys.argv[-1]) if len(sys.argv) > 1 and not sys.argv[-1].endswith((".hip", ".cpp")) else None
tmp = tempfile.mkdtemp()
asm, rem = os.path.join(tmp, "k.s"), os.path.join(tmp, "usage.txt")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "hyper-greco_amd", "csrc"), "-Rpass-analysis=kernel-resource-usage", "-o", asm]
with open(rem, "w") as f:
    subprocess.run(cmd, stderr=f, check=True)
usage = open(rem).read()
lines = open(asm).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
starts.append((len(lines), "end"))
def demangle(n):
    try: return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception: return n
print("%-64s %5s %3s %7s %5s %9s %7s %7s %6s %6s" % ("kernel", "vgpr", "occ", "scratch", "flat", "vm0+lgkm0", "vmcnt0", "scr.ops", "s_nop", "VALU"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    body = lines[a:b]
    end = [k for k, l in enumerate(body) if "s_endpgm" in l]
    if not end: continue
    body = body[:end[0]]
    dn = demangle(name).replace("hg::dev::", "").replace("hg::bn::", "bn::").replace("hg::", "")
    if flt and not flt.search(dn): continue
    m = re.search(re.escape(name) + r".*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", usage, re.S)
    c = collections.Counter()
    for l in body:
        mm = re.match(r"\s+([a-z_0-9]+)", l)
        if mm: c[mm.group(1)] += 1
    flat = sum(v for k, v in c.items() if k.startswith("flat_"))
    both = sum(1 for l in body if "vmcnt(0) lgkmcnt(0)" in l)
    vm0 = sum(1 for l in body if re.search(r"s_waitcnt vmcnt\(0\)\s*$", l))
    scr = sum(v for k, v in c.items() if k.startswith("scratch_"))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print("%-64s %5s %3s %7s %5d %9d %7d %7d %6d %6d" % (dn[:64], m.group(1) if m else "?", m.group(3) if m else "?", m.group(2) if m else "?", flat, both, vm0, scr, c["s_nop"], valu))
