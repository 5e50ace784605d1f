#!/usr/bin/env python3
"""What hipcc made of the kernels (no GPU needed): per kernel of csrc/kernels.hip (or the file given) the VGPR count, occupancy and
scratch bytes of the resource-usage remarks, and from the gfx950 assembly the counts that decide whether an inner loop keeps its
loads in flight: flat accesses, `s_waitcnt vmcnt(0) lgkmcnt(0)`, plain `vmcnt(0)`, scratch accesses, s_nop.
With --json <out>: the static VALU mix per kernel in the two issue classes scripts/ub/ratebench.hip measures on gfx950 (profiles/r06_ratebench.txt):
class A = the VOP1 / VOP2 forms that issue at the v_mov_b32 rate (v_mov_b32, v_add_u32, v_sub_u32, v_subrev_u32, v_and_b32, v_or_b32, v_xor_b32,
v_lshrrev_b32, v_lshlrev_b32, v_ashrrev_i32, v_not_b32), class B = every other VALU instruction (VOP3, 64-bit, carries, multiplies,
compares, selects: about half that rate); MFMA and the wait states (s_nop) are counted apart. bench.py prices a kernel's VALU instructions
with this mix (roofline.issue_frac).
usage: scripts/isa_census.py [file.hip ...] [name-filter-regex] [--json out.json]"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
json_out = None
if "--json" in argv:
    json_out = argv[argv.index("--json") + 1]
    del argv[argv.index("--json"):argv.index("--json") + 2]
srcs = [a for a in argv if a.endswith((".hip", ".cpp"))] or [os.path.join(ROOT, "hyper-greco_amd", "csrc", "kernels.hip")]
flt = re.compile(argv[-1]) if argv and not argv[-1].endswith((".hip", ".cpp")) else None
CLASS_A = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_not_b32"}
mix = {}
print("%-64s %5s %3s %7s %5s %9s %7s %7s %6s %6s" % ("kernel", "vgpr", "occ", "scratch", "flat", "vm0+lgkm0", "vmcnt0", "scr.ops", "s_nop", "VALU"))
for src in srcs:
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
      nA = sum(v for k, v in c.items() if k.startswith("v_") and re.sub(r"_e(32|64)$", "", k) in CLASS_A and not k.endswith("_e64"))
      nM = sum(v for k, v in c.items() if "mfma" in k)
      mix[dn] = {"valu": valu - nM, "class_a": nA, "class_b": valu - nM - nA, "mfma": nM, "s_nop": c["s_nop"], "vgpr": int(m.group(1)) if m else None}
      print("%-64s %5s %3s %7s %5d %9d %7d %7d %6d %6d" % (dn[:64], m.group(1) if m else "?", m.group(3) if m else "?", m.group(2) if m else "?", flat, both, vm0, scr, c["s_nop"], valu))

if json_out:
    import json
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from code_hash import code_hash
    mix["_meta"] = {"code_hash": code_hash(), "class_a": sorted(CLASS_A), "note": "static instruction counts of the whole kernel body (hipcc -O3 --offload-arch=gfx950 -S)"}
    json.dump(mix, open(json_out, "w"), indent=1)
