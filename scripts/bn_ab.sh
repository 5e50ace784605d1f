#!/bin/bash
# Interleaved A/B of library tunables for the BN254 prove (config 5): R passes (default 3) over the settings, 7 proves each (the
# first two of a process dropped); prints the median prove time per setting. usage: [R=3] scripts/bn_ab.sh "VAR=val ..." ...
R=${R:-3}
tmp=$(mktemp -d)
cat > $tmp/run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
hg = entry.load_package()
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 5)
for i in range(7):
    proof, wms, pms = ctx.prove_bn254(pk, w, cap=1 << 25)
    if i >= 2: print("PMS %.3f %.3f" % (pms, wms))
PY
for r in $(seq $R); do
  i=0
  for s in "$@"; do env $s python $tmp/run.py 2>&1 | grep "^PMS" >> $tmp/$i.txt; i=$((i+1)); done
done
i=0
for s in "$@"; do
  echo -n "$s: "
  python3 -c "
import statistics
g=[float(l.split()[1]) for l in open('$tmp/$i.txt')]; w=[float(l.split()[2]) for l in open('$tmp/$i.txt')]
print('prove median %.2f ms  min %.2f  (witness %.2f)  n=%d'%(statistics.median(g),min(g),statistics.median(w),len(g)))"
  i=$((i+1))
done
rm -rf $tmp
