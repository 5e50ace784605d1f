#!/bin/bash
# Interleaved A/B of environment settings on the bn254 prove (n=32768 k=16): R passes, the median of the prove times per setting.
R=${R:-3}
tmp=$(mktemp -d)
for r in $(seq $R); do i=0; for s in "$@"; do env $s python scripts/bn254_prove_bench.py 2>&1 | grep "prove " >> $tmp/$i.txt; i=$((i+1)); done; done
i=0
for s in "$@"; do echo -n "$s: "; python3 -c "
import re,statistics
v=[float(re.search(r'prove ([0-9.]+) ms', l).group(1)) for l in open('$tmp/$i.txt')]
print('prove median %.2f min %.2f n=%d' % (statistics.median(v), min(v), len(v)))"; i=$((i+1)); done
rm -rf $tmp
