#!/bin/bash
# Interleaved A/B of library tunables on one box: R passes over the settings (default 3), 24 graph replays each; prints the median
# GPU time per setting over all passes. usage: [R=3] scripts/ab.sh "VAR=val ..." "VAR=val ..."   (a bare "A=1" is the default build)
R=${R:-3}
tmp=$(mktemp -d)
for r in $(seq $R); do
  i=0
  for s in "$@"; do
    env $s python scripts/prove_once.py 32768 16 28 2>&1 | tail -24 >> $tmp/$i.txt
    i=$((i+1))
  done
done
i=0
for s in "$@"; do
  echo -n "$s: "
  python3 -c "
import sys,ast,statistics
g=[ast.literal_eval(l.strip())['gpu_ms'] for l in open('$tmp/$i.txt') if l.startswith('{')]
print('gpu median %.3f  min %.3f  n=%d'%(statistics.median(g),min(g),len(g)))"
  i=$((i+1))
done
rm -rf $tmp
