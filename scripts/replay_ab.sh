#!/bin/bash
# replay_ms / prove_ms medians of graph-replayed proves per setting. usage: scripts/replay_ab.sh "VAR=val" ...
R=${R:-3}
tmp=$(mktemp -d)
for r in $(seq $R); do i=0; for s in "$@"; do env $s python scripts/prove_once.py 32768 16 28 2>&1 | tail -24 >> $tmp/$i.txt; i=$((i+1)); done; done
i=0
for s in "$@"; do echo -n "$s: "; python3 -c "
import ast,statistics
rows=[ast.literal_eval(l.strip()) for l in open('$tmp/$i.txt') if l.startswith('{')]
print('replay median %.3f  prove %.3f  gpu %.3f  n=%d'%(statistics.median(r['replay_ms'] for r in rows), statistics.median(r['prove_ms'] for r in rows), statistics.median(r['gpu_ms'] for r in rows), len(rows)))"; i=$((i+1)); done
rm -rf $tmp
