#!/bin/bash
# A/B of library tunables on one box: median GPU time of 8 graph replays per setting. usage: sweep_tunables.sh VAR=val [VAR=val ...]
run() { echo -n "$1: "; env $1 python scripts/prove_once.py 32768 16 14 2>&1 | tail -8 | python3 -c "
import sys,ast,statistics
g=[ast.literal_eval(l.strip())['gpu_ms'] for l in sys.stdin]
print('gpu median %.3f min %.3f'%(statistics.median(g),min(g)))"; }
run A=1
for s in "$@"; do run "$s"; done
run A=1
