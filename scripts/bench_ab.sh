#!/bin/bash
# A/B of library tunables through bench.py itself (value = wall ms per proof over fresh witnesses): R passes, one bench run per setting and pass
R=${R:-2}
for r in $(seq $R); do
for s in "$@"; do
  env $s python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$s', 'value', d['value'], 'gpu_ms', d['whole_prove']['gpu_ms'], 'frac', d['roofline']['frac'], 'launch_us', d['roofline']['avg_launch_us'])"
done
done
