#!/bin/bash
# value, step time and stage times of bench.py for configs 1 and 4 (no legs, no CPU baseline, no file leg)
for c in ${1:-1 4}; do
  st=20; [ $c = 4 ] && st=3
  python bench.py --config $c --steps $st --warmup 2 --no-legs --e2e-regions 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', d['value'], d['ms_per_step'], d['config']['stage_ms'])"
done
