#!/bin/bash
# quick A/B: bench at N regions (default 4000), prints rate and stage times
N=${1:-4000}
timeout -k 10 300 python3 bench.py --regions $N --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['stage_ms'])"
