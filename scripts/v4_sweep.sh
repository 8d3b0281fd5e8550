#!/bin/bash
# experiment: waves per alignment in the LDS affine tiers
run() { echo "$1"; env $1 timeout -k 10 300 python3 bench.py --regions 4000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['stage_ms'])" || exit 1; }
run "OTG_V4_NWS=2 OTG_V4_NWM=2"
run "OTG_V4_NWS=2 OTG_V4_NWM=4"
run "OTG_V4_NWS=1 OTG_V4_NWM=4"
