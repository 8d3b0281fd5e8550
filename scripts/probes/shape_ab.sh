#!/bin/bash
# A/B of register-tier shapes on a bench workload: affine stage time per OTG_REG_SHAPE value
# usage: bash scripts/probes/shape_ab.sh <config> <regions> <shape> [<shape> ...]     (shape = OTG_REG_SHAPE digits, 0 = defaults)
cfg=$1; reg=$2; shift 2
for sh in "$@"; do
  OTG_REG_SHAPE=$sh timeout -k 10 400 python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --regions $reg --steps 2 --warmup 1 --no-legs --no-cpu-baseline --e2e-regions 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['config']['stage_ms']
print('shape %6s  value %9.1f  affine %8.1f ms  total %8.1f ms' % ('$sh', d['value'], s['ms_affine'], s['ms_total']))"
done
