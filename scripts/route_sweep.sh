#!/bin/bash
for m in 80 90 100 110; do
  echo "margin=$m"
  OTG_DEBUG=1 OTG_EDIT_ROUTE_MARGIN=$m timeout -k 10 300 python3 bench.py --regions 2000 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/route_$m.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['stage_ms'])" || exit 1
  grep "edit:" gpurun_out/route_$m.err | tail -2 | cut -c1-200
done
echo "no route"
OTG_NO_EDIT_ROUTE=1 timeout -k 10 300 python3 bench.py --regions 2000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['stage_ms'])"
