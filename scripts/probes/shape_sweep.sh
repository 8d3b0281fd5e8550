for sh in 0 1000 100 1100; do
  OTG_REG_SHAPE=$sh python bench.py --config 1 --steps 5 --warmup 2 --no-legs --e2e-regions 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shape $sh', d['value'], d['config']['stage_ms']['ms_affine'])"
done
