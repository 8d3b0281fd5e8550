#!/bin/bash
# usage: [CFG=1] scripts/probes/adaptive_ab.sh "<ENV=val ...>" ...   — one adaptive bench run of configs[CFG] per environment setting: tier statistics + stage times
cd $GRAFT_REPO_ROOT
CFG=${CFG:-1}
for envs in "$@"; do
  echo "== $envs"
  env $envs OTG_DEBUG=1 timeout -k 10 400 python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --e2e-regions 0 --heuristic wfadaptive --no-legs > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
  grep "affine, wfadaptive" gpurun_out/ab.err | tail -1
  grep "edit, wfadaptive" gpurun_out/ab.err | tail -2
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab.json") if l.startswith("{")][-1])
print(d["value"], d["config"]["stage_ms"])
PY
done
