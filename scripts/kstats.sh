#!/bin/bash
# usage: scripts/kstats.sh <tag> <config> [regions] [extra bench.py arguments, e.g. "--heuristic wfadaptive --no-legs"]
# rocprofv3 kernel stats of the bench command; prints the top kernels (per-launch average in ms) and keeps the csv + bench line
tag=$1; cfg=${2:-1}; regions=${3:-}; extra=${4:-}
cd /tmp && export TMPDIR=/tmp
ARGS="--config $cfg --steps 2 --warmup 1 --no-cpu-baseline --e2e-regions 0"
if [ -n "$regions" ]; then ARGS="$ARGS --regions $regions"; fi
ARGS="$ARGS $extra"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, shutil
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(f, "gpurun_out/${tag}_kernel_stats.csv")
rows = list(csv.reader(open(f)))
for r in rows[1:22]:
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print(name[:52].ljust(52), r[1].rjust(4), "total %8.1f ms" % (float(r[2]) / 1e6), "avg %8.2f ms" % (float(r[3]) / 1e6), r[4] + "%")
PY
grep metric gpurun_out/prof_$tag.log > gpurun_out/${tag}_bench_profiled.json
cut -c1-140 gpurun_out/${tag}_bench_profiled.json
