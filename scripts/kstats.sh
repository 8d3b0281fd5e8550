#!/bin/bash
# usage: scripts/kstats.sh <tag> <regions> [extra env assignments are inherited]
# rocprofv3 kernel stats of the bench command; prints the top kernels (per-launch average in ms)
tag=$1; regions=${2:-1000}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --regions $regions --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
for r in rows[1:16]:
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print(name[:48].ljust(48), r[1].rjust(4), "total %8.1f ms" % (float(r[2]) / 1e6), "avg %8.2f ms" % (float(r[3]) / 1e6), r[4] + "%")
PY
grep metric gpurun_out/prof_$tag.log | cut -c1-140
