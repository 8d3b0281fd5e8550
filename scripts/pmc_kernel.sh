#!/bin/bash
# ad-hoc counter passes over the bench command, summed per kernel matching a pattern
# usage: bash scripts/pmc_kernel.sh <config> <kernel substring> "<counters pass 1>" ["<counters pass 2>" ...]
CFG=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pk_$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pk_$i -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --e2e-regions 0 > /tmp/pk_$i.log 2>&1 || { tail -5 /tmp/pk_$i.log; exit 1; }
  python3 - "$PAT" /tmp/pk_$i <<'PY'
import csv, glob, sys, collections
pat, d = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print("  %-28s %.4g  (%d dispatch rows)" % (k, agg[k], n[k]))
PY
done
