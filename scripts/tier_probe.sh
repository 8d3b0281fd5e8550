#!/bin/bash
# kernel times of scripts/tier_probe.py under rocprofv3 for a list of "mask:shape" settings
# usage: bash scripts/tier_probe.sh <read_len> <n_pairs> <rate> "<mask:shape> ..."
cd /tmp && export TMPDIR=/tmp
L=$1; N=$2; R=$3
for ms in $4; do
  export OTG_AFFINE_REG=${ms%%:*} OTG_REG_SHAPE=${ms##*:}
  rm -rf /tmp/tp_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp_prof -- python3 $GRAFT_REPO_ROOT/scripts/tier_probe.py $L $N $R > /tmp/tp.log 2>&1 || { tail -5 /tmp/tp.log; exit 1; }
  grep "^len" /tmp/tp.log
  python3 - <<'PY'
import csv, glob, re
for f in glob.glob("/tmp/tp_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "wfa_affine" in n and float(r["TotalDurationNs"]) > 2e5:
            m = re.search(r"(wfa_affine\w*<[^>]*>)", n)
            print("    %-50s calls %s avg %.2f ms" % (m.group(1) if m else n[:50], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
