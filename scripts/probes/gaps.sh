#!/bin/bash
# Timeline of the hot path on a small batch: rocprofv3 kernel trace of bench.py --regions N, then per step the span, the sum of kernel
# durations, and the largest gaps between consecutive kernels (with the kernels on either side).
# usage: bash scripts/probes/gaps.sh <regions>
N=${1:-1000}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python3 $GRAFT_REPO_ROOT/bench.py --config 1 --regions $N --steps 3 --warmup 2 --no-legs --e2e-regions 0 --no-cpu-baseline > /tmp/gp.log 2>&1 || { tail -5 /tmp/gp.log; exit 1; }
grep metric /tmp/gp.log | cut -c1-200
python3 - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob("/tmp/gp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:44]))
for f in glob.glob("/tmp/gp/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "")[:30]))
rows.sort()
# the last step = from the last K_pair_tasks to the end
starts = [i for i, r in enumerate(rows) if r[2].startswith("K_pair_tasks")]
i0 = starts[-1]
step = rows[i0:]
span = (step[-1][1] - step[0][0]) / 1e6
busy = 0; cur_end = step[0][0]; gaps = []
for a, b, n in step:
    if a > cur_end: gaps.append(((a - cur_end) / 1e3, n)); 
    busy += max(0, b - max(a, cur_end)); cur_end = max(cur_end, b)
print("last step: %d launches/copies, span %.2f ms, GPU busy (union) %.2f ms, idle %.2f ms" % (len(step), span, busy / 1e6, span - busy / 1e6))
gaps.sort(reverse=True)
print("largest gaps (us, before):", [(round(g), n) for g, n in gaps[:14]])
print("gaps > 10 us: %d totalling %.2f ms" % (sum(1 for g, _ in gaps if g > 10), sum(g for g, _ in gaps if g > 10) / 1e3))
agg = {}
for a, b, n in step: agg[n] = agg.get(n, 0) + (b - a) / 1e6
for n, t in sorted(agg.items(), key=lambda kv: -kv[1])[:18]: print("  %-46s %8.2f ms" % (n, t))
PY
