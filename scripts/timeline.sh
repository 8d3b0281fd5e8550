#!/bin/bash
# usage: scripts/timeline.sh <tag> <config> <regions> [min_ms]
# rocprofv3 kernel trace of ONE timed step of the bench command, printed as a timeline (start offset, duration, gap to the previous end) of the
# kernels that last at least min_ms (default 0.3): where a small batch spends its time between the kernels and in their tails
tag=$1; cfg=${2:-1}; regions=${3:-1000}; minms=${4:-0.3}
cd /tmp && export TMPDIR=/tmp
ARGS="--config $cfg --steps 1 --warmup 2 --no-cpu-baseline --e2e-regions 0 --no-legs --regions $regions"
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/tl_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/tl_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the last step = everything after the last K_pair_tasks launch that starts an edit stage of a run (first kernel of otg_assemble_run)
starts = [i for i, e in enumerate(ev) if "K_pair_tasks" in e[2]]
i0 = starts[-1] if starts else 0
# walk back to the true beginning of that run (kernels launched before K_pair_tasks in the same run are within 2 ms)
while i0 > 0 and ev[i0][0] - ev[i0 - 1][1] < 2_000_000 and ev[i0][0] - ev[i0-1][0] < 5_000_000: i0 -= 1
t0 = ev[i0][0]; last_end = t0
out = open("gpurun_out/${tag}_timeline.txt", "w")
busy = 0
for s, e, n in ev[i0:]:
    d = (e - s) / 1e6
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:60]
    if d >= $minms or (s - last_end) / 1e6 > 0.2:
        line = "%9.2f ms  +%8.2f ms  gap %7.2f  %s" % ((s - t0) / 1e6, d, (s - last_end) / 1e6, n)
        print(line); out.write(line + "\n")
    last_end = max(last_end, e)
print("total %.2f ms, %d kernels" % ((last_end - t0) / 1e6, len(ev) - i0))
out.write("total %.2f ms, %d kernels\n" % ((last_end - t0) / 1e6, len(ev) - i0))
PY
grep -h metric gpurun_out/tl_$tag.log | cut -c1-200
