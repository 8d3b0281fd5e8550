set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu_peak_pmc2
rm -rf $OUT; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_peak $GRAFT_REPO_ROOT/scripts/probes/valu_peak.hip
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_IOPS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- /tmp/valu_peak 8 > $OUT/p1.log 2>&1
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"probe<(\d+)>", r["Kernel_Name"])
        if m: agg[int(m.group(1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = re.findall(r'"([^"]*)"', re.search(r'NAME\[N_CLS\] = \{(.*?)\};', open("$GRAFT_REPO_ROOT/scripts/probes/valu_peak.hip").read(), re.S).group(1))
for c in sorted(agg):
    a = {k: v[-1] for k, v in agg[c].items()}
    i = max(a.get("SQ_INSTS_VALU", 1), 1)
    print("%-44s insts %.4g thread_cyc/inst %.2f  active2/inst %.3f int32/inst %.2f iops/inst %.2f busy_cu %.4g any/inst %.2f" % (names[c][:44], i, a.get("SQ_THREAD_CYCLES_VALU", 0) / i, a.get("SQ_ACTIVE_INST_VALU2", 0) / i,
          a.get("SQ_INSTS_VALU_INT32", 0) / i, a.get("SQ_INSTS_VALU_IOPS", 0) / i, a.get("SQ_BUSY_CU_CYCLES", 0), a.get("SQ_ACTIVE_INST_ANY", 0) / i))
PY
