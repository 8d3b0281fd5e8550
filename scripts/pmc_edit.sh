#!/bin/bash
# PMC passes for the edit kernels (each --pmc set in its own rocprofv3 run; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_edit
rm -rf $OUT; mkdir -p $OUT
N=${1:-600}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/bench_edit.py $N > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_edit/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:30]
        if "edit" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(agg, open("gpurun_out/pmc_edit/summary.json", "w"), indent=1)
for k, v in agg.items():
    if v.get("SQ_WAVE_CYCLES", 0) < 1e8: continue
    wc = v["SQ_WAVE_CYCLES"]
    print(k)
    print("   VALU busy per SIMD %.1f%%  waves/SIMD %.1f" % (100 * v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), wc * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)))
    for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
        print("   %-22s %5.1f%% of wave-cycles" % (c, 100 * v[c] / wc))
    print("   insts VALU %.3g SALU %.3g LDS %.3g VMEM_RD %.3g; LDS bank conflict cycles %.3g" % (v["SQ_INSTS_VALU"], v["SQ_INSTS_SALU"], v["SQ_INSTS_LDS"], v["SQ_INSTS_VMEM_RD"], v["SQ_LDS_BANK_CONFLICT"]))
PY
