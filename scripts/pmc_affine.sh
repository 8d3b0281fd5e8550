#!/bin/bash
# PMC passes for the affine kernels (each --pmc set in its own rocprofv3 run; kernel-trace only)
# usage: scripts/pmc_affine.sh [regions]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_affine
rm -rf $OUT; mkdir -p $OUT
N=${1:-1500}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/bench_affine.py $N 1 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_affine/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:34]
        if "affine" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(agg, open("gpurun_out/pmc_affine/summary.json", "w"), indent=1)
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()): print("   %-44s %.4g" % (c, x))
PY
