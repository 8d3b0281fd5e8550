#!/bin/bash
# One batch of same-sized gap-affine alignments through the register tier(s) a mask selects (scripts/tier_probe.py), under rocprofv3:
# kernel time (kernel-trace) and two PMC passes (SQ activity; instruction counts), summed per affine kernel.
# usage: bash scripts/tier_pmc.sh <read_len> <n_pairs> <rate> <mask> [tag]     -> gpurun_out/tier_pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
L=$1; N=$2; R=$3; export OTG_AFFINE_REG=$4; TAG=${5:-t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tier_pmc_$TAG.txt
: > $OUT
rm -rf /tmp/tpm_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tpm_0 -- python3 $GRAFT_REPO_ROOT/scripts/tier_probe.py $L $N $R > /tmp/tpm_0.log 2>&1 || { tail -5 /tmp/tpm_0.log; exit 1; }
grep "^len" /tmp/tpm_0.log >> $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/tpm_$i -- python3 $GRAFT_REPO_ROOT/scripts/tier_probe.py $L $N $R > /tmp/tpm_$i.log 2>&1 || { echo "pmc pass $i failed" >> $OUT; tail -3 /tmp/tpm_$i.log >> $OUT; }
done
python3 - >> $OUT <<'PY'
import csv, glob, re, collections
for f in glob.glob("/tmp/tpm_0/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "wfa_affine" in n and float(r["TotalDurationNs"]) > 2e5:
            m = re.search(r"(wfa_affine\w*<[^>]*>)", n)
            print("time  %-46s calls %s avg %.2f ms" % (m.group(1) if m else n[:46], r["Calls"], float(r["AverageNs"]) / 1e6))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for i in (1, 2, 3):
    for f in glob.glob("/tmp/tpm_%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "wfa_affine" not in n: continue
            m = re.search(r"(wfa_affine\w*<[^>]*>)", n)
            agg[m.group(1) if m else n[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if v.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    print(k)
    for c, x in sorted(v.items()): print("   %-28s %.5g" % (c, x))
PY
cat $OUT
