#!/bin/bash
# PMC passes over the bench command (one BASELINE config), each counter set in its own rocprofv3 run with --kernel-trace only
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with other trace domains).
# usage: [PMC_EXTRA="--heuristic wfadaptive"] bash scripts/pmc_bench.sh <config> [regions]   -> gpurun_out/pmc_bench_c<config>/<pass>/...counter_collection.csv
# then:  python3 scripts/pmc_summarize.py <config> [regions]  -> profiles/pmc_summary.json (+ per-kernel table on stdout)
CFG=${1:-1}
REG=${2:-}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench_c$CFG
rm -rf $OUT; mkdir -p $OUT
ARGS="--config $CFG --steps 1 --warmup 0 --no-cpu-baseline --e2e-regions 0 --no-legs"
if [ -n "$REG" ]; then ARGS="$ARGS --regions $REG"; fi
ARGS="$ARGS ${PMC_EXTRA:-}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU2 SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
echo done
