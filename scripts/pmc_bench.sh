#!/bin/bash
# HBM traffic (PMC) of the bench command, separate --pmc passes as MI355X_MICROARCH.md prescribes.
# usage: bash scripts/pmc_bench.sh <regions>     -> gpurun_out/pmc_bench/{fetch,write}/...counter_collection.csv
set -e
R=${1:-10000}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --regions $R --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --regions $R --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1
echo done
