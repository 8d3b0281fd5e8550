#!/bin/bash
# usage: scripts/gpu_busy.sh <tag> <loci> <batch_regions>   (kernel trace of scripts/dispatch_trace.py, then scripts/gpu_busy.py on its last pass)
tag=$1; loci=${2:-10000}; batch=${3:-1000}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 560 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gb_$tag -- python3 $GRAFT_REPO_ROOT/scripts/dispatch_trace.py $loci 16 $batch > $GRAFT_REPO_ROOT/gpurun_out/gb_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/gb_$tag -name '*kernel_trace.csv' | head -1)
nb=$(grep -c "ingest" gpurun_out/gb_$tag.log); nb=$((nb / 3))
grep wall gpurun_out/gb_$tag.log
python3 scripts/gpu_busy.py $f $nb 0.5 | tee gpurun_out/${tag}_gpu_busy.txt
rm -rf gpurun_out/gb_$tag
