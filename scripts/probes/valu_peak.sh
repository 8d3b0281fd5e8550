#!/bin/bash
# builds and runs the VALU ceiling probe on the GPU box -> gpurun_out/r04_valu_peak.json (copy the result to profiles/)
set -e
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_peak $GRAFT_REPO_ROOT/scripts/probes/valu_peak.hip
timeout -k 10 300 /tmp/valu_peak 8 > $GRAFT_REPO_ROOT/gpurun_out/r04_valu_peak.json
