#!/bin/bash
# the VALU ceiling probe under the counters bench.py's physical figures come from: what do SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / GRBM_GUI_ACTIVE read
# when a SIMD is KNOWN to be saturated with one instruction class?  -> gpurun_out/valu_peak_pmc/ (+ a table on stdout)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu_peak_pmc
rm -rf $OUT; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_peak $GRAFT_REPO_ROOT/scripts/probes/valu_peak.hip
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- /tmp/valu_peak 8 > $OUT/p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- /tmp/valu_peak 8 > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"probe<(\d+)>", r["Kernel_Name"])
        if m: agg[int(m.group(1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = re.search(r'NAME\[N_CLS\] = \{(.*?)\};', open("$GRAFT_REPO_ROOT/scripts/probes/valu_peak.hip").read(), re.S).group(1)
names = re.findall(r'"([^"]*)"', names)
print("%-60s %12s %12s %10s %10s %10s" % ("class (8 waves per SIMD, last launch)", "INSTS_VALU", "ACTIVE_VALU", "GUI/8", "act*4/slots", "act/inst"))
for c in sorted(agg):
    a = {k: v[-1] for k, v in agg[c].items()}       # the last (full-length) launch of the class
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print("%-60s %12.4g %12.4g %10.4g %10.3f %10.3f" % (names[c][:60], a.get("SQ_INSTS_VALU", 0), a.get("SQ_ACTIVE_INST_VALU", 0), cyc,
          4.0 * a.get("SQ_ACTIVE_INST_VALU", 0) / (1024.0 * cyc) if cyc else 0, a.get("SQ_ACTIVE_INST_VALU", 0) / max(a.get("SQ_INSTS_VALU", 1), 1)))
PY
