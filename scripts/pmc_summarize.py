"""Summarises gpurun_out/pmc_bench_c<config> (scripts/pmc_bench.sh) into profiles/pmc_summary.json, the file bench.py takes
`roofline.traffic` and `roofline.physical` from (only when its workload is the one profiled here).

Per kernel and for the two kernel chains bench.py reports on (edit = router + capped wavefront pass + bit-parallel tiers; affine = bound pass +
exact tiers), per LAUNCH of the chain (edit: 2 per step = distance matrix + reassignment; affine: 1 per step without -r, 2 with):
  traffic  = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; the x2 is gfx950's FETCH_SIZE correction, MI355X_MICROARCH.md §HBM)
  valu_busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
              (SQ_ACTIVE_INST_* count quad-cycles; one wave64 VALU instruction holds its SIMD for one quad-cycle)
  salu_per_valu, lds_bank_conflict_rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (north_star's figure), wait fractions of wave-cycles.
usage: python3 scripts/pmc_summarize.py <config> [regions] [tag]"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from otter_amd import synth  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
regions = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else (synth.CONFIGS[cfg]["n_regions"] // 8 if cfg == 4 else synth.CONFIGS[cfg]["n_regions"])
tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
src = os.path.join(ROOT, "gpurun_out", "pmc_bench_c%d" % cfg)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+(<[0-9, a-z]+>)?)", n)
    return m.group(1) if m else n[:40]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
# gpurun merges a call's files into gpurun_out/ without removing those of earlier calls: one file per pass, the newest
newest = {}
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    p = os.path.relpath(f, src).split(os.sep)[0]
    if p not in newest or os.path.getmtime(f) > os.path.getmtime(newest[p]):
        newest[p] = f
for f in newest.values():
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1


def group_of(k):
    if "affine" in k or k.startswith("K_tsort") or k.startswith("K_seg_copy"):
        return "wfa_affine_kernel"
    if "edit" in k or k.startswith("K_sort"):
        return "wfa_edit_kernel"
    if "poa" in k:
        return "poa"
    if "cluster" in k or "genotype" in k:
        return "cluster"
    return None


def derive(c, n_launch):
    out = {"launches": n_launch}
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc > 0:
        out["kernel_cycles"] = cyc
        out["valu_busy"] = round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * cyc), 4)
        out["salu_busy_per_cu"] = round(4.0 * c.get("SQ_ACTIVE_INST_SCA", 0.0) / (256.0 * cyc) / 4.0, 4)
        out["lds_busy"] = round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * cyc), 4)
    if c.get("SQ_INSTS_VALU"):
        out["valu_insts"] = c["SQ_INSTS_VALU"]; out["salu_insts"] = c.get("SQ_INSTS_SALU", 0.0); out["lds_insts"] = c.get("SQ_INSTS_LDS", 0.0)
        out["salu_per_valu"] = round(c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"], 4)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_rate"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 5)
    if c.get("SQ_WAVE_CYCLES"):
        out["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        out["wait_inst_any_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        if cyc > 0:
            out["waves_per_simd"] = round(4.0 * c["SQ_WAVE_CYCLES"] / (1024.0 * cyc), 3)
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        out["fetch_kb"] = c.get("FETCH_SIZE", 0.0); out["write_kb"] = c.get("WRITE_SIZE", 0.0)
        out["traffic_bytes"] = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
    return out


kernels, groups = {}, collections.defaultdict(lambda: collections.defaultdict(float))
for k, c in agg.items():
    g = group_of(k)
    if g is None:
        continue
    n = max(launches[k].values())
    kernels[k] = derive(c, n)
    for name, v in c.items():
        groups[g][name] += v
# launches of each chain inside the profiled process (bench.py --steps 1 --warmup 0 = the timed step + the host-to-host pass): counted on a kernel
# that runs exactly once per chain launch
def n_of(name, default):
    for k in kernels:
        if k.startswith(name):
            return max(1, kernels[k]["launches"])
    return default
chain_launches = {"wfa_edit_kernel": n_of("edit_route_kernel", 2), "wfa_affine_kernel": n_of("wfa_affine_bound1_kernel", 1), "poa": n_of("poa_count_kernel", 1),
                  "cluster": n_of("cluster_kernel", 1)}
# cells the exact gap-affine tiers visited per chain launch: the device counter in the bench line of the profiled process (pass 1)
visited = None
try:
    for ln in open(os.path.join(src, "p1.log")):
        if ln.startswith("{") and '"roofline"' in ln:
            visited = json.loads(ln)["roofline"].get("affine_visited_cells")
except Exception:
    pass
physical, traffic = {}, {}
for g, c in groups.items():
    d = derive(c, chain_launches[g])
    if g == "wfa_affine_kernel" and visited and d.get("valu_insts"):
        d["visited_cells_per_launch"] = visited
        d["valu_insts_per_visited_cell"] = round(d["valu_insts"] / (float(visited) * chain_launches[g]), 4)
    d["bound"] = "valu"
    d["frac"] = d.get("valu_busy")
    d["what"] = "share of the chip's VALU issue slots (1024 SIMDs x kernel cycles) the chain's kernels used, summed over its kernels"
    physical[g] = d
    if "traffic_bytes" in d:
        traffic[g] = d["traffic_bytes"] / chain_launches[g]
path = os.path.join(ROOT, "profiles", "pmc_summary.json")
allc = json.load(open(path)) if os.path.exists(path) else {}
allc["config%d" % cfg] = {
    "regions": regions, "workload": synth.config_workload(cfg, regions),
    "source": "profiles/%s_pmc_c%d/ (scripts/pmc_bench.sh: rocprofv3 --kernel-trace --pmc <set> over bench.py --config %d --steps 1 --warmup 0)" % (tag, cfg, cfg),
    "traffic_bytes_per_launch": traffic, "physical": physical, "kernels": kernels}
json.dump(allc, open(path, "w"), indent=1, sort_keys=True)
dst = os.path.join(ROOT, "profiles", "%s_pmc_c%d" % (tag, cfg))
os.makedirs(dst, exist_ok=True)
json.dump({k: dict(v) for k, v in agg.items()}, open(os.path.join(dst, "counters_by_kernel.json"), "w"), indent=1, sort_keys=True)
print("%-48s %5s %9s %7s %7s %7s %8s %9s" % ("kernel", "n", "valu_busy", "salu/v", "lds_bc", "wait", "waves/S", "traffic"))
for k, d in sorted(kernels.items(), key=lambda kv: -kv[1].get("kernel_cycles", 0)):
    print("%-48s %5d %9s %7s %7s %7s %8s %9.3g" % (k[:48], d["launches"], d.get("valu_busy"), d.get("salu_per_valu"), d.get("lds_bank_conflict_rate"),
                                              d.get("wait_any_frac"), d.get("waves_per_simd"), d.get("traffic_bytes", 0.0)))
for g, d in physical.items():
    print(g, json.dumps({k: v for k, v in d.items() if k != "what"}))
