"""Summarises gpurun_out/pmc_bench_c<config> (scripts/pmc_bench.sh) into profiles/pmc_summary.json, the file bench.py takes
`roofline.traffic` and `roofline.physical` from (only when its workload is the one profiled here).

Per kernel and for the two kernel chains bench.py reports on (edit = router + capped wavefront pass + bit-parallel tiers; affine = bound pass +
exact tiers), per LAUNCH of the chain (edit: 2 per step = distance matrix + reassignment; affine: 1 per step without -r, 2 with):
  traffic  = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; the x2 is gfx950's FETCH_SIZE correction, MI355X_MICROARCH.md §HBM)
  valu_busy = SIMD cycles the vector instructions held / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.  A wave64 vector
              instruction does not always hold its SIMD for 4 cycles on gfx950: the plain 32-bit VOP1 / VOP2 integer operations take 2.24, everything
              else 4.15 (v_swap_b32 twice that) — measured by scripts/probes/valu_peak.hip (profiles/r04_valu_peak.json).  The counters tell the two
              classes apart: SQ_ACTIVE_INST_VALU counts one per instruction pass whatever its class, SQ_ACTIVE_INST_VALU2 reads 0.464 per fast
              instruction and 0 for the others (the same probe under rocprofv3: scripts/probes/valu_peak_pmc2.sh, profiles/r04_valu_peak_pmc.txt).
              So: fast = VALU2 / 0.464; cycles = 2.24 x fast + 4.15 x (ACTIVE_INST_VALU - fast).  (Round 3 multiplied every instruction by 4 and
              read 1.07-1.33 for kernels rich in fast instructions.)
  salu_busy = SQ_INSTS_SALU x 1.09 cycles / (256 CUs x kernel cycles): the scalar unit is one per CU and issues ~0.92 instructions per cycle
              (the probe's s_add_u32 loop) — the second ceiling of these integer kernels
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
tag = sys.argv[3] if len(sys.argv) > 3 else "r04"
mode = sys.argv[4] if len(sys.argv) > 4 else ""          # "adaptive": the passes were taken with PMC_EXTRA="--heuristic wfadaptive"; kept under its own key / directory
sfx = "_adaptive" if mode == "adaptive" else ""
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


C_FAST, C_SLOW, VALU2_PER_FAST, SALU_CYC = 2.24, 4.15, 0.464, 1.09      # profiles/r04_valu_peak.json, profiles/r04_valu_peak_pmc.txt


def derive(c, n_launch):
    out = {"launches": n_launch}
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc > 0:
        out["kernel_cycles"] = cyc
        act = c.get("SQ_ACTIVE_INST_VALU", 0.0)
        if "SQ_ACTIVE_INST_VALU2" in c:
            fast = min(act, c["SQ_ACTIVE_INST_VALU2"] / VALU2_PER_FAST)
            out["valu_fast_share"] = round(fast / act, 4) if act else 0.0
            out["valu_simd_cycles"] = C_FAST * fast + C_SLOW * (act - fast)
            out["valu_busy"] = round(out["valu_simd_cycles"] / (1024.0 * cyc), 4)
        else:                     # no class split measured: every instruction at the slow class's cost = an upper bound of the busy share
            out["valu_busy_upper_bound"] = round(C_SLOW * act / (1024.0 * cyc), 4)
        if c.get("SQ_INSTS_SALU"):
            out["salu_busy"] = round(SALU_CYC * c["SQ_INSTS_SALU"] / (256.0 * cyc), 4)
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
chain_launches = {"wfa_edit_kernel": n_of("wfa_edit_adaptive_lds_kernel<1024" if sfx else "edit_route_kernel", 2),
                  "wfa_affine_kernel": n_of("wfa_affine_adaptive_lds_kernel<256" if sfx else "wfa_affine_bound1_kernel", 1), "poa": n_of("poa_count_kernel", 1),
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
    d["what"] = "share of the chip's SIMD cycles (1024 SIMDs x kernel cycles) the chain's vector instructions held, each class at its measured issue cost"
    physical[g] = d
    if "traffic_bytes" in d:
        traffic[g] = d["traffic_bytes"] / chain_launches[g]
path = os.path.join(ROOT, "profiles", "pmc_summary.json")
allc = json.load(open(path)) if os.path.exists(path) else {}
allc["config%d%s" % (cfg, sfx)] = {
    "regions": regions, "workload": synth.config_workload(cfg, regions) + (" under wfadaptive(10,50,1)" if sfx else ""),
    "source": "profiles/%s_pmc_c%d%s/ (scripts/pmc_bench.sh: rocprofv3 --kernel-trace --pmc <set> over bench.py --config %d --steps 1 --warmup 0%s)" % (tag, cfg, sfx, cfg, " --heuristic wfadaptive" if sfx else ""),
    "traffic_bytes_per_launch": traffic, "physical": physical, "kernels": kernels}
json.dump(allc, open(path, "w"), indent=1, sort_keys=True)
dst = os.path.join(ROOT, "profiles", "%s_pmc_c%d%s" % (tag, cfg, sfx))
os.makedirs(dst, exist_ok=True)
json.dump({k: dict(v) for k, v in agg.items()}, open(os.path.join(dst, "counters_by_kernel.json"), "w"), indent=1, sort_keys=True)
lines = ["%-48s %5s %9s %9s %8s %7s %7s %7s %8s %9s" % ("kernel", "n", "valu_busy", "salu_busy", "fast_shr", "salu/v", "lds_bc", "wait", "waves/S", "traffic")]
for k, d in sorted(kernels.items(), key=lambda kv: -kv[1].get("kernel_cycles", 0)):
    lines.append("%-48s %5d %9s %9s %8s %7s %7s %7s %8s %9.3g" % (k[:48], d["launches"], d.get("valu_busy", d.get("valu_busy_upper_bound")), d.get("salu_busy"), d.get("valu_fast_share"),
                                                             d.get("salu_per_valu"), d.get("lds_bank_conflict_rate"), d.get("wait_any_frac"), d.get("waves_per_simd"), d.get("traffic_bytes", 0.0)))
print("\n".join(lines))
open(os.path.join(dst, "table.txt"), "w").write("\n".join(lines) + "\n")
for g, d in physical.items():
    print(g, json.dumps({k: v for k, v in d.items() if k != "what"}))
