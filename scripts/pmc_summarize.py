"""Summarises gpurun_out/pmc_bench into profiles/r01_pmc_traffic.json: HBM bytes per launch of each hot-path kernel
and of the two kernel groups bench.py reports on (edit = capped wavefront + bit-parallel tiers, affine = bound pass +
exact tiers).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md §HBM) — we report the corrected read figure (x2) and the raw one.  A "launch" of a group is one
call of its tier chain (edit: 2 per step = distance matrix + reassignment; affine: 1 per step)."""
import csv, glob, json, collections, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(which):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_bench", which, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][0] += float(r["Counter_Value"]); agg[r["Kernel_Name"]][1] += 1
    return agg
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+(<[0-9, ]+>)?)", n)
    return m.group(1) if m else n[:40]
fe, wr = load("fetch"), load("write")
out = {"_note": "bytes per launch; traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes); raw counters alongside", "kernels": {}}
groups = {"wfa_affine_kernel": [0.0, 0], "wfa_edit_kernel": [0.0, 0]}
for name in fe:
    k = short(name)
    if not any(t in k for t in ("affine", "edit", "poa", "cluster")): continue
    f, nf = fe[name]; w, nw = wr.get(name, [0.0, 1])
    tr = (2 * f / nf + w / max(nw, 1)) * 1024
    out["kernels"][k] = {"launches": nf, "fetch_kb_per_launch": f / nf, "write_kb_per_launch": w / max(nw, 1), "traffic_bytes_per_launch": tr}
    if "affine" in k: groups["wfa_affine_kernel"][0] += tr
    elif "edit" in k: groups["wfa_edit_kernel"][0] += tr
out["wfa_affine_kernel"] = groups["wfa_affine_kernel"][0]
out["wfa_edit_kernel"] = groups["wfa_edit_kernel"][0]
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else "...") for k, v in out.items()}, indent=1))
for k, v in out["kernels"].items(): print("%-46s launches %d  traffic/launch %.3g B" % (k, v["launches"], v["traffic_bytes_per_launch"]))
