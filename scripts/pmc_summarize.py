"""Summarises gpurun_out/pmc_bench into profiles/r01_pmc_traffic.json: HBM bytes per launch of each WFA kernel.
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md
§HBM) — we report the corrected read figure (x2) and the raw one."""
import csv, glob, json, collections, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(which):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_bench", which, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][0] += float(r["Counter_Value"]); agg[r["Kernel_Name"]][1] += 1
    return agg
fe, wr = load("fetch"), load("write")
out = {"_note": "bytes per launch; traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes); raw counters alongside", "kernels": {}}
def short(n):
    for key in ("wfa_affine_kernel_v3<4096", "wfa_affine_kernel_v3<12288", "wfa_affine_kernel<", "myers_edit_kernel<1", "myers_edit_kernel<2", "myers_edit_kernel<4", "wfa_edit_kernel_v2<2048", "wfa_edit_kernel_v2<8192", "poa_graph_kernel", "cluster_kernel"):
        if key in n: return key
    return None
for name in fe:
    k = short(name)
    if not k: continue
    f, nf = fe[name]; w, nw = wr.get(name, [0.0, 1])
    out["kernels"][k] = {"launches": nf, "fetch_kb_per_launch": f / nf, "write_kb_per_launch": w / max(nw, 1),
                         "traffic_bytes_per_launch": (2 * f / nf + w / max(nw, 1)) * 1024}
a = out["kernels"].get("wfa_affine_kernel_v3<4096", {}).get("traffic_bytes_per_launch", 0) + out["kernels"].get("wfa_affine_kernel_v3<12288", {}).get("traffic_bytes_per_launch", 0)
e = sum(out["kernels"].get(k, {}).get("traffic_bytes_per_launch", 0) for k in ("myers_edit_kernel<1", "myers_edit_kernel<2", "myers_edit_kernel<4", "wfa_edit_kernel_v2<2048"))
out["wfa_affine_kernel"] = a
out["wfa_edit_kernel"] = e
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
