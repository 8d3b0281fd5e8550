import csv, glob, collections, re, sys, os
src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.match(r"([A-Za-z0-9_]+(<[0-9, a-z]+>)?)", k).group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("%-46s %4s %9s %9s %8s %7s %7s %7s %7s %7s" % ("kernel", "n", "Mcyc", "VALU(M)", "SALU/V", "LDS/V", "busy4", "wait", "w/SIMD", "ldsbc"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc < 1e5: continue
    v = c.get("SQ_INSTS_VALU", 0)
    print("%-46s %4d %9.1f %9.1f %8.2f %7.2f %7.3f %7.3f %7.2f %7.3f" % (k[:46], max(n[k].values()), cyc / 1e6, v / 1e6, c.get("SQ_INSTS_SALU", 0) / max(v, 1), c.get("SQ_INSTS_LDS", 0) / max(v, 1),
          v * 4.1 / (1024.0 * cyc) if cyc else 0, c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 4.0 * c.get("SQ_WAVE_CYCLES", 0) / (1024.0 * cyc) if cyc else 0,
          c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
