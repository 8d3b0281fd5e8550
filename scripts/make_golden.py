"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE (oracle/_ref/libotter_ref.so, built from
/root/reference/src/{andistmat,ankde}.cpp, include/hclust-cpp/fastcluster.cpp and src/anppoa.hpp) on seeded
inputs.  Fixtures are data only (inputs + expected outputs).  Run in the build container:
    python scripts/make_golden.py"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from helpers import cluster_cases, random_poa_specs, build_poa_batch  # noqa: E402

assert O.ref() is not None, "oracle/_ref not built (needs /root/reference)"
rng = np.random.default_rng(20241008)
out = os.path.join(ROOT, "tests", "golden")

# G2: hclust_fast(AVERAGE) / cutree_k / cutree_cdist / get_medoid
cases = [c for c in cluster_cases(rng, 54, nmax=28) if len(c[1]) >= 2]
rec = {}
for i, (d, lens) in enumerate(cases):
    n = len(lens)
    merge, height = O.hclust_average(n, d, which="ref")
    cd = float(np.median(d)) if d.size else 0.0
    rec["d%d" % i] = d
    rec["merge%d" % i] = merge
    rec["height%d" % i] = height
    rec["cut_k2_%d" % i] = O.cutree_k(n, merge, 2, which="ref")
    rec["cut_k3_%d" % i] = O.cutree_k(n, merge, 3, which="ref")
    rec["cut_c_%d" % i] = O.cutree_cdist(n, merge, height, cd, which="ref")
    rec["cd%d" % i] = np.array([cd])
    ind = np.arange(0, n, 2, dtype=np.uint32)
    rec["medoid%d" % i] = np.array([O.medoid(n, d, ind, which="ref")])
rec["n_cases"] = np.array([len(cases)])
np.savez_compressed(os.path.join(out, "hclust_ref.npz"), **rec)

# G1 (KDE part): KDE::f on grid points and KDE::maximas on the normalised densities
rec = {}
kc = [c for c in cluster_cases(rng, 36, nmax=20) if c[0].size >= 3]
for i, (d, lens) in enumerate(kc):
    h = 0.015 if i % 2 else 0.01
    xs = np.array([0.0, 0.0025, 0.1, 0.25, 0.5, 0.99999999999998967])
    rec["d%d" % i] = d
    rec["h%d" % i] = np.array([h])
    rec["xs%d" % i] = xs
    rec["f%d" % i] = np.array([O.kde_f(h, d, float(x), which="ref") for x in xs])
    _, _, dens = O.find_clustering_dist(d, h)          # oracle densities (same KDE::f formula), fed to the reference's maximas
    mx, mn = O.kde_maximas(dens, which="ref")
    rec["dens%d" % i] = dens
    rec["max_i%d" % i] = np.array([m[0] for m in mx], dtype=np.int32)
    rec["max_v%d" % i] = np.array([m[1] for m in mx])
    rec["min_i%d" % i] = np.array([m[0] for m in mn], dtype=np.int32)
    rec["min_v%d" % i] = np.array([m[1] for m in mn])
rec["n_cases"] = np.array([len(kc)])
np.savez_compressed(os.path.join(out, "kde_ref.npz"), **rec)

# G3: PPOA consensus (reference anppoa.hpp) on op strings
specs = random_poa_specs(rng, O, 40, 5, 160, err=0.08)
sarena, carena, members, graphs = build_poa_batch(specs)
cons = O.poa_consensus_batch(sarena, carena, members, graphs, which="ref")
np.savez_compressed(os.path.join(out, "poa_ref.npz"), sarena=sarena, carena=carena, members=members, graphs=graphs,
                    cons=np.frombuffer(b"\n".join(cons), dtype=np.uint8))
print("golden fixtures written to", out)


# G5: record emit — the reference's own ANALLELE::stdout_sam / stdout_fa (oracle/_ref/libotter_ref_io.so) on synthetic records;
# the header lines are three literal stream inserts (src/assemble.cpp:171-174), written here from the oracle's restatement
import json  # noqa: E402
from test_emit import synthetic_records, TARGETS  # noqa: E402
assert O.ref_io() is not None, "oracle/_ref/libotter_ref_io.so not built"
beds, carena, res = synthetic_records()
gold = {k: O.emit_alleles(beds, carena, res, rg, fa, which="ref").decode() for k, rg, fa in
        (("sam", "", False), ("sam_rg", "sampleA", False), ("fa", "", True), ("fa_rg", "sampleA", True))}
gold["header"] = O.emit_sam_header(TARGETS, "sampleA", 30, 31).decode()
json.dump(gold, open(os.path.join(out, "emit_ref.json"), "w"))
print("emit_ref.json", {k: len(v) for k, v in gold.items()})


# G6: BAM ingest — a small BAM/BAI written by the reference's htslib-lite from synthetic SAM text, and what the reference's own
# parse_anreads (oracle/_ref/libotter_ref_io.so) returns for a fixed region list under three option sets
import shutil, tempfile  # noqa: E402
import test_ingest as TI  # noqa: E402
rng = np.random.default_rng(20241009)
tmp = tempfile.mkdtemp()
chroms = [("chr1", 300_000), ("chr2_random:alt", 80_000), ("chrBig", 400_000_000)]
sam, bam = os.path.join(tmp, "g.sam"), os.path.join(tmp, "g.bam")
n = TI._random_sam(sam, rng, chroms, 900, read_len=(20, 150))
assert O.ref_io().ref_sam_to_bam(sam.encode(), bam.encode()) == n
shutil.copy(bam, os.path.join(out, "ingest_small.bam"))
shutil.copy(bam + ".bai", os.path.join(out, "ingest_small.bam.bai"))
regions = []
for _ in range(120):
    name, clen = chroms[int(rng.integers(0, len(chroms)))]
    s0 = int(rng.integers(0, min(clen, 300_000) - 3000)) if name != "chrBig" or rng.random() < 0.5 else int(rng.integers(0, clen - 3000))
    regions.append((name, s0, s0 + int(rng.integers(1, 2500))))
regions += [("chr1", 0, 10), ("chr1", 5, 5), ("nochr", 10, 20), ("chr1", 299_000, 310_000)]
rec = {"regions_chr": np.array([r[0] for r in regions]), "regions_start": np.array([r[1] for r in regions], dtype=np.int64),
       "regions_end": np.array([r[2] for r in regions], dtype=np.int64)}
OPTS = [dict(), dict(offset_l=1, offset_r=1, mapq=10), dict(offset_l=50000, offset_r=7, nonprimary=True), dict(omit_nonspanning=True, mapq=3),
             dict(read_quality=0.4, nonprimary=True)]
for i, kw in enumerate(OPTS):
    b = TI._ref_ingest(bam, regions, **kw)
    rec["reads%d" % i] = b["reads"]; rec["arena%d" % i] = np.ascontiguousarray(b["arena"]); rec["n_reads%d" % i] = b["regions"]["n_reads"]
np.savez_compressed(os.path.join(out, "ingest_ref.npz"), **rec)
print("ingest_small.bam", os.path.getsize(os.path.join(out, "ingest_small.bam")), "bytes;", {k: len(v) for k, v in rec.items() if k.startswith("reads")})
