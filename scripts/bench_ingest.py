"""Ingest throughput on the host: otg_ingest_regions (this repo) vs the reference's parse_anreads (oracle/_ref build) on the same
BAM: R regions x D reads of 1-5 kb with ONT-like CIGARs (random ops; only the ingest semantics matter here)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import otter_amd, oracle_lib
import test_ingest as TI

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(5)
tmp = tempfile.mkdtemp()
chrom = ("chr1", 10_000 * R + 100_000)
regions = [("chr1", 50_000 + 10_000 * i, 50_000 + 10_000 * i + int(rng.integers(300, 1500))) for i in range(R)]
sam, bam = os.path.join(tmp, "b.sam"), os.path.join(tmp, "b.bam")
with open(sam, "w") as f:
    f.write("@HD\tVN:1.4\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % chrom)
    for i, (c, s, e) in enumerate(regions):
        recs = []
        for d in range(D):
            L = int(rng.integers(1000, 5000))
            pos = s - int(rng.integers(100, L - 100)) if L > (e - s) + 300 else s - 50
            ops, q = [], 0
            while q < L:
                m = int(rng.integers(5, 40)); ops.append("%dM" % m); q += m
                k = rng.random()
                if k < 0.4: ops.append("%dI" % int(rng.integers(1, 4))); q += int(ops[-1][:-1])
                elif k < 0.8: ops.append("%dD" % int(rng.integers(1, 4)))
            recs.append((max(pos, 1), "r%d_%d" % (i, d), "".join(ops), q))
        recs.sort()
        for pos, nm, cg, q in recs:
            f.write("%s\t0\tchr1\t%d\t60\t%s\t*\t0\t0\t%s\t*\n" % (nm, pos, cg, "".join("ACGT"[x] for x in rng.integers(0, 4, q))))
n = oracle_lib.ref_io().ref_sam_to_bam(sam.encode(), bam.encode())
print("BAM: %d records, %.1f MB" % (n, os.path.getsize(bam) / 1e6), flush=True)
for rep in range(2):
    t = time.perf_counter(); a = otter_amd.Bam(bam).ingest(regions, offset_l=1, offset_r=1, mapq=10); t1 = time.perf_counter() - t
    t = time.perf_counter(); b = TI._ref_ingest(bam, regions, offset_l=1, offset_r=1, mapq=10); t2 = time.perf_counter() - t
    t = time.perf_counter(); a8 = otter_amd.Bam(bam).ingest(regions, offset_l=1, offset_r=1, mapq=10, threads=8); t8 = time.perf_counter() - t
    print("rep %d: product %.2f s (%.0f regions/s, %d reads) on 1 thread, %.2f s (%.0f regions/s) on 8 threads; reference %.2f s (%.0f regions/s) on 1 thread" % (
        rep, t1, R / t1, len(a["reads"]), t8, R / t8, t2, R / t2), flush=True)
    TI._same(a8, b)
TI._same(a, b)
print("identical batches")
