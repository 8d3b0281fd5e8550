"""File-to-text timing of the `otter assemble` drop-in on one GPU box: BED file + BAM/BAI -> otg_parse_bed_file -> otg_ingest_regions
(host threads) -> otg_assemble_submit / run / collect (MI355X) -> otg_emit_alleles, with the time of every stage.  The BAM is
synthetic: tandem-repeat loci with two alleles, ONT-like reads whose CIGARs are written alongside the errors that make them (no
aligner needed), converted to BAM/BAI by the reference's htslib-lite (oracle/_ref/libotter_ref_io.so, prebuilt).
usage: python scripts/bench_e2e.py [regions=1000] [reads=30] [threads=16]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import otter_amd
from otter_amd import abi
import oracle_lib

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 30
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = np.random.default_rng(7)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def noisy(seq, rate=0.07):
    """seq (uint8 array) with substitutions / insertions / deletions at `rate`; returns (read bytes, CIGAR ops as (len, op) list)."""
    n = len(seq)
    kind = rng.random(n)
    out, ops = [], []

    def push(l, o):
        if ops and ops[-1][1] == o:
            ops[-1][0] += l
        else:
            ops.append([l, o])
    sub = kind < rate * 0.45
    ins = (kind >= rate * 0.45) & (kind < rate * 0.72)
    dele = (kind >= rate * 0.72) & (kind < rate)
    s2 = seq.copy()
    s2[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
    i = 0
    edges = np.flatnonzero(ins | dele)
    for e in edges:
        if e > i:
            out.append(s2[i:e]); push(int(e - i), "M")
        if ins[e]:
            k = int(rng.integers(1, 4))
            out.append(ACGT[rng.integers(0, 4, k)]); push(k, "I")
            out.append(s2[e:e + 1]); push(1, "M")
        else:
            push(1, "D")
        i = e + 1
    if i < n:
        out.append(s2[i:]); push(n - i, "M")
    return np.concatenate(out) if out else np.zeros(0, np.uint8), ops


tmp = tempfile.mkdtemp()
t0 = time.perf_counter()
ref_parts, regions, recs = [], [], []
pos = 0
flank = 1200
for r in range(R):
    motif = ACGT[rng.integers(0, 4, int(rng.integers(2, 7)))]
    L = int(rng.integers(1000, 5000))
    tr = np.tile(motif, L // len(motif) + 1)[:L]
    fl, fr = ACGT[rng.integers(0, 4, flank)], ACGT[rng.integers(0, 4, flank)]
    start = pos + flank
    ref_parts += [fl, tr, fr]
    regions.append(("chrS", start, start + L))
    delta = [0, int(rng.integers(-40, 41)) * len(motif)]          # allele 2 differs by whole copies
    for d in range(D):
        a = d % 2
        lf, rf = int(rng.integers(200, 900)), int(rng.integers(200, 900))
        body = tr if delta[a] >= 0 else tr[:L + delta[a]]
        left, ops_l = noisy(fl[flank - lf:])
        mid, ops_m = noisy(body)
        right, ops_r = noisy(fr[:rf])
        ops = ops_l + ops_m
        extra = np.zeros(0, np.uint8)
        if delta[a] > 0:
            extra = np.tile(motif, delta[a] // len(motif))
            ops = ops + [[len(extra), "I"]]
        elif delta[a] < 0:
            ops = ops + [[-delta[a], "D"]]
        ops = ops + ops_r
        merged = []
        for l, o in ops:
            if merged and merged[-1][1] == o:
                merged[-1][0] += l
            else:
                merged.append([l, o])
        read = np.concatenate([left, mid, extra, right])
        recs.append((start - lf, "r%d_%d" % (r, d), "".join("%d%s" % (l, o) for l, o in merged), read.tobytes().decode()))
    pos += flank + L + flank
ref_len = pos
recs.sort(key=lambda x: x[0])
sam, bam, bed = os.path.join(tmp, "e.sam"), os.path.join(tmp, "e.bam"), os.path.join(tmp, "e.bed")
with open(sam, "w") as f:
    f.write("@HD\tVN:1.4\tSO:coordinate\n@SQ\tSN:chrS\tLN:%d\n" % ref_len)
    for p, nm, cg, sq in recs:
        f.write("%s\t0\tchrS\t%d\t60\t%s\t*\t0\t0\t%s\t*\n" % (nm, p + 1, cg, sq))
with open(bed, "w") as f:
    for c, s, e in regions:
        f.write("%s\t%d\t%d\n" % (c, s, e))
n = oracle_lib.ref_io().ref_sam_to_bam(sam.encode(), bam.encode())
print("fixture: %d regions, %d records, BAM %.1f MB (%.1f s to build)" % (R, n, os.path.getsize(bam) / 1e6, time.perf_counter() - t0), flush=True)

import ctypes as C
L_ = otter_amd.load()
ctx = otter_amd.Context(0)
P = abi.default_params()
for rep in range(3):
    t = [time.perf_counter()]
    beds, carena, _ = otter_amd.parse_bed_file(bed); t.append(time.perf_counter())
    b = otter_amd.Bam(bam); targets = b.targets(); t.append(time.perf_counter())
    # ingest through the C-ABI with buffers sized once (a caller would reuse them)
    if rep == 0:
        reads = np.zeros(R * D + 1024, dtype=abi.read_dt); arena = np.zeros(R * D * 7000, dtype=np.uint8); regs = np.zeros(R, dtype=abi.region_dt)
    opts = np.zeros(1, dtype=abi.ingest_opts_dt); opts[0]["offset_l"] = 1; opts[0]["offset_r"] = 1; opts[0]["mapq"] = 10; opts[0]["threads"] = T
    used, nr = C.c_uint64(0), C.c_uint32(0)
    rc = L_.otg_ingest_regions(b._h, abi.ptr(beds), abi.ptr(carena, C.c_char_p), C.c_uint32(R), abi.ptr(opts), abi.ptr(arena), C.c_uint64(arena.size), C.byref(used),
                               abi.ptr(reads), C.c_uint32(len(reads)), C.byref(nr), abi.ptr(regs))
    assert rc == 0, rc
    batch = {"arena": arena[:used.value + 64], "reads": reads[:nr.value], "regions": regs}
    t.append(time.perf_counter())
    ctx.assemble_submit(P, batch); t.append(time.perf_counter())
    ctx.assemble_run(); t.append(time.perf_counter())
    res = ctx.assemble_collect(); t.append(time.perf_counter())
    text = otter_amd.emit_sam_header(targets, "s1", 1, 1) + otter_amd.emit_alleles(beds, carena, res, "s1", False); t.append(time.perf_counter())
    names = ["bed", "bam open", "ingest (%d threads)" % T, "submit (H2D)", "hot path (GPU)", "collect (D2H)", "emit"]
    dt = [t[i + 1] - t[i] for i in range(len(names))]
    tot = t[-1] - t[0]
    print("rep %d: %d reads, %d alleles, %.1f MB of SAM text; total %.3f s = %.0f regions/s end to end | " % (rep, nr.value, len(res["alleles"]), len(text) / 1e6, tot, R / tot)
          + ", ".join("%s %.0f ms" % (nm, 1000 * x) for nm, x in zip(names, dt)), flush=True)
ok = int((res["regions"]["status"] == 0).sum())
fc = res["regions"]["fc"]
print("regions OK %d / %d; alleles per region: %s" % (ok, R, np.bincount(fc[fc >= 0], minlength=3)[:4].tolist()))

# ---- the same work in two halves, the ingest of the second half overlapping the GPU run of the first (the C-ABI calls release the
# interpreter lock; a C++ host would use a thread the same way)
import threading
half = R // 2
bh = otter_amd.Bam(bam)


def ingest_range(lo, hi, slot):
    rd = np.zeros((hi - lo) * D + 1024, dtype=abi.read_dt); ar = np.zeros((hi - lo) * D * 7000, dtype=np.uint8); rg = np.zeros(hi - lo, dtype=abi.region_dt)
    o = np.zeros(1, dtype=abi.ingest_opts_dt); o[0]["offset_l"] = 1; o[0]["offset_r"] = 1; o[0]["mapq"] = 10; o[0]["threads"] = T
    u, k = C.c_uint64(0), C.c_uint32(0)
    sub = np.ascontiguousarray(beds[lo:hi])
    rc_ = L_.otg_ingest_regions(bh._h, abi.ptr(sub), abi.ptr(carena, C.c_char_p), C.c_uint32(hi - lo), abi.ptr(o), abi.ptr(ar), C.c_uint64(ar.size), C.byref(u),
                                abi.ptr(rd), C.c_uint32(len(rd)), C.byref(k), abi.ptr(rg))
    assert rc_ == 0
    slot.append((sub, {"arena": ar[:u.value + 64], "reads": rd[:k.value], "regions": rg}))


for rep in range(2):
    t0 = time.perf_counter()
    s1, s2 = [], []
    ingest_range(0, half, s1)
    th = threading.Thread(target=ingest_range, args=(half, R, s2)); th.start()
    out = []
    for slot in (s1, s2):
        if slot is s2:
            th.join()
        sub, bt = slot[0]
        ctx.assemble_submit(P, bt); ctx.assemble_run(); rs = ctx.assemble_collect()
        out.append(otter_amd.emit_alleles(sub, carena, rs, "s1", False))
    tot = time.perf_counter() - t0
    print("overlapped, 2 batches of %d: total %.3f s = %.0f regions/s end to end (%d MB of text)" % (half, tot, R / tot, sum(len(x) for x in out) // 1000000), flush=True)
