"""File-to-text timing of the `otter assemble` drop-in on one GPU box: BED file + BAM/BAI -> otg_parse_bed_file -> otg_ingest_regions
(host threads) -> otg_assemble_submit / run / collect (MI355X) -> otg_emit_alleles, with the time of every stage.  The BAM is
synthetic: tandem-repeat loci with two alleles, ONT-like reads whose CIGARs are written alongside the errors that make them (no
aligner needed), written as BAM/BAI by otter_amd/bamwrite.py.
usage: python scripts/bench_e2e.py [regions=1000] [reads=30] [threads=16]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import otter_amd
from otter_amd import abi

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 30
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = np.random.default_rng(7)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


from otter_amd import bamwrite
tmp = tempfile.mkdtemp()
t0 = time.perf_counter()
fx = bamwrite.make_tr_fixture(tmp, R, depth=D, len_range=(1000, 5000), seed=7)
bam, bed, regions = fx["bam"], fx["bed"], fx["regions"]
print("fixture: %d regions, %d records, BAM %.1f MB (%.1f s to build)" % (R, fx["n_records"], os.path.getsize(bam) / 1e6, time.perf_counter() - t0), flush=True)

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

# ---- the library's dispatcher (otg_assemble_files): bounded batches, ingest / hot path / emit on concurrent host threads
for batch in (R, max(64, R // 4), max(64, R // 8)):
    for rep in range(2):
        t0 = time.perf_counter()
        txt, st = otter_amd.assemble_files(bam, bed, read_group="s1", batch_regions=batch, offset_l=1, offset_r=1, mapq=10, threads=T)
        tot = time.perf_counter() - t0
    print("dispatcher, batches of %d: total %.3f s = %.0f regions/s end to end (%d MB of text); stage busy ms: ingest %.0f, hot path %.0f, emit %.0f" % (
        batch, tot, R / tot, len(txt) // 1000000, st["ms_ingest"], st["ms_hot_path"], st["ms_emit"]), flush=True)
