"""End-to-end fixture of the integration seam: a synthetic reference + coordinate-sorted SAM of reads
over tandem-repeat regions, converted to BAM/BAI and ingested by the REFERENCE's own code (parse_anreads through
oracle/_ref/libotter_ref_io.so, built from the reference sources), giving exactly the region batch the reference's
worker would hand to the five hot-path calls.  Test infrastructure."""
import ctypes as C
import os
import numpy as np
from otter_amd import abi, synth
import oracle_lib

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rand(rng, n):
    return _ACGT[rng.integers(0, 4, n)].tobytes()


def _mut(rng, s, rate):
    code = np.searchsorted(_ACGT, np.frombuffer(s, dtype=np.uint8)).astype(np.uint8)
    return _ACGT[synth._mutate(rng, code, rate, (0.4, 0.25, 0.35))].tobytes()


def _sam_cigar(ops):
    """op string of (pattern = reference window, text = read) -> (leading reference bases to skip, SAM CIGAR).
    'I' (read only) -> I, 'D' (reference only) -> D, M / X -> M; gaps at the ends become soft clips / position shifts."""
    ops = ops.decode()
    lead_d = len(ops) - len(ops.lstrip("D"))
    ops = ops[lead_d:].rstrip("D")
    lead_i = len(ops) - len(ops.lstrip("I"))
    core = ops[lead_i:]
    trail_i = len(core) - len(core.rstrip("I"))
    core = core[:len(core) - trail_i] if trail_i else core
    out = []
    if lead_i:
        out.append("%dS" % lead_i)
    run, prev = 0, None
    for o in core:
        o = "M" if o in "MX" else o
        if o == prev:
            run += 1
        else:
            if prev:
                out.append("%d%s" % (run, prev))
            prev, run = o, 1
    if prev:
        out.append("%d%s" % (run, prev))
    if trail_i:
        out.append("%dS" % trail_i)
    return lead_d, "".join(out)


def make_dataset(tmpdir, n_regions=8, seed=61, err=0.01, depth=12):
    """Writes ref.fa, reads.sam (sorted), regions (list of (chr, start, end)); returns paths + regions."""
    rng = np.random.default_rng(seed)
    chrom = "chrT"
    ref = bytearray(_rand(rng, 3000))
    regions, tracts = [], []
    for r in range(n_regions):
        motif = _rand(rng, int(rng.integers(2, 7)))
        ncopy = int(rng.integers(40, 120))
        start = len(ref)
        ref += motif * ncopy
        regions.append((chrom, start, len(ref)))
        tracts.append((motif, ncopy))
        ref += _rand(rng, 2500)
    ref = bytes(ref)
    recs = []
    pairs, meta = [], []
    for r, ((c, start, end), (motif, ncopy)) in enumerate(zip(regions, tracts)):
        alleles = [ncopy + int(rng.integers(-6, 7)), ncopy + int(rng.integers(-25, 26))]
        for i in range(depth):
            a = alleles[i % 2]
            fl, fr = int(rng.integers(300, 1200)), int(rng.integers(300, 1200))
            w0, w1 = start - fl, end + fr
            read = ref[w0:start] + motif * max(a, 1) + ref[end:w1]
            kind = i % 6
            if kind == 4:      # starts inside the tract: spans only the right side
                cut = fl + int(rng.integers(5, len(motif) * max(a, 1) // 2))
                read = read[cut:]; w0 = start + (cut - fl) * ncopy // max(a, 1)
                w0 = min(max(w0, start), end - 1)
            elif kind == 5:    # ends inside the tract: spans only the left side
                cut = fr + int(rng.integers(5, len(motif) * max(a, 1) // 2))
                read = read[:len(read) - cut]; w1 = max(end - (cut - fr) * ncopy // max(a, 1), start + 1)
            read = _mut(rng, read, err)
            if kind == 3:      # a mapper's soft clip: foreign bases at the left end
                read = _rand(rng, int(rng.integers(5, 40))) + read
            pairs.append((ref[w0:w1], read))
            meta.append((r, i, w0, read, kind))
    from helpers import pair_tasks
    arena, tasks = pair_tasks(pairs)
    _, cigs = oracle_lib.affine_align_batch(arena, tasks)
    for (r, i, w0, read, kind), ops in zip(meta, cigs):
        skip, cigar = _sam_cigar(ops)
        if not cigar or all(ch in "0123456789S" for ch in cigar):
            continue
        tags = ""
        if i % 5 == 1:
            tags = "\tHP:i:%d\tPS:i:%d" % (1 + i % 2, 1000 + r)
        if i % 7 == 2:
            tags += "\trq:f:0.99"
        flag = 0 if i % 11 else 256          # a few secondary alignments (dropped unless --non-primary)
        mapq = 60 if i % 13 else 3
        recs.append((w0 + skip, "r%d_%d" % (r, i), flag, mapq, cigar, read, tags))
    recs.sort(key=lambda x: x[0])
    fa = os.path.join(tmpdir, "ref.fa")
    with open(fa, "w") as f:
        f.write(">%s\n" % chrom)
        for j in range(0, len(ref), 60):
            f.write(ref[j:j + 60].decode() + "\n")
    sam = os.path.join(tmpdir, "reads.sam")
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.4\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (chrom, len(ref)))
        for pos, name, flag, mapq, cigar, read, tags in recs:
            f.write("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t*%s\n" % (name, flag, chrom, pos + 1, mapq, cigar, read.decode(), tags))
    return {"fasta": fa, "sam": sam, "regions": regions, "chrom": chrom, "ref_len": len(ref), "n_records": len(recs)}


def ingest_with_reference(ds, tmpdir, offset_l=1, offset_r=1, mapq=10, nonprimary=False, read_quality=0.0, omitnonspanning=False,
                          flank=0):
    """BAM/BAI through the reference's htslib-lite, then parse_anreads per region -> batch dict for assemble_submit."""
    R = oracle_lib.ref_io()
    bam = os.path.join(tmpdir, "reads.bam")
    n = R.ref_sam_to_bam(ds["sam"].encode(), bam.encode())
    assert n == ds["n_records"], n
    R.ref_ingest_open.restype = C.c_void_p
    R.ref_ingest_region.restype = C.c_int64
    h = C.c_void_p(R.ref_ingest_open(bam.encode(), ds["fasta"].encode() if flank else b""))
    cap_reads, cap_arena = 4096, 64 << 20
    reads = np.zeros(cap_reads, dtype=abi.read_dt)
    arena = np.zeros(cap_arena, dtype=np.uint8)
    used = C.c_uint64(0)
    regions = np.zeros(len(ds["regions"]), dtype=abi.region_dt)
    nr = 0
    for r, (c, s, e) in enumerate(ds["regions"]):
        sub = reads[nr:]
        k = R.ref_ingest_region(h, c.encode(), C.c_int(s), C.c_int(e), C.c_int(offset_l), C.c_int(offset_r), C.c_int(mapq), C.c_int(int(nonprimary)),
                                C.c_double(read_quality), C.c_int(int(omitnonspanning)), abi.ptr(sub), C.c_uint64(len(sub)), abi.ptr(arena),
                                C.c_uint64(cap_arena), C.byref(used))
        assert k >= 0, k
        regions[r]["first_read"] = nr; regions[r]["n_reads"] = k
        nr += k
        if flank:
            buf = C.create_string_buffer(flank + 8)
            for side, (b, e2) in (("l", (s - offset_l - flank, s - offset_l)), ("r", (e + offset_r, e + offset_r + flank))):
                ln = R.ref_fetch(h, c.encode(), C.c_int(b), C.c_int(e2), buf, C.c_int(flank + 8))
                regions[r]["flank_%s_off" % side] = used.value; regions[r]["flank_%s_len" % side] = ln
                arena[used.value:used.value + ln] = np.frombuffer(buf.raw[:ln], dtype=np.uint8)
                used.value += ln
    R.ref_ingest_close(h)
    return {"arena": np.ascontiguousarray(arena[:used.value + 64]), "reads": np.ascontiguousarray(reads[:nr]), "regions": regions}
