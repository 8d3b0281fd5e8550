"""Minimal BAM + BAI writer for synthetic fixtures (bench.py's end-to-end leg, scripts/bench_e2e.py): coordinate-sorted alignments ->
BGZF-compressed BAM and its `.bai`.  Written from the SAM/BAM specification (records, BGZF framing, the binning index with its
16-kb linear index); numpy + zlib only — no reference code, no oracle.  tests/test_ingest.py checks that the reference's own reader
(built from its sources, test infrastructure) and the product's reader see the same alignments in files written here."""
import struct
import zlib

import numpy as np

_CIGAR_OPS = {c: i for i, c in enumerate("MIDNSHP=X")}
_NT16 = np.full(256, 15, dtype=np.uint8)
for _i, _c in enumerate("=ACMGRSVTWYHKDBN"):
    _NT16[ord(_c)] = _i
    _NT16[ord(_c.lower())] = _i
_EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


class _Bgzf:
    def __init__(self, path, level=1):
        self.f = open(path, "wb")
        self.buf = bytearray()
        self.level = level
        self.block_start = 0            # file offset of the block being filled

    def tell(self):                     # virtual offset of the next byte
        return (self.block_start << 16) | len(self.buf)

    def _flush_block(self, n):
        data = bytes(self.buf[:n])
        del self.buf[:n]
        co = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        bsize = len(comp) + 25
        self.f.write(struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize))
        self.f.write(comp)
        self.f.write(struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
        self.block_start += bsize + 1

    def write(self, b):
        self.buf += b
        while len(self.buf) >= 0xff00:
            self._flush_block(0xff00)

    def flush(self):
        if self.buf:
            self._flush_block(len(self.buf))

    def close(self):
        self.flush()
        self.f.write(_EOF_BLOCK)
        self.f.close()


def _parse_cigar(cigar):
    ops, n = [], 0
    for ch in cigar:
        if ch.isdigit():
            n = n * 10 + ord(ch) - 48
        else:
            ops.append((n, _CIGAR_OPS[ch]))
            n = 0
    return ops


def write_bam(path, targets, records, level=1, extra_header=""):
    """targets: [(name, length)]; records: iterable of (tid, pos0, name, flag, mapq, cigar (str, [(len, opchar)] or (lengths, BAM op codes) as arrays), seq (bytes or uint8 array), tags bytes)
    sorted by (tid, pos0).  Writes `path` and `path + '.bai'`.  Returns the number of records."""
    bg = _Bgzf(path, level)
    text = "@HD\tVN:1.4\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in targets) + extra_header
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(targets))
    for n, l in targets:
        nb = n.encode() + b"\x00"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    bg.write(hdr)
    bg.flush()                                   # records start on a block boundary
    bins = [dict() for _ in targets]             # bin -> list of [beg, end] chunks
    linear = [dict() for _ in targets]           # window -> min voffset
    count = 0
    for tid, pos, name, flag, mapq, cigar, seq, tags in records:
        if isinstance(cigar, tuple):          # (lengths, BAM op codes) as arrays
            cl, co = cigar
            n_ops = int(len(cl))
            ops_bytes = ((cl.astype(np.uint32) << 4) | co.astype(np.uint32)).astype("<u4").tobytes()
            rlen = int(cl[np.isin(co, (0, 2, 3, 7, 8))].sum())
        else:
            ops = _parse_cigar(cigar) if isinstance(cigar, str) else [(l, _CIGAR_OPS[o]) for l, o in cigar]
            n_ops = len(ops)
            ops_bytes = b"".join(struct.pack("<I", (l << 4) | o) for l, o in ops)
            rlen = sum(l for l, o in ops if o in (0, 2, 3, 7, 8))
        s = np.frombuffer(seq, dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else np.asarray(seq, dtype=np.uint8)
        lseq = int(s.size)
        end = pos + (rlen if rlen > 0 else 1)
        b = reg2bin(pos, end)
        codes = _NT16[s]
        if lseq & 1:
            codes = np.concatenate([codes, np.zeros(1, dtype=np.uint8)])
        packed = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()
        nb = name.encode() + b"\x00"
        if n_ops > 65535:
            raise ValueError("more than 65535 CIGAR operations: not supported by this writer")
        body = struct.pack("<iiBBHHHiiii", tid, pos, len(nb), mapq, b, n_ops, flag, lseq, -1, -1, 0) + nb + \
            ops_bytes + packed + b"\xff" * lseq + (tags or b"")
        v0 = bg.tell()
        bg.write(struct.pack("<i", len(body)) + body)
        v1 = bg.tell()
        if tid >= 0:
            ch = bins[tid].setdefault(b, [])
            if ch and ch[-1][1] == v0:
                ch[-1][1] = v1
            else:
                ch.append([v0, v1])
            for w in range(pos >> 14, ((end - 1) >> 14) + 1):
                if w not in linear[tid]:
                    linear[tid][w] = v0
        count += 1
    bg.close()
    with open(path + ".bai", "wb") as f:
        f.write(b"BAI\x01" + struct.pack("<i", len(targets)))
        for tid in range(len(targets)):
            f.write(struct.pack("<i", len(bins[tid])))
            for b in sorted(bins[tid]):
                ch = bins[tid][b]
                f.write(struct.pack("<Ii", b, len(ch)))
                for beg, en in ch:
                    f.write(struct.pack("<QQ", beg, en))
            nw = (max(linear[tid]) + 1) if linear[tid] else 0
            f.write(struct.pack("<i", nw))
            last = 0
            for w in range(nw):                  # windows without a record take the previous offset, as samtools writes them
                last = linear[tid].get(w, last)
                f.write(struct.pack("<Q", last))
    return count


def make_tr_fixture(dirname, n_regions, depth=30, len_range=(1000, 5000), seed=7, rate=0.07, flank=1200):
    """Synthetic tandem-repeat loci with two alleles and ONT-like reads whose CIGARs are written alongside the errors that make them (no
    aligner needed): writes reads.bam (+ .bai), regions.bed and ref.fa (+ nothing else) into `dirname`.  Returns dict(bam, bed, fasta, regions)."""
    import os
    rng = np.random.default_rng(seed)
    ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

    def noisy(seq):
        """ONT-like errors on `seq` with the op list that describes them, built without a Python loop over the bases: every reference
        base contributes its inserted bases (0-3, op I) and then itself (op M) or nothing (op D)."""
        n = len(seq)
        kind = rng.random(n)
        sub = kind < rate * 0.45
        ins = (kind >= rate * 0.45) & (kind < rate * 0.72)
        dele = (kind >= rate * 0.72) & (kind < rate)
        s2 = seq.copy()
        s2[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
        ilen = np.zeros(n, dtype=np.int64)
        ilen[ins] = rng.integers(1, 4, int(ins.sum()))
        keep = ~dele
        # read bases: per reference base `ilen` random bases, then the base itself unless deleted
        cnt = ilen + keep
        end = np.cumsum(cnt)
        out = ACGT[rng.integers(0, 4, int(end[-1]) if n else 0)]
        if n:
            out[(end - 1)[keep]] = s2[keep]
        # ops: per reference base `ilen` x I, then M or D; run-length encoded
        ocnt = ilen + 1
        oend = np.cumsum(ocnt)
        codes = np.ones(int(oend[-1]) if n else 0, dtype=np.uint8)          # 1 = I
        if n:
            codes[oend - 1] = np.where(keep, 0, 2)                         # 0 = M, 2 = D
        return out, codes

    def rle(codes):
        """(lengths, BAM op codes 0 = M, 1 = I, 2 = D) of a per-op code array."""
        if codes.size == 0:
            return np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8)
        starts = np.concatenate(([0], np.flatnonzero(np.diff(codes)) + 1))
        lens = np.diff(np.concatenate((starts, [codes.size]))).astype(np.uint32)
        return lens, codes[starts]

    ref_parts, regions, recs = [], [], []
    pos = 0
    for r in range(n_regions):
        motif = ACGT[rng.integers(0, 4, int(rng.integers(2, 7)))]
        L = int(rng.integers(len_range[0], len_range[1]))
        tr = np.tile(motif, L // len(motif) + 1)[:L]
        fl, fr = ACGT[rng.integers(0, 4, flank)], ACGT[rng.integers(0, 4, flank)]
        start = pos + flank
        ref_parts += [fl, tr, fr]
        regions.append(("chrS", start, start + L))
        delta = [0, int(rng.integers(-40, 41)) * len(motif)]
        for d in range(depth):
            a = d % 2
            lf, rf = int(rng.integers(200, 900)), int(rng.integers(200, 900))
            body = tr if delta[a] >= 0 else tr[:L + delta[a]]
            left, c_l = noisy(fl[flank - lf:])
            mid, c_m = noisy(body)
            right, c_r = noisy(fr[:rf])
            extra = np.zeros(0, np.uint8)
            c_x = np.zeros(0, np.uint8)
            if delta[a] > 0:
                extra = np.tile(motif, delta[a] // len(motif))
                c_x = np.full(len(extra), 1, dtype=np.uint8)
            elif delta[a] < 0:
                c_x = np.full(-delta[a], 2, dtype=np.uint8)
            recs.append((0, start - lf, "r%d_%d" % (r, d), 0, 60, rle(np.concatenate([c_l, c_m, c_x, c_r])), np.concatenate([left, mid, extra, right]), b""))
        pos += flank + L + flank
    recs.sort(key=lambda x: x[1])
    ref = np.concatenate(ref_parts)
    bam, bed, fa = os.path.join(dirname, "reads.bam"), os.path.join(dirname, "regions.bed"), os.path.join(dirname, "ref.fa")
    write_bam(bam, [("chrS", int(ref.size))], recs)
    with open(bed, "w") as f:
        for c, s, e in regions:
            f.write("%s\t%d\t%d\n" % (c, s, e))
    with open(fa, "w") as f:
        f.write(">chrS\n")
        rb = ref.tobytes().decode()
        for i in range(0, len(rb), 60):
            f.write(rb[i:i + 60] + "\n")
    return {"bam": bam, "bed": bed, "fasta": fa, "regions": regions, "n_records": len(recs)}


def make_genotype_fixture(dirname, n_regions, n_samples=50, len_range=(1000, 5000), seed=11, flank=300):
    """The input of `otter genotype` at BASELINE configs[3]'s shape: ONE merged allele BAM as `otter assemble` writes it per sample (a read group per
    sample, `@PG ID:otter OF:l,r`, records = allele sequences with the tags RG / ta / tc / ac / sc / ic / se; src/anseqs.cpp:42-63) holding
    2 x n_samples allele records per tandem-repeat locus (otter_amd.synth.make_genotype_batch: population alleles + consensus-level errors), a BED
    file and a reference FASTA whose sequence under each region is the locus's reference allele.  Writes alleles.bam (+ .bai), regions.bed, ref.fa."""
    import os
    import struct as st
    from . import synth
    gb = synth.make_genotype_batch(n_regions, n_samples=n_samples, len_range=len_range, seed=seed)
    rng = np.random.default_rng(seed + 1)
    ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
    arena, off, ln, first, smp = gb["arena"], gb["seq_off"], gb["seq_len"], gb["first_allele"], gb["sample"]
    ref_parts, regions, recs = [], [], []
    pos = 0
    names = ["smp%02d" % i for i in range(n_samples)]
    for r in range(n_regions):
        a0, a1 = int(first[r]), int(first[r + 1])
        refal = arena[int(off[a1 - 1]):int(off[a1 - 1]) + int(ln[a1 - 1])]        # the last allele of a region is the reference allele
        start = pos + flank
        end = start + int(refal.size)
        ref_parts += [ACGT[rng.integers(0, 4, flank)], refal, ACGT[rng.integers(0, 4, flank)]]
        regions.append(("chrG", start, end))
        ta = ("chrG:%d-%d" % (start, end)).encode()
        seen = {}
        for a in range(a0, a1 - 1):
            sm = int(smp[a]); k = seen.get(sm, 0); seen[sm] = k + 1
            seq = arena[int(off[a]):int(off[a]) + int(ln[a])]
            cov = int(rng.integers(8, 20))
            tags = (b"RGZ" + names[sm].encode() + b"\0" + b"taZ" + ta + b"\0" + b"tcC" + bytes([cov * 2]) + b"acC" + bytes([cov]) + b"scC" + bytes([max(1, cov - 2)]) +
                    b"icC" + bytes([2]) + b"sef" + st.pack("<f", float(rng.integers(0, 50)) / 1000.0))
            recs.append((0, start - 1, "chrG:%d-%d_%d" % (start, end, k), 0, 0, [(int(seq.size), "M")], seq, tags))
        pos = end + flank
    ref = np.concatenate(ref_parts)
    bam, bed, fa = os.path.join(dirname, "alleles.bam"), os.path.join(dirname, "regions.bed"), os.path.join(dirname, "ref.fa")
    extra = "".join("@RG\tID:%s\n" % n for n in names) + "@PG\tID:otter\tOF:1,0\n"
    write_bam(bam, [("chrG", int(ref.size))], recs, extra_header=extra)             # generated in coordinate order already
    with open(bed, "w") as f:
        for c, s_, e in regions:
            f.write("%s\t%d\t%d\n" % (c, s_, e))
    with open(fa, "w") as f:
        f.write(">chrG\n")
        rb = ref.tobytes().decode()
        for i in range(0, len(rb), 60):
            f.write(rb[i:i + 60] + "\n")
    return {"bam": bam, "bed": bed, "fasta": fa, "regions": regions, "n_records": len(recs), "n_samples": n_samples}
