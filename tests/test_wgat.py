"""`otter wgat`: the product's otg_wgat against the REFERENCE's own wgat() (src/wgat.cpp + src/opinterval.cpp + its
interval tree, compiled from its sources into oracle/_ref/libotter_ref_io.so) on synthetic whole-genome alignments: contigs with every
CIGAR operation, clips at either end, regions inside deletions, at alignment edges, overlapping one another, equal coordinates on two
chromosomes, many regions (the tree's inner nodes), offsets.  Byte-identical text, SAM and FASTA.  Host code: runs without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest
import otter_amd
from otter_amd import bamwrite
import oracle_lib

needs_ref = pytest.mark.skipif(oracle_lib.ref_io() is None, reason="oracle/_ref/libotter_ref_io.so not built")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _ref_wgat(bam, bed, rg, fasta, ol, orr):
    R = oracle_lib.ref_io()
    R.ref_wgat.restype = C.c_uint64
    R.ref_wgat.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_uint64]
    cap = 64 << 20
    buf = C.create_string_buffer(cap)
    n = R.ref_wgat(bam.encode(), bed.encode(), rg.encode(), int(fasta), ol, orr, buf, cap)
    assert n <= cap
    return buf.raw[:n]


def _make_wga(tmpdir, seed, n_contigs=40, n_beds=300):
    rng = np.random.default_rng(seed)
    chroms = [("chrA", 400_000), ("chrB", 250_000), ("chrC", 30_000)]
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    recs = []
    for c in range(n_contigs):
        tid = int(rng.integers(0, len(chroms)))
        clen = chroms[tid][1]
        pos = int(rng.integers(0, clen - 25_000))
        ops = []
        # contigs are clipped at both ends (an alignment that starts inside a region WITHOUT a clip makes the reference read outside the
        # sequence — undefined output; the product skips those, test_wgat_unclipped_edges_are_skipped)
        ops.append((int(rng.integers(1, 300)), "SH"[int(rng.integers(0, 2))]))
        n_mid = int(rng.integers(3, 60))
        for j in range(n_mid):
            ops.append((int(rng.integers(1, 700)), "M=X"[int(rng.choice([0, 0, 0, 1, 2]))]))
            if j + 1 < n_mid:
                k = rng.random()
                if k < 0.35:
                    ops.append((int(rng.integers(1, 400)), "I"))
                elif k < 0.8:
                    ops.append((int(rng.integers(1, 900)), "D"))
                elif k < 0.85:
                    ops.append((int(rng.integers(1, 200)), "N"))
                elif k < 0.9:
                    ops.append((int(rng.integers(1, 5)), "P"))
        ops.append((int(rng.integers(1, 300)), "SH"[int(rng.integers(0, 2))]))
        if rng.random() < 0.1:                      # both kinds of clip at one end: equal (start, stop) keys for the sort
            ops = [(7, "H"), (9, "S")] + [o for o in ops if o[1] not in "SH"] + [(4, "S"), (2, "H")]
        qlen = sum(l for l, o in ops if o in "MIS=X")
        seq = acgt[rng.choice(5, qlen, p=[0.24, 0.24, 0.24, 0.24, 0.04])]
        if rng.random() < 0.05:
            seq = np.zeros(0, np.uint8); ops = [(l, o) for l, o in ops if o not in "MIS=X"] or [(5, "D")]      # no sequence: skipped, not counted
        recs.append((tid, pos, "contig%d" % c, int(rng.choice([0, 16, 2048])), 60, ops, seq, b""))
    recs.sort(key=lambda r: (r[0], r[1]))
    bam = os.path.join(tmpdir, "wga.bam")
    bamwrite.write_bam(bam, chroms, recs)
    beds = []
    for _ in range(n_beds):
        tid = int(rng.integers(0, len(chroms)))
        s = int(rng.integers(0, chroms[tid][1] - 3000))
        beds.append((chroms[tid][0], s, s + int(rng.integers(1, 2500))))
    beds += [("chrA", 1000, 1500), ("chrB", 1000, 1500), ("chrC", 1000, 1500), ("chrA", 1000, 1500), ("chrZ", 5, 9)]
    for r in recs[:10]:                             # regions hugging alignment edges and inner op boundaries
        rp = r[1]
        beds.append((chroms[r[0]][0], max(0, rp - 3), rp + 50))
        for l, o in r[5][:6]:
            if o in "M=XDN":
                beds.append((chroms[r[0]][0], rp + l - 1, rp + l + 1)); rp += l
    bed = os.path.join(tmpdir, "wga.bed")
    with open(bed, "w") as f:
        for c, s, e in beds:
            f.write("%s\t%d\t%d\n" % (c, s, e))
    return bam, bed


@needs_ref
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_wgat_matches_reference(tmp_path, seed):
    bam, bed = _make_wga(str(tmp_path), seed)
    b = otter_amd.Bam(bam)
    regions = otter_amd.parse_bed_file(bed)[:2]
    total = 0
    for fasta in (False, True):
        for ol, orr in ((1, 0), (0, 0), (25, 40)):
            exp = _ref_wgat(bam, bed, "asm1", fasta, ol, orr)
            got, n = otter_amd.wgat(b, regions, "asm1", fasta, ol, orr)
            assert got == exp, (fasta, ol, orr)
            total += n
    assert total > 200
    b.close()


def test_wgat_committed_golden():
    """Runs everywhere: tests/golden/wgat_small.bam (+ .bai, .bed) and the text the reference's wgat() printed for it (scripts/make_golden_wgat.py)."""
    b = otter_amd.Bam(os.path.join(GOLD, "wgat_small.bam"))
    regions = otter_amd.parse_bed_file(os.path.join(GOLD, "wgat_small.bed"))[:2]
    for fasta, name in ((False, "wgat_small.sam.txt"), (True, "wgat_small.fa.txt")):
        got, n = otter_amd.wgat(b, regions, "asm1", fasta, 1, 0)
        assert got == open(os.path.join(GOLD, name), "rb").read() and n > 20
    b.close()


def test_wgat_unclipped_edges_are_skipped(tmp_path):
    """An alignment that begins / ends inside a region without a clip: no record (the reference reads outside the sequence there)."""
    seq = np.frombuffer(b"ACGT" * 50, dtype=np.uint8)
    bam = os.path.join(str(tmp_path), "e.bam")
    bamwrite.write_bam(bam, [("chrA", 10000)], [(0, 1000, "c1", 0, 60, [(200, "M")], seq, b"")])
    b = otter_amd.Bam(bam)
    got, n = otter_amd.wgat(b, [("chrA", 990, 1050), ("chrA", 1150, 1250), ("chrA", 1050, 1100)], "a", True, 0, 0)
    b.close()
    assert n == 1 and got == b">a#c1#chrA:1050-1100#0#tc:i:1#ac:i:1#sc:i:1#sp:A:b\n" + seq[50:100].tobytes() + b"\n"
