"""BAM/BAI region ingest: the product's own reader (otg_bam_open / otg_ingest_regions: BGZF on zlib, BAI,
CIGAR walk) against the REFERENCE's ingest built from its sources (parse_anreads over its htslib-lite,
oracle/_ref/libotter_ref_io.so), on BAM files written by that htslib-lite from synthetic SAM text.  Host code: runs
without a GPU.  Skipped when the reference build is absent."""
import ctypes as C
import os
import numpy as np
import pytest
import otter_amd
from otter_amd import abi
import oracle_lib

needs_ref = pytest.mark.skipif(oracle_lib.ref_io() is None, reason="oracle/_ref/libotter_ref_io.so not built")
GOLD = os.path.join(os.path.dirname(__file__), "golden")
GOLD_OPTS = [dict(), dict(offset_l=1, offset_r=1, mapq=10), dict(offset_l=50000, offset_r=7, nonprimary=True), dict(omit_nonspanning=True, mapq=3),
             dict(read_quality=0.4, nonprimary=True)]


def test_ingest_against_committed_golden():
    """Runs everywhere: tests/golden/ingest_small.bam(.bai) was written by the reference's htslib-lite and
    tests/golden/ingest_ref.npz holds what the reference's parse_anreads returned for it (scripts/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "ingest_ref.npz"))
    regions = [(str(c), int(s), int(e)) for c, s, e in zip(g["regions_chr"], g["regions_start"], g["regions_end"])]
    bam = otter_amd.Bam(os.path.join(GOLD, "ingest_small.bam"))
    assert [t[0] for t in bam.targets()] == ["chr1", "chr2_random:alt", "chrBig"]
    for i, kw in enumerate(GOLD_OPTS):
        for threads in (1, 3):
            got = bam.ingest(regions, threads=threads, **kw)
            exp = g["reads%d" % i]
            assert np.array_equal(got["regions"]["n_reads"], g["n_reads%d" % i])
            assert len(got["reads"]) == len(exp)
            for f in ("seq_off", "seq_len", "spanning_l", "spanning_r", "ps", "hp", "ccoord_first", "ccoord_second"):
                assert np.array_equal(got["reads"][f], exp[f]), (i, f)
            n = int(exp["seq_len"].astype(np.int64).sum())
            assert got["arena"][:n].tobytes() == g["arena%d" % i][:n].tobytes()
    bam.close()



def _random_sam(path, rng, chroms, n_records, read_len=(20, 400)):
    """Coordinate-sorted SAM with arbitrary (not biologically meaningful) CIGARs: every op the reference's walk
    distinguishes (M I D N S H P = X), clips at either end, zero-length reads, tags HP/PS/rq in several integer widths."""
    recs = []
    for i in range(n_records):
        ci = int(rng.integers(0, len(chroms)))
        name, clen = chroms[ci]
        pos = int(rng.integers(1, clen - 2000))
        ops = []
        if rng.random() < 0.3:
            ops.append((int(rng.integers(1, 40)), "SH"[int(rng.integers(0, 2))]))
        n_mid = int(rng.integers(1, 9))
        for j in range(n_mid):
            ops.append((int(rng.integers(1, 120)), "M=X"[int(rng.choice([0, 0, 0, 1, 2]))]))
            if j + 1 < n_mid:
                k = rng.random()
                if k < 0.35:
                    ops.append((int(rng.integers(1, 60)), "I"))
                elif k < 0.7:
                    ops.append((int(rng.integers(1, 300)), "D"))
                elif k < 0.8:
                    ops.append((int(rng.integers(1, 500)), "N"))
                elif k < 0.85:
                    ops.append((int(rng.integers(1, 5)), "P"))
        if rng.random() < 0.3:
            ops.append((int(rng.integers(1, 40)), "SH"[int(rng.integers(0, 2))]))
        qlen = sum(l for l, o in ops if o in "MIS=X")
        seq = "".join("ACGTN"[int(x)] for x in rng.choice(5, qlen, p=[0.24, 0.24, 0.24, 0.24, 0.04])) if qlen else "*"
        cigar = "".join("%d%s" % (l, o) for l, o in ops)
        tags = ""
        t = int(rng.integers(0, 6))
        if t == 1:
            tags = "\tHP:i:%d\tPS:i:%d" % (int(rng.integers(0, 3)), int(rng.integers(0, 70000)))
        elif t == 2:
            tags = "\tPS:i:%d" % int(rng.integers(0, 200))
        elif t == 3:
            tags = "\trq:f:%.4f\tHP:i:1" % float(rng.random())
        elif t == 4:
            tags = "\tXZ:Z:foo\tXB:B:c,1,2,3\trq:f:0.5\tXA:A:q"
        flag = int(rng.choice([0, 16, 256, 2048, 4, 1024]))
        recs.append((ci, pos, "q%d" % i, flag, int(rng.integers(0, 61)), cigar, seq, tags))
    recs.sort(key=lambda r: (r[0], r[1]))
    with open(path, "w") as f:
        f.write("@HD\tVN:1.4\tSO:coordinate\n")
        for name, clen in chroms:
            f.write("@SQ\tSN:%s\tLN:%d\n" % (name, clen))
        for ci, pos, qn, flag, mq, cigar, seq, tags in recs:
            f.write("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t*%s\n" % (qn, flag, chroms[ci][0], pos, mq, cigar, seq, tags))
    return len(recs)


def _ref_ingest(bam, regions, **kw):
    R = oracle_lib.ref_io()
    R.ref_ingest_open.restype = C.c_void_p
    R.ref_ingest_region.restype = C.c_int64
    h = C.c_void_p(R.ref_ingest_open(bam.encode(), b""))
    reads = np.zeros(1 << 21, dtype=abi.read_dt)
    arena = np.zeros(256 << 20, dtype=np.uint8)
    used = C.c_uint64(0)
    regs = np.zeros(len(regions), dtype=abi.region_dt)
    nr = 0
    for r, (c, s, e) in enumerate(regions):
        sub = reads[nr:]
        k = R.ref_ingest_region(h, c.encode(), C.c_int(s), C.c_int(e), C.c_int(kw.get("offset_l", 0)), C.c_int(kw.get("offset_r", 0)),
                                C.c_int(kw.get("mapq", 0)), C.c_int(int(kw.get("nonprimary", False))), C.c_double(kw.get("read_quality", 0.0)),
                                C.c_int(int(kw.get("omit_nonspanning", False))), abi.ptr(sub), C.c_uint64(len(sub)), abi.ptr(arena), C.c_uint64(arena.size), C.byref(used))
        assert k >= 0
        regs[r]["first_read"] = nr; regs[r]["n_reads"] = k
        nr += k
    R.ref_ingest_close(h)
    return {"arena": arena[:used.value], "reads": reads[:nr].copy(), "regions": regs}


def _same(a, b):
    assert np.array_equal(a["regions"]["n_reads"], b["regions"]["n_reads"]), (a["regions"]["n_reads"][:20], b["regions"]["n_reads"][:20])
    assert np.array_equal(a["regions"]["first_read"], b["regions"]["first_read"])
    for f in ("seq_len", "spanning_l", "spanning_r", "ps", "hp", "ccoord_first", "ccoord_second", "seq_off"):
        assert np.array_equal(a["reads"][f], b["reads"][f]), f
    n = int(b["reads"]["seq_len"].astype(np.int64).sum())
    assert a["arena"][:n].tobytes() == b["arena"][:n].tobytes()


@needs_ref
@pytest.mark.parametrize("seed,n_records", [(71, 3000), (72, 40000)])
def test_ingest_matches_reference_on_random_bam(tmp_path, seed, n_records):
    rng = np.random.default_rng(seed)
    chroms = [("chr1", 2_000_000), ("chr2_random:alt", 300_000), ("chrM", 17_000), ("chrBig", 400_000_000)]
    sam, bam = str(tmp_path / "x.sam"), str(tmp_path / "x.bam")
    n = _random_sam(sam, rng, chroms, n_records)
    assert oracle_lib.ref_io().ref_sam_to_bam(sam.encode(), bam.encode()) == n
    regions = []
    for _ in range(300):
        name, clen = chroms[int(rng.integers(0, len(chroms)))]
        if name == "chrBig" and rng.random() < 0.5:
            s = int(rng.integers(0, clen - 3000))
        else:
            s = int(rng.integers(0, min(clen, 2_000_000) - 3000))
        regions.append((name, s, s + int(rng.integers(1, 2500))))
    regions += [("chr1", 0, 10), ("chr1", 5, 5), ("chrM", 16_990, 17_050), ("nochr", 10, 20), ("chr1", 1_999_000, 2_100_000)]
    bamh = otter_amd.Bam(bam)
    assert bamh.targets() == chroms
    for kw in (dict(), dict(offset_l=1, offset_r=1, mapq=10), dict(offset_l=30, offset_r=31, nonprimary=True, read_quality=0.4),
               dict(omit_nonspanning=True, mapq=1), dict(offset_l=100000)):
        ref = _ref_ingest(bam, regions, **kw)
        _same(bamh.ingest(regions, **kw), ref)
        _same(bamh.ingest(regions, threads=5, **kw), ref)          # region slices on host threads, merged in region order
    bamh.close()


@needs_ref
def test_ingest_on_realistic_tr_bam(tmp_path):
    """The end-to-end fixture (reads aligned over tandem repeats): same batch as the reference's ingest."""
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=8, seed=64)
    ref = e2e_bam.ingest_with_reference(ds, str(tmp_path), offset_l=1, offset_r=1, mapq=10)
    got = otter_amd.Bam(os.path.join(str(tmp_path), "reads.bam")).ingest(ds["regions"], offset_l=1, offset_r=1, mapq=10)
    _same(got, {"arena": ref["arena"], "reads": ref["reads"], "regions": ref["regions"]})


def test_bam_open_errors(tmp_path):
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.Bam(str(tmp_path / "missing.bam"))
    p = tmp_path / "junk.bam"
    p.write_bytes(b"not a bam")
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.Bam(str(p))


@needs_ref
def test_ingest_long_cigar_in_cg_tag(tmp_path):
    """An alignment with more than 65535 CIGAR operations: BAM stores `<l_seq>S<rlen>N` in the record and the real CIGAR in tag
    CG:B,I (the reference's writer, src/sam.c:323-352); the reference's reader moves it back (bam_tag2cigar, src/sam.c:243-285).  The
    product reads the ops from the tag in place; dropped silently it would mark the read unsuccessful."""
    rng = np.random.default_rng(5)
    n_pairs = 34000                                   # 1M1D x 34000 + tails = 68003 ops
    ops = "5S" + "1M1D" * n_pairs + "40M" + "3S"
    qlen = 5 + n_pairs + 40 + 3
    seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, qlen))
    short = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 300))
    sam, bam = str(tmp_path / "cg.sam"), str(tmp_path / "cg.bam")
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.4\tSO:coordinate\n@SQ\tSN:chr1\tLN:1000000\n")
        f.write("long\t0\tchr1\t1000\t60\t%s\t*\t0\t0\t%s\t*\tHP:i:2\tPS:i:77\n" % (ops, seq))
        f.write("short\t0\tchr1\t30000\t60\t300M\t*\t0\t0\t%s\t*\n" % short)
    assert oracle_lib.ref_io().ref_sam_to_bam(sam.encode(), bam.encode()) == 2
    regions = [("chr1", 1500, 1600), ("chr1", 29990, 30100), ("chr1", 68000, 69050), ("chr1", 900, 1010), ("chr1", 69030, 69100)]
    bamh = otter_amd.Bam(bam)
    for kw in (dict(), dict(offset_l=1, offset_r=1)):
        ref = _ref_ingest(bam, regions, **kw)
        assert ref["regions"]["n_reads"][0] == 1 and ref["reads"]["seq_len"][0] > 40     # the long read is used, with its real CIGAR
        _same(bamh.ingest(regions, **kw), ref)
        _same(bamh.ingest(regions, threads=3, **kw), ref)
    bamh.close()


def test_ingest_rejects_corrupt_and_truncated_bam(tmp_path):
    """Malformed input must come back as an error code through the C-ABI — never a crash, never a silent short read: a BGZF block that
    does not inflate, a truncated file, aux fields that run past the record."""
    import shutil
    src = os.path.join(GOLD, "ingest_small.bam")
    g = np.load(os.path.join(GOLD, "ingest_ref.npz"))
    regions = [(str(c), int(s), int(e)) for c, s, e in zip(g["regions_chr"], g["regions_start"], g["regions_end"])]
    raw = bytearray(open(src, "rb").read())
    rng = np.random.default_rng(9)
    n_err = 0
    for trial in range(12):
        bad = bytearray(raw)
        if trial < 6:
            for _ in range(40):                                   # flip bytes in the compressed payload (past the header block)
                p = int(rng.integers(len(bad) // 3, len(bad) - 40))
                bad[p] ^= 0x5a
        else:
            bad = bad[:int(rng.integers(len(bad) // 2, len(bad) - 30))]   # truncated inside a block
        p = tmp_path / ("bad%d.bam" % trial)
        p.write_bytes(bytes(bad))
        shutil.copy(src + ".bai", str(p) + ".bai")
        try:
            b = otter_amd.Bam(str(p))
            try:
                b.ingest(regions, threads=1 + trial % 3)
            finally:
                b.close()
        except otter_amd.OtterGpuError:
            n_err += 1
    assert n_err >= 6          # most corruptions are detected (a flipped byte may by chance leave a decodable stream); none crashed


@needs_ref
def test_python_bam_writer_is_read_alike_by_both_readers(tmp_path):
    """otter_amd/bamwrite.py (the fixture writer of bench.py's end-to-end leg, numpy + zlib, no reference code): the BAM / BAI it writes is
    read identically by the reference's own reader (htslib-lite + parse_anreads, oracle/_ref) and by the product's reader."""
    from otter_amd import bamwrite
    fx = bamwrite.make_tr_fixture(str(tmp_path), 40, depth=12, len_range=(300, 1500), seed=3)
    regions = fx["regions"] + [("chrS", 0, 50), ("chrS", 100, 100000)]
    bamh = otter_amd.Bam(fx["bam"])
    assert bamh.targets()[0][0] == "chrS"
    for kw in (dict(), dict(offset_l=1, offset_r=1, mapq=10)):
        ref = _ref_ingest(fx["bam"], regions, **kw)
        assert ref["regions"]["n_reads"][:40].min() >= 10
        _same(bamh.ingest(regions, **kw), ref)
        _same(bamh.ingest(regions, threads=4, **kw), ref)
    bamh.close()


def test_ingest_short_cg_tag_leaves_the_placeholder(tmp_path):
    """bam_tag2cigar (src/sam.c:243-285) only moves a CG:B,I array into place when it is at least as long as the placeholder CIGAR and shorter than
    2^29 entries; a record whose CG tag is shorter than its two-operation placeholder keeps `<l_seq>S<rlen>N` — no aligned base, nothing to take
    from it.  (The reference's own region walk reads out of range on such a record — it segfaults in the reference build here — so the pin is the
    rule itself: a record with a one-operation CG, and one without the tag, must change nothing; a reader that took any non-empty CG array would
    use the first as `300M`.)"""
    import struct
    from otter_amd import bamwrite
    rng = np.random.default_rng(9)
    def seq(n):
        return bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, n))
    def cg(ops):
        return b"CGBI" + struct.pack("<I", len(ops)) + b"".join(struct.pack("<I", (l << 4) | o) for l, o in ops)
    plain = (0, 1005, "plain", 0, 60, "280M", seq(280), b"")
    real = (0, 1000, "real", 0, 60, "300S260N", seq(300), cg([(20, 4), (260, 0), (20, 4)]))
    short_cg = (0, 1000, "short_cg", 0, 60, "300S260N", seq(300), cg([(300, 0)]))
    no_cg = (0, 1000, "no_cg", 0, 60, "300S260N", seq(300), b"")
    regions = [("chr1", 1050, 1150), ("chr1", 990, 1010), ("chr1", 1200, 1262)]
    def run(name, recs):
        bam = str(tmp_path / (name + ".bam"))
        assert bamwrite.write_bam(bam, [("chr1", 100000)], recs) == len(recs)
        bamh = otter_amd.Bam(bam)
        out = bamh.ingest(regions), bamh.ingest(regions, offset_l=1, offset_r=1, threads=2)
        bamh.close()
        return out
    base = run("base", [real, plain])
    assert base[0]["regions"]["n_reads"].tolist() == [2, 2, 2]            # the real CIGAR of the tag is used
    assert run("plain", [plain])[0]["regions"]["n_reads"].tolist() == [1, 1, 1]
    with_junk = run("junk", [real, short_cg, no_cg, plain])
    for a, b in zip(base, with_junk):
        _same(a, b)
