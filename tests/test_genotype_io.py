"""`otter genotype` IO: sample index and allele ingest of an allele BAM (otg_bam_sample_index,
otg_ingest_alleles) against the REFERENCE's own SampleIndex / parse_analleles / FaidxInstance built from its sources
(oracle/_ref/libotter_ref_io.so) and against committed golden outputs of that build (tests/golden/genotype_small.bam,
genotype_ref.json; scripts/make_golden_genotype.py); VCF text (otg_emit_vcf_header / otg_emit_vcf_lines /
otg_emit_genotype_lengths) against the oracle's restatement of src/genotype.cpp:16-78,103-157 and literal expectations (the
reference's genotype.cpp itself cannot be built here: it includes the absent WFA2-lib header).  The GPU test runs the chain
allele BAM + BED + FASTA -> otg_genotype_cluster_batch -> VCF against the same chain through the oracle."""
import ctypes as C
import json
import os
import numpy as np
import pytest
import otter_amd
from otter_amd import abi
import oracle_lib

needs_ref = pytest.mark.skipif(oracle_lib.ref_io() is None, reason="oracle/_ref/libotter_ref_io.so not built")
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SAMPLES = ["HG001", "HG002", "s3"]


def make_allele_sam(path, rng, n_regions=24, ref_len=60000, offsets=(31, 7), chrom="chrG"):
    """SAM text as `otter assemble` writes it (one read group per sample, merged and sorted): tandem-repeat loci with 0-2 alleles
    per sample, nested / overlapping loci (records of one region inside another's query window), an allele without sequence ('*'
    -> "N"), haplotagged alleles, a record without the ta tag.  Returns (regions, reference bytes)."""
    ref = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), ref_len))
    regions, recs = [], []
    pos = 500
    for r in range(n_regions):
        L = int(rng.integers(20, 400))
        s = pos if r % 5 else max(1, pos - 150)          # every fifth locus reaches back into the previous one
        e = s + L
        regions.append((chrom, s, e))
        pos = e + int(rng.integers(50, 900))
        motif = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(2, 6))))
        for si, sm in enumerate(SAMPLES):
            na = [2, 2, 1, 0, 2][(r + si) % 5]
            for a in range(na):
                n = int(rng.integers(3, 60))
                seq = (motif * n)[:int(rng.integers(5, 300))]
                if r == 3 and si == 0 and a == 0:
                    seq = b""
                tags = "\tRG:Z:%s\tta:Z:%s:%d-%d\ttc:i:%d\tac:i:%d\tsc:i:%d\tic:i:%d\tse:f:%s" % (
                    sm, chrom, s, e, int(rng.integers(2, 90)), int(rng.integers(1, 40)), int(rng.integers(1, 30)), int(rng.integers(1, 3)),
                    ["0", "0.25", "1.5e-05", "12.5", "0.333333"][int(rng.integers(0, 5))])
                if (r + a) % 4 == 0:
                    tags += "\tPS:i:%d\tHP:i:%d" % (int(rng.integers(0, 9000)), 1 + a)
                recs.append((s, "%s:%d-%d_%d" % (chrom, s, e, a), "%dM" % len(seq) if seq else "*", seq.decode() if seq else "*", tags))
    recs.append((regions[2][1], "stray", "10M", "ACGTACGTAC", "\tRG:Z:s3"))          # no ta tag: belongs to no region
    recs.sort(key=lambda x: x[0])
    with open(path, "w") as f:
        f.write("@HD\tVN:1.4\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n@SQ\tSN:chrOther\tLN:1000\n" % (chrom, ref_len))
        for sm in SAMPLES:
            f.write("@RG\tID:%s\n" % sm)
        f.write("@PG\tID:otter\tOF:%d,%d\n@CO\tcontact user@example.org about RG things\n" % offsets)
        for p, name, cigar, seq, tags in recs:
            f.write("%s\t0\t%s\t%d\t0\t%s\t*\t0\t0\t%s\t%s%s\n" % (name, chrom, p, cigar, seq, "!" * len(seq) if seq != "*" else "*", tags))
    return regions + [(chrom, 40000, 40100), ("chrOther", 5, 50), ("nochr", 1, 2)], ref


def write_fasta(path, chrom, ref):
    with open(path, "w") as f:
        f.write(">%s\n" % chrom)
        for j in range(0, len(ref), 70):
            f.write(ref[j:j + 70].decode() + "\n")
        f.write(">chrOther\n" + "ACGT" * 250 + "\n")


def _ref_sample_index(bam):
    R = oracle_lib.ref_io()
    R.ref_sample_index.restype = C.c_int64
    buf = C.create_string_buffer(1 << 16)
    n = R.ref_sample_index(bam.encode(), buf, C.c_uint64(1 << 16))
    lines = buf.raw[:n].decode("latin-1").split("\n")[:-1]
    ol, orr = lines[0].split("\t")
    return lines[1:], int(ol), int(orr)


def _ref_ingest_alleles(bam, fasta, regions):
    R = oracle_lib.ref_io()
    R.ref_ingest_open.restype = C.c_void_p
    R.ref_ingest_alleles.restype = C.c_int64
    h = C.c_void_p(R.ref_ingest_open(bam.encode(), (fasta or "").encode()))
    alleles = np.zeros(1 << 16, dtype=abi.allele_dt)
    arena = np.zeros(32 << 20, dtype=np.uint8)
    used = C.c_uint64(0)
    first = np.zeros(len(regions) + 1, dtype=np.uint32)
    na = 0
    for r, (c, s, e) in enumerate(regions):
        sub = alleles[na:]
        k = R.ref_ingest_alleles(h, bam.encode(), c.encode(), C.c_uint32(s), C.c_uint32(e), C.c_uint32(r), abi.ptr(sub), C.c_uint64(len(sub)),
                                 abi.ptr(arena), C.c_uint64(arena.size), C.byref(used))
        assert k >= 0
        first[r] = na
        na += k
    first[len(regions)] = na
    R.ref_ingest_close(h)
    return {"alleles": alleles[:na].copy(), "first_allele": first, "arena": arena[:used.value + 64].copy()}


def _same_alleles(a, b):
    assert np.array_equal(a["first_allele"], b["first_allele"])
    assert len(a["alleles"]) == len(b["alleles"])
    for f in ("seq_off", "seq_len", "scov", "acov", "tcov", "se", "ic", "ps", "hp", "region", "label"):
        assert np.array_equal(a["alleles"][f], b["alleles"][f]), f
    n = int(b["alleles"]["seq_len"].astype(np.int64).sum())
    assert a["arena"][:n].tobytes() == b["arena"][:n].tobytes()


def _golden():
    g = json.load(open(os.path.join(GOLD, "genotype_ref.json")))
    regions = [tuple(x) for x in g["regions"]]
    return g, regions, os.path.join(GOLD, "genotype_small.bam"), os.path.join(GOLD, "genotype_small.fa")


def _blk_from_json(d):
    al = np.zeros(len(d["alleles"]), dtype=abi.allele_dt)
    for i, row in enumerate(d["alleles"]):
        for k, v in row.items():
            al[i][k] = v
    return {"alleles": al, "first_allele": np.array(d["first_allele"], dtype=np.uint32), "arena": np.frombuffer(d["arena"].encode("latin-1") + b"\0" * 64, dtype=np.uint8).copy()}


def test_allele_ingest_against_committed_golden():
    g, regions, bam_path, fa_path = _golden()
    bam = otter_amd.Bam(bam_path)
    assert bam.sample_index() == (g["samples"], g["offset_l"], g["offset_r"])
    _same_alleles(bam.ingest_alleles(regions), _blk_from_json(g["without_reference"]))
    fa = otter_amd.Fasta(fa_path)
    for threads in (1, 4):
        _same_alleles(bam.ingest_alleles(regions, reference=fa, threads=threads), _blk_from_json(g["with_reference"]))


@needs_ref
def test_allele_ingest_matches_reference(tmp_path):
    rng = np.random.default_rng(91)
    sam, bam_path, fa_path = str(tmp_path / "a.sam"), str(tmp_path / "a.bam"), str(tmp_path / "ref.fa")
    regions, ref = make_allele_sam(sam, rng, n_regions=60, ref_len=120000, offsets=(5, 9))
    write_fasta(fa_path, "chrG", ref)
    assert oracle_lib.ref_io().ref_sam_to_bam(sam.encode(), bam_path.encode()) > 0
    bam = otter_amd.Bam(bam_path)
    assert bam.sample_index() == _ref_sample_index(bam_path) == (SAMPLES, 5, 9)
    _same_alleles(bam.ingest_alleles(regions), _ref_ingest_alleles(bam_path, None, regions))
    fa = otter_amd.Fasta(fa_path)
    ref_blk = _ref_ingest_alleles(bam_path, fa_path, regions)
    _same_alleles(bam.ingest_alleles(regions, reference=fa), ref_blk)
    _same_alleles(bam.ingest_alleles(regions, reference=fa, threads=5), ref_blk)
    assert int((ref_blk["alleles"]["label"] == len(SAMPLES)).sum()) > 40          # reference alleles were appended


def test_sample_index_header_shapes(tmp_path):
    """Header lines SampleIndex reads or skips: single-number and malformed offsets, no read group at all, an unknown read group on a record."""
    if oracle_lib.ref_io() is None:
        pytest.skip("needs the reference's htslib-lite to write BAM files")
    R = oracle_lib.ref_io()

    def bam_of(name, header, body="a\t0\tc1\t5\t0\t4M\t*\t0\t0\tACGT\t!!!!\tRG:Z:zz\tta:Z:c1:5-9\n"):
        sam, bam = str(tmp_path / (name + ".sam")), str(tmp_path / (name + ".bam"))
        open(sam, "w").write("@HD\tVN:1.4\tSO:coordinate\n@SQ\tSN:c1\tLN:1000\n" + header + body)
        assert R.ref_sam_to_bam(sam.encode(), bam.encode()) >= 0
        return bam
    b = otter_amd.Bam(bam_of("one", "@RG\tID:x\n@PG\tID:otter\tOF:12\n"))
    assert b.sample_index() == (["x"], 12, 12) == _ref_sample_index(b._path)
    b = otter_amd.Bam(bam_of("more", "@RG\tID:x\tSM:y\n@RG\tID:z\n"))          # the reference keeps everything after "ID:" as the name
    assert b.sample_index() == (["x\tSM:y", "z"], 1, 0) == _ref_sample_index(b._path)
    b = otter_amd.Bam(bam_of("dflt", "@RG\tID:x\n@RG\tSM:noid\n@PG\tID:other\tOF:3,4\n"))
    assert b.sample_index() == (["x"], 1, 0) == _ref_sample_index(b._path)
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.Bam(bam_of("norg", "@PG\tID:otter\tOF:1,1\n")).sample_index()
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.Bam(bam_of("bad", "@RG\tID:x\n@PG\tID:otter\tOF:1,2,3\n")).sample_index()
    with pytest.raises(otter_amd.OtterGpuError):          # record with a read group the header does not list: the reference exits
        otter_amd.Bam(bam_of("unk", "@RG\tID:x\n")).ingest_alleles([("c1", 5, 9)])


def _tiny_case():
    """Two regions, two samples + reference: hand-made genotype numbers so that every branch of the line format shows."""
    regions = [("chr1", 100, 110), ("chr1", 4294967295, 5), ("chr2", 7, 9)]
    beds, carena = abi.make_beds(regions)
    seqs = [b"CAGCAG", b"CAGCAGCAG", b"N", b"CAGCAG", b"CAGCAGCA",     # region 0: s0 a0,a1; s1 a0 (deleted), a1; reference
            b"TT", b"TTT"]                                             # region 2: s1 one allele; reference
    al = np.zeros(len(seqs), dtype=abi.allele_dt)
    arena = bytearray()
    meta = [(0, 0, 20, 9, 8, 0.25, 77, 1), (0, 0, 20, 11, 10, 0.0, 77, 2), (0, 1, 31, 15, 3, 1.5e-5, -1, -1), (0, 1, 31, 16, 12, 12.5, -1, -1), (0, 2, 1, 1, 1, 0.0, -1, -1),
            (2, 1, 8, 8, 7, 0.5, -1, -1), (2, 2, 1, 1, 1, 0.0, -1, -1)]
    for i, (s_, m) in enumerate(zip(seqs, meta)):
        al[i]["seq_off"] = len(arena); al[i]["seq_len"] = len(s_); arena += s_
        al[i]["region"], al[i]["label"], al[i]["tcov"], al[i]["acov"], al[i]["scov"], al[i]["se"], al[i]["ps"], al[i]["hp"] = m
        al[i]["ic"] = 1
    blk = {"alleles": al, "first_allele": np.array([0, 5, 5, 7], dtype=np.uint32), "arena": np.frombuffer(bytes(arena) + b"\0" * 64, dtype=np.uint8).copy()}
    # anallele_cluster output: region 0 has three genotypes (0: CAGCAG x2 incl. a1 of s1; 1: CAGCAGCAG; 2: N); the reference allele has gt 1
    gt = np.array([0, 1, 2, 0, 1, 0, 1], dtype=np.int32)
    hsd = np.array([1.0, 2.5, 3.0, 1.0, 1.25, 1.0, 1.99184], dtype=np.float64)
    n_gt = np.array([3, 0, 2], dtype=np.int32)
    reps = np.array([0, 1, 2, 0, 0, 0, 1], dtype=np.int32)
    return regions, beds, carena, blk, gt, hsd, n_gt, reps


def test_vcf_lines_literal_and_oracle():
    regions, beds, carena, blk, gt, hsd, n_gt, reps = _tiny_case()
    got = otter_amd.emit_vcf_lines(beds, carena, blk, 2, gt, hsd, n_gt, reps, 31, 7)
    assert got == oracle_lib.emit_vcf_lines(beds, carena, blk, 2, gt, hsd, n_gt, reps, 31, 7)
    lines = got.decode().split("\n")
    # reference allele gt 1 becomes 0, former 0 becomes 1; ALT order: old gt 0's representative, then gt 2 ("N" -> <DEL>)
    assert lines[0] == ("chr1\t70\tchr1:100-110\tCAGCAGCA\tCAGCAG,<DEL>\t.\t.\tHSD=1.25,1,3\tGT:PS:HP:TC:AC:SC:SE"
                        "\t1/0:77:1:20:9,11:8,10:0.25,0\t2/1:-1:-1:31:15,16:3,12:1.5e-05,12.5")
    assert lines[1] == "chr2\t4294967273\tchr2:7-9\tTTT\tTT\t.\t.\tHSD=1.99184,1\tGT:PS:HP:TC:AC:SC:SE\t./.:.:.:.:.:.:.\t1/1:-1:-1:8:8,8:7,7:0.5,0.5"
    assert lines[2] == "" and len(lines) == 3          # the region without alleles prints nothing
    # a region whose alleles all carry the reference genotype: ALT is '.'
    gt1 = gt.copy(); gt1[5] = 0; gt1[6] = 0
    n1 = n_gt.copy(); n1[2] = 1
    one = otter_amd.emit_vcf_lines(beds, carena, blk, 2, gt1, hsd, n1, reps, 0, 0)
    assert one == oracle_lib.emit_vcf_lines(beds, carena, blk, 2, gt1, hsd, n1, reps, 0, 0)
    assert one.decode().split("\n")[1].startswith("chr2\t8\tchr2:7-9\tTTT\t.\t.\t.\tHSD=1.99184\tGT")


@needs_ref
def test_vcf_header_and_length_table(tmp_path):
    rng = np.random.default_rng(92)
    sam, bam_path = str(tmp_path / "a.sam"), str(tmp_path / "a.bam")
    regions, ref = make_allele_sam(sam, rng, n_regions=10)
    assert oracle_lib.ref_io().ref_sam_to_bam(sam.encode(), bam_path.encode()) > 0
    bam = otter_amd.Bam(bam_path)
    samples, ol, orr = bam.sample_index()
    hdr = otter_amd.emit_vcf_header(bam)
    assert hdr == oracle_lib.emit_vcf_header(bam.targets(), samples)
    assert hdr.startswith(b"##fileformat=VCFv4.2\n##contig=<ID=chrG,length=60000>\n##contig=<ID=chrOther,length=1000>\n##INFO=<ID=HSD,")
    assert hdr.endswith(b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tHG001\tHG002\ts3\n") and hdr.count(b"\n") == 13
    beds, carena = abi.make_beds(regions)
    blk = bam.ingest_alleles((beds, carena))
    txt = otter_amd.emit_genotype_lengths(bam, beds, carena, blk, len(samples)).decode("latin-1")
    exp = []
    for r, (c, s, e) in enumerate(regions):
        a0, a1 = int(blk["first_allele"][r]), int(blk["first_allele"][r + 1])
        for si, sm in enumerate(samples):
            ls = [int(blk["alleles"][i]["seq_len"]) for i in range(a0, a1) if int(blk["alleles"][i]["label"]) == si]
            if ls:
                exp.append("%s:%d-%d\t%s\t%d\t%d\n" % (c, s, e, sm, min(ls[0], ls[-1]), max(ls[0], ls[-1])))
    assert txt == "".join(exp) and len(exp) > 10


@needs_ref
@pytest.mark.gpu
def test_genotype_chain_bam_to_vcf_gpu_vs_oracle(gpu, oracle, tmp_path):
    """`otter genotype -b regions.bed -r ref.fa alleles.bam` through the C-ABI: allele ingest -> GPU anallele_cluster -> VCF, against the
    reference's own ingest -> oracle anallele_cluster -> oracle VCF text."""
    rng = np.random.default_rng(93)
    sam, bam_path, fa_path = str(tmp_path / "a.sam"), str(tmp_path / "a.bam"), str(tmp_path / "ref.fa")
    regions, ref = make_allele_sam(sam, rng, n_regions=40, ref_len=90000, offsets=(3, 2))
    write_fasta(fa_path, "chrG", ref)
    assert oracle_lib.ref_io().ref_sam_to_bam(sam.encode(), bam_path.encode()) > 0
    bam = otter_amd.Bam(bam_path)
    samples, ol, orr = bam.sample_index()
    beds, carena = abi.make_beds(regions)
    blk = bam.ingest_alleles((beds, carena), reference=otter_amd.Fasta(fa_path), threads=2)
    ref_blk = _ref_ingest_alleles(bam_path, fa_path, regions)
    P = abi.default_params()
    so, sl, fa_, na_ = otter_amd.genotype_blocks(blk)
    gt, gl, gk, hsd, ngt, reps = gpu.genotype_cluster_batch(P, blk["arena"], so, sl, fa_, na_)
    so2, sl2, fa2, na2 = otter_amd.genotype_blocks(ref_blk)
    ogt, ogl, ogk, ohsd, ongt, oreps = oracle.genotype_cluster_batch(P, ref_blk["arena"], so2, sl2, fa2, na2)
    assert np.array_equal(gt, ogt) and np.array_equal(ngt, ongt)
    got = otter_amd.emit_vcf_header(bam) + otter_amd.emit_vcf_lines(beds, carena, blk, len(samples), gt, hsd, ngt, reps, ol, orr)
    exp = oracle.emit_vcf_header(bam.targets(), samples) + oracle.emit_vcf_lines(beds, carena, ref_blk, len(samples), ogt, ohsd, ongt, oreps, ol, orr)
    assert got == exp
    assert got.count(b"\n") == 13 + int((np.diff(blk["first_allele"].astype(np.int64)) > 0).sum())
    # the one-call dispatcher (genotype() / genotype_process(), src/genotype.cpp:69-192): same text whatever the batch size; and the
    # two-length table the reference prints without -r (src/genotype.cpp:112-121)
    bed_path = str(tmp_path / "r.bed")
    with open(bed_path, "w") as f:
        for c, s_, e_ in regions:
            f.write("%s\t%d\t%d\n" % (c, s_, e_))
    for batch in (0, 7):
        text, st = otter_amd.genotype_files(bam_path, bed_path, fasta=fa_path, threads=3, batch_regions=batch)
        assert text == got, batch
        assert st["n_regions"] == len(regions) and st["n_alleles"] == len(blk["alleles"])
    blk0 = bam.ingest_alleles((beds, carena), reference=None, threads=2)
    table, _ = otter_amd.genotype_files(bam_path, bed_path, fasta=None, threads=2, batch_regions=9)
    assert table == otter_amd.emit_genotype_lengths(bam, beds, carena, blk0, len(samples)) and table.count(b"\n") > 10


@needs_ref
def test_bench_genotype_fixture_reads_alike(tmp_path):
    """the allele BAM bench.py's file-to-VCF leg runs on (otter_amd.bamwrite.make_genotype_fixture): the reference's own SampleIndex / parse_analleles and
    the product's ingest return the same samples, offsets and alleles for it"""
    from otter_amd import bamwrite
    fx = bamwrite.make_genotype_fixture(str(tmp_path), 12, n_samples=5, len_range=(150, 700))
    bam = otter_amd.Bam(fx["bam"])
    samples, ol, orr = bam.sample_index()
    rs, rol, ror = _ref_sample_index(fx["bam"])
    assert (samples, ol, orr) == (rs, rol, ror) and len(samples) == 5 and (ol, orr) == (1, 0)
    beds, carena = abi.make_beds(fx["regions"])
    for fasta in (None, fx["fasta"]):
        blk = bam.ingest_alleles((beds, carena), reference=otter_amd.Fasta(fasta) if fasta else None, threads=2)
        ref_blk = _ref_ingest_alleles(fx["bam"], fasta, fx["regions"])
        assert np.array_equal(blk["first_allele"], ref_blk["first_allele"])
        assert np.array_equal(np.diff(blk["first_allele"].astype(np.int64)), np.full(12, 11 if fasta else 10))
        for f in ("seq_len", "scov", "acov", "tcov", "ic", "ps", "hp", "region", "label"):
            assert np.array_equal(blk["alleles"][f], ref_blk["alleles"][f]), f
        assert np.array_equal(blk["alleles"]["se"], ref_blk["alleles"]["se"])
        n = int(blk["alleles"]["seq_len"].astype(np.int64).sum())
        assert blk["arena"][:n].tobytes() == ref_blk["arena"][:n].tobytes()


@pytest.mark.gpu
def test_genotype_files_on_the_bench_fixture(gpu, oracle, tmp_path):
    """otg_genotype_files on the bench fixture's shape (small): the VCF text of the one-call path == product ingest -> ORACLE anallele_cluster -> oracle text"""
    from otter_amd import bamwrite
    fx = bamwrite.make_genotype_fixture(str(tmp_path), 30, n_samples=6, len_range=(200, 900))
    bam = otter_amd.Bam(fx["bam"])
    samples, ol, orr = bam.sample_index()
    beds, carena = abi.make_beds(fx["regions"])
    blk = bam.ingest_alleles((beds, carena), reference=otter_amd.Fasta(fx["fasta"]), threads=2)
    so, sl, fa_, na_ = otter_amd.genotype_blocks(blk)
    assert (na_ == 13).all()
    P = abi.default_params()
    ogt, ogl, ogk, ohsd, ongt, oreps = oracle.genotype_cluster_batch(P, blk["arena"], so, sl, fa_, na_)
    exp = oracle.emit_vcf_header(bam.targets(), samples) + oracle.emit_vcf_lines(beds, carena, blk, len(samples), ogt, ohsd, ongt, oreps, ol, orr)
    text, st = otter_amd.genotype_files(fx["bam"], fx["bed"], fasta=fx["fasta"], threads=3)
    assert text == exp and st["n_regions"] == 30 and st["n_alleles"] == 30 * 13
    assert (ongt > 1).sum() > 10                  # the loci are polymorphic: the clustering has something to do
