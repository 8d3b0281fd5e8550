"""BED file parsing, FASTA flank fetch and the --reads-only records (the remaining assemble-side IO): the
product's host code (otg_parse_bed_file, otg_fasta_*, otg_ingest_regions_named + otg_emit_reads) against the REFERENCE's own
code built from its sources (parse_bed_file, FaidxInstance::fetch over its faidx.c, parse_anreads + ANREAD::stdout_*;
oracle/_ref/libotter_ref_io.so) and against committed golden outputs of that build (tests/golden/bedfa_ref.json,
scripts/make_golden.py).  Host code: runs without a GPU."""
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

# every line shape parse_bed / parse_sc_bed distinguishes (src/anbed.cpp:23-63)
BED_TEXT = (
    "# comment\n"
    "chr1\t100\t200\n"
    "chr1\t300\t400\tname\t0\t+\n"
    "chr2_random:alt\t5\t6\n"
    "chr1:1000-2000\n"
    "chrX:7-9:extra:fields\n"
    "chr1:10-20-30\n"
    "\n"
    "chr1\t50\n"
    "chr1:\n"
    "chr1:5\n"
    ":5-6\n"
    "chr3\t 12\t+13\n"
    "chr4\t4294967295\t4294967296\n"
    "chr5:4294967295-10\n"
    "chr6\t-5\t20\n"
    "chr7\t10abc\t20xyz\ttrailing\t\n"
    "track name=foo\n"
    "chr8\t1\t2\r\n"
    "chr9:3-4\r\n"
    "\t\n"
    "chr10\t7\t8")          # no final newline


def _ref_parse_bed(path):
    R = oracle_lib.ref_io()
    R.ref_parse_bed_file.restype = C.c_int64
    buf = C.create_string_buffer(1 << 20)
    n = R.ref_parse_bed_file(path.encode(), buf, C.c_uint64(1 << 20))
    if n < 0:
        return None
    return [(c, int(s), int(e)) for c, s, e in (ln.split("\t") for ln in buf.raw[:n].decode("latin-1").split("\n")[:-1])]


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_bytes(text.encode("latin-1"))
    return str(p)


def test_bed_file_against_committed_golden(tmp_path):
    g = json.load(open(os.path.join(GOLD, "bedfa_ref.json")))
    beds, carena, skipped = otter_amd.parse_bed_file(_write(tmp_path, "a.bed", g["bed_text"]))
    assert otter_amd.bed_tuples(beds, carena) == [tuple(x) for x in g["bed_regions"]]
    assert g["bed_text"] == BED_TEXT


@needs_ref
def test_bed_file_matches_reference(tmp_path):
    p = _write(tmp_path, "a.bed", BED_TEXT)
    beds, carena, skipped = otter_amd.parse_bed_file(p)
    assert otter_amd.bed_tuples(beds, carena) == _ref_parse_bed(p)
    assert skipped > 0
    # random files out of the same token soup
    rng = np.random.default_rng(5)
    toks = ["chr1", "chrUn_x", "12", "0", "99999", "4000000000", "", " 7", "c:1-2", "c:1", "c", ":", "-", "1-2", "#x", "9\r"]
    for k in range(60):
        lines = []
        for _ in range(int(rng.integers(0, 30))):
            n = int(rng.integers(1, 5))
            lines.append(["\t", ":", "-"][int(rng.integers(0, 3)) if n > 1 and rng.random() < 0.2 else 0].join(toks[int(rng.integers(0, len(toks)))] for _ in range(n)))
        p = _write(tmp_path, "r%d.bed" % k, "\n".join(lines) + ("\n" if k % 2 else ""))
        ref = _ref_parse_bed(p)
        if ref is None:                     # the reference terminates on a coordinate that is not a number
            with pytest.raises(otter_amd.OtterGpuError):
                otter_amd.parse_bed_file(p)
        else:
            beds, carena, _ = otter_amd.parse_bed_file(p)
            assert otter_amd.bed_tuples(beds, carena) == ref, lines


def test_bed_file_errors(tmp_path):
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.parse_bed_file(str(tmp_path / "missing.bed"))
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.parse_bed_file(_write(tmp_path, "bad.bed", "chr1\tx\t5\n"))
    beds, carena, skipped = otter_amd.parse_bed_file(_write(tmp_path, "empty.bed", ""))
    assert len(beds) == 0 and skipped == 0
    # more regions than the binding's first guess: the capacity protocol
    many = "".join("chr%d\t%d\t%d\n" % (i % 23, i, i + 10) for i in range(5000))
    beds, carena, _ = otter_amd.parse_bed_file(_write(tmp_path, "many.bed", many))
    assert len(beds) == 5000 and otter_amd.bed_tuples(beds, carena)[4999] == ("chr%d" % (4999 % 23), 4999, 5009)


def _write_fasta(path, rng, width=60, lower=True, with_index=False):
    recs = [("chrA", 1000), ("chrB desc text", 61), ("chrC", 60), ("chrD", 1), ("chrE", 12345)]
    seqs = {}
    with open(path, "w") as f:
        for name, n in recs:
            s = "".join("ACGTNacgtn"[int(x)] for x in rng.integers(0, 10 if lower else 5, n))
            seqs[name.split()[0]] = s
            f.write(">%s\n" % name)
            for j in range(0, n, width):
                f.write(s[j:j + width] + "\n")
    return seqs


FETCHES = [("chrA", 0, 100), ("chrA", 59, 60), ("chrA", 60, 60), ("chrA", 899, 1200), ("chrA", -50, 50), ("chrA", 5000, 6000),
           ("chrA", 100, 50), ("chrA", -10, -5), ("chrB", 0, 60), ("chrB", 60, 61), ("chrC", 0, 59), ("chrC", 59, 200), ("chrD", 0, 0),
           ("chrD", 0, 100), ("chrE", 12000, 12345), ("chrE", 119, 241), ("chrE", 6000, 6100)]


@needs_ref
@pytest.mark.parametrize("width,prebuilt", [(60, False), (60, True), (7, False), (1000, False)])
def test_fasta_fetch_matches_reference(tmp_path, width, prebuilt):
    rng = np.random.default_rng(11 + width)
    fa = str(tmp_path / "ref.fa")
    seqs = _write_fasta(fa, rng, width)
    R = oracle_lib.ref_io()
    R.ref_ingest_open.restype = C.c_void_p
    bam = os.path.join(GOLD, "ingest_small.bam")
    if prebuilt:            # index written by the reference's fai_build first, read back by the product
        h = C.c_void_p(R.ref_ingest_open(bam.encode(), fa.encode()))
        fah = otter_amd.Fasta(fa)
    else:                   # index built (and written) by the product, read back by the reference's fai_load
        fah = otter_amd.Fasta(fa)
        assert os.path.exists(fa + ".fai")
        h = C.c_void_p(R.ref_ingest_open(bam.encode(), fa.encode()))
    assert fah.seqs() == [(k, len(v)) for k, v in seqs.items()]
    buf = C.create_string_buffer(1 << 16)
    for c, b, e in FETCHES + [("chrE", int(x), int(x) + 100) for x in rng.integers(-200, 12500, 40)]:
        n = R.ref_fetch(h, c.encode(), C.c_int(b), C.c_int(e), buf, C.c_int(1 << 16))
        assert fah.fetch(c, b, e) == buf.raw[:n], (c, b, e)
    assert fah.fetch("nochr", 0, 10) == b""
    R.ref_ingest_close(h)
    fah.close()


def test_fasta_index_text_and_fetch_against_plain_slicing(tmp_path):
    """Runs without the reference build: the .fai the product writes has the five faidx columns, and fetches equal plain slices."""
    rng = np.random.default_rng(3)
    fa = str(tmp_path / "ref.fa")
    seqs = _write_fasta(fa, rng, 50)
    fah = otter_amd.Fasta(fa)
    rows = [ln.split("\t") for ln in open(fa + ".fai").read().splitlines()]
    assert [r[0] for r in rows] == list(seqs) and all(len(r) == 5 for r in rows)
    assert [int(r[1]) for r in rows] == [len(s) for s in seqs.values()]
    assert rows[0][3:] == ["50", "51"]
    for c, b, e in FETCHES:
        s = seqs[c]
        if e < b:
            b = e
        b = min(max(b, 0), len(s) - 1); e = min(max(e, 0), len(s) - 1)
        assert fah.fetch(c, b, e) == s[b:e + 1].upper().encode(), (c, b, e)
    g = json.load(open(os.path.join(GOLD, "bedfa_ref.json")))
    fa2 = _write(tmp_path, "g.fa", g["fasta_text"])
    f2 = otter_amd.Fasta(fa2)
    for (c, b, e), exp in zip(g["fetches"], g["fetched"]):
        assert f2.fetch(c, b, e) == exp.encode()
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.Fasta(str(tmp_path / "missing.fa"))
    gz = tmp_path / "x.fa.gz"
    gz.write_bytes(b"\x1f\x8b\x08\x00rest")
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.Fasta(str(gz))
    ragged = _write(tmp_path, "ragged.fa", ">a\nACGT\nAC\nACGT\n")
    with pytest.raises(otter_amd.OtterGpuError):
        otter_amd.Fasta(ragged)


@needs_ref
def test_region_flanks_match_reference_helper(tmp_path):
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=6, seed=66)
    ref = e2e_bam.ingest_with_reference(ds, str(tmp_path), offset_l=1, offset_r=1, mapq=10, flank=100)
    regions = ds["regions"] + [("chrT", 20, 90), ("chrT", ds["ref_len"] - 30, ds["ref_len"] - 5), ("nochr", 5, 9)]
    beds, carena = abi.make_beds(regions)
    bam = otter_amd.Bam(os.path.join(str(tmp_path), "reads.bam"))
    got = bam.ingest((beds, carena), offset_l=1, offset_r=1, mapq=10)
    fa = otter_amd.Fasta(ds["fasta"])
    fa.region_flanks(beds, carena, got, flank=100, offset_l=1, offset_r=1)
    for r in range(len(ds["regions"])):
        for side in ("l", "r"):
            a = got["arena"][int(got["regions"][r]["flank_%s_off" % side]):][:int(got["regions"][r]["flank_%s_len" % side])].tobytes()
            b = ref["arena"][int(ref["regions"][r]["flank_%s_off" % side]):][:int(ref["regions"][r]["flank_%s_len" % side])].tobytes()
            assert a == b and len(a) == 101, (r, side)
    # near the contig ends the flank is clamped; an unknown contig has none
    k = len(ds["regions"])
    assert int(got["regions"][k]["flank_l_len"]) == 20 and int(got["regions"][k + 1]["flank_r_len"]) == 4      # [len - 4, len - 1]
    assert int(got["regions"][k + 2]["flank_l_len"]) == 0 and int(got["regions"][k + 2]["flank_r_len"]) == 0


def _ref_reads_only(bam, regions, read_group, fasta, max_cov=200, **kw):
    R = oracle_lib.ref_io()
    R.ref_ingest_open.restype = C.c_void_p
    R.ref_reads_only.restype = C.c_uint64
    h = C.c_void_p(R.ref_ingest_open(bam.encode(), b""))
    buf = C.create_string_buffer(64 << 20)
    out = []
    for c, s, e in regions:
        n = R.ref_reads_only(h, c.encode(), C.c_int(s), C.c_int(e), C.c_int(kw.get("offset_l", 0)), C.c_int(kw.get("offset_r", 0)),
                             C.c_int(kw.get("mapq", 0)), C.c_int(int(kw.get("nonprimary", False))), C.c_double(kw.get("read_quality", 0.0)),
                             C.c_int(int(kw.get("omit_nonspanning", False))), C.c_int(max_cov), read_group.encode(), C.c_int(int(fasta)),
                             buf, C.c_uint64(64 << 20))
        out.append(buf.raw[:n])
    R.ref_ingest_close(h)
    return b"".join(out)


def _golden_regions():
    g = np.load(os.path.join(GOLD, "ingest_ref.npz"))
    return [(str(c), int(s), int(e)) for c, s, e in zip(g["regions_chr"], g["regions_start"], g["regions_end"])]


GOLD_READS_REGIONS = 30


def test_reads_only_against_committed_golden():
    g = json.load(open(os.path.join(GOLD, "bedfa_ref.json")))
    regions = _golden_regions()[:GOLD_READS_REGIONS]
    beds, carena = abi.make_beds(regions)
    bam = otter_amd.Bam(os.path.join(GOLD, "ingest_small.bam"))
    for case in g["reads_only"]:
        kw = case["opts"]
        batch = bam.ingest((beds, carena), names=True, threads=case["threads"], **kw)
        txt = otter_amd.emit_reads(beds, carena, batch, read_group=case["read_group"], fasta=case["fasta"], max_cov=case["max_cov"])
        assert txt.decode("latin-1") == case["text"], case["opts"]


@needs_ref
def test_reads_only_matches_reference(tmp_path):
    regions = _golden_regions()
    beds, carena = abi.make_beds(regions)
    path = os.path.join(GOLD, "ingest_small.bam")
    bam = otter_amd.Bam(path)
    for kw in (dict(), dict(offset_l=50000, offset_r=7, nonprimary=True), dict(read_quality=0.4, nonprimary=True)):
        for fasta in (False, True):
            for rg, max_cov in (("", 200), ("sampleA", 3)):
                batch = bam.ingest((beds, carena), names=True, threads=3, **kw)
                got = otter_amd.emit_reads(beds, carena, batch, read_group=rg, fasta=fasta, max_cov=max_cov)
                assert got == _ref_reads_only(path, regions, rg, fasta, max_cov=max_cov, **kw), (kw, fasta, rg)
                assert len(got) > 0
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=5, seed=67)
    e2e_bam.ingest_with_reference(ds, str(tmp_path))          # writes reads.bam with the reference's htslib-lite
    p2 = os.path.join(str(tmp_path), "reads.bam")
    beds, carena = abi.make_beds(ds["regions"])
    batch = otter_amd.Bam(p2).ingest((beds, carena), names=True, offset_l=1, offset_r=1)
    assert otter_amd.emit_reads(beds, carena, batch, read_group="x") == _ref_reads_only(p2, ds["regions"], "x", False, offset_l=1, offset_r=1)
