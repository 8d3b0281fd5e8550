"""BAM -> SAM text, end to end: the reference's own ingest (parse_anreads, built from its sources) feeds the hot path, and
the record text comes out of otg_emit_alleles — the reference worker's role (src/assemble.cpp:53-149) with the five
compute calls replaced by the C-ABI.  Needs oracle/_ref/libotter_ref_io.so (prebuilt; it travels with the tree)."""
import numpy as np
import pytest
import otter_amd
from otter_amd import abi
import oracle_lib

needs_ref = pytest.mark.skipif(oracle_lib.ref_io() is None, reason="oracle/_ref/libotter_ref_io.so not built")


@needs_ref
def test_reference_ingest_semantics(tmp_path):
    """CPU: what the reference's ingest produces for the synthetic BAM — the invariants the hot path relies on
    (SURVEY Appendix D): flags and clip coordinates consistent with the extracted sub-sequence, filters honoured."""
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path))
    b = e2e_bam.ingest_with_reference(ds, str(tmp_path))
    reads, regions = b["reads"], b["regions"]
    assert regions["n_reads"].sum() == len(reads) and (regions["n_reads"] >= 6).all()
    assert (reads["seq_len"] > 0).all()
    assert ((reads["ccoord_first"] >= 0) & (reads["ccoord_second"] <= reads["seq_len"].astype(np.int64)) & (reads["ccoord_first"] <= reads["ccoord_second"])).all()
    both = (reads["spanning_l"] == 1) & (reads["spanning_r"] == 1)
    assert both.sum() > len(reads) // 2 and (~both).sum() > 0
    assert (reads["hp"] >= -1).all() and (reads["hp"] > 0).sum() > 0
    # MAPQ 3 records and secondary alignments are dropped at --mapq 10 without --non-primary
    all_in = e2e_bam.ingest_with_reference(ds, str(tmp_path), mapq=0, nonprimary=True)
    assert len(all_in["reads"]) > len(reads)


@needs_ref
@pytest.mark.gpu
def test_bam_to_sam_text_gpu_vs_oracle(gpu, oracle, tmp_path):
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=10, seed=62)
    batch = e2e_bam.ingest_with_reference(ds, str(tmp_path))
    P = abi.default_params()
    ora = oracle.assemble_batch(P, batch)
    res = gpu.assemble(P, batch)
    assert np.array_equal(res["labels"], ora["labels"])
    beds, carena = abi.make_beds(ds["regions"])
    hdr = otter_amd.emit_sam_header([(ds["chrom"], ds["ref_len"])], "s1", 1, 1)
    assert hdr == oracle.emit_sam_header([(ds["chrom"], ds["ref_len"])], "s1", 1, 1)
    text = otter_amd.emit_alleles(beds, carena, res, "s1", False)
    assert text == oracle.emit_alleles(beds, carena, ora, "s1", False)
    assert text.count(b"\n") == len(res["alleles"]) and len(res["alleles"]) >= len(ds["regions"])


@needs_ref
@pytest.mark.gpu
def test_bam_with_reference_flanks_realign(gpu, oracle, tmp_path):
    """-r given: local_realignment gets the reference flanks the reference's own FASTA helper fetches."""
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=6, seed=63)
    batch = e2e_bam.ingest_with_reference(ds, str(tmp_path), flank=100)
    P = abi.default_params(realign=1)
    ora = oracle.assemble_batch(P, batch)
    res = gpu.assemble(P, batch)
    beds, carena = abi.make_beds(ds["regions"])
    assert np.array_equal(res["labels"], ora["labels"])
    assert otter_amd.emit_alleles(beds, carena, res, "", True) == oracle.emit_alleles(beds, carena, ora, "", True)


@needs_ref
@pytest.mark.gpu
def test_bam_to_sam_text_all_native(gpu, oracle, tmp_path):
    """The whole chain without the reference at run time: otg_bam_open / otg_ingest_regions -> GPU hot path ->
    otg_emit_alleles, against reference ingest -> oracle -> oracle emit (the BAM itself is written by the reference's
    htslib-lite, which is why the prebuilt _ref library is still needed to make the fixture)."""
    import os
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=10, seed=65)
    ref_batch = e2e_bam.ingest_with_reference(ds, str(tmp_path), offset_l=1, offset_r=1, mapq=10)
    bam = otter_amd.Bam(os.path.join(str(tmp_path), "reads.bam"))
    batch = bam.ingest(ds["regions"], offset_l=1, offset_r=1, mapq=10)
    targets = bam.targets()
    bam.close()
    P = abi.default_params()
    res = gpu.assemble(P, batch)
    ora = oracle.assemble_batch(P, ref_batch)
    beds, carena = abi.make_beds(ds["regions"])
    assert otter_amd.emit_sam_header(targets, "s1", 1, 1) + otter_amd.emit_alleles(beds, carena, res, "s1", False) == \
        oracle.emit_sam_header([(ds["chrom"], ds["ref_len"])], "s1", 1, 1) + oracle.emit_alleles(beds, carena, ora, "s1", False)


@needs_ref
@pytest.mark.gpu
def test_bed_bam_fasta_to_sam_text_all_native_with_realign(gpu, oracle, tmp_path):
    """`otter assemble -b regions.bed -r ref.fa reads.bam` through the C-ABI alone: otg_parse_bed_file -> otg_ingest_regions ->
    otg_fasta_region_flanks -> GPU hot path with local_realignment -> otg_emit_alleles, against the reference's own ingest + FASTA
    helper -> oracle -> oracle emit."""
    import os
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=8, seed=68)
    ref_batch = e2e_bam.ingest_with_reference(ds, str(tmp_path), offset_l=1, offset_r=1, mapq=10, flank=100)
    bed_path = os.path.join(str(tmp_path), "regions.bed")
    with open(bed_path, "w") as f:
        f.write("# synthetic tandem-repeat loci\n")
        for i, (c, s, e) in enumerate(ds["regions"]):
            f.write("%s:%d-%d\n" % (c, s, e) if i % 3 == 0 else "%s\t%d\t%d\tlocus%d\n" % (c, s, e, i))
    beds, carena, _ = otter_amd.parse_bed_file(bed_path)
    assert otter_amd.bed_tuples(beds, carena) == ds["regions"]
    bam = otter_amd.Bam(os.path.join(str(tmp_path), "reads.bam"))
    batch = bam.ingest((beds, carena), offset_l=1, offset_r=1, mapq=10, threads=2)
    otter_amd.Fasta(ds["fasta"]).region_flanks(beds, carena, batch, flank=100, offset_l=1, offset_r=1)
    P = abi.default_params(realign=1)
    res = gpu.assemble(P, batch)
    ora = oracle.assemble_batch(P, ref_batch)
    assert np.array_equal(res["labels"], ora["labels"])
    assert otter_amd.emit_alleles(beds, carena, res, "s1", False) == oracle.emit_alleles(beds, carena, ora, "s1", False)


def _write_bed(path, regions):
    with open(path, "w") as f:
        for c, s, e in regions:
            f.write("%s\t%d\t%d\n" % (c, s, e))


@needs_ref
@pytest.mark.gpu
def test_dispatcher_files_to_text(gpu, oracle, tmp_path):
    """The library's dispatcher (otg_assemble_files, the role of assemble() / assemble_process(), src/assemble.cpp:39-179): BED + BAM
    [+ FASTA] -> SAM / FASTA text in one call, bounded batches flowing through ingest / hot path / emit threads.  The text must not
    depend on the batch size and must equal the chain the reference's own ingest -> oracle -> oracle emit produces."""
    import os
    import subprocess
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=14, seed=71)
    bam = os.path.join(str(tmp_path), "reads.bam")
    bed = os.path.join(str(tmp_path), "regions.bed")
    _write_bed(bed, ds["regions"])
    beds, carena = abi.make_beds(ds["regions"])
    hdr = oracle.emit_sam_header([(ds["chrom"], ds["ref_len"])], "s1", 1, 1)
    for fasta_ref, realign in ((None, 0), (ds["fasta"], 1)):
        ref_batch = e2e_bam.ingest_with_reference(ds, str(tmp_path), offset_l=1, offset_r=1, mapq=10, flank=100 if realign else 0)
        ora = oracle.assemble_batch(abi.default_params(realign=realign), ref_batch)
        for is_fa in (False, True):
            expect = (b"" if is_fa else hdr) + oracle.emit_alleles(beds, carena, ora, "s1", is_fa)
            for batch in (3, 5, 0):
                text, st = otter_amd.assemble_files(bam, bed, fasta=fasta_ref, read_group="s1", is_fasta=is_fa, batch_regions=batch, offset_l=1, offset_r=1,
                                                    mapq=10, threads=3)
                assert text == expect, (realign, is_fa, batch)
                assert st["n_regions"] == len(ds["regions"]) and st["n_alleles"] == len(ora["alleles"]) and st["output_bytes"] == len(text)
    # the command-line host over the same entry point
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "otter_assemble")
    if os.path.exists(tool):
        p = subprocess.run([tool, "-b", bed, "-R", "s1", "-o", "1,1", "-m", "10", "-t", "2", "--batch", "4", "-r", ds["fasta"], bam], capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-500:]
        assert p.stdout == expect_sam_with_ref(hdr, oracle, beds, carena, ora)


def expect_sam_with_ref(hdr, oracle, beds, carena, ora):
    return hdr + oracle.emit_alleles(beds, carena, ora, "s1", False)


@needs_ref
@pytest.mark.gpu
def test_dispatcher_reads_only(gpu, oracle, tmp_path):
    """--reads-only through the dispatcher (with and without -r): the read records of otg_ingest_regions_named [+ otg_assemble_realign]
    + otg_emit_reads, batch by batch, equal to the one-batch chain."""
    import os
    import e2e_bam
    ds = e2e_bam.make_dataset(str(tmp_path), n_regions=9, seed=72)
    bam = os.path.join(str(tmp_path), "reads.bam")
    assert oracle_lib.ref_io().ref_sam_to_bam(ds["sam"].encode(), bam.encode()) == ds["n_records"]
    bed = os.path.join(str(tmp_path), "regions.bed")
    _write_bed(bed, ds["regions"])
    for fasta_ref in (None, ds["fasta"]):
        whole, _ = otter_amd.assemble_files(bam, bed, fasta=fasta_ref, read_group="s1", reads_only=True, batch_regions=0, offset_l=1, offset_r=1, mapq=10, threads=2)
        parts, st = otter_amd.assemble_files(bam, bed, fasta=fasta_ref, read_group="s1", reads_only=True, batch_regions=2, offset_l=1, offset_r=1, mapq=10, threads=2)
        assert whole == parts and whole.count(b"\n") > 60
        # against the pieces called by hand
        b = otter_amd.Bam(bam)
        batch = b.ingest(ds["regions"], offset_l=1, offset_r=1, mapq=10, names=True)
        hdr = otter_amd.emit_sam_header(b.targets(), "s1", 1, 1)
        b.close()
        beds, carena = abi.make_beds(ds["regions"])
        if fasta_ref:
            otter_amd.Fasta(fasta_ref).region_flanks(beds, carena, batch, flank=100, offset_l=1, offset_r=1)
            batch = dict(batch, reads=gpu.realign_reads(abi.default_params(realign=1), batch))
        assert whole == hdr + otter_amd.emit_reads(beds, carena, batch, read_group="s1", fasta=False, max_cov=200)


@pytest.mark.gpu
def test_dispatcher_shards_on_one_device(gpu, tmp_path):
    """The dispatcher's multi-shard path (one contiguous BED shard per entry of the device list = the reference's static split over worker
    threads, src/BS_thread_pool.hpp:183-198) exercised on ONE card by naming device 0 two and three times: every shard has its own ingest
    thread, hot-path contexts and back-pressure bound, the writer drains the shards in order.  The text must be byte-identical to the
    single-shard run whatever the number of shards and the batch size — also when there are fewer regions than shards (block == 0)."""
    import os
    from otter_amd import bamwrite
    fx = bamwrite.make_tr_fixture(str(tmp_path), 23, depth=12, len_range=(300, 900), seed=5)
    kw = dict(read_group="s1", offset_l=1, offset_r=1, mapq=10, threads=4)
    gpu.trim()                              # the session context's aligner workspaces: room for the dispatcher's own contexts
    one, st1 = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=0, **kw)
    assert st1["n_regions"] == 23 and st1["n_regions_ok"] >= 20 and one.count(b"\n") > 23
    for devs in ([0, 0], [0, 0, 0]):
        for batch in (2, 5, 0):
            text, st = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=batch, devices=devs, **kw)
            assert text == one, (devs, batch)
            assert st["n_devices"] == len(devs) and st["n_alleles"] == st1["n_alleles"] and st["n_regions_ok"] == st1["n_regions_ok"]
    # fewer regions than shards: the first two regions alone, three shards
    two_bed = os.path.join(str(tmp_path), "two.bed")
    with open(fx["bed"]) as f:
        lines = f.readlines()
    with open(two_bed, "w") as f:
        f.writelines(lines[:2])
    a, _ = otter_amd.assemble_files(fx["bam"], two_bed, batch_regions=0, **kw)
    b3, st3 = otter_amd.assemble_files(fx["bam"], two_bed, batch_regions=1, devices=[0, 0, 0], **kw)
    assert a == b3 and st3["n_regions"] == 2
    fa1, _ = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=0, is_fasta=True, **kw)
    fa2, _ = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=3, is_fasta=True, devices=[0, 0], **kw)
    assert fa1 == fa2 and fa1.startswith(b">")
    otter_amd.assemble_files_release()


@pytest.mark.gpu
def test_dispatcher_two_devices(gpu, tmp_path):
    """otg_assemble_files over two real devices (n_devices = 2: the C++ host's multi-GPU form, no collective — the writer thread orders the
    shards as the reference's mutex section orders its threads' output, src/assemble.cpp:143-149): byte-identical to the one-device text.
    Needs two visible devices; the round's box has one."""
    import otter_amd
    if otter_amd.device_count() < 2:
        pytest.skip("one HIP device visible")
    from otter_amd import bamwrite
    fx = bamwrite.make_tr_fixture(str(tmp_path), 40, depth=14, len_range=(400, 1500), seed=6)
    kw = dict(read_group="s1", offset_l=1, offset_r=1, mapq=10, threads=4)
    one, _ = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=0, **kw)
    two, st = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=7, devices=[0, 1], **kw)
    assert one == two and st["n_devices"] == 2
    otter_amd.assemble_files_release()


@pytest.mark.gpu
def test_dispatcher_batch_plan_of_the_library(gpu, tmp_path):
    """With batch_regions = 0 the library cuts a shard by its own plan (small batches first, full ones, two equal halves at the end:
    include/otter_gpu.h, otg_assemble_batch_plan).  On a BED long enough for the plan to have several batches of different sizes — also per
    shard of a two-shard run — the text must be what one batch of the whole BED gives."""
    from otter_amd import bamwrite
    n = 1300
    fx = bamwrite.make_tr_fixture(str(tmp_path), n, depth=8, len_range=(300, 700), seed=8)
    plan = otter_amd.assemble_batch_plan(n, 0)
    assert len(plan) >= 3 and sum(plan) == n and len(set(plan)) >= 2
    kw = dict(read_group="s1", offset_l=1, offset_r=1, mapq=10, threads=8)
    gpu.trim()
    whole, st0 = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=n, **kw)
    auto, st1 = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=0, **kw)
    assert st0["n_regions"] == n and st0["n_regions_ok"] > 0.9 * n
    assert auto == whole and st1["n_alleles"] == st0["n_alleles"]
    two, st2 = otter_amd.assemble_files(fx["bam"], fx["bed"], batch_regions=0, devices=[0, 0], **kw)
    assert two == whole and st2["n_devices"] == 2
    otter_amd.assemble_files_release()


@pytest.mark.gpu
def test_dispatcher_adaptive_heuristic(gpu, oracle, tmp_path):
    """The aligner mode travels through the one-call path: otg_assemble_job.params.heuristic = wfadaptive(10,50,1) (and the command line's
    --wfa-heuristic) gives the text of library ingest -> ORACLE in adaptive mode -> oracle emit; the default job stays exact, and the two differ."""
    import os
    import subprocess
    from otter_amd import bamwrite
    fx = bamwrite.make_tr_fixture(str(tmp_path), 12, depth=16, len_range=(500, 1600), seed=9)
    b = otter_amd.Bam(fx["bam"])
    batch = b.ingest(fx["regions"], offset_l=1, offset_r=1, mapq=10)
    hdr = otter_amd.emit_sam_header(b.targets(), "s1", 1, 1)
    b.close()
    beds, carena = abi.make_beds(fx["regions"])
    texts = {}
    gpu.trim()
    for name, heur in (("exact", abi.OTG_HEURISTIC_NONE), ("adaptive", abi.OTG_HEURISTIC_WFADAPTIVE)):
        P = abi.default_params(heuristic=heur)
        expect = hdr + oracle.emit_alleles(beds, carena, oracle.assemble_batch(P, batch), "s1", False)
        text, st = otter_amd.assemble_files(fx["bam"], fx["bed"], read_group="s1", params=P, batch_regions=5, offset_l=1, offset_r=1, mapq=10, threads=2)
        assert text == expect, name
        texts[name] = text
    assert texts["exact"] != texts["adaptive"]          # ONT reads of 0.5-1.6 kb: the adaptive cut moves some consensus bases
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "otter_assemble")
    if os.path.exists(tool):
        otter_amd.assemble_files_release()
        for flag, want in ((["--wfa-heuristic", "wfadaptive"], "adaptive"), (["--wfa-heuristic", "wfadaptive:10,50,1"], "adaptive"), (["--wfa-heuristic", "none"], "exact"), ([], "exact")):
            p = subprocess.run([tool, "-b", fx["bed"], "-R", "s1", "-o", "1,1", "-m", "10", "-t", "2"] + flag + [fx["bam"]], capture_output=True, timeout=300)
            assert p.returncode == 0, p.stderr.decode()[-500:]
            assert p.stdout == texts[want], flag
