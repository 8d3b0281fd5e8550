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
