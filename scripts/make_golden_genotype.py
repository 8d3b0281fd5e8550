#!/usr/bin/env python3
"""G8 golden fixture for the `otter genotype` ingest rows (tests/golden/genotype_small.bam/.bai/.fa/.fai, genotype_ref.json): a small
allele BAM written by the reference's htslib-lite and what the REFERENCE's own SampleIndex / parse_analleles / FaidxInstance
(oracle/_ref/libotter_ref_io.so) return for it.  Run in the build container (needs /root/reference for `make -C oracle`)."""
import json
import os
import shutil
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import test_genotype_io as TG  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
assert O.ref_io() is not None, "oracle/_ref/libotter_ref_io.so not built"
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(20241011)
sam, bam, fa = os.path.join(tmp, "g.sam"), os.path.join(tmp, "g.bam"), os.path.join(tmp, "g.fa")
regions, ref = TG.make_allele_sam(sam, rng, n_regions=14, ref_len=12000)
TG.write_fasta(fa, "chrG", ref)
assert O.ref_io().ref_sam_to_bam(sam.encode(), bam.encode()) > 0


def dump(blk):
    n = int(blk["alleles"]["seq_len"].astype(np.int64).sum())
    return {"alleles": [{k: (float(a[k]) if k == "se" else int(a[k])) for k in ("seq_off", "seq_len", "scov", "acov", "tcov", "se", "ic", "ps", "hp", "region", "label")}
                        for a in blk["alleles"]],
            "first_allele": [int(x) for x in blk["first_allele"]], "arena": blk["arena"][:n].tobytes().decode("latin-1")}


samples, ol, orr = TG._ref_sample_index(bam)
gold = {"regions": [list(r) for r in regions], "samples": samples, "offset_l": ol, "offset_r": orr,
        "without_reference": dump(TG._ref_ingest_alleles(bam, None, regions)),
        "with_reference": dump(TG._ref_ingest_alleles(bam, fa, regions))}       # the reference's fai_load writes g.fa.fai here
for src, dst in ((bam, "genotype_small.bam"), (bam + ".bai", "genotype_small.bam.bai"), (fa, "genotype_small.fa"), (fa + ".fai", "genotype_small.fa.fai")):
    shutil.copy(src, os.path.join(out, dst))
json.dump(gold, open(os.path.join(out, "genotype_ref.json"), "w"))
print("genotype_small.bam", os.path.getsize(os.path.join(out, "genotype_small.bam")), "bytes;", len(gold["with_reference"]["alleles"]), "alleles with reference,",
      len(gold["without_reference"]["alleles"]), "without")
