#!/usr/bin/env python3
"""G7 golden fixture for the BED / FASTA / --reads-only rows (tests/golden/bedfa_ref.json): outputs of the REFERENCE's own code
(oracle/_ref/libotter_ref_io.so: parse_bed_file, FaidxInstance::fetch over its faidx.c, parse_anreads + ANREAD::stdout_*) on small
committed inputs.  Run in the build container (needs /root/reference for `make -C oracle`); the fixture then travels."""
import ctypes as C
import json
import os
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import test_bedfa as TB  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
R = O.ref_io()
assert R is not None, "oracle/_ref/libotter_ref_io.so not built"
tmp = tempfile.mkdtemp()
gold = {"bed_text": TB.BED_TEXT}
bed = os.path.join(tmp, "g.bed")
open(bed, "wb").write(TB.BED_TEXT.encode("latin-1"))
gold["bed_regions"] = TB._ref_parse_bed(bed)

rng = np.random.default_rng(20241010)
fa = os.path.join(tmp, "g.fa")
TB._write_fasta(fa, rng, 60)
gold["fasta_text"] = open(fa).read()
R.ref_ingest_open.restype = C.c_void_p
h = C.c_void_p(R.ref_ingest_open(os.path.join(out, "ingest_small.bam").encode(), fa.encode()))
buf = C.create_string_buffer(1 << 16)
gold["fetches"] = [list(x) for x in TB.FETCHES]
gold["fetched"] = []
for c, b, e in TB.FETCHES:
    n = R.ref_fetch(h, c.encode(), C.c_int(b), C.c_int(e), buf, C.c_int(1 << 16))
    gold["fetched"].append(buf.raw[:n].decode())
R.ref_ingest_close(h)

regions = TB._golden_regions()[:TB.GOLD_READS_REGIONS]
gold["reads_only"] = []
for kw, rg, fasta, max_cov, threads in ((dict(), "", False, 200, 1), (dict(offset_l=300, offset_r=7, nonprimary=True), "sampleA", False, 200, 3),
                                        (dict(read_quality=0.4, nonprimary=True), "", True, 200, 2), (dict(mapq=10), "rg", True, 2, 1)):
    txt = TB._ref_reads_only(os.path.join(out, "ingest_small.bam"), regions, rg, fasta, max_cov=max_cov, **kw)
    gold["reads_only"].append({"opts": kw, "read_group": rg, "fasta": fasta, "max_cov": max_cov, "threads": threads, "text": txt.decode("latin-1")})
json.dump(gold, open(os.path.join(out, "bedfa_ref.json"), "w"))
print("bedfa_ref.json", os.path.getsize(os.path.join(out, "bedfa_ref.json")), "bytes;", len(gold["bed_regions"]), "regions;",
      [len(c["text"]) for c in gold["reads_only"]])
