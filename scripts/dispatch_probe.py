"""File-to-text rate of otg_assemble_files for several batch sizes (0 = the library's own batch plan) and numbers of hot-path contexts per device
(OTG_DISPATCH_CONTEXTS).
usage: [OTG_PROBE_BATCHES=0,2048 OTG_PROBE_CONTEXTS=2,3 OTG_PROBE_LEN=1000,10000] python scripts/dispatch_probe.py [loci] [ingest_threads]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import otter_amd  # noqa: E402
from otter_amd import bamwrite  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tmp = tempfile.mkdtemp(prefix="otg_dp_")
t0 = time.perf_counter()
LEN = tuple(int(x) for x in os.environ.get("OTG_PROBE_LEN", "1000,5000").split(","))       # locus length range of the fixture
fx = bamwrite.make_tr_fixture(tmp, R, depth=30, len_range=LEN, seed=7)
print("fixture: %d loci in %.0f s" % (R, time.perf_counter() - t0), flush=True)
for batch in (int(b) for b in os.environ.get("OTG_PROBE_BATCHES", "250,500,1000").split(",")):
    for nctx in (int(c) for c in os.environ.get("OTG_PROBE_CONTEXTS", "1,2,3,4").split(",")):
        os.environ["OTG_DISPATCH_CONTEXTS"] = str(nctx)
        best = None
        for rep in range(3):
            t1 = time.perf_counter()
            txt, st = otter_amd.assemble_files(fx["bam"], fx["bed"], read_group="s1", batch_regions=batch, offset_l=1, offset_r=1, mapq=10, threads=T)
            dt = time.perf_counter() - t1
            best = dt if best is None or dt < best else best
        print("batch %4d contexts %d: %.3f s = %.0f regions/s (busy ms: ingest %.0f hot %.0f)" % (batch, nctx, best, R / best, st["ms_ingest"], st["ms_hot_path"]), flush=True)
