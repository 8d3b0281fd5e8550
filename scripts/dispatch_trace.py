"""Host-side timeline of one otg_assemble_files job (OTG_DISPATCH_TRACE=1 must be set in the environment: one stderr line per stage and batch).
usage: OTG_DISPATCH_TRACE=1 python scripts/dispatch_trace.py [loci] [ingest_threads] [batch,batch,...]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import otter_amd  # noqa: E402
from otter_amd import bamwrite  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
batches = [int(b) for b in (sys.argv[3] if len(sys.argv) > 3 else "1000").split(",")]
tmp = tempfile.mkdtemp(prefix="otg_dt_")
fx = bamwrite.make_tr_fixture(tmp, R, depth=30, len_range=(1000, 5000), seed=7)
for batch in batches:
    for rep in range(3):
        sys.stderr.write("[otg trace] ==== batch_regions %d, pass %d\n" % (batch, rep)); sys.stderr.flush()
        t1 = time.perf_counter()
        txt, st = otter_amd.assemble_files(fx["bam"], fx["bed"], read_group="s1", batch_regions=batch, offset_l=1, offset_r=1, mapq=10, threads=T)
        dt = time.perf_counter() - t1
        sys.stderr.write("[otg trace] ==== wall %.1f ms = %.0f regions/s\n" % (dt * 1e3, R / dt)); sys.stderr.flush()
