"""Affine-only micro-benchmark (rep vs members, config-1-like reads) for profiling the gap-affine WFA kernel."""
import sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import otter_amd
from otter_amd import abi, synth

nreg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
b = synth.make_batch(nreg, len_range=(1000, 5000), n_reads=30, err="ont", seed=7, frac_het=0.0, frac_partial=0.0)
reads, regions = b["reads"], b["regions"]
rows = []
for r in regions:
    f = int(r["first_read"])
    for a in range(1, 14):
        x, y = reads[f], reads[f + a]
        rows.append((int(x["seq_off"]), int(x["seq_len"]), int(y["seq_off"]), int(y["seq_len"])))
tasks = abi.make_tasks(rows)
ctx = otter_amd.Context(0)
for rep in range(reps):
    t = time.time()
    sc, cigs, cells = ctx.affine_align_batch(b["arena"], tasks, want_cells=True)
    dt = time.time() - t
    print("affine rep%d: %d aln %.3fs  %.1f Kaln/s  %.2f Gcells/s  mean score=%.0f mean len=%.0f" % (
        rep, len(tasks), dt, len(tasks) / dt / 1e3, cells.sum() / dt / 1e9, sc.mean(), np.mean([len(c) for c in cigs])), flush=True)
