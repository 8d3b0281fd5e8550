"""Micro-benchmark of the L1 batched aligners on config-2-like pairs (ONT, 1-5 kb)."""
import sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import otter_amd
from otter_amd import abi, synth

nreg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = synth.make_batch(nreg, len_range=(1000, 5000), n_reads=30, err="ont", seed=7)
reads, regions = b["reads"], b["regions"]
rows = []
aff_rows = []
for r in regions:
    idx = [i for i in range(r["first_read"], r["first_read"] + r["n_reads"]) if reads[i]["spanning_l"] and reads[i]["spanning_r"]]
    for a in range(len(idx)):
        for c in range(a + 1, len(idx)):
            x, y = reads[idx[a]], reads[idx[c]]
            if x["seq_len"] < y["seq_len"]:
                x, y = y, x
            rows.append((int(x["seq_off"]), int(x["seq_len"]), int(y["seq_off"]), int(y["seq_len"])))
    for a in range(1, min(len(idx), 14)):
        x, y = reads[idx[0]], reads[idx[a]]
        aff_rows.append((int(x["seq_off"]), int(x["seq_len"]), int(y["seq_off"]), int(y["seq_len"])))
tasks = abi.make_tasks(rows)
atasks = abi.make_tasks(aff_rows)
print("regions", nreg, "edit tasks", len(tasks), "affine tasks", len(atasks), flush=True)
ctx = otter_amd.Context(0)
for rep in range(3):
    t = time.time()
    sc, cells = ctx.edit_distance_batch(b["arena"], tasks, want_cells=True)
    dt = time.time() - t
    print("edit rep%d: %.3fs  %.1f Mpairs/s  %.2f Gcells/s  mean s=%.0f  (%.1f regions/s edit-only incl. H2D)" % (
        rep, dt, len(tasks) / dt / 1e6, cells.sum() / dt / 1e9, sc.mean(), nreg / dt), flush=True)
for rep in range(2):
    t = time.time()
    sc, cigs, cells = ctx.affine_align_batch(b["arena"], atasks, want_cells=True)
    dt = time.time() - t
    print("affine rep%d: %.3fs  %.1f Kaln/s  %.2f Gcells/s mean score=%.0f" % (rep, dt, len(atasks) / dt / 1e3, cells.sum() / dt / 1e9, sc.mean()), flush=True)
