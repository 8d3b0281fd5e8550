"""One batch of same-sized gap-affine alignments through whichever exact tier the environment switches select (OTG_AFFINE_REG mask,
OTG_REG_SHAPE): times the second call of otg_affine_align_batch.  Used to compare tiers on identical work.
usage: python scripts/tier_probe.py [read_len] [n_pairs] [error_rate]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import otter_amd  # noqa: E402
from helpers import rand_seq, mutate, pair_tasks  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
rate = float(sys.argv[3]) if len(sys.argv) > 3 else 0.07
rng = np.random.default_rng(5)
pairs = []
for _ in range(256):
    base = rand_seq(rng, L)
    pairs.append((mutate(rng, base, rate), mutate(rng, base, rate)))
pairs = [pairs[i % 256] for i in range(N)]
arena, tasks = pair_tasks(pairs)
gpu = otter_amd.Context(0)
gpu.affine_align_batch(arena, tasks)
t0 = time.time()
sc, _ = gpu.affine_align_batch(arena, tasks)
dt = time.time() - t0
print("len %d x %d pairs, rate %.2f: mean score %.0f (reduced %.0f), mask %s shape %s: %.1f ms" % (
    L, N, rate, sc.mean(), sc.mean() / 2, os.environ.get("OTG_AFFINE_REG", "default"), os.environ.get("OTG_REG_SHAPE", "0"), dt * 1e3))
