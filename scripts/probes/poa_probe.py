"""How long ONE graph (one wave) takes in the POA kernel, by backbone length and member count, and how that scales with the number of such graphs.
usage: python scripts/probes/poa_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import otter_amd
from helpers import rand_seq, mutate, pair_tasks, build_poa_batch

rng = np.random.default_rng(3)
gpu = otter_amd.Context(0)

def spec(L, n):
    base = rand_seq(rng, L)
    seqs = [mutate(rng, base, 0.07) for _ in range(n)]
    arena, tasks = pair_tasks([(seqs[0], s) for s in seqs])
    _, cigs = gpu.affine_align_batch(arena, tasks)
    return (seqs[0], [(seqs[i], cigs[i], True, True) for i in range(n)], np.float32(n * 0.4), np.float32(0.3))

for L, n, copies in ((2000, 15, 1), (5000, 30, 1), (5000, 30, 64), (5000, 30, 1500), (3000, 22, 1500)):
    sp = spec(L, n)
    sarena, carena, members, graphs = build_poa_batch([sp] * copies)
    gpu.poa_consensus_batch(sarena, carena, members, graphs)
    t0 = time.time(); gpu.poa_consensus_batch(sarena, carena, members, graphs); dt = time.time() - t0
    ops = sum(len(m[1]) for m in sp[1])
    print("backbone %d, %d members (%d ops), %d graphs: %.2f ms" % (L, n, ops, copies, dt * 1e3), flush=True)
