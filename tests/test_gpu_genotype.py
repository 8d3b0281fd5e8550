"""GPU parity: anallele_cluster (otter genotype) kernel vs the CPU oracle.  gt/gt_l/gt_k/reps bit-exact,
hsd (Hill-Shannon diversity, FP64 log/pow) within 1e-9 relative."""
import numpy as np
import pytest
from otter_amd import abi
from helpers import rand_seq, mutate, tr_seq

pytestmark = pytest.mark.gpu


def _regions(rng, n_regions, amax, with_short=False):
    seqs, first, counts = [], [], []
    for r in range(n_regions):
        A = int(rng.integers(1, amax))
        first.append(len(seqs)); counts.append(A)
        base = [tr_seq(rng, int(rng.integers(60, 900))) for _ in range(int(rng.integers(1, 5)))]
        for a in range(A):
            s = mutate(rng, base[int(rng.integers(0, len(base)))], [0.0, 0.002, 0.02][a % 3])
            if with_short and a % 7 == 3:
                s = s[:int(rng.integers(0, 3))] or b"N"
            if a % 11 == 5:
                s = s[:10] + b"NNN" + s[10:]
            seqs.append(s)
    arena, off, ln = abi.pack_seqs(seqs)
    return arena, off, ln, np.asarray(first, dtype=np.uint32), np.asarray(counts, dtype=np.uint32)


def _check(gpu, oracle, args):
    P = abi.default_params()
    e = oracle.genotype_cluster_batch(P, *args)
    g = gpu.genotype_cluster_batch(P, *args)
    for i in (0, 1, 2, 4, 5):
        assert np.array_equal(g[i], e[i]), i
    assert np.allclose(g[3], e[3], rtol=1e-9, atol=0, equal_nan=True)


def test_genotype_small(gpu, oracle):
    rng = np.random.default_rng(51)
    _check(gpu, oracle, _regions(rng, 120, 24, with_short=True))


def test_genotype_config4_sized(gpu, oracle):
    """A = 101 alleles per region (50 samples x 2 + reference), BASELINE config 3."""
    rng = np.random.default_rng(52)
    seqs, first, counts = [], [], []
    for r in range(6):
        pop = [tr_seq(rng, int(rng.integers(800, 2500))) for _ in range(4)]
        first.append(len(seqs)); counts.append(101)
        for a in range(101):
            seqs.append(mutate(rng, pop[int(rng.integers(0, 4))], 0.003))
    arena, off, ln = abi.pack_seqs(seqs)
    _check(gpu, oracle, (arena, off, ln, np.asarray(first, dtype=np.uint32), np.asarray(counts, dtype=np.uint32)))
