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


def test_genotype_config3_full_size(gpu, oracle):
    """BASELINE configs[3] at full size — 5 000 regions x 101 alleles (50 samples x 2 + reference) of 1-5 kb, the bytes bench.py's genotype
    leg runs (synth.config_batch(3)) — through otg_genotype_cluster_batch, EVERY region against the oracle's anallele_cluster
    (src/otterclust.cpp:463-527): gt / gt_l / gt_k / representatives bit-exact, hsd within 1e-9 relative.  Plus what the generator implies:
    the samples' alleles come from four population alleles per locus, so a region has at most a handful of genotypes and the reference
    allele shares one with the samples that carry population allele 0."""
    from otter_amd import synth
    b = synth.config_batch(3)
    n = synth.CONFIGS[3]["n_regions"]
    assert len(b["n_alleles"]) == n == 5000 and (b["n_alleles"] == 101).all()
    args = (b["arena"], b["seq_off"], b["seq_len"], np.ascontiguousarray(b["first_allele"][:-1]), b["n_alleles"])
    P = abi.default_params()
    g = gpu.genotype_cluster_batch(P, *args)
    g2 = gpu.genotype_cluster_batch(P, *args)
    e = oracle.genotype_cluster_batch(P, *args)
    for i in (0, 1, 2, 4, 5):
        assert np.array_equal(g[i], e[i]), i
        assert np.array_equal(g[i], g2[i]), i
    assert np.allclose(g[3], e[3], rtol=1e-9, atol=0, equal_nan=True)
    ngt = g[4]
    assert ngt.min() >= 1 and ngt.max() <= 12 and (ngt >= 2).mean() > 0.9
    # a shard alone gives the same records as inside the whole batch
    lo, hi = 2000, 2064
    f0, f1 = int(b["first_allele"][lo]), int(b["first_allele"][hi])
    part = gpu.genotype_cluster_batch(P, b["arena"], b["seq_off"][f0:f1].copy(), b["seq_len"][f0:f1].copy(),
                                      (b["first_allele"][lo:hi] - b["first_allele"][lo]).astype(np.uint32), b["n_alleles"][lo:hi].copy())
    assert np.array_equal(part[0], g[0][f0:f1]) and np.array_equal(part[4], ngt[lo:hi]) and np.array_equal(part[5], g[5][f0:f1])
