"""GPU parity: KDE bound + average-linkage clustering + coverage repair vs the CPU oracle.
Integer outputs (labels, ic, fc) bit-exact; decision bounds (grid index * 0.0025) bit-exact as doubles."""
import numpy as np
import pytest
from otter_amd import abi
from helpers import cluster_cases, pack_cluster_cases

pytestmark = pytest.mark.gpu


def _compare(gpu, oracle, cases, **kw):
    P = abi.default_params(**kw)
    packed = pack_cluster_cases(cases)
    rc, el, eic, efc, eb = oracle.cluster_batch(P, *packed)
    assert rc == 0
    gl, gic, gfc, gb = gpu.cluster_batch(P, *packed)
    assert np.array_equal(gic, eic)
    assert np.array_equal(gfc, efc)
    assert np.array_equal(gl, el)
    assert np.array_equal(np.isnan(gb), np.isnan(eb))
    assert np.array_equal(np.nan_to_num(gb, nan=-7.0), np.nan_to_num(eb, nan=-7.0))


def test_cluster_synthetic_matrices(gpu, oracle):
    rng = np.random.default_rng(31)
    _compare(gpu, oracle, cluster_cases(rng, 360))


def test_cluster_max_alleles_variants(gpu, oracle):
    rng = np.random.default_rng(32)
    cases = cluster_cases(rng, 90)
    for ma in (1, 3, 0):
        _compare(gpu, oracle, cases, max_alleles=ma)


def test_cluster_max_cov_sized(gpu, oracle):
    """V = 200 valid reads (reference default max_cov), 19 900 distances per region."""
    rng = np.random.default_rng(33)
    cases = []
    for _ in range(3):
        n = 200
        g = rng.integers(0, 2, n)
        full = np.abs(g[:, None] - g[None, :]) * 0.2 + 0.13 + rng.random((n, n)) * 0.03
        full = np.triu(full, 1)
        cases.append((full[np.triu_indices(n, 1)].copy(), rng.integers(900, 1200, n).astype(np.uint32)))
    _compare(gpu, oracle, cases)


def test_cluster_from_real_distances(gpu, oracle):
    """Distance matrices produced by the alignment stage on synthetic TR regions (oracle pipeline output)."""
    from otter_amd import synth
    b = synth.make_batch(12, len_range=(300, 700), n_reads=20, err="hifi", seed=5)
    P = abi.default_params()
    r = oracle.assemble_batch(P, b)
    cases = []
    for i, reg in enumerate(b["regions"]):
        nv = int(r["regions"][i]["n_valid"])
        d = r["dist"][int(r["dist_off"][i]):int(r["dist_off"][i + 1])]
        idx = [j for j in range(reg["first_read"], reg["first_read"] + reg["n_reads"])
               if b["reads"][j]["spanning_l"] and b["reads"][j]["spanning_r"]]
        assert len(idx) == nv
        cases.append((d.copy(), b["reads"]["seq_len"][idx].astype(np.uint32)))
    _compare(gpu, oracle, cases)


def test_cluster_large_v_ties_and_time(gpu, oracle):
    """V = 100-200 with structure that stresses the nearest-neighbour tie rules of the wave-parallel NN-chain (quantised distances: many equal
    minima; several groups; noise), 48 regions.  The NN search and the `s*a + t*b` update run across the 64 lanes of a wave; results must not move."""
    import time
    rng = np.random.default_rng(34)
    cases = []
    for c in range(48):
        n = int(rng.integers(100, 201))
        k = int(rng.integers(1, 5))
        g = rng.integers(0, k, n)
        cen = np.sort(rng.uniform(0.0, 0.6, k))
        full = np.abs(cen[g][:, None] - cen[g][None, :]) + 0.05 + rng.random((n, n)) * 0.04
        if c % 3 == 0:
            full = np.round(full, 2)                     # ties everywhere
        elif c % 3 == 1:
            full = np.round(full, 3)
        full = np.triu(full, 1)
        cases.append((full[np.triu_indices(n, 1)].copy(), rng.integers(300, 3000, n).astype(np.uint32)))
    _compare(gpu, oracle, cases)
    P = abi.default_params()
    packed = pack_cluster_cases(cases)
    t0 = time.perf_counter()
    gpu.cluster_batch(P, *packed)
    dt = time.perf_counter() - t0
    print("48 regions of V=100..200: %.1f ms through otg_cluster_batch (upload + kernel + download)" % (dt * 1e3))
    assert dt < 2.0
