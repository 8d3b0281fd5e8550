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
