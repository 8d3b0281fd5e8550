"""GPU parity: POA consensus kernel vs the CPU oracle (and the reference's own PPOA when oracle/_ref is built)."""
import json
import os
import numpy as np
import pytest
from helpers import build_poa_batch, random_poa_specs, pair_tasks

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _check(gpu, oracle, specs):
    sarena, carena, members, graphs = build_poa_batch(specs)
    exp = oracle.poa_consensus_batch(sarena, carena, members, graphs)
    got = gpu.poa_consensus_batch(sarena, carena, members, graphs)
    assert got == exp
    if oracle.ref() is not None:
        assert got == oracle.poa_consensus_batch(sarena, carena, members, graphs, which="ref")
    return got


def test_poa_reference_kats_on_gpu(gpu, oracle):
    """The reference's 4 known-answer tests (test/ppoa_test.cpp:39-105): GPU affine WFA op strings -> GPU POA."""
    kats = json.load(open(os.path.join(GOLD, "ppoa_kats.json")))
    for kat in kats:
        seqs = [s.encode() for s in kat["sequences"]]
        arena, tasks = pair_tasks([(seqs[0], s) for s in seqs])
        _, cigs = gpu.affine_align_batch(arena, tasks)
        n = len(seqs)
        spec = (seqs[0], [(seqs[i], cigs[i], True, True) for i in range(n)], np.float32(n * 0.4), np.float32(0.3))
        got = _check(gpu, oracle, [spec])
        assert got[0].decode() == kat["expected"]


def test_poa_random_small(gpu, oracle):
    rng = np.random.default_rng(41)
    _check(gpu, oracle, random_poa_specs(rng, oracle, 200, 1, 120, err=0.1))


def test_poa_random_ont_kb(gpu, oracle):
    rng = np.random.default_rng(42)
    _check(gpu, oracle, random_poa_specs(rng, oracle, 24, 800, 2500, err=0.07))


def test_poa_hifi(gpu, oracle):
    rng = np.random.default_rng(43)
    _check(gpu, oracle, random_poa_specs(rng, oracle, 60, 300, 700, err=0.002))
