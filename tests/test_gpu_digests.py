"""Oracle-compared coverage of the bench workloads without oracle time on the GPU box: tests/golden/digest_c{1,2,4}.npz hold one
digest per region and per allele record (labels, ic / fc, scov / acov / tcov, `se` bits, sequence hash — tests/digests.py) of the CPU oracle
run in the build container over the first 1 000+ regions of synth.config_batch(1), (2) and (4) — the bytes bench.py runs (same chunk seeds,
rank 0's shard); scripts/make_golden_digests.py made them.  Here the same regions go through the C-ABI on the device and every digest must
be equal: `otter assemble`'s region loop body, src/assemble.cpp:71-150, on >= 3 000 regions of BASELINE configs[1], [2] and [4]."""
import os

import numpy as np
import pytest

import digests
from otter_amd import abi, synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_digest(cfg, adaptive=False):
    with np.load(os.path.join(GOLDEN, "digest_c%d%s.npz" % (cfg, "_adaptive" if adaptive else ""))) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("adaptive", [False, True], ids=["exact", "wfadaptive"])
@pytest.mark.parametrize("cfg", [1, 2, 4])
def test_bench_workload_regions_equal_the_oracle_digests(gpu, cfg, adaptive):
    """exact: the contract; wfadaptive: both aligners under WFA2-lib's adaptive reduction (10, 50, 1) — the mode the reference's binary runs in if
    its WFA2-lib build defaults to it (otg_params.heuristic; digest_c*_adaptive.npz = the oracle in that mode)"""
    want = load_digest(cfg, adaptive)
    n = int(want["n_regions"][0])
    assert n >= 1000 and int(want["seed"][0]) == synth.SEED and int(want["first_chunk"][0]) == 0
    b = synth.config_batch(cfg, n, workers=4)
    # the fixture and this run must have seen the same input bytes (numpy's generators are stable across versions; this says so if not)
    assert np.array_equal(digests._sha16(b["arena"].tobytes() + b["reads"].tobytes() + b["regions"].tobytes()), want["input_sha"])
    P = abi.default_params(realign=1 if synth.CONFIGS[cfg].get("realign") else 0,
                           heuristic=abi.OTG_HEURISTIC_WFADAPTIVE if adaptive else abi.OTG_HEURISTIC_NONE)
    res = gpu.assemble(P, b)
    got = digests.digest(res, b, 0, n)
    n_reg, n_al = digests.compare(got, want, "configs[%d]" % cfg)
    assert n_reg == n and n_al >= n                      # diploid loci: more allele records than regions
    # liveness of what the digest pins: two-allele regions, reassigned reads, POA-built alleles
    assert (want["fc"] == 2).sum() > 0.4 * n and (want["alleles"][:, 4] > 2).sum() > 0.8 * n_al
