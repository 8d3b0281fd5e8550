"""bench.py's baseline A plumbing: the oracle pipeline with every consensus handed to the reference's own PPOA (oracle/_ref, built from
/root/reference/src/anppoa.hpp where it lies) must give the records the port gives — the hook changes who computes the consensus, not
what it is."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
from otter_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_ppoa_hook_gives_the_same_records(oracle):
    refp = os.path.join(ROOT, "oracle", "_ref", "libotter_ref.so")
    if not os.path.exists(refp):
        pytest.skip("oracle/_ref not built (reference sources absent)")
    b = synth.make_batch(10, len_range=(200, 700), n_reads=12, err="ont", seed=5, frac_partial=0.2)
    P = abi.default_params()
    plain = oracle.assemble_batch(P, b)
    L = oracle.lib()
    R = C.CDLL(refp)
    L.oto_set_poa_hook.argtypes = [C.c_void_p]
    L.oto_set_poa_hook(C.cast(R.ref_poa_consensus_one, C.c_void_p))
    try:
        hooked = oracle.assemble_batch(P, b)
    finally:
        L.oto_set_poa_hook(None)
    assert (plain["alleles"]["acov"] > 2).sum() >= 8          # POA-built alleles are present
    for k in ("regions", "alleles", "labels"):
        assert plain[k].tobytes() == hooked[k].tobytes(), k
    n = int(plain["alleles"]["seq_len"].astype(np.int64).sum())
    assert plain["seqs"][:n].tobytes() == hooked["seqs"][:n].tobytes()


def test_bench_refuses_a_world_size_mismatch():
    """`--gpus N` must mean N ranks: under a launcher that set another WORLD_SIZE the bench exits non-zero before touching a GPU."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, timeout=120)
    assert p.returncode == 2 and b"WORLD_SIZE=1" in p.stderr
