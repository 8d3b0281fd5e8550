"""Lengths beyond the 16-bit tiers.  The reference caps nothing but the flank size (src/otter_opts.cpp:148-150; region loop src/assemble.cpp:51-154), so a 40 kb
or a 70 kb locus inside a normal batch must come out like any other: bit-parallel edit tiers stop at 16 384 rows, the 16-bit wavefront tiers at 65 535
/ 32 766 bases — what lies beyond runs in the int32 tiers (HBM-resident wavefronts, 32-bit queue entries).  And when even those run out of provenance
storage, only that REGION drops out (status OTG_REGION_ALIGN_CAPACITY): the rest of the batch is delivered."""
import os
import subprocess
import sys

import numpy as np
import pytest

from otter_amd import abi, synth
from test_gpu_pipeline import compare

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def long_batch():
    """normal regions + a 36 kb locus at ONT divergence (beyond the bit-parallel tiers and the 32 766-base register / HBM-row tiers) + a 70 kb locus at HiFi
    divergence (pattern + text beyond 65 535: 32-bit offsets everywhere), three spanning reads each, one allele"""
    parts = [synth.make_batch(6, len_range=(400, 1500), n_reads=12, err="ont", seed=71, frac_partial=0.15),
             synth.make_batch(1, len_range=(36000, 36000), n_reads=3, err="ont", seed=72, frac_partial=0.0, frac_het=0.0),
             synth.make_batch(1, len_range=(70000, 70000), n_reads=3, err="hifi", seed=73, frac_partial=0.0, frac_het=0.0),
             synth.make_batch(4, len_range=(400, 1500), n_reads=12, err="ont", seed=74, frac_partial=0.15)]
    return synth.concat_batches(parts)


def test_long_loci_inside_a_normal_batch(gpu, oracle):
    b = long_batch()
    assert int(b["reads"]["seq_len"].max()) > 65535 // 2 and sorted(b["regions"]["n_reads"].tolist())[:2] == [3, 3]
    P = abi.default_params()
    res = gpu.assemble(P, b)
    ora = oracle.assemble_batch(P, b)
    compare(res, ora, b)
    assert (res["regions"]["status"] == abi.OTG_REGION_OK).all()
    long_regions = np.nonzero(b["regions"]["n_reads"] == 3)[0]
    for r in long_regions:                                  # the long loci did produce POA-built alleles
        a = res["alleles"][res["alleles"]["region"] == r]
        assert len(a) >= 1 and int(a["seq_len"].max()) > 30000


def test_long_loci_adaptive(gpu, oracle):
    b = long_batch()
    P = abi.default_params(heuristic=abi.OTG_HEURISTIC_WFADAPTIVE)
    compare(gpu.assemble(P, b), oracle.assemble_batch(P, b), b)


CHILD = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import otter_amd, oracle_lib
from otter_amd import abi
from test_gpu_long_reads import long_batch
b = long_batch()
P = abi.default_params()
with otter_amd.Context(0) as ctx:
    res = ctx.assemble(P, b)                    # must not fail as a batch
ora = oracle_lib.assemble_batch(P, b)
st = res["regions"]["status"]
bad = np.nonzero(st == abi.OTG_REGION_ALIGN_CAPACITY)[0]
assert len(bad) >= 1 and set(bad.tolist()) <= set(np.nonzero(b["regions"]["n_reads"] == 3)[0].tolist()), (st.tolist(), bad.tolist())
assert (res["regions"]["n_alleles"][bad] == 0).all()
ok = np.nonzero(st == abi.OTG_REGION_OK)[0]
assert len(ok) >= 10
for r in ok:                                    # every other region: the oracle's records
    ga = res["alleles"][res["alleles"]["region"] == r]; oa = ora["alleles"][ora["alleles"]["region"] == r]
    assert len(ga) == len(oa) and len(ga) == int(ora["regions"]["n_alleles"][r])
    for x, y in zip(ga, oa):
        for f in ("seq_len", "scov", "acov", "tcov", "ic", "label"):
            assert int(x[f]) == int(y[f]), (r, f)
        assert res["seqs"][int(x["seq_off"]):int(x["seq_off"]) + int(x["seq_len"])].tobytes() == ora["seqs"][int(y["seq_off"]):int(y["seq_off"]) + int(y["seq_len"])].tobytes()
print("ok: regions", bad.tolist(), "dropped with OTG_REGION_ALIGN_CAPACITY,", len(ok), "regions delivered")
"""


def test_a_region_beyond_the_workspaces_drops_out_alone(gpu):
    """the last-resort tier with a 1 MB provenance slab (test switch, read once per process: a child): the 36 kb ONT locus cannot be held"""
    gpu.trim()
    env = dict(os.environ, OTG_AFFINE_LAST_SLAB_MB="1")
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tests"))], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "dropped with OTG_REGION_ALIGN_CAPACITY" in r.stdout
