"""GPU parity of the alternative kernel paths.  The library picks its tier chains by itself; environment switches
(read once per process) force the fallback chains — un-bounded affine pass, HBM-row affine tiers, the generic affine kernel, wavefront-only edit
distance, un-routed / un-sorted bit-parallel tiers, POA graphs in global memory only.  Every chain must be bit-exact, so the
aligner / POA parity tests are re-run in a child process per switch."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# a subset of the aligner tests without the (oracle-heavy) admission sweep, for the chains whose difference lies outside the tier admission
AFFINE_CORE = ["tests/test_gpu_affine.py::test_affine_small_mixed", "tests/test_gpu_affine.py::test_affine_long_ont",
               "tests/test_gpu_affine.py::test_affine_probe_boundaries", "tests/test_gpu_affine.py::test_affine_packed_sequence_capacity_sweep"]

# test_gpu_affine.py without the admission sweep of the HBM-row tiers (11 s of oracle work; kept where those tiers are what the switch forces)
AFFINE_ALL_BUT_ADMISSION = ["tests/test_gpu_affine.py::" + t for t in (
    "test_affine_small_mixed", "test_affine_other_penalties", "test_affine_long_ont", "test_affine_cigar_valid_full_size", "test_affine_bounded_tandem_repeats",
    "test_affine_non_acgt_bytes", "test_affine_packed_sequence_capacity_sweep", "test_affine_probe_boundaries", "test_affine_wide_free_begin")]

CASES = [
    ("OTG_NO_AFFINE_BOUND", ["tests/test_gpu_affine.py"]),                               # no score bound: everything on the HBM-row tiers, un-pruned
    ("OTG_AFFINE_REG=0", ["tests/test_gpu_affine.py"]),                                  # bound + HBM-row tiers only (what the register tiers fall back to)
    ("OTG_AFFINE_REG=25", AFFINE_ALL_BUT_ADMISSION + ["tests/test_gpu_poa.py"]),        # register tiers 1024 / 4096 / 8192 only: the 1536 / 2048 windows on the multi-wave tier
    ("OTG_NO_AFFINE_V3", AFFINE_CORE),                                                   # generic kernel only
    ("OTG_NO_MYERS", ["tests/test_gpu_edit.py::test_edit_small_mixed", "tests/test_gpu_edit.py::test_edit_long_ont"]),
    ("OTG_NO_EDIT_ROUTE OTG_NO_EDIT_SORT", ["tests/test_gpu_edit.py"]),
    ("OTG_NO_EDIT_SAMPLE", ["tests/test_gpu_edit.py"]),
    ("OTG_EDIT_TIERS=255", ["tests/test_gpu_edit.py"]),                                 # all eight bit-parallel tiers (the two three-block ones only run on passes with millions of pairs otherwise)
    ("OTG_POA_V1 OTG_POA_NO_LDS", ["tests/test_gpu_poa.py", "tests/test_gpu_pipeline.py::test_ont_kb"]),   # first-generation POA (serial threading, Kahn sweep), global memory only
    ("OTG_POA_NO_LDS", ["tests/test_gpu_poa.py"]),                                       # second-generation POA on every graph (the op-string fuzz and the insertion stretches in global memory)
    ("OTG_POA_PIECE_MB=1", ["tests/test_gpu_poa.py", "tests/test_gpu_pipeline.py::test_ont_kb"]),         # graph images in many small pieces that reuse the work arrays
    ("OTG_NO_REASSIGN_REV", ["tests/test_gpu_pipeline.py::test_haps_mode", "tests/test_gpu_pipeline.py::test_ont_kb"]),
    ("OTG_REG_SHAPE=11210", AFFINE_CORE),                                                # the other instantiation of every register window: <2,4>, <1,12> at 4 waves, <1,16>, <8,4>
    ("OTG_REG_SHAPE=2200", AFFINE_CORE),                                                 # <2,6> for the 1536 window, <1,16> without spills for the 2048 one
    ("OTG_AFFINE_CONCURRENT=1", AFFINE_CORE),                                            # register tiers side by side on three streams whatever the batch size
    ("OTG_AFFINE_CONCURRENT=0", ["tests/test_gpu_affine.py::test_affine_small_mixed", "tests/test_gpu_pipeline.py::test_ont_kb"]),   # ... and one after the other on a small batch
    # the adaptive mode's tier chains (wfa_adaptive.hip): byte-probe tiers only; the wide packed tier first; the 1024-diagonal LDS tier / the int32 tier alone for the gap-affine aligner
    ("OTG_ADAPTIVE_EDIT_TIERS=12", ["tests/test_gpu_adaptive.py::test_adaptive_edit_small", "tests/test_gpu_adaptive.py::test_adaptive_edit_long", "tests/test_gpu_adaptive.py::test_adaptive_edit_wide_and_huge"]),
    ("OTG_ADAPTIVE_NO_WIDE_START", ["tests/test_gpu_adaptive.py::test_adaptive_edit_small", "tests/test_gpu_adaptive.py::test_adaptive_edit_wide_and_huge", "tests/test_gpu_adaptive.py::test_adaptive_pipeline_hifi_and_haps"]),
    ("OTG_ADAPTIVE_EDIT_TIERS=16", ["tests/test_gpu_adaptive.py::test_adaptive_edit_small", "tests/test_gpu_adaptive.py::test_adaptive_edit_long", "tests/test_gpu_adaptive.py::test_adaptive_edit_parameters", "tests/test_gpu_adaptive.py::test_adaptive_edit_wide_and_huge", "tests/test_gpu_adaptive.py::test_adaptive_pipeline[0]"]),
    ("OTG_ADAPTIVE_EDIT_TIERS=2", ["tests/test_gpu_adaptive.py::test_adaptive_edit_small", "tests/test_gpu_adaptive.py::test_adaptive_edit_long", "tests/test_gpu_adaptive.py::test_adaptive_edit_parameters", "tests/test_gpu_adaptive.py::test_adaptive_pipeline_hifi_and_haps"]),
    ("OTG_ADAPTIVE_AFFINE_TIERS=2", ["tests/test_gpu_adaptive.py::test_adaptive_affine_small", "tests/test_gpu_adaptive.py::test_adaptive_affine_long", "tests/test_gpu_adaptive.py::test_adaptive_pipeline[0]"]),
    ("OTG_ADAPTIVE_AFFINE_TIERS=8", ["tests/test_gpu_adaptive.py::test_adaptive_affine_small", "tests/test_gpu_adaptive.py::test_adaptive_affine_long", "tests/test_gpu_adaptive.py::test_adaptive_affine_parameters", "tests/test_gpu_adaptive.py::test_adaptive_pipeline[0]"]),
    ("OTG_ADAPTIVE_AFFINE_TIERS=10", ["tests/test_gpu_adaptive.py::test_adaptive_affine_small", "tests/test_gpu_adaptive.py::test_adaptive_affine_long", "tests/test_gpu_adaptive.py::test_adaptive_affine_other_penalties_and_wide", "tests/test_gpu_adaptive.py::test_adaptive_pipeline_hifi_and_haps"]),
    ("OTG_ADAPTIVE_AFFINE_TIERS=18", ["tests/test_gpu_adaptive.py::test_adaptive_affine_small", "tests/test_gpu_adaptive.py::test_adaptive_affine_long", "tests/test_gpu_adaptive.py::test_adaptive_pipeline[0]"]),
    ("OTG_ADAPTIVE_AFFINE_TIERS=0", ["tests/test_gpu_adaptive.py::test_adaptive_affine_small", "tests/test_gpu_adaptive.py::test_adaptive_affine_other_penalties_and_wide"]),
]


@pytest.mark.parametrize("switch,targets", CASES, ids=[c[0].replace(" ", "+") for c in CASES])
def test_alternative_path(gpu, switch, targets):
    gpu.trim()          # the session context's aligner workspaces (tens of GB after the long-read tests): the child needs the room
    env = dict(os.environ)
    for kv in switch.split():
        k, _, v = kv.partition("=")
        env[k] = v or "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + targets,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:       # the whole output of the child, where a GPU box run leaves it behind
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "alt_path_%s.log" % switch.replace(" ", "+").replace("=", "-")), "w") as f:
                f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
        except OSError:
            pass
    assert r.returncode == 0, (switch, r.stdout[-3000:], r.stderr[-1000:])
