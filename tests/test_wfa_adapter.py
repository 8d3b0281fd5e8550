"""The operator-level adapter: include/wfa_adapter/bindings/cpp/WFAligner.hpp gives the six `wfa::` symbols the reference links
(src/analignments.cpp:25,31,37,70-71,88-97,268-280; Makefile:4-5) over the C-ABI.  CPU: the header and a driver written like the reference's
call sites compile with g++ -std=c++17 -Wall against the C header alone and link with libotter_gpu.so; without a device every call reports
StatusOOM.  GPU: every score and op string the driver prints equals the CPU oracle's."""
import os
import subprocess

import numpy as np
import pytest

from helpers import rand_seq, mutate, tr_seq, pair_tasks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "adapter", "wfa_adapter_driver.cpp")


def build(tmp):
    exe = os.path.join(tmp, "wfa_adapter_driver")
    lib = os.path.join(ROOT, "otter_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include", "wfa_adapter"), "-I" + os.path.join(ROOT, "include"),
                           "-o", exe, SRC, "-L" + lib, "-lotter_gpu", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_adapter_builds_and_fails_loudly_without_a_device(tmp_path):
    import otter_amd
    exe = build(str(tmp_path))
    if otter_amd.device_count() > 0:
        pytest.skip("a device is present: the no-device behaviour is what this test is about")
    r = subprocess.run([exe], input=b"ACGT ACGA 0 0 0 0 0\n", capture_output=True, timeout=120)
    assert r.returncode == 3 and b"adapter:" in r.stderr and b"HIP" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(), ("none",), ("wfadaptive", "10", "50", "1"), ("wfadaptive", "4", "20", "2")], ids=["default", "none", "wfadaptive", "wfadaptive-4-20-2"])
def test_adapter_equals_the_oracle(oracle, tmp_path, mode):
    """setHeuristicNone / setHeuristicWFadaptive on the adapter against the oracle in the same mode (the default of the adapter is exact)"""
    exe = build(str(tmp_path))
    rng = np.random.default_rng(61)
    pairs, forms = [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"ACTGGA", b"ACCGA")], [None] * 4
    for i in range(150):
        n = int(rng.integers(1, 900))
        a = tr_seq(rng, n) if i % 2 else rand_seq(rng, n)
        b = mutate(rng, a, [0.01, 0.07, 0.2][i % 3])
        f = None
        if i % 3 == 0:
            d = len(a) - len(b)
            f = [(0, d, 0, 0), (d, 0, 0, 0), (d // 2, d // 2, 0, 0)][(i // 3) % 3] if d >= 0 else [(0, 0, 0, -d), (0, 0, -d, 0)][(i // 3) % 2]
        pairs.append((a, b)); forms.append(f)
    text = b"".join(b"%s %s %d %d %d %d %d\n" % ((a or b"-"), (b or b"-"), *((1,) + tuple(f) if f else (0, 0, 0, 0, 0))) for (a, b), f in zip(pairs, forms))
    r = subprocess.run([exe] + list(mode), input=text, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    lines = r.stdout.decode().split("\n")[:-1]
    assert len(lines) == len(pairs)
    arena, tasks = pair_tasks(pairs, forms)
    if mode and mode[0] == "wfadaptive":
        oracle.set_heuristic(1, *[int(x) for x in mode[1:]])
    try:
        ed = oracle.edit_distance_batch(arena, tasks)
        sc, cg = oracle.affine_align_batch(arena, tasks)
    finally:
        oracle.set_heuristic(0)
    for i, ln in enumerate(lines):
        st1, e1, st2, s2, c = ln.split(" ")
        assert (int(st1), int(st2)) == (0, 0)
        assert int(e1) == int(ed[i]), i                      # edit distance: positive
        assert int(s2) == -int(sc[i]), i                     # gap-affine, match 0: WFA2-lib reports the negative penalty
        assert (b"" if c == "-" else c.encode()) == cg[i], i
