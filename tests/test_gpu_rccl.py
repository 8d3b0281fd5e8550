"""The N>1 path on hardware: `python bench.py --gpus 2` starts two ranks itself (fresh processes, one per GPU), shards the BED list
statically, and gathers the allele records to rank 0 over RCCL from the library's device buffers.  Needs two visible devices; the round's
1-GPU box skips it (tests/test_distributed_gloo.py covers the same gather code over gloo, world size 2)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_over_rccl():
    import otter_amd
    if otter_amd.device_count() < 2:
        pytest.skip("one visible device: RCCL world 2 needs two")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "1", "--regions", "250", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size_rccl"] == 2
    assert line["config"]["allele_records"] >= 500 and line["config"]["gather"]["bytes"] > 0
    assert line["value"] > 0 and line["scaling"] == "weak"


def _records(res):
    out = []
    for r, g in enumerate(res["regions"]):
        for a in res["alleles"][int(g["first_allele"]):int(g["first_allele"]) + int(g["n_alleles"])]:
            assert int(a["region"]) == r
            out.append((r, int(a["label"]), int(a["scov"]), int(a["acov"]), int(a["tcov"]), float(a["se"]),
                        res["seqs"][int(a["seq_off"]):int(a["seq_off"]) + int(a["seq_len"])].tobytes()))
    return out


def test_library_gather_world_1(gpu):
    """The library's own RCCL gather (otg_comm_* / otg_gather_sizes / otg_gather_records: no PyTorch involved) with a world of one rank: RCCL is
    loaded, a communicator is made, the size all-gather and the record hand-over run on the library's stream; rank 0's result must equal
    otg_assemble_collect's."""
    import numpy as np
    import otter_amd
    from otter_amd import abi, synth
    b = synth.make_batch(12, len_range=(200, 500), n_reads=10, err="hifi", seed=33)
    res = gpu.assemble(abi.default_params(), b)
    comm = otter_amd.Comm(0, 0, 1, otter_amd.Comm.unique_id())
    try:
        g = comm.gather_records(gpu)
    finally:
        comm.close()
    assert g["counts"].tolist() == [[12, len(res["alleles"]), int(res["alleles"]["seq_len"].astype(np.int64).sum())]]
    assert _records(g) == _records(res) and len(_records(g)) >= 12


_RANK_CODE = r"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, %(root)r)
import otter_amd
from otter_amd import abi, synth
rank, world, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
if rank == 0:
    uid = otter_amd.Comm.unique_id()
    open(idfile + ".tmp", "wb").write(uid); os.rename(idfile + ".tmp", idfile)
else:
    for _ in range(600):
        if os.path.exists(idfile): break
        time.sleep(0.1)
    uid = open(idfile, "rb").read()
b = synth.make_batch(40, len_range=(200, 600), n_reads=10, err="hifi", seed=34)
lo, hi = synth.shard_bounds(40, world, rank)
ctx = otter_amd.Context(rank)
comm = otter_amd.Comm(rank, rank, world, uid)
ctx.assemble(abi.default_params(), b, region_range=(lo, hi))
g = comm.gather_records(ctx)
if rank == 0:
    np.savez(out, regions=g["regions"], alleles=g["alleles"], seqs=g["seqs"], counts=g["counts"])
comm.close(); ctx.close()
"""


def test_library_gather_two_ranks(tmp_path):
    """Two processes, two devices, static BED split, the library's RCCL gather (ncclSend / ncclRecv group): rank 0 holds the records of all 40
    regions in BED order, equal to one device running the whole batch."""
    import numpy as np
    import otter_amd
    from otter_amd import abi, synth
    if otter_amd.device_count() < 2:
        pytest.skip("one visible device: RCCL world 2 needs two")
    idfile, out = str(tmp_path / "rccl.id"), str(tmp_path / "gathered.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = [subprocess.Popen([sys.executable, "-c", _RANK_CODE % {"root": ROOT}, str(r), "2", idfile, out], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    g = dict(np.load(out))
    b = synth.make_batch(40, len_range=(200, 600), n_reads=10, err="hifi", seed=34)
    with otter_amd.Context(0) as ctx:
        whole = ctx.assemble(abi.default_params(), b)
    assert g["counts"][:, 0].tolist() == [20, 20]
    assert _records(g) == _records(whole)


@pytest.mark.gpu
def test_bench_multi_rank_control_flow_rehearsal(gpu):
    """bench.py's N > 1 path executed on the one-GPU box (OTG_BENCH_REHEARSAL=1: two ranks, both contexts on device 0, gloo, records gathered from host
    memory): static BED shards per rank, the timed steps with their gather, the configs[4] leg that rides along at N > 1, the max / sum over ranks, rank 0
    printing ONE line.  A control-flow check, not a measurement (the line says so)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gpu.trim()
    env = dict(os.environ, OTG_BENCH_REHEARSAL="1", OTG_BENCH_REHEARSAL_REGIONS="250")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--leg-steps", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "rehearsal" in d["config"]
    assert d["config"]["regions_per_gpu"] == 250 and "static BED shard of 500" in d["config"]["workload"]
    leg = d["config"]["legs"]["configs[4]"]
    assert leg["n_gpus"] == 2 and leg["value"] > 0 and leg["allele_records"] >= 500 and "1000-10000 bp" in leg["workload"]
    assert d["config"]["gather"]["bytes"] > 0
