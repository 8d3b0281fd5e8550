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
