"""Times the end-of-run gather (one rank over RCCL) on a config-1 batch: device-resident path vs host round trip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import otter_amd
from otter_amd import abi, synth, parallel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29733")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = otter_amd.Context(0)
batch = synth.make_batch(n, len_range=(1000, 5000), n_reads=30, err="ont", seed=synth.SEED)
ctx.assemble_submit(abi.default_params(), batch)
ctx.assemble_run()
dev = torch.device("cuda", 0)
for rep in range(3):
    t = time.perf_counter(); g = parallel.gather_records(ctx.assemble_device_results(), dist, 0, 1, dev); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); g2 = parallel.gather_records(ctx.assemble_collect(), dist, 0, 1, dev); torch.cuda.synchronize(); t2 = time.perf_counter() - t
    print("rep %d: device path %.1f ms, host path %.1f ms, %d alleles, %.1f MB of sequence" % (rep, t1 * 1e3, t2 * 1e3, len(g["alleles"]), len(g["seqs"]) / 1e6), flush=True)
dist.destroy_process_group()
