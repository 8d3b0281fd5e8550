import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import otter_amd
ctx = otter_amd.Context(0)
import torch
print("avail", torch.cuda.is_available(), "count", torch.cuda.device_count(), flush=True)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29711")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("pg ok", flush=True)
dist.destroy_process_group()
