import numpy as np, sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import otter_amd
from helpers import pair_tasks, rand_seq
rng=np.random.default_rng(1)
ctx=otter_amd.Context(0)
print("ctx ok, exp variant", ctx.exp_variant, flush=True)
arena,tasks=pair_tasks([(b"ACGTACGTAC", b"ACGTTCGTAC")])
print(ctx.edit_distance_batch(arena,tasks), flush=True)
