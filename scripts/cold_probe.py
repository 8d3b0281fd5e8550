"""Stage times of the first (cold context) and later runs of one batch: what a fresh context costs the dispatcher."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otter_amd
from otter_amd import abi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
b = synth.config_batch(1, n)
P = abi.default_params()
for c in range(2):
    ctx = otter_amd.Context(0)
    for rep in range(3):
        t0 = time.perf_counter(); ctx.assemble_submit(P, b); t1 = time.perf_counter(); ctx.assemble_run(); t2 = time.perf_counter(); ctx.assemble_collect(); t3 = time.perf_counter()
        st = ctx.assemble_stats()
        print("ctx %d rep %d: submit %.0f ms, run %.0f ms, collect %.0f ms | " % (c, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3) +
              ", ".join("%s %.0f" % (k[3:], st[k]) for k in ("ms_edit", "ms_cluster", "ms_reassign", "ms_affine", "ms_poa")), flush=True)
    ctx.close()
