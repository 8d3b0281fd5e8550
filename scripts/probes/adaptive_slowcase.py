import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import otter_amd
from otter_amd import abi, synth
ctx = otter_amd.Context(0)
batch = synth.make_batch(600, len_range=(3000, 9000), n_reads=16, err="ont", seed=8103)
for hp in [(10,50,1),(20,30,2)]:
    P = abi.default_params(heuristic=abi.OTG_HEURISTIC_WFADAPTIVE, heur_min_wavefront_length=hp[0], heur_max_distance_threshold=hp[1], heur_steps_between_cutoffs=hp[2])
    for it in range(2):
        t0=time.time(); res = ctx.assemble(P, batch); print(hp, "gpu %.2fs"%(time.time()-t0), flush=True)
