"""How wide do wavefronts get under wfadaptive(10,50,1)?  (CPU only; sizing of the adaptive-mode kernels' windows.)

Runs the oracle's whole pipeline in adaptive mode over the first N regions of a bench workload with the oracle's width statistics on and
prints, per aligner kind, the histogram of each alignment's widest wavefront (edit: the widest computed M wavefront; gap-affine: the widest
union of the M wavefronts of the last gap_open + gap_ext scores, i.e. the window a band-following kernel must hold).

    python scripts/heuristic_widths.py <config 1|2|4> <regions> [threads]
"""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from otter_amd import abi, synth  # noqa: E402


def main():
    cfg, n = int(sys.argv[1]), int(sys.argv[2])
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
    b = synth.config_batch(cfg, n, workers=min(8, threads))
    P = abi.default_params(realign=1 if synth.CONFIGS[cfg].get("realign") else 0)
    L = O.lib()
    L.oto_set_heuristic(1, 10, 50, 1)
    L.oto_width_stats(1, None)
    jobs = list(range(0, n, 2))
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= len(jobs):
                return
            O.assemble_batch(P, b, region_range=(jobs[i], min(jobs[i] + 2, n)))
    th = [threading.Thread(target=work) for _ in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    out = np.zeros(4 * 18, dtype=np.uint64)
    L.oto_width_stats(0, abi.ptr(out))
    ph = np.zeros(16, dtype=np.uint64)
    L.oto_width_phase.argtypes = [C.c_void_p]; L.oto_width_phase.restype = None
    L.oto_width_phase(abi.ptr(ph))
    L.oto_set_heuristic(0, 10, 50, 1)
    kinds = ["edit end-to-end", "edit ends-free", "affine end-to-end", "affine ends-free"]
    res = {"config": cfg, "regions": n}
    for k, name in enumerate(kinds):
        h = out[k * 18:k * 18 + 16].tolist()
        res[name] = {"alignments": int(out[k * 18 + 16]), "scores": int(out[k * 18 + 17]),
                     "widest_wavefront_le": {str(8 << i): int(c) for i, c in enumerate(h) if c}}
    for k, name in enumerate(kinds):          # the alignments that outgrow the fast tier's window (1 020 diagonals edit, 252 gap-affine): is the wide phase a prefix?
        n_w, wide, last, allsc = (int(x) for x in ph[4 * k:4 * k + 4])
        res[name]["wider_than_window"] = {"alignments": n_w, "mean_scores": round(allsc / max(1, n_w), 1), "mean_wide_scores": round(wide / max(1, n_w), 1),
                                        "mean_last_wide_score": round(last / max(1, n_w), 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
