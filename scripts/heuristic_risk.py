"""How much would otter's results move if the author's WFA2-lib build ran its adaptive heuristic?  (CPU only; the oracle in both modes.)

otter never calls setHeuristic* (src/assemble.cpp:49-50), and the default of that WFA2-lib build cannot be recovered offline:
`none` (exact — the contract of this repository) or wf_heuristic_wfadaptive(10, 50, 1).  This script runs the CPU oracle over the first N
regions of a bench workload twice — exact, then with the adaptive reduction switched on in BOTH aligners (oto_set_heuristic) — and counts the
regions whose pairwise distances, final read labels or allele records differ.

    python scripts/heuristic_risk.py <config 1|2|4> <regions> [threads]      -> one JSON line (also appended to profiles/r03_heuristic_risk.jsonl)
"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import digests  # noqa: E402
from otter_amd import abi, synth  # noqa: E402


def run_all(P, b, n, threads):
    step = 2
    jobs = [(a, min(a + step, n)) for a in range(0, n, step)]
    out = [None] * len(jobs)
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= len(jobs):
                return
            lo, hi = jobs[i]
            res = O.assemble_batch(P, b, region_range=(lo, hi))
            d = digests.digest(res, b, lo, hi)
            d["dist"] = [res["dist"][int(res["dist_off"][r]):int(res["dist_off"][r + 1]) if r + 1 < len(res["dist_off"]) else None].copy() for r in range(lo, hi)]
            out[i] = d
    th = [threading.Thread(target=work) for _ in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    return out


def main():
    cfg, n = int(sys.argv[1]), int(sys.argv[2])
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
    b = synth.config_batch(cfg, n, workers=min(8, threads))
    P = abi.default_params(realign=1 if synth.CONFIGS[cfg].get("realign") else 0)
    L = O.lib()
    L.oto_set_heuristic.argtypes = [C.c_int] * 4
    t0 = time.time()
    L.oto_set_heuristic(0, 10, 50, 1)
    exact = run_all(P, b, n, threads)
    t1 = time.time()
    L.oto_set_heuristic(1, 10, 50, 1)
    adapt = run_all(P, b, n, threads)
    t2 = time.time()
    L.oto_set_heuristic(0, 10, 50, 1)
    reg_dist = reg_lab = reg_al = reg_seq = reg_se = n_pairs = n_pairs_diff = 0
    worst = 0.0
    for e, a in zip(exact, adapt):
        k = len(e["status"])
        for i in range(k):
            de, da = e["dist"][i], a["dist"][i]
            n_pairs += len(de)
            nd = int((de != da).sum()) if len(de) == len(da) else len(de)
            n_pairs_diff += nd
            reg_dist += nd > 0
            if nd and len(de) == len(da):
                worst = max(worst, float(np.abs(de - da).max()))
            reg_lab += bool((e["labels_sha"][i] != a["labels_sha"][i]).any())
        # allele records of these regions (digest rows are per allele, in region order): integer fields, sequences, se separately
        ea, aa = e["alleles"], a["alleles"]
        for rr in set(ea[:, 0].tolist()) | set(aa[:, 0].tolist()):
            me, ma = ea[:, 0] == rr, aa[:, 0] == rr
            cov = [0, 1, 3, 4, 5, 6, 7, 8]                     # region, label, scov, acov, tcov, ic, ps, hp (column 2 is the sequence length)
            if me.sum() != ma.sum() or (ea[me][:, cov] != aa[ma][:, cov]).any():
                reg_al += 1; reg_seq += 1; continue
            if (ea[me][:, 2] != aa[ma][:, 2]).any() or (e["seq_sha"][me] != a["seq_sha"][ma]).any():
                reg_seq += 1; continue
            if (e["se_bits"][me] != a["se_bits"][ma]).any():
                reg_se += 1
    line = {"config": cfg, "regions": n, "heuristic": "wfadaptive(10,50,1) in the edit and the gap-affine aligner", "regions_with_a_changed_distance": int(reg_dist),
            "pair_distances": int(n_pairs), "pair_distances_changed": int(n_pairs_diff), "largest_distance_change": round(worst, 6),
            "regions_with_changed_final_labels": int(reg_lab), "regions_with_changed_allele_count_or_coverage": int(reg_al),
            "regions_with_a_changed_allele_sequence": int(reg_seq), "regions_with_only_se_changed": int(reg_se),
            "oracle_seconds": {"exact": round(t1 - t0, 1), "adaptive": round(t2 - t1, 1)}, "threads": threads}
    print(json.dumps(line))
    with open(os.path.join(ROOT, "profiles", "r03_heuristic_risk.jsonl"), "a") as f:
        f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
