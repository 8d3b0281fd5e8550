"""One-off differential check at config-1 shapes: the HIP pipeline vs the CPU oracle (multi-threaded over regions) on
N regions per case; prints one line per case.  Heavier than the -m gpu tests (the oracle runs ~0.3 regions/s/thread).
usage: python scripts/diff_check.py [regions per case] [seed offset] [none | wfadaptive[:min_wf_len,max_dist,steps]]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import otter_amd
from otter_amd import abi, synth
import oracle_lib
from test_gpu_pipeline import compare

oracle_lib.lib()
ctx = otter_amd.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
seed_off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
heur = sys.argv[3] if len(sys.argv) > 3 else "none"
hkw = {}
if heur.startswith("wfadaptive"):
    a, b, c = (int(x) for x in heur.split(":")[1].split(",")) if ":" in heur else (10, 50, 1)
    hkw = dict(heuristic=abi.OTG_HEURISTIC_WFADAPTIVE, heur_min_wavefront_length=a, heur_max_distance_threshold=b, heur_steps_between_cutoffs=c)
cases = [dict(len_range=(1000, 5000), n_reads=30, err="ont", seed=101),
         dict(len_range=(1000, 5000), n_reads=30, err="ont", seed=102, realign=True),
         dict(len_range=(3000, 9000), n_reads=16, err="ont", seed=103),
         dict(len_range=(300, 1500), n_reads=40, err="hifi", seed=104),
         dict(len_range=(1000, 10000), n_reads=30, err="ont", seed=105),
         dict(len_range=(100, 800), n_reads=60, err="ont", seed=106, realign=True)]
nth = min(16, os.cpu_count() or 1)
for c in cases:
    kw = dict(c); realign = kw.pop("realign", False); kw["seed"] += seed_off
    batch = synth.make_batch(n, realign=realign, **kw)
    P = abi.default_params(realign=1 if realign else 0, **hkw)
    t0 = time.time()
    res = ctx.assemble(P, batch)
    tg = time.time() - t0
    parts = [None] * nth
    def work(i):
        a, b = i * n // nth, (i + 1) * n // nth
        if b > a: parts[i] = (a, b, oracle_lib.assemble_batch(P, batch, region_range=(a, b)))
    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nth)]
    [t.start() for t in th]; [t.join() for t in th]
    to = time.time() - t0
    bad = 0
    for p in parts:
        if p is None: continue
        a, b, ora = p
        # compare region by region: alleles of regions [a,b)
        gr = res["regions"][a:b]; orr = ora["regions"][a:b]
        for f in ("status", "ic", "fc", "n_valid", "n_alleles"):
            if not np.array_equal(gr[f], orr[f]): bad += 1; print("  mismatch field", f, "regions", a, b)
        ga = res["alleles"][(res["alleles"]["region"] >= a) & (res["alleles"]["region"] < b)]
        oa = ora["alleles"]
        if len(ga) != len(oa): bad += 1; print("  allele count", len(ga), len(oa)); continue
        for f in ("seq_len", "scov", "acov", "tcov", "ic", "ps", "hp", "label"):
            if not np.array_equal(ga[f], oa[f]): bad += 1; print("  mismatch allele field", f)
        if not np.allclose(ga["se"], oa["se"], rtol=0, atol=1e-6): bad += 1; print("  se differs")
        for i in range(len(ga)):
            gs = res["seqs"][int(ga[i]["seq_off"]):int(ga[i]["seq_off"]) + int(ga[i]["seq_len"])].tobytes()
            os_ = ora["seqs"][int(oa[i]["seq_off"]):int(oa[i]["seq_off"]) + int(oa[i]["seq_len"])].tobytes()
            if gs != os_: bad += 1
    print("case %s: %d regions, %d alleles, gpu %.2fs oracle %.1fs (%d threads): %s" % (c, n, len(res["alleles"]), tg, to, nth, "OK" if bad == 0 else "%d MISMATCHES" % bad), flush=True)
