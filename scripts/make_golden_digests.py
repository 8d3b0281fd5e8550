"""Per-region oracle digests of the bench workloads, for tests/test_gpu_digests.py.

Runs the CPU oracle (oracle/libotter_oracle.so) in THIS build container over the first N regions of otter_amd.synth.config_batch(cfg) —
the same chunk-seeded bytes bench.py runs (first_chunk 0 = rank 0's shard) — and stores one digest per region and per allele record
(tests/digests.py) under tests/golden/digest_c<cfg>.npz.  The GPU box then checks every one of these regions against the oracle without
spending oracle time there.  Fixtures are data only: sizes, flags, hashes.

    python scripts/make_golden_digests.py 1 1250          # configs[1], first 1250 regions (5 chunks)
    python scripts/make_golden_digests.py 2 1000
    python scripts/make_golden_digests.py 4 1000
    python scripts/make_golden_digests.py 1 1250 --heuristic      # the same under wfadaptive(10, 50, 1) -> digest_c1_adaptive.npz
"""
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


def main():
    adaptive = "--heuristic" in sys.argv
    argv = [a for a in sys.argv if a != "--heuristic"]
    cfg, n = int(argv[1]), int(argv[2])
    threads = int(argv[3]) if len(argv) > 3 else (os.cpu_count() or 1)
    assert n % synth.CHUNK == 0, "whole chunks only: a chunk is the unit the generator seeds"
    b = synth.config_batch(cfg, n, workers=min(8, threads))
    P = abi.default_params(realign=1 if synth.CONFIGS[cfg].get("realign") else 0,
                           heuristic=abi.OTG_HEURISTIC_WFADAPTIVE if adaptive else abi.OTG_HEURISTIC_NONE)      # (10, 50, 1): WFA2-lib's own values
    O.lib()
    step = 4
    jobs = [(a, min(a + step, n)) for a in range(0, n, step)]
    parts = [None] * len(jobs)
    nxt = [0]
    lock = threading.Lock()
    t0 = time.time()

    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= len(jobs):
                return
            lo, hi = jobs[i]
            res = O.assemble_batch(P, b, region_range=(lo, hi))
            parts[i] = digests.digest(res, b, lo, hi)
            if i % 25 == 0:
                sys.stderr.write("  regions %d / %d  (%.0f s)\n" % (lo, n, time.time() - t0))
    th = [threading.Thread(target=work) for _ in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    out.update(cfg=np.array([cfg]), n_regions=np.array([n]), seed=np.array([synth.SEED]), first_chunk=np.array([0]),
               input_sha=digests._sha16(b["arena"].tobytes() + b["reads"].tobytes() + b["regions"].tobytes()))
    path = os.path.join(ROOT, "tests", "golden", "digest_c%d%s.npz" % (cfg, "_adaptive" if adaptive else ""))
    np.savez_compressed(path, **out)
    print("%s: %d regions, %d allele records, oracle time %.0f s on %d threads" % (path, n, len(out["alleles"]), time.time() - t0, threads))


if __name__ == "__main__":
    main()
