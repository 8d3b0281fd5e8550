#!/usr/bin/env python3
"""bench.py — regions/sec of the otter assemble hot path on MI355X (contract in the task brief).

A step = one pass of the whole hot path ([local_realignment] -> fill_dist_matrix -> otter_hclust ->
invalid_reassignment -> rapid_consensus) over one resident batch of synthetic TR regions; inputs are
uploaded (otg_assemble_submit) before the timed region.  N=1 workload: BASELINE.json configs[1]
(10k regions x 1-5 kb TR, 30x ONT-error reads).  N>1: regions are sharded statically (each rank owns a
contiguous shard of N x per-GPU regions, weak scaling) with no data-path collective; the per-rank allele
records are gathered to rank 0 over RCCL at the end of every step, as north_star specifies.

Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(batch, params, n_sample, n_threads):
    """Times the CPU oracle (a port of the reference algorithm, kind 'port') on the first n_sample regions
    of the same workload, n_threads host threads (ctypes releases the GIL), static contiguous split."""
    import oracle_lib
    oracle_lib.lib()
    n_sample = min(n_sample, len(batch["regions"]))
    bounds = [(i * n_sample // n_threads, (i + 1) * n_sample // n_threads) for i in range(n_threads)]
    done = [0] * n_threads

    def work(i):
        a, b = bounds[i]
        if b > a:
            r = oracle_lib.assemble_batch(params, batch, region_range=(a, b))
            done[i] = int((r["regions"]["n_alleles"][a:b] > 0).sum())

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": round(sum(done) / dt, 4), "unit": "regions/s", "cores": n_threads, "kind": "port",
            "sample": "first %d regions of the same synthetic workload, oracle/libotter_oracle.so (scalar C++ WFA + O(N+E) "
                      "consensus), %d threads, %.1f s" % (n_sample, n_threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--regions", type=int, default=10000, help="regions per GPU (config 1: 10000)")
    ap.add_argument("--len-min", type=int, default=1000)
    ap.add_argument("--len-max", type=int, default=5000)
    ap.add_argument("--reads", type=int, default=30)
    ap.add_argument("--err", default="ont")
    ap.add_argument("--realign", action="store_true", help="config 2: -r given, soft-clipped flanks")
    ap.add_argument("--cpu-sample", type=int, default=192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import otter_amd
    from otter_amd import abi, synth, parallel

    ctx = otter_amd.Context(local_rank)
    params = abi.default_params(realign=1 if args.realign else 0)
    # static BED split: rank r owns the r-th contiguous shard of world*regions regions (seeded per shard)
    batch = synth.make_batch(args.regions, len_range=(args.len_min, args.len_max), n_reads=args.reads, err=args.err,
                             realign=args.realign, seed=synth.SEED + rank)
    t_sub = time.perf_counter()
    ctx.assemble_submit(params, batch)       # H2D: inputs are resident in HBM from here on
    submit_ms = (time.perf_counter() - t_sub) * 1000.0     # not part of `value`: reported so that the PCIe-inclusive rate can be derived

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    gathered = {}

    def step():
        ctx.assemble_run()
        if dist is not None:
            # end-of-run gather of the per-region allele records to rank 0 (RCCL over xGMI), straight from the
            # library's device-resident result buffers: GPU -> GPU, one device-to-host copy on rank 0
            try:
                res = ctx.assemble_device_results()
            except Exception:                      # same records through the host (otg_assemble_collect) if wrapping fails
                res = ctx.assemble_collect()
            g = parallel.gather_records(res, dist, rank, world, torch.device("cuda", local_rank))
            if rank == 0:
                gathered["records"] = len(g["alleles"])

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    kstats = []
    for _ in range(args.steps):
        step()
        kstats.append(ctx.assemble_stats().copy())
    sync()
    dt = time.perf_counter() - t0
    st = kstats[-1]
    regions_ok = int(st["n_regions_ok"])
    tt = torch.tensor([dt, float(regions_ok)], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); total_regions = float(tsum[1])
    else:
        total_regions = float(regions_ok)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = dt * 1000.0 / args.steps
    value = total_regions * args.steps / dt
    # dominant kernel: the WFA kernel group (edit or affine) with the larger HIP-event time; algorithmic bytes per launch
    # = Σ(a+b) + 4·W (+ W/2 for the CIGAR-scope aligner), SURVEY.md §8d / DESIGN.md §6
    ek = float(np.mean([s["ms_edit_kernel"] for s in kstats])); el = max(1, int(st["edit_kernel_launches"]))
    ak = float(np.mean([s["ms_affine_kernel"] for s in kstats])); al = max(1, int(st["affine_kernel_launches"]))
    e_bytes = int(st["edit_seq_bytes"]) + 4 * int(st["edit_cells"])
    a_bytes = int(st["affine_seq_bytes"]) + 4 * int(st["affine_cells"]) + (int(st["affine_cells"]) + 1) // 2
    if ek >= ak:
        kname, kbytes, kms, kl = "wfa_edit_kernel", e_bytes, ek, el
    else:
        kname, kbytes, kms, kl = "wfa_affine_kernel", a_bytes, ak, al
    achieved = (kbytes / kl) / (kms / kl * 1e-3) / 1e9 if kms > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(kname)
        except Exception:
            traffic = None
    out = {
        "metric": "regions/sec (otter assemble hot path) on synthetic TR regions",
        "value": round(value, 3), "unit": "regions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32", "data": "synthetic",
        "config": {"workload": "otter assemble hot path, %d regions/GPU x %d-%d bp TR, %dx %s-error reads%s (BASELINE configs[%d])" % (
                       args.regions, args.len_min, args.len_max, args.reads, args.err.upper(), ", -r local re-alignment" if args.realign else "",
                       2 if args.realign else 1),
                   "regions_per_gpu": args.regions, "reads_per_region": args.reads, "parallelism": "static BED shard x%d + RCCL gather" % world,
                   "stage_ms": {k: round(float(st[k]), 2) for k in ("ms_realign", "ms_edit", "ms_cluster", "ms_reassign", "ms_affine", "ms_poa", "ms_total")},
                   "edit_pairs": int(st["edit_tasks"]), "affine_alignments": int(st["affine_tasks"]),
                   "wavefront_cells": int(st["edit_cells"]) + int(st["affine_cells"]),
                   "exp_variant": "glibc-fma" if ctx.exp_variant else "glibc-nofma",
                   "h2d_submit_ms": round(submit_ms, 2), "input_bytes": int(batch["arena"].size + batch["reads"].nbytes + batch["regions"].nbytes)},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": kbytes // kl, "avg_launch_ms": round(kms / kl, 3),
                     "other_kernel": {"edit_ms": round(ek, 2), "affine_ms": round(ak, 2)}},
    }
    if not args.no_cpu_baseline:
        try:
            nth = max(1, min(16, os.cpu_count() or 1))
            out["cpu_baseline"] = cpu_baseline(batch, params, max(args.cpu_sample, nth), nth)
        except Exception as e:  # the oracle is only a reported baseline; never fail the bench line on it
            out["cpu_baseline"] = {"value": None, "unit": "regions/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if gathered:
        out["config"]["gathered_allele_records"] = gathered["records"]
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
