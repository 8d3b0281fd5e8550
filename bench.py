#!/usr/bin/env python3
"""bench.py — regions/sec of the otter assemble hot path on MI355X (contract in the task brief).

A step = one pass of the whole hot path ([local_realignment] -> fill_dist_matrix -> otter_hclust ->
invalid_reassignment -> rapid_consensus) over one batch of synthetic TR regions that is already resident in HBM
(otg_assemble_submit ran before the timed region), INCLUDING the hand-over of the allele records to the host:
otg_assemble_collect at N=1, the RCCL gather to rank 0 + its one device-to-host copy at N>1.

Workloads come from otter_amd.synth.CONFIGS (= BASELINE.json configs): --config 1 (default at every N: 10 000 regions x 1-5 kb,
30x ONT per GPU), --config 2 (the same with -r and soft-clipped divergent flanks), --config 4 (1-10 kb regions, 12 500 per GPU = the
per-GPU shard of the 100 000-region 8-GPU job).  N>1: static contiguous BED split (rank r owns the
r-th shard, weak scaling), no data-path collective, end-of-run gather of the records over RCCL as north_star specifies.

`python bench.py --gpus N` without a launcher starts the N ranks itself: the parent never imports torch or touches the
GPU, it starts N fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set before anything is imported) and
relays rank 0's line; under torchrun (WORLD_SIZE set) it is simply one rank.

Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_summary.json")   # written by scripts/pmc_bench.sh + scripts/pmc_summarize.py


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=(1, 2, 4), help="BASELINE.json configs index (default 1 at every N: weak scaling over identical per-GPU shards)")
    ap.add_argument("--regions", type=int, default=None, help="regions per GPU (default: the config's own count; config 4: 100000/8)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU work per cpu_baseline run (3 runs per kind)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e-regions", type=int, default=4000, help="regions of the file-to-text leg (BED + BAM -> SAM text through otg_assemble_files); 0: skip")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ N-rank launch
def spawn_ranks(n):
    """Parent of a self-launched N-rank run.  Imports nothing that touches the GPU."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stdin=subprocess.DEVNULL))
    out0 = []
    t = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()))
    t.start()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            c = procs[r].poll()
            if c is None:
                continue
            pending.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, c))
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    t.join()
    sys.stdout.write((out0[0] or b"").decode(errors="replace"))
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline(batch, params, seconds, n_threads):
    """The CPU oracle (oracle/libotter_oracle.so, a port of the reference algorithm) on a bounded sample of the same workload, all host
    threads (ctypes releases the GIL; static contiguous split as the reference's thread pool).  Two kinds, three runs each, median:
      port                 — the port as it is (O(N+E) consensus): BASELINE.md §3 baseline B
      reference_consensus  — the same regions with every consensus computed by the REFERENCE'S OWN PPOA (src/anppoa.hpp, quadratic
                             heaviest path :254-288) compiled into oracle/_ref/libotter_ref.so: baseline A.  Neither is the reference
                             binary (WFA2-lib is absent); alignment and clustering are the port in both."""
    import ctypes as C
    import numpy as np
    import oracle_lib
    L = oracle_lib.lib()
    n_regions = len(batch["regions"])

    def run(n_sample):
        bounds = [(i * n_sample // n_threads, (i + 1) * n_sample // n_threads) for i in range(n_threads)]
        done = [0] * n_threads

        def work(i):
            a, b = bounds[i]
            if b > a:
                r = oracle_lib.assemble_batch(params, batch, region_range=(a, b))
                done[i] = int((r["regions"]["n_alleles"][a:b] > 0).sum())
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        [t.start() for t in th]; [t.join() for t in th]
        return sum(done), time.perf_counter() - t0

    def measure(label):
        ok, dt = run(n_threads)                                   # pilot: one region per thread sizes the sample
        n = int(min(n_regions, max(n_threads, n_threads * round(seconds / max(dt, 1e-3)))))
        rates, times = [], []
        for _ in range(3):
            ok, dt = run(n)
            rates.append(ok / dt); times.append(dt)
        return {"value": round(float(np.median(rates)), 4), "unit": "regions/s", "cores": n_threads, "runs": [round(x, 4) for x in rates],
                "sample": "first %d regions of the same synthetic workload, %d threads, %s; 3 runs of %.1f-%.1f s, median" % (
                    n, n_threads, label, min(times), max(times))}

    out = dict(measure("oracle/libotter_oracle.so: scalar C++ WFA + O(N+E) consensus (BASELINE.md baseline B)"), kind="port")
    refp = os.path.join(ROOT, "oracle", "_ref", "libotter_ref.so")
    if os.path.exists(refp):
        R = C.CDLL(refp)
        L.oto_set_poa_hook.argtypes = [C.c_void_p]
        L.oto_set_poa_hook(C.cast(R.ref_poa_consensus_one, C.c_void_p))
        try:
            a = measure("the same port with every consensus through the reference's own PPOA (oracle/_ref/libotter_ref.so, quadratic "
                        "heaviest path of src/anppoa.hpp:254-288; BASELINE.md baseline A)")
            out["reference_consensus"] = dict(a, kind="port + reference PPOA")
        finally:
            L.oto_set_poa_hook(None)
    else:
        out["reference_consensus"] = None
    return out


# ------------------------------------------------------------------------------------------------ file-to-text leg
def e2e_leg(n_regions, threads):
    """BASELINE.json quotes the metric "on synthetic TR BED+BAM": the same path from files — BED + BAM/BAI -> otg_assemble_files (the library's
    dispatcher: BAM ingest on host threads, batches through the GPU hot path, SAM text) — on a fixture of configs[1]'s shape written by
    otter_amd/bamwrite.py.  Ingest-bound on the host; reported beside `value`, never as `value`."""
    import shutil
    import tempfile
    import otter_amd
    from otter_amd import bamwrite
    tmp = tempfile.mkdtemp(prefix="otg_e2e_")
    try:
        t0 = time.perf_counter()
        fx = bamwrite.make_tr_fixture(tmp, n_regions, depth=30, len_range=(1000, 5000), seed=7)
        build_s = time.perf_counter() - t0
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            text, st = otter_amd.assemble_files(fx["bam"], fx["bed"], read_group="s1", batch_regions=max(64, n_regions // 4), offset_l=1, offset_r=1, mapq=10, threads=threads)
            dt = time.perf_counter() - t1
            if best is None or dt < best[0]:
                best = (dt, st, len(text))
        dt, st, nbytes = best
        # the same job on the first half of the BED: the difference of the two walls is what the second half cost once the pipeline was full
        # (a job of a few thousand loci is short against ingest of its first batch and the drain of its last)
        half_bed = os.path.join(tmp, "half.bed")
        with open(fx["bed"]) as f:
            lines = f.readlines()
        with open(half_bed, "w") as f:
            f.writelines(lines[:n_regions // 2])
        half = None
        for _ in range(3):
            t1 = time.perf_counter()
            otter_amd.assemble_files(fx["bam"], half_bed, read_group="s1", batch_regions=max(64, n_regions // 4), offset_l=1, offset_r=1, mapq=10, threads=threads)
            h = time.perf_counter() - t1
            half = h if half is None or h < half else half
        marginal = (n_regions - n_regions // 2) / (dt - half) if dt > half else None
        return {"regions_per_s": round(n_regions / dt, 1), "marginal_regions_per_s": round(marginal, 1) if marginal else None, "half_job_wall_ms": round(half * 1000.0, 1), "regions": n_regions, "reads": int(st["n_reads"]), "alleles": int(st["n_alleles"]), "sam_bytes": nbytes,
                "bam_bytes": os.path.getsize(fx["bam"]), "host_threads": threads, "batch_regions": max(64, n_regions // 4), "wall_ms": round(dt * 1000.0, 1),
                "stage_busy_ms": {"ingest": round(st["ms_ingest"], 1), "hot_path": round(st["ms_hot_path"], 1), "emit": round(st["ms_emit"], 1)},
                "what": "BED file + BAM/BAI -> otg_assemble_files -> SAM text (header + allele records), best of 3; fixture: %d two-allele TR loci x 30 ONT-like reads of "
                        "1-5 kb written by otter_amd/bamwrite.py in %.0f s" % (n_regions, build_s)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d — start it as `python bench.py --gpus N` (it launches the ranks itself) "
                         "or under torchrun with --nproc-per-node equal to --gpus\n" % (args.gpus, world))
        return 2
    import numpy as np
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import otter_amd
    from otter_amd import abi, synth, parallel

    # Same per-GPU workload at every N (weak scaling): the driver derives scaling efficiency from the per-N values, so N = 1 and N > 1 must run the
    # same shape — configs[1], the configuration the metric is quoted on.  `--config 4` selects the 1-10 kb shard of the 8-GPU configs[4] at any N.
    cfg = args.config if args.config is not None else 1
    n_regions = args.regions if args.regions is not None else (synth.CONFIGS[cfg]["n_regions"] // 8 if cfg == 4 else synth.CONFIGS[cfg]["n_regions"])
    ctx = otter_amd.Context(local_rank)
    params = abi.default_params(realign=1 if synth.CONFIGS[cfg].get("realign") else 0)
    # static BED split: rank r owns the r-th contiguous shard of world * n_regions regions (chunk-seeded generator: the shard is the
    # same bytes whatever the world size)
    if n_regions % synth.CHUNK and world > 1:
        sys.stderr.write("bench.py: --regions must be a multiple of %d for N>1\n" % synth.CHUNK)
        return 2
    batch = synth.config_batch(cfg, n_regions, first_chunk=rank * (n_regions // synth.CHUNK) if world > 1 else 0,
                               workers=max(1, min(16, (os.cpu_count() or 1) // max(1, world))))
    t_sub = time.perf_counter()
    ctx.assemble_submit(params, batch)       # H2D: inputs are resident in HBM from here on
    submit_ms = (time.perf_counter() - t_sub) * 1000.0

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    info = {}

    def step():
        ctx.assemble_run()
        if dist is not None:
            # end-of-run gather of the per-region allele records to rank 0 (RCCL over xGMI), straight from the library's
            # device-resident result buffers: GPU -> GPU, one device-to-host copy on rank 0
            t0 = time.perf_counter()
            res = ctx.assemble_device_results()
            g = parallel.gather_records(res, dist, rank, world, torch.device("cuda", local_rank))
            info["gather_ms"] = (time.perf_counter() - t0) * 1000.0
            if rank == 0:
                info["records"] = len(g["alleles"])
                info["gather_bytes"] = int(g["alleles"].nbytes + g["seqs"].nbytes + g["regions"].nbytes)
        else:
            t0 = time.perf_counter()
            res = ctx.assemble_collect()       # allele records + sequences + region results to host memory
            info["collect_ms"] = (time.perf_counter() - t0) * 1000.0
            info["records"] = len(res["alleles"])

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
    world_seen = 1
    if dist is not None:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); total_regions = float(tsum[1])
        world_seen = dist.get_world_size()
    else:
        total_regions = float(regions_ok)
    # host memory -> host memory (SURVEY §8d's kernel-path metric: upload + run + download), one extra untimed-for-`value` pass per rank
    t1 = time.perf_counter()
    ctx.assemble_submit(params, batch); ctx.assemble_run(); ctx.assemble_collect()
    h2h_s = time.perf_counter() - t1
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    ms_per_step = dt * 1000.0 / args.steps
    value = total_regions * args.steps / dt
    # dominant kernel group: the WFA kernel chain (edit or affine) with the larger HIP-event time (events on the library's own stream);
    # algorithmic bytes per launch = Σ(a+b) + 4·W (+ W/2 for the CIGAR-scope aligner), SURVEY.md §8d / DESIGN.md §6
    ek = float(np.mean([s["ms_edit_kernel"] for s in kstats])); el = max(1, int(st["edit_kernel_launches"]))
    ak = float(np.mean([s["ms_affine_kernel"] for s in kstats])); al = max(1, int(st["affine_kernel_launches"]))
    e_bytes = int(st["edit_seq_bytes"]) + 4 * int(st["edit_cells"])
    a_bytes = int(st["affine_seq_bytes"]) + 4 * int(st["affine_cells"]) + (int(st["affine_cells"]) + 1) // 2
    if ek >= ak:
        kname, kbytes, kms, kl = "wfa_edit_kernel", e_bytes, ek, el
    else:
        kname, kbytes, kms, kl = "wfa_affine_kernel", a_bytes, ak, al
    achieved = (kbytes / kl) / (kms / kl * 1e-3) / 1e9 if kms > 0 else 0.0
    workload = synth.config_workload(cfg, n_regions, world)
    # PMC-derived figures are only attached when the committed summary was taken on exactly this workload
    traffic, physical = None, None
    if os.path.exists(PMC_SUMMARY):
        try:
            pm = json.load(open(PMC_SUMMARY)).get("config%d" % cfg)
            if pm and int(pm.get("regions", -1)) == n_regions:
                traffic = pm.get("traffic_bytes_per_launch", {}).get(kname)
                physical = dict(pm.get("physical", {}).get(kname, {}), source=pm.get("source"), workload=pm.get("workload"))
        except Exception as e:
            physical = {"error": "profiles/pmc_summary.json unreadable: %r" % (e,)}
    visited = int(st["affine_visited_cells"]) if "affine_visited_cells" in st.dtype.names else 0
    out = {
        "metric": "regions/sec (otter assemble hot path) on synthetic TR regions",
        "value": round(value, 3), "unit": "regions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32", "data": "synthetic",
        "config": {"workload": workload, "baseline_config": cfg,
                   "regions_per_gpu": n_regions, "reads_per_region": synth.CONFIGS[cfg]["n_reads"],
                   "parallelism": "static BED shard x%d + RCCL gather" % world, "world_size_rccl": world_seen,
                   "timed_region": "otg_assemble_run + " + ("RCCL gather of the records to rank 0 (device buffers) + one D2H" if world > 1 else "otg_assemble_collect (D2H of the records)") + "; inputs resident in HBM",
                   "stage_ms": {k: round(float(st[k]), 2) for k in ("ms_realign", "ms_edit", "ms_cluster", "ms_reassign", "ms_affine", "ms_poa", "ms_total")},
                   "edit_pairs": int(st["edit_tasks"]), "affine_alignments": int(st["affine_tasks"]),
                   "wavefront_cells": int(st["edit_cells"]) + int(st["affine_cells"]),
                   "exp_variant": "glibc-fma" if ctx.exp_variant else "glibc-nofma",
                   "h2d_submit_ms": round(submit_ms, 2), "input_bytes": int(batch["arena"].size + batch["reads"].nbytes + batch["regions"].nbytes),
                   "allele_records": info.get("records"),
                   "host_to_host": {"regions_per_s": round(regions_ok / h2h_s, 2), "ms": round(h2h_s * 1000.0, 2),
                                    "what": "otg_assemble_submit (H2D) + otg_assemble_run + otg_assemble_collect (D2H), one pass on rank 0's shard"}},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": kbytes // kl, "avg_launch_ms": round(kms / kl, 3),
                     "note": "achieved = SURVEY §8(d) algorithmic bytes (every cell of the reference's un-pruned, HBM-resident wavefronts) / "
                             "HIP-event time of the kernel chain: a work-equivalent rate, NOT bytes moved — the kernels keep wavefronts in registers / LDS "
                             "and prune cells; the binding resource is in `physical` (PMC, profiles/)",
                     "physical": physical,
                     "other_kernel": {"edit_ms": round(ek, 2), "affine_ms": round(ak, 2)}},
    }
    if visited:
        out["roofline"]["affine_visited_cells_per_s"] = round(visited / (float(st["ms_affine_kernel"]) * 1e-3), 1) if float(st["ms_affine_kernel"]) > 0 else None
        out["roofline"]["affine_visited_cells"] = visited
    if world > 1:
        out["config"]["gather"] = {"path": "device buffers (otg_assemble_device_results) -> RCCL gather -> one D2H on rank 0",
                                   "bytes": info.get("gather_bytes"), "ms_last_step": round(info.get("gather_ms", 0.0), 2)}
    else:
        out["config"]["collect_ms_last_step"] = round(info.get("collect_ms", 0.0), 2)
    if not args.no_cpu_baseline and world == 1:        # rank 0 at N=1 only
        nth = max(1, min(32, os.cpu_count() or 1))       # the reference caps -t at 32 (src/otter_opts.cpp:93)
        try:
            out["cpu_baseline"] = cpu_baseline(batch, params, args.cpu_seconds, nth)
        except Exception as e:  # the oracle is only a reported baseline; never fail the bench line on it
            out["cpu_baseline"] = {"value": None, "unit": "regions/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if args.e2e_regions > 0 and world == 1:
        try:
            ctx.close()                       # the dispatcher creates its own contexts
            out["e2e"] = e2e_leg(args.e2e_regions, max(1, min(16, os.cpu_count() or 1)))
        except Exception as e:
            out["e2e"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
