#!/usr/bin/env python3
"""bench.py — regions/sec of the otter assemble hot path on MI355X (contract in the task brief).

A step = one pass of the whole hot path ([local_realignment] -> fill_dist_matrix -> otter_hclust ->
invalid_reassignment -> rapid_consensus; src/assemble.cpp:71-150) over one batch of synthetic TR regions that is already resident in HBM
(otg_assemble_submit ran before the timed region), INCLUDING the hand-over of the allele records to the host:
otg_assemble_collect at N=1, the RCCL gather to rank 0 + its one device-to-host copy at N>1.

`value` is BASELINE configs[1] (10 000 regions x 1-5 kb, 30x ONT per GPU; weak scaling over identical per-GPU shards at every N).  At N = 1 the
line also carries, under `config`:
  host_to_host   the same batch from host memory to records in host memory (SURVEY §8d's kernel-path metric: submit + run + collect);
  legs           configs[2] (the same with -r and soft-clipped divergent flanks), configs[4] (north_star's shape: 1-10 kb regions, the
                 12 500-region shard one GPU owns of the 100 000-region 8-GPU job) and configs[3] (otter genotype's allele clustering,
                 5 000 regions x 101 alleles), each with value / ms_per_step / stage_ms / roofline (/ cpu_baseline); and
                 configs[1]_adaptive: the timed workload again with both aligners under wfadaptive(10, 50, 1) (otg_params.heuristic — the mode
                 the reference's binary runs in if its WFA2-lib build defaults to it; `cpu_baseline.adaptive` is the oracle in that mode);
  e2e            BED + BAM -> SAM text through the library's dispatcher on 10 000 loci of configs[1]'s shape.
--config N makes another configuration the timed one; --no-legs / --e2e-regions 0 / --no-cpu-baseline switch the extras off.

`python bench.py --gpus N` without a launcher starts the N ranks itself: the parent never imports torch or touches the
GPU, it starts N fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set before anything is imported) and
relays rank 0's line; under torchrun (WORLD_SIZE set) it is simply one rank.  Every batch is generated (worker processes) BEFORE torch is
imported or a context exists: nothing forks from a process that has initialised the GPU.

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
METRIC = "regions/sec (otter assemble) on synthetic TR BED+BAM, 1/2/4/8 MI355X"      # BASELINE.json's metric, verbatim


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=(1, 2, 3, 4), help="BASELINE.json configs index of the timed workload (default 1 at every N)")
    ap.add_argument("--regions", type=int, default=None, help="regions per GPU (default: the config's own count; config 4: 100000/8)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU work per cpu_baseline run (3 runs per kind)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the configs[2] / [3] / [4] / adaptive legs (N = 1 has them all; N > 1 the configs[4] shard)")
    ap.add_argument("--heuristic", choices=("none", "wfadaptive"), default="none",
                    help="aligner mode of the TIMED workload: none = exact (the contract), wfadaptive = WFA2-lib's adaptive reduction (10, 50, 1)")
    ap.add_argument("--leg-steps", type=int, default=3)
    ap.add_argument("--e2e-regions", type=int, default=10000, help="regions of the file-to-text leg (BED + BAM -> SAM text through otg_assemble_files); 0: skip")
    ap.add_argument("--vcf-regions", type=int, default=5000, help="regions of the file-to-VCF leg (50-sample allele BAM + BED + FASTA -> VCF through otg_genotype_files; configs[3]'s size); 0: skip")
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
def cpu_baseline(batch, params, seconds, n_threads, runs=3, with_reference_consensus=True, adaptive_params=None):
    """The CPU oracle (oracle/libotter_oracle.so, a port of the reference algorithm) on a bounded sample of the same workload, all host
    threads (ctypes releases the GIL; static contiguous split as the reference's thread pool).  Two kinds, `runs` runs each, median:
      port                 — the port as it is (O(N+E) consensus): BASELINE.md §3 baseline B
      reference_consensus  — the same regions with every consensus computed by the REFERENCE'S OWN PPOA (src/anppoa.hpp, quadratic
                             heaviest path :254-288) compiled into oracle/_ref/libotter_ref.so: baseline A.  Neither is the reference
                             binary: WFA2-lib is absent from the reference tree, so BASELINE.md §3.2's calibration against one cannot be made."""
    import ctypes as C
    import numpy as np
    import oracle_lib
    L = oracle_lib.lib()
    n_regions = len(batch["regions"])

    cur = [params]

    def run(n_sample):
        bounds = [(i * n_sample // n_threads, (i + 1) * n_sample // n_threads) for i in range(n_threads)]
        done = [0] * n_threads

        def work(i):
            a, b = bounds[i]
            if b > a:
                r = oracle_lib.assemble_batch(cur[0], batch, region_range=(a, b))
                done[i] = int((r["regions"]["n_alleles"][a:b] > 0).sum())
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        [t.start() for t in th]; [t.join() for t in th]
        return sum(done), time.perf_counter() - t0

    def measure(label):
        ok, dt = run(n_threads)                                   # pilot: one region per thread sizes the sample
        n = int(min(n_regions, max(n_threads, n_threads * round(seconds / max(dt, 1e-3)))))
        rates, times = [(ok / dt)], [dt]
        if runs > 1 or n > n_threads:
            rates, times = [], []
            for _ in range(runs):
                ok, dt = run(n)
                rates.append(ok / dt); times.append(dt)
        return {"value": round(float(np.median(rates)), 4), "unit": "regions/s", "cores": n_threads, "runs": [round(x, 4) for x in rates],
                "sample": "first %d regions, %d threads, %s; %d run(s) of %.1f-%.1f s" % (n if (runs > 1 or n > n_threads) else n_threads, n_threads, label, len(rates), min(times), max(times))}

    out = dict(measure("oracle port (scalar C++ WFA, O(N+E) consensus)"), kind="port")
    out["note"] = ("not the reference binary: its aligner (WFA2-lib, an un-vendored submodule) is absent from the image, so BASELINE.md 3.2's "
                   "calibration of the port against a reference build cannot be done here")
    refp = os.path.join(ROOT, "oracle", "_ref", "libotter_ref.so")
    if with_reference_consensus and os.path.exists(refp):
        R = C.CDLL(refp)
        L.oto_set_poa_hook.argtypes = [C.c_void_p]
        L.oto_set_poa_hook(C.cast(R.ref_poa_consensus_one, C.c_void_p))
        try:
            a = measure("port + the reference's own PPOA (quadratic consensus)")
            out["reference_consensus"] = dict(a, kind="port + reference PPOA")
        finally:
            L.oto_set_poa_hook(None)
    if adaptive_params is not None:      # the same port with both aligners under wfadaptive(10, 50, 1): what a reference whose WFA2-lib defaults to it would cost
        cur[0] = adaptive_params
        try:
            out["adaptive"] = dict(measure("oracle port under wfadaptive(10,50,1) in both aligners"), kind="port, adaptive aligners")
        finally:
            cur[0] = params
    return out


# ------------------------------------------------------------------------------------------------ file-to-text leg
FIXTURE_CODE = ("import sys, json, time; sys.path.insert(0, %r); from otter_amd import bamwrite; t0 = time.time(); "
                "fx = bamwrite.make_tr_fixture(sys.argv[1], int(sys.argv[2]), depth=30, len_range=(1000, 5000), seed=7); "
                "json.dump({'bam': fx['bam'], 'bed': fx['bed'], 'build_s': time.time() - t0}, open(sys.argv[1] + '/fixture.json', 'w'))")


GT_FIXTURE_CODE = ("import sys, json, time; sys.path.insert(0, %r); from otter_amd import bamwrite; t0 = time.time(); "
                   "fx = bamwrite.make_genotype_fixture(sys.argv[1], int(sys.argv[2]), n_samples=50, len_range=(1000, 5000), seed=11); "
                   "json.dump({'bam': fx['bam'], 'bed': fx['bed'], 'fasta': fx['fasta'], 'records': fx['n_records'], 'build_s': time.time() - t0}, open(sys.argv[1] + '/fixture.json', 'w'))")


def start_gt_fixture(n_regions):
    """the 50-sample allele BAM of the file-to-VCF leg, written by a child process started before this process touches the GPU"""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="otg_vcf_")
    p = subprocess.Popen([sys.executable, "-c", GT_FIXTURE_CODE % ROOT, tmp, str(n_regions)], stdin=subprocess.DEVNULL)
    return tmp, p


def vcf_leg(tmp, proc, n_regions, threads):
    """BASELINE configs[3] from files: `otter genotype -b regions.bed -r ref.fa alleles.bam` as ONE library call (otg_genotype_files: allele ingest on
    host threads -> anallele_cluster on the GPU -> VCF text; src/genotype.cpp:69-171), on a generated 50-sample x n_regions allele BAM.  The
    reference's own judgement of this configuration is that it is BAM-I/O-bound (SURVEY finding 9): the stage times say where the wall goes."""
    import shutil
    import otter_amd
    try:
        if proc.wait(timeout=1200) != 0:
            raise RuntimeError("fixture writer failed")
        fx = json.load(open(os.path.join(tmp, "fixture.json")))
        best = None
        for _ in range(2):
            t1 = time.perf_counter()
            text, st = otter_amd.genotype_files(fx["bam"], fx["bed"], fasta=fx["fasta"], threads=threads)
            dt = time.perf_counter() - t1
            if best is None or dt < best[0]:
                best = (dt, st, len(text), text.count(b"\n"))
        dt, st, nbytes, nlines = best
        return {"workload": "BASELINE configs[3] from files: %d regions x 50 samples x 2 alleles of 1-5 kb (+ the reference allele), one merged allele BAM" % n_regions,
                "regions_per_s": round(n_regions / dt, 1), "regions": n_regions, "allele_records": int(fx["records"]), "alleles_clustered": int(st["n_alleles"]),
                "vcf_bytes": nbytes, "vcf_lines": nlines, "bam_bytes": os.path.getsize(fx["bam"]), "host_threads": threads, "wall_ms": round(dt * 1000.0, 1),
                "stage_busy_ms": {"ingest": round(st["ms_ingest"], 1), "cluster": round(st["ms_hot_path"], 1), "emit": round(st["ms_emit"], 1)},
                "what": "allele BAM/BAI + BED + FASTA -> otg_genotype_files -> VCF text, best of 2", "fixture_build_s": round(fx["build_s"], 1)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def start_fixture(n_regions):
    """The BED + BAM fixture of the file-to-text leg is written by a child process (started before this process touches the GPU) while the
    resident-batch legs run."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="otg_e2e_")
    p = subprocess.Popen([sys.executable, "-c", FIXTURE_CODE % ROOT, tmp, str(n_regions)], stdin=subprocess.DEVNULL)
    return tmp, p


def e2e_leg(tmp, proc, n_regions, threads):
    """BASELINE.json quotes the metric "on synthetic TR BED+BAM": the same path from files — BED + BAM/BAI -> otg_assemble_files (the library's
    dispatcher: BAM ingest on host threads, batches through the GPU hot path, SAM text) — on a fixture of configs[1]'s shape and size written
    by otter_amd/bamwrite.py.  Reported beside `value`, never as `value`."""
    import shutil
    import otter_amd
    try:
        if proc.wait(timeout=900) != 0:
            raise RuntimeError("fixture writer failed")
        fx = json.load(open(os.path.join(tmp, "fixture.json")))
        batch = 0          # the library's own batch plan
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            text, st = otter_amd.assemble_files(fx["bam"], fx["bed"], read_group="s1", batch_regions=batch, offset_l=1, offset_r=1, mapq=10, threads=threads)
            dt = time.perf_counter() - t1
            if best is None or dt < best[0]:
                best = (dt, st, len(text))
        dt, st, nbytes = best
        # the same job on the first half of the BED: the difference of the two walls is what the second half cost once the pipeline was full
        half_bed = os.path.join(tmp, "half.bed")
        with open(fx["bed"]) as f:
            lines = f.readlines()
        with open(half_bed, "w") as f:
            f.writelines(lines[:n_regions // 2])
        half = None
        for _ in range(2):
            t1 = time.perf_counter()
            otter_amd.assemble_files(fx["bam"], half_bed, read_group="s1", batch_regions=batch, offset_l=1, offset_r=1, mapq=10, threads=threads)
            h = time.perf_counter() - t1
            half = h if half is None or h < half else half
        marginal = (n_regions - n_regions // 2) / (dt - half) if dt > half else None
        # the same job with both aligners under wfadaptive(10, 50, 1) (otg_params.heuristic), best of 2
        adaptive = None
        try:
            from otter_amd import abi
            pa = abi.default_params(heuristic=abi.OTG_HEURISTIC_WFADAPTIVE)
            for _ in range(2):
                t1 = time.perf_counter()
                otter_amd.assemble_files(fx["bam"], fx["bed"], read_group="s1", params=pa, batch_regions=batch, offset_l=1, offset_r=1, mapq=10, threads=threads)
                a = time.perf_counter() - t1
                adaptive = a if adaptive is None or a < adaptive else adaptive
        except Exception:
            adaptive = None
        return {"regions_per_s": round(n_regions / dt, 1), "adaptive_regions_per_s": round(n_regions / adaptive, 1) if adaptive else None, "marginal_regions_per_s": round(marginal, 1) if marginal else None, "regions": n_regions,
                "reads": int(st["n_reads"]), "alleles": int(st["n_alleles"]), "sam_bytes": nbytes, "bam_bytes": os.path.getsize(fx["bam"]),
                "host_threads": threads, "batch_regions": batch, "wall_ms": round(dt * 1000.0, 1), "half_job_wall_ms": round(half * 1000.0, 1),
                "stage_busy_ms": {"ingest": round(st["ms_ingest"], 1), "hot_path": round(st["ms_hot_path"], 1), "emit": round(st["ms_emit"], 1)},
                "what": "BED + BAM/BAI -> otg_assemble_files -> SAM text, best of 3", "fixture_build_s": round(fx["build_s"], 1)}
    finally:
        try:
            otter_amd.assemble_files_release()
        except Exception:
            pass
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ roofline of an assemble workload
def pmc_for(cfg, n_regions, adaptive=False):
    """PMC-derived figures of profiles/pmc_summary.json, only when it was measured on exactly this workload (and aligner mode)."""
    if not os.path.exists(PMC_SUMMARY):
        return None
    try:
        pm = json.load(open(PMC_SUMMARY)).get("config%d%s" % (cfg, "_adaptive" if adaptive else ""))
        if pm and int(pm.get("regions", -1)) == n_regions:
            return pm
    except Exception:
        pass
    return None


VALU_PEAK = os.path.join(ROOT, "profiles", "r04_valu_peak.json")     # scripts/probes/valu_peak.hip on the MI355X


def valu_ceiling():
    """What the chip's vector ALUs sustain, from the committed micro-benchmark: wave64 instructions per second of the 4-cycle class at 8 waves per
    SIMD (packed 16-bit, three-operand, DPP, compares, min / max ...; the plain 32-bit VOP2 class issues 1.85 x as fast and is counted at its own
    cost, see scripts/pmc_summarize.py).  The unit of `achieved` / `peak` is therefore 4-cycle-class instruction slots per second."""
    try:
        d = json.load(open(VALU_PEAK))
        r = [c["wave_insts_per_s"] for c in d["classes"] if c["class"] in ("v_pk_add_u16", "v_pk_max_i16", "v_alignbit_b32", "v_max_i32", "v_cmp_gt_i32 (vcc)") and c["waves_per_simd"] == 8]
        clk = [c["shader_clock_mhz"] for c in d["classes"] if c["waves_per_simd"] == 8]
        return {"slots_per_s": sum(r) / len(r), "clock_mhz": sum(clk) / len(clk), "source": "profiles/r04_valu_peak.json (scripts/probes/valu_peak.hip)"}
    except Exception:
        return None


def roofline(kstats, cfg, n_regions, adaptive=False):
    """The dominant kernel chain by HIP-event time (events on the library's own stream).  These kernels are integer wavefront sweeps whose state lives
    in registers / LDS: what binds them is vector-instruction issue, not HBM.  So the headline object is `bound: "valu"`: `achieved` = the SIMD issue
    slots the chain's vector instructions held per second (PMC instruction counts of exactly this workload, each instruction class at its measured issue
    cost — profiles/pmc_summary.json — divided by the chain's duration measured live here), `peak` = what the micro-benchmark sustains.  SURVEY 8(d)'s
    figure — algorithmic bytes of the REFERENCE's un-pruned wavefronts / time against 8 TB/s, a work-equivalent rate that can exceed 1 — and the
    physical HBM traffic stay beside it under `hbm`."""
    import numpy as np
    st = kstats[-1]
    ek = float(np.mean([s["ms_edit_kernel"] for s in kstats])); el = max(1, int(st["edit_kernel_launches"]))
    ak = float(np.mean([s["ms_affine_kernel"] for s in kstats])); al = max(1, int(st["affine_kernel_launches"]))
    e_bytes = int(st["edit_seq_bytes"]) + 4 * int(st["edit_cells"])
    a_bytes = int(st["affine_seq_bytes"]) + 4 * int(st["affine_cells"]) + (int(st["affine_cells"]) + 1) // 2
    kname, kbytes, kms, kl = ("wfa_edit_kernel", e_bytes, ek, el) if ek >= ak else ("wfa_affine_kernel", a_bytes, ak, al)
    hbm_rate = (kbytes / kl) / (kms / kl * 1e-3) / 1e9 if kms > 0 else 0.0
    visited = int(st["affine_visited_cells"])
    hbm = {"algorithmic_bytes_per_launch": kbytes // kl, "achieved": round(hbm_rate, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_work_equivalent": round(hbm_rate / HBM_PEAK_GBS, 4),
           "note": "SURVEY 8(d): bytes of the reference's un-pruned, HBM-resident wavefronts / time — work-equivalent, not bytes moved (the kernels keep wavefronts in registers / LDS and "
                   + ("evaluate only the cells the adaptive cut leaves)" if adaptive else "visit a fraction of the counted cells)")}
    out = {"bound": "valu", "kernel": kname, "achieved": None, "peak": None, "unit": "G issue slots/s (wave64 instructions at the 4-cycle class's cost)", "frac": None,
           "traffic": None, "avg_launch_ms": round(kms / kl, 3), "chain_ms": {"edit": round(ek, 2), "affine": round(ak, 2)}, "hbm": hbm}
    if kname == "wfa_affine_kernel" and not adaptive:
        impl_min = int(st["affine_seq_bytes"]) + visited + int(st["affine_seq_bytes"]) // 2 + int(st["allele_bytes"])
        hbm["impl_min_bytes"] = impl_min // kl
        out["affine_visited_cells"] = visited
        out["affine_visited_cells_per_s"] = round(visited / (ak * 1e-3), 1) if ak > 0 else None
    ceil = valu_ceiling()
    if ceil:
        out["peak"] = round(ceil["slots_per_s"] / 1e9, 2)
        out["peak_source"] = ceil["source"]
    pm = pmc_for(cfg, n_regions, adaptive)
    if pm:
        ph = pm.get("physical", {}).get(kname, {})
        out["traffic"] = pm.get("traffic_bytes_per_launch", {}).get(kname)
        cyc = ph.get("valu_simd_cycles")                 # SIMD cycles the chain's vector instructions held, all launches of the profiled process
        nl = max(1, int(ph.get("launches", 1)))
        if cyc and ceil and kms > 0:
            slots = cyc / 4.15 / nl                      # per chain launch, in 4-cycle-class slots
            out["achieved"] = round(slots / (kms / kl * 1e-3) / 1e9, 2)
            out["frac"] = round(out["achieved"] / out["peak"], 4)
            out["valu_slots_per_launch"] = round(slots, 1)
        out["binding"] = {"resource": "valu issue", "valu_busy_in_the_profiled_run": ph.get("valu_busy"), "valu_fast_class_share": ph.get("valu_fast_share"), "salu_busy": ph.get("salu_busy"),
                          "wait_any_frac": ph.get("wait_any_frac"), "valu_insts_per_visited_cell": ph.get("valu_insts_per_visited_cell"),
                          "lds_bank_conflict_rate": ph.get("lds_bank_conflict_rate"), "source": pm.get("source")}
        if hbm.get("impl_min_bytes") and out["traffic"]:
            hbm["traffic_over_impl_min"] = round(float(out["traffic"]) / hbm["impl_min_bytes"], 2)
        if out["traffic"] and kms > 0:       # the physical HBM fraction: bytes the counters saw / chain time / peak
            hbm["traffic_frac_of_hbm_peak"] = round(float(out["traffic"]) / (kms / kl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    else:
        out["note"] = "no PMC instruction counts for exactly this workload in profiles/pmc_summary.json (scripts/pmc_bench.sh): achieved / frac not computed"
    return out


def stage_ms(st):
    return {k: round(float(st[k]), 2) for k in ("ms_realign", "ms_edit", "ms_cluster", "ms_reassign", "ms_affine", "ms_poa", "ms_total")}


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
    from otter_amd import abi, synth          # numpy only: nothing here touches the GPU
    # Rehearsal of the N > 1 code path on a box with ONE GPU (OTG_BENCH_REHEARSAL=1, optionally OTG_BENCH_REHEARSAL_REGIONS=<multiple of 250>): every
    # rank opens its context on device 0, the process group is gloo, the records are gathered from host memory.  Not a measurement — it exists so
    # that the control flow of the multi-rank run (shards, legs, collectives, who prints) has been executed before a real 8-GPU node runs it.
    rehearse = os.environ.get("OTG_BENCH_REHEARSAL") == "1" and world > 1
    reh_regions = int(os.environ.get("OTG_BENCH_REHEARSAL_REGIONS", "0")) if rehearse else 0

    # ---- every synthetic input first, by worker processes, before torch / HIP are initialised in this process
    cfg = args.config if args.config is not None else 1
    n_regions = args.regions if args.regions is not None else (synth.CONFIGS[cfg]["n_regions"] // 8 if cfg == 4 else synth.CONFIGS[cfg]["n_regions"])
    if reh_regions and args.regions is None:
        n_regions = reh_regions
    if n_regions % synth.CHUNK and world > 1:
        sys.stderr.write("bench.py: --regions must be a multiple of %d for N>1\n" % synth.CHUNK)
        return 2
    workers = max(1, min(16, (os.cpu_count() or 1) // max(1, world)))
    t_gen = time.perf_counter()
    # static BED split: rank r owns the r-th contiguous shard of world * n_regions regions (chunk-seeded generator: the shard is the
    # same bytes whatever the world size)
    batch = synth.config_batch(cfg, n_regions, first_chunk=rank * (n_regions // synth.CHUNK) if world > 1 else 0, workers=workers)
    legs_in = {}
    want_legs = world == 1 and not args.no_legs and args.config is None and args.regions is None
    if want_legs:
        legs_in[2] = synth.config_batch(2, workers=workers)
        legs_in[4] = synth.config_batch(4, synth.CONFIGS[4]["n_regions"] // 8, workers=workers)
        legs_in[3] = synth.config_batch(3, workers=workers)
    # N > 1: north_star's own shape rides along — configs[4], the 100 000-region job of 1-10 kb loci, 12 500 regions per GPU (rank r's contiguous shard)
    want_leg4_multi = world > 1 and not args.no_legs and args.config is None and args.regions is None
    if want_leg4_multi:
        n4 = reh_regions if reh_regions else synth.CONFIGS[4]["n_regions"] // 8
        legs_in[4] = synth.config_batch(4, n4, first_chunk=rank * (n4 // synth.CHUNK), workers=workers)
    gen_s = time.perf_counter() - t_gen
    fixture = start_fixture(args.e2e_regions) if (args.e2e_regions > 0 and world == 1) else None
    gt_fixture = start_gt_fixture(args.vcf_regions) if (args.vcf_regions > 0 and want_legs) else None

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import otter_amd
    from otter_amd import parallel
    ctx = otter_amd.Context(local_rank)
    tdev = torch.device("cpu") if rehearse else torch.device("cuda", local_rank)      # where the timing tensors and the gathered records live
    nth = max(1, min(32, os.cpu_count() or 1))       # the reference caps -t at 32 (src/otter_opts.cpp:93)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- configs[3]: otter genotype's allele clustering (operator-level entry: host buffers in, host buffers out)
    def genotype_run(gb, steps, warmup):
        P = abi.default_params()
        ga = (gb["arena"], gb["seq_off"], gb["seq_len"], np.ascontiguousarray(gb["first_allele"][:-1]), gb["n_alleles"])
        for _ in range(warmup):
            ctx.genotype_cluster_batch(P, *ga)
        kms = []
        t0 = time.perf_counter()
        for _ in range(steps):
            res = ctx.genotype_cluster_batch(P, *ga)
            kms.append(ctx.last_kernel_ms())
        dt = time.perf_counter() - t0
        nreg = len(gb["n_alleles"])
        nbytes = int(gb["seq_len"].astype(np.int64).sum()) + 65 * 8 * len(gb["seq_len"])        # allele bytes + one 65-bin usage vector per allele
        k = float(np.mean(kms))
        achieved = nbytes / (k * 1e-3) / 1e9 if k > 0 else 0.0
        out = {"value": round(nreg * steps / dt, 2), "unit": "regions/s", "ms_per_step": round(dt * 1000.0 / steps, 3), "steps": steps,
               "timed_region": "otg_genotype_cluster_batch: allele sequences in host memory -> genotypes in host memory (H2D + kernel + D2H)",
               "kernel_ms": round(k, 3), "genotypes": int(res[4].sum()), "input_bytes": int(gb["arena"].size),
               "roofline": {"bound": "hbm", "kernel": "genotype_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": round(k, 3)}}
        return out, P, ga

    def genotype_cpu(P, ga):
        import oracle_lib
        n = min(len(ga[4]), 1000)
        first, na = ga[3][:n], ga[4][:n]
        t0 = time.perf_counter()
        oracle_lib.genotype_cluster_batch(P, ga[0], ga[1], ga[2], first, na)
        dt = time.perf_counter() - t0
        return {"value": round(n / dt, 2), "unit": "regions/s", "cores": 1, "kind": "port", "sample": "first %d regions, 1 thread, oracle anallele_cluster, %.1f s" % (n, dt)}

    if cfg == 3:          # --config 3: the genotype leg as the timed workload
        out3, P3, ga3 = genotype_run(batch, args.steps, args.warmup)
        line = {"metric": "regions/sec (otter genotype allele clustering) on synthetic allele sets", "value": out3["value"], "unit": "regions/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": out3["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic", "config": dict(workload=synth.config_workload(3, n_regions, world), **{k: v for k, v in out3.items() if k not in ("roofline", "value", "unit", "ms_per_step", "steps")}),
                "roofline": out3["roofline"]}
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = genotype_cpu(P3, ga3)
        if rank == 0:
            print(json.dumps(line), flush=True)
        return 0

    # ---- an assemble workload resident in HBM: K timed steps (+ the host-to-host passes)
    def assemble_run(c, b, steps, warmup, h2h_passes, heuristic="none"):
        params = abi.default_params(realign=1 if synth.CONFIGS[c].get("realign") else 0,
                                    heuristic=abi.OTG_HEURISTIC_WFADAPTIVE if heuristic == "wfadaptive" else abi.OTG_HEURISTIC_NONE)
        t_sub = time.perf_counter()
        ctx.assemble_submit(params, b)       # H2D: inputs are resident in HBM from here on
        submit_ms = (time.perf_counter() - t_sub) * 1000.0
        info = {}

        def step():
            ctx.assemble_run()
            if dist is not None:
                # end-of-run gather of the per-region allele records to rank 0 (RCCL over xGMI), straight from the library's
                # device-resident result buffers: GPU -> GPU, one device-to-host copy on rank 0
                t0 = time.perf_counter()
                res = ctx.assemble_collect() if rehearse else ctx.assemble_device_results()
                g = parallel.gather_records(res, dist, rank, world, tdev)
                info["gather_ms"] = (time.perf_counter() - t0) * 1000.0
                if rank == 0:
                    info["records"] = len(g["alleles"])
                    info["gather_bytes"] = int(g["alleles"].nbytes + g["seqs"].nbytes + g["regions"].nbytes)
            else:
                t0 = time.perf_counter()
                res = ctx.assemble_collect()       # allele records + sequences + region results to host memory
                info["collect_ms"] = (time.perf_counter() - t0) * 1000.0
                info["records"] = len(res["alleles"])

        for _ in range(warmup):
            step()
        sync()
        t0 = time.perf_counter()
        kstats = []
        for _ in range(steps):
            step()
            kstats.append(ctx.assemble_stats().copy())
        sync()
        dt = time.perf_counter() - t0
        # host memory -> host memory (upload + run + download), untimed for `value`
        h2h = None
        if h2h_passes > 0:
            t1 = time.perf_counter()
            for _ in range(h2h_passes):
                ctx.assemble_submit(params, b); ctx.assemble_run(); ctx.assemble_collect()
            h2h = (time.perf_counter() - t1) / h2h_passes
        return {"dt": dt, "kstats": kstats, "info": info, "submit_ms": submit_ms, "h2h_s": h2h, "params": params}

    r = assemble_run(cfg, batch, args.steps, args.warmup, min(3, max(1, args.steps)), args.heuristic)
    st = r["kstats"][-1]
    regions_ok = int(st["n_regions_ok"])
    dt = r["dt"]
    tt = torch.tensor([dt, float(regions_ok)], dtype=torch.float64, device=tdev)
    world_seen = 1
    if dist is not None:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); total_regions = float(tsum[1])
        world_seen = dist.get_world_size()
    else:
        total_regions = float(regions_ok)
    leg4_multi = None
    if want_leg4_multi:
        lr = assemble_run(4, legs_in[4], args.leg_steps, 1, 0)
        ls = lr["kstats"][-1]
        t4 = torch.tensor([lr["dt"], float(int(ls["n_regions_ok"]))], dtype=torch.float64, device=tdev)
        t4max = t4.clone(); dist.all_reduce(t4max, op=dist.ReduceOp.MAX)
        t4sum = t4.clone(); dist.all_reduce(t4sum, op=dist.ReduceOp.SUM)
        dt4 = float(t4max[0])
        leg4_multi = {"workload": synth.config_workload(4, len(legs_in[4]["regions"]), world), "value": round(float(t4sum[1]) * args.leg_steps / dt4, 2), "unit": "regions/s",
                      "n_gpus": world, "ms_per_step": round(dt4 * 1000.0 / args.leg_steps, 2), "steps": args.leg_steps, "scaling": "weak",
                      "timed_region": "otg_assemble_run + RCCL gather to rank 0 + one D2H; barrier on both sides, max over ranks", "stage_ms_rank0": stage_ms(ls),
                      "allele_records": lr["info"].get("records"), "gather_bytes": lr["info"].get("gather_bytes")}
        legs_in[4] = None
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    info = r["info"]
    out = {
        "metric": METRIC, "value": round(total_regions * args.steps / dt, 3), "unit": "regions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1000.0 / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i16", "data": "synthetic",
        "config": {"workload": synth.config_workload(cfg, n_regions, world), "baseline_config": cfg,
                   "regions_per_gpu": n_regions, "reads_per_region": synth.CONFIGS[cfg]["n_reads"],
                   "parallelism": "static BED shard x%d + RCCL gather" % world, "world_size_rccl": world_seen,
                   **({"rehearsal": "OTG_BENCH_REHEARSAL=1: all ranks on device 0, gloo, host gather — control-flow check, NOT a measurement"} if rehearse else {}),
                   "aligner_heuristic": "none (exact)" if args.heuristic == "none" else "wfadaptive(10,50,1)",
                   "timed_region": "otg_assemble_run + " + ("RCCL gather to rank 0 + one D2H" if world > 1 else "otg_assemble_collect (D2H of the records)") + "; inputs resident in HBM",
                   "stage_ms": stage_ms(st),
                   "edit_pairs": int(st["edit_tasks"]), "affine_alignments": int(st["affine_tasks"]),
                   "wavefront_cells": int(st["edit_cells"]) + int(st["affine_cells"]),
                   "exp_variant": "glibc-fma" if ctx.exp_variant else "glibc-nofma",
                   "h2d_submit_ms": round(r["submit_ms"], 2), "input_bytes": int(batch["arena"].size + batch["reads"].nbytes + batch["regions"].nbytes),
                   "allele_records": info.get("records"), "synthetic_input_generation_s": round(gen_s, 1),
                   "host_to_host": {"regions_per_s": round(regions_ok / r["h2h_s"], 2), "ms": round(r["h2h_s"] * 1000.0, 2),
                                    "what": "SURVEY 8(d): otg_assemble_submit (H2D) + run + collect (D2H), mean of %d passes" % min(3, max(1, args.steps))}},
        "roofline": roofline(r["kstats"], cfg, n_regions, adaptive=args.heuristic != "none"),
    }
    if leg4_multi is not None:
        out["config"]["legs"] = {"configs[4]": leg4_multi}
    if world > 1:
        out["config"]["gather"] = {"path": "device buffers -> RCCL gather -> one D2H on rank 0", "bytes": info.get("gather_bytes"), "ms_last_step": round(info.get("gather_ms", 0.0), 2)}
    else:
        out["config"]["collect_ms_last_step"] = round(info.get("collect_ms", 0.0), 2)
    if not args.no_cpu_baseline and world == 1:        # rank 0 at N=1 only
        try:
            pa = abi.default_params(realign=r["params"].realign, heuristic=abi.OTG_HEURISTIC_WFADAPTIVE)
            out["cpu_baseline"] = cpu_baseline(batch, r["params"], args.cpu_seconds, nth, adaptive_params=pa if args.heuristic == "none" else None)
        except Exception as e:  # the oracle is only a reported baseline; never fail the bench line on it
            out["cpu_baseline"] = {"value": None, "unit": "regions/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    # ---- the other BASELINE configurations (N = 1): same timed region, fewer steps
    if want_legs:
        legs = {}
        try:         # the timed workload again under the adaptive heuristic
            lr = assemble_run(cfg, batch, args.leg_steps, 1, 1, "wfadaptive" if args.heuristic == "none" else "none")
            ls = lr["kstats"][-1]
            legs["configs[%d]_%s" % (cfg, "adaptive" if args.heuristic == "none" else "exact")] = {
                "workload": synth.config_workload(cfg, n_regions, 1), "aligner_heuristic": "wfadaptive(10,50,1)" if args.heuristic == "none" else "none (exact)",
                "value": round(int(ls["n_regions_ok"]) * args.leg_steps / lr["dt"], 2), "unit": "regions/s",
                "ms_per_step": round(lr["dt"] * 1000.0 / args.leg_steps, 2), "steps": args.leg_steps, "stage_ms": stage_ms(ls),
                "host_to_host_regions_per_s": round(int(ls["n_regions_ok"]) / lr["h2h_s"], 2), "allele_records": lr["info"].get("records"),
                "edit_pairs": int(ls["edit_tasks"]), "affine_alignments": int(ls["affine_tasks"]), "wavefront_cells": int(ls["edit_cells"]) + int(ls["affine_cells"]),
                "roofline": roofline(lr["kstats"], cfg, n_regions, adaptive=args.heuristic == "none")}
        except Exception as e:
            legs["configs[%d]_adaptive" % cfg] = {"error": repr(e)}
        for c in (2, 4):
            try:
                b = legs_in[c]
                nr = len(b["regions"])
                lr = assemble_run(c, b, args.leg_steps, 1, 1)
                ls = lr["kstats"][-1]
                leg = {"workload": synth.config_workload(c, nr, 1), "value": round(int(ls["n_regions_ok"]) * args.leg_steps / lr["dt"], 2), "unit": "regions/s",
                       "ms_per_step": round(lr["dt"] * 1000.0 / args.leg_steps, 2), "steps": args.leg_steps, "stage_ms": stage_ms(ls),
                       "host_to_host_regions_per_s": round(int(ls["n_regions_ok"]) / lr["h2h_s"], 2), "allele_records": lr["info"].get("records"),
                       "roofline": roofline(lr["kstats"], c, nr)}
                if c == 4 and not args.no_cpu_baseline:
                    leg["cpu_baseline"] = cpu_baseline(b, lr["params"], 0.0, nth, runs=1, with_reference_consensus=False)
                legs["configs[%d]" % c] = leg
            except Exception as e:
                legs["configs[%d]" % c] = {"error": repr(e)}
            legs_in[c] = None
        try:
            g3, P3, ga3 = genotype_run(legs_in[3], args.leg_steps, 1)
            g3["workload"] = synth.config_workload(3, len(legs_in[3]["n_alleles"]), 1)
            if not args.no_cpu_baseline:
                g3["cpu_baseline"] = genotype_cpu(P3, ga3)
            legs["configs[3]"] = g3
        except Exception as e:
            legs["configs[3]"] = {"error": repr(e)}
        out["config"]["legs"] = legs
    if fixture is not None:
        try:
            ctx.close()                       # the dispatcher creates its own contexts
            out["config"]["e2e"] = e2e_leg(fixture[0], fixture[1], args.e2e_regions, max(1, min(16, os.cpu_count() or 1)))
        except Exception as e:
            out["config"]["e2e"] = {"error": repr(e)}
    if gt_fixture is not None:
        try:
            ctx.close()
            out["config"]["legs"]["configs[3]_files_to_vcf"] = vcf_leg(gt_fixture[0], gt_fixture[1], args.vcf_regions, max(1, min(16, os.cpu_count() or 1)))
        except Exception as e:
            out["config"]["legs"]["configs[3]_files_to_vcf"] = {"error": repr(e)}
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
