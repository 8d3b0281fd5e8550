"""CPU: the N>1 path (static BED shard per rank + gather of allele records to rank 0) with world_size 2 over gloo.
The per-rank compute is stood in by the CPU oracle here (no GPU in this container); sharding and the gather are the
code under test (otter_amd/parallel.py), the same code bench.py runs over RCCL."""
import os
import sys
import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, as_tensors=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib
    from otter_amd import abi, synth, parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = synth.make_batch(9, len_range=(120, 260), n_reads=8, err="hifi", seed=21)
    P = abi.default_params()
    a, b = parallel.shard_bounds(len(batch["regions"]), world, rank)
    full = oracle_lib.assemble_batch(P, batch, region_range=(a, b))
    # what Context.assemble_collect returns for a shard: only the shard's regions
    res = {"regions": full["regions"][a:b].copy(), "alleles": full["alleles"].copy(), "seqs": full["seqs"]}
    res["alleles"]["region"] -= a
    if as_tensors:
        # the form Context.assemble_device_results hands over on a GPU: flat uint8 tensors, sequence arena = exactly
        # the bytes the records point at
        nseq = int(res["alleles"]["seq_len"].astype(np.int64).sum())
        res = {"alleles": torch.from_numpy(res["alleles"].view(np.uint8).reshape(-1).copy()),
               "seqs": torch.from_numpy(np.ascontiguousarray(res["seqs"][:nseq]).copy()),
               "regions": torch.from_numpy(res["regions"].view(np.uint8).reshape(-1).copy())}
    g = parallel.gather_records(res, dist, rank, world, torch.device("cpu"))
    if rank == 0:
        whole = oracle_lib.assemble_batch(P, batch)
        ok = (len(g["alleles"]) == len(whole["alleles"]) and np.array_equal(g["alleles"]["seq_len"], whole["alleles"]["seq_len"])
              and np.array_equal(g["alleles"]["region"], whole["alleles"]["region"])
              and np.array_equal(g["alleles"]["seq_off"], whole["alleles"]["seq_off"])
              and g["seqs"].tobytes() == whole["seqs"][:len(g["seqs"])].tobytes()
              and np.array_equal(g["regions"]["fc"], whole["regions"]["fc"])
              and np.array_equal(g["regions"]["first_allele"], whole["regions"]["first_allele"]))
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def _run(as_tensors, port_off):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + port_off
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, as_tensors)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_two_rank_shard_and_gather():
    _run(False, 0)


def test_two_rank_gather_from_device_style_tensors():
    _run(True, 2000)
