"""Multi-GPU plumbing of the hot path: static contiguous BED split (the reference's BS::thread_pool
parallelize_loop, src/BS_thread_pool.hpp:183-198, with ranks in place of threads) and the end-of-run gather
of per-region allele records to rank 0 (the single-process analogue is the mutex-guarded stdout section,
src/assemble.cpp:143-149).  One process per GPU; torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X, "gloo" in CPU tests).  There is no data-path collective: regions are independent."""
import numpy as np
from . import abi
from .synth import shard_bounds  # noqa: F401  (re-exported)


def gather_records(res, dist, rank, world, device):
    """All ranks call this with their own `res` (dict from Context.assemble_collect: regions, alleles, seqs).
    Returns on rank 0 a dict with the records of all ranks concatenated in rank (= BED) order, sequence
    offsets and region indices rebased; None on other ranks."""
    import torch
    rec = torch.from_numpy(np.ascontiguousarray(res["alleles"]).view(np.uint8).reshape(-1).copy()).to(device)
    nseq = int(res["alleles"]["seq_len"].astype(np.int64).sum()) if len(res["alleles"]) else 0
    seq = torch.from_numpy(np.ascontiguousarray(res["seqs"][:nseq]).copy()).to(device)
    reg = torch.from_numpy(np.ascontiguousarray(res["regions"]).view(np.uint8).reshape(-1).copy()).to(device)
    sizes = torch.tensor([rec.numel(), seq.numel(), reg.numel()], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros(3, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    mx = torch.stack(all_sizes).max(dim=0).values.tolist()
    out = []
    for t, m in ((rec, mx[0]), (seq, mx[1]), (reg, mx[2])):
        pad = torch.zeros(int(m), dtype=torch.uint8, device=device)
        pad[:t.numel()] = t
        lst = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, lst, dst=0)
        out.append(lst)
    if rank != 0:
        return None
    alleles, seqs, regions = [], [], []
    seq_base = allele_base = region_base = 0
    for r in range(world):
        nrec, nsq, nrg = (int(x) for x in all_sizes[r].tolist())
        a = out[0][r][:nrec].cpu().numpy().view(abi.allele_dt).copy()
        s = out[1][r][:nsq].cpu().numpy().copy()
        g = out[2][r][:nrg].cpu().numpy().view(abi.region_result_dt).copy()
        a["seq_off"] += seq_base
        a["region"] += region_base
        g["first_allele"] += allele_base
        alleles.append(a); seqs.append(s); regions.append(g)
        seq_base += nsq; allele_base += len(a); region_base += len(g)
    return {"alleles": np.concatenate(alleles), "seqs": np.concatenate(seqs) if seqs else np.zeros(0, np.uint8),
            "regions": np.concatenate(regions)}
