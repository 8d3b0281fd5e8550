"""Multi-GPU plumbing of the hot path: static contiguous BED split (the reference's BS::thread_pool
parallelize_loop, src/BS_thread_pool.hpp:183-198, with ranks in place of threads) and the end-of-run gather
of per-region allele records to rank 0 (the single-process analogue is the mutex-guarded stdout section,
src/assemble.cpp:143-149).  One process per GPU; torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X, "gloo" in CPU tests).  There is no data-path collective: regions are independent."""
import numpy as np
from . import abi
from .synth import shard_bounds  # noqa: F401  (re-exported)


_PINNED = {}


def _pinned(slot, nbytes, torch):
    """Grow-only page-locked staging buffers of rank 0 (allocating hundreds of MB of pinned memory per step would cost
    more than the copy)."""
    buf = _PINNED.get(slot)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes + nbytes // 8, 1 << 20), dtype=torch.uint8, pin_memory=True)
        _PINNED[slot] = buf
    return buf[:nbytes]


def gather_records(res, dist, rank, world, device):
    """All ranks call this with their own `res`: either the dict of Context.assemble_collect (host numpy arrays) or
    the dict of Context.assemble_device_results (uint8 tensors already on `device`; the records then travel
    GPU -> GPU and touch the host once, on rank 0).  Returns on rank 0 a dict with the records of all ranks
    concatenated in rank (= BED) order, sequence offsets and region indices rebased; None on other ranks."""
    import torch
    if isinstance(res["alleles"], torch.Tensor):
        rec, seq, reg = res["alleles"], res["seqs"], res["regions"]
        if rec.numel():
            # the sequence arena holds exactly the bytes the allele records point at (compacted on the device)
            pass
    else:
        rec = torch.from_numpy(np.ascontiguousarray(res["alleles"]).view(np.uint8).reshape(-1).copy()).to(device)
        nseq = int(res["alleles"]["seq_len"].astype(np.int64).sum()) if len(res["alleles"]) else 0
        seq = torch.from_numpy(np.ascontiguousarray(res["seqs"][:nseq]).copy()).to(device)
        reg = torch.from_numpy(np.ascontiguousarray(res["regions"]).view(np.uint8).reshape(-1).copy()).to(device)
    sizes = torch.tensor([rec.numel(), seq.numel(), reg.numel()], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros(3, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = torch.stack(all_sizes).cpu()
    mx = all_sizes.max(dim=0).values.tolist()
    out = []
    for t, m in ((rec, mx[0]), (seq, mx[1]), (reg, mx[2])):
        pad = torch.empty(int(m), dtype=torch.uint8, device=device)
        pad[:t.numel()] = t
        lst = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, lst, dst=0)
        out.append(lst)
    if device.type == "cuda":
        # the collectives above run on torch's stream; the library rewrites (or frees) the forwarded result buffers on ITS stream at the
        # next submit / run, so every rank waits here until its sends have left the buffers
        torch.cuda.current_stream(device).synchronize()
    if rank != 0:
        return None
    # rank 0: trim + concatenate on the device, ONE device-to-host copy per array, then rebase on the host
    parts = []
    for j in range(3):
        cat = torch.cat([out[j][r][:int(all_sizes[r][j])] for r in range(world)])
        if cat.is_cuda:
            host = _pinned(j, cat.numel(), torch)
            host.copy_(cat, non_blocking=False)
            cat = host
        parts.append(cat.numpy())
    alleles = parts[0].view(abi.allele_dt).copy()
    seqs = parts[1].copy()
    regions = parts[2].view(abi.region_result_dt).copy()
    seq_base = allele_base = region_base = 0
    ai = gi = 0
    for r in range(world):
        nrec = int(all_sizes[r][0]) // abi.allele_dt.itemsize
        nsq = int(all_sizes[r][1])
        nrg = int(all_sizes[r][2]) // abi.region_result_dt.itemsize
        alleles["seq_off"][ai:ai + nrec] += seq_base
        alleles["region"][ai:ai + nrec] += region_base
        regions["first_allele"][gi:gi + nrg] += allele_base
        ai += nrec; gi += nrg
        seq_base += nsq; allele_base += nrec; region_base += nrg
    return {"alleles": alleles, "seqs": seqs, "regions": regions}
