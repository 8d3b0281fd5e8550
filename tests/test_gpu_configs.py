"""BASELINE.json configs[2] (`otter assemble -r`, divergent soft-clipped flanks) and configs[4] (1-10 kb regions, the per-GPU shard
of the 8-GPU job) at full size through the C-ABI, checked the way tests/test_gpu_pipeline.py checks configs[1]: size-independent
properties over every region (two runs agree bit for bit, a shard run alone equals the same regions inside the whole batch,
coverage / label / length invariants) and oracle windows record for record — chosen so that the -r windows hold rescued reads and the
1-10 kb windows hold reads beyond 6 kb and 8 kb (the 4096-diagonal LDS tier and the HBM-row tiers of the gap-affine aligner run inside
the pipeline, src/analignments.cpp:11-60 and :268-280)."""
import threading

import numpy as np
import pytest
from otter_amd import abi, synth

pytestmark = pytest.mark.gpu


def alleles_of(res, lo, hi, base=0):
    out = []
    for r in range(lo, hi):
        g = res["regions"][r - base]
        for a in res["alleles"][int(g["first_allele"]):int(g["first_allele"]) + int(g["n_alleles"])]:
            out.append((r, int(a["label"]), int(a["scov"]), int(a["acov"]), int(a["tcov"]), float(a["se"]), int(a["ic"]),
                        res["seqs"][int(a["seq_off"]):int(a["seq_off"]) + int(a["seq_len"])].tobytes()))
    return out


def full_size_properties(gpu, oracle, b, P, windows, shard, n_reads_max):
    N = len(b["regions"])
    r1 = gpu.assemble(P, b)
    st = gpu.assemble_stats().copy()
    r2 = gpu.assemble(P, b)
    for k in ("regions", "alleles", "labels"):
        assert r1[k].tobytes() == r2[k].tobytes(), k
    nseq = int(r1["alleles"]["seq_len"].astype(np.int64).sum())
    assert r1["seqs"][:nseq].tobytes() == r2["seqs"][:nseq].tobytes()

    part = gpu.assemble(P, b, region_range=shard)
    assert alleles_of(part, shard[0], shard[1], base=shard[0]) == alleles_of(r1, shard[0], shard[1])

    reg, al = r1["regions"], r1["alleles"]
    ok = reg["status"] == 0
    assert ok.sum() > 0.99 * N
    assert ((reg["fc"][ok] >= 1) & (reg["fc"][ok] <= P.max_alleles) & (reg["n_alleles"][ok] == reg["fc"][ok])).all()
    assert ((al["scov"] <= al["acov"]) & (al["acov"] <= al["tcov"]) & (al["tcov"] <= n_reads_max) & (al["scov"] >= 1)).all()
    lens = b["reads"]["seq_len"].astype(np.int64)
    first = b["regions"]["first_read"].astype(np.int64)
    rmin = np.minimum.reduceat(lens, first); rmax = np.maximum.reduceat(lens, first)
    ar = al["region"].astype(np.int64)
    assert ((al["seq_len"] >= rmin[ar] // 2) & (al["seq_len"] <= rmax[ar] * 3 // 2)).all()
    assert np.isin(r1["seqs"][:nseq], np.frombuffer(b"ACGTN", dtype=np.uint8)).all()
    fc_per_read = np.repeat(reg["fc"], b["regions"]["n_reads"])
    assert ((r1["labels"] >= -1) & (r1["labels"] < np.maximum(fc_per_read, 1))).all()

    got = [None] * len(windows)

    def work(i):
        got[i] = oracle.assemble_batch(P, b, region_range=windows[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(windows))]
    [t.start() for t in th]; [t.join() for t in th]
    for (lo, hi), ora in zip(windows, got):
        assert alleles_of(ora, lo, hi) == alleles_of(r1, lo, hi), (lo, hi)
        f0, f1 = int(first[lo]), int(first[hi - 1] + b["regions"]["n_reads"][hi - 1])
        assert np.array_equal(ora["labels"][f0:f1], r1["labels"][f0:f1])
    return r1, st


def windows_with(score, width, count, n):
    """`count` disjoint windows of `width` consecutive regions with the largest window sums of `score` (plus the first and the last
    window of the batch, which exercise the batch edges)."""
    w = np.convolve(score.astype(np.float64), np.ones(width), mode="valid")
    out = [(0, width), (n - width, n)]
    for s in np.argsort(-w):
        s = int(s)
        if all(s + width <= a or s >= bnd for a, bnd in out):
            out.append((s, s + width))
        if len(out) >= count + 2:
            break
    return out


def test_full_size_config2_realign_properties(gpu, oracle):
    """BASELINE configs[2]: 10 000 divergent-flank regions x 30 ONT reads of 1-5 kb with -r (local re-alignment triggered: a quarter of the
    reads carry a soft-clipped flank of 150-400 bp; 60 % of those flanks are the true flank read with ONT errors, and at 7 % error about half of
    them still reach min_sim 0.9 over the 100-bp flank: ~27 % of the clipped reads = 6.8 % of all reads are rescued, measured)."""
    b = synth.config_batch(2)
    N = len(b["regions"])
    assert N == synth.CONFIGS[2]["n_regions"] == 10000
    P = abi.default_params(realign=1)
    # reads that local_realignment rescues, per region: windows are taken where most of them are
    trimmed = gpu.realign_reads(P, b)
    changed = (trimmed["seq_len"] != b["reads"]["seq_len"]) | (trimmed["seq_off"] != b["reads"]["seq_off"])
    assert changed.sum() > 0.05 * len(changed)                     # measured share: 6.8 % of all reads (see the docstring); the floor only says the rescue is live
    assert (trimmed["spanning_l"][changed] == 1).all() and (trimmed["spanning_r"][changed] == 1).all()
    first = b["regions"]["first_read"].astype(np.int64)
    per_region = np.add.reduceat(changed.astype(np.int64), first)
    windows = windows_with(per_region, 6, 2, N)
    assert per_region[windows[2][0]:windows[2][1]].sum() >= 12
    r1, st = full_size_properties(gpu, oracle, b, P, windows, (4000, 4064), 30)
    assert float(st["ms_realign"]) > 0 and int(st["n_regions_ok"]) > 0.99 * N
    # the first 1 000 regions against the committed oracle digests (computed four regions at a time; here inside the 10 000-region batch)
    import digests
    from test_gpu_digests import load_digest
    want = load_digest(2)
    nd = int(want["n_regions"][0])
    assert digests.compare(digests.digest(r1, b, 0, nd), want, "configs[2] inside the full batch")[0] == nd
    # the rescue matters: without -r the same batch gives different alleles somewhere in the windows' regions
    lo, hi = windows[2]
    r0 = gpu.assemble(abi.default_params(), b, region_range=(lo, hi))
    assert alleles_of(r0, lo, hi, base=lo) != alleles_of(r1, lo, hi)


def test_full_size_config4_shard_properties(gpu, oracle):
    """BASELINE configs[4], one GPU's shard of the 8-GPU job: 12 500 regions x 30 ONT reads of 1-10 kb (rank 3's chunks of the 100 000)."""
    n_shard = synth.CONFIGS[4]["n_regions"] // 8
    assert n_shard == 12500
    b = synth.config_batch(4, n_shard, first_chunk=3 * n_shard // synth.CHUNK)
    N = len(b["regions"])
    P = abi.default_params()
    lens = b["reads"]["seq_len"].astype(np.int64)
    first = b["regions"]["first_read"].astype(np.int64)
    rmax = np.maximum.reduceat(lens, first)
    assert rmax.max() > 9500
    windows = windows_with(rmax, 4, 2, N)                           # the longest regions: reads of 9-10 kb
    assert min(rmax[windows[2][0]:windows[2][1]].max(), rmax[windows[3][0]:windows[3][1]].max()) > 8000
    mid = int(np.argmin(np.abs(rmax - 6500)))                       # and a region of ~6.5 kb reads
    mid = min(max(mid, 8), N - 12)
    if all(mid + 2 <= a or mid >= bnd for a, bnd in windows):
        windows.append((mid, mid + 2))
    r1, st = full_size_properties(gpu, oracle, b, P, windows, (6000, 6048), 30)
    assert int(st["n_regions_ok"]) > 0.99 * N
