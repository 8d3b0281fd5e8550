"""GPU parity of the whole hot path (region loop body of assemble_process, src/assemble.cpp:71-150):
HIP pipeline through the C-ABI vs the CPU oracle on the same seeded region batches.
Integer fields and sequences bit-exact; `se` within 1e-6 (north_star tolerance; in practice identical)."""
import numpy as np
import pytest
from otter_amd import abi, synth

pytestmark = pytest.mark.gpu


def compare(gpu_res, ora, batch):
    gr, orr = gpu_res["regions"], ora["regions"]
    for f in ("status", "ic", "fc", "n_valid", "n_alleles", "first_allele"):
        assert np.array_equal(gr[f], orr[f]), f
    assert np.array_equal(gpu_res["labels"], ora["labels"])
    ga, oa = gpu_res["alleles"], ora["alleles"]
    assert len(ga) == len(oa)
    for f in ("seq_len", "scov", "acov", "tcov", "ic", "ps", "hp", "region", "label"):
        assert np.array_equal(ga[f], oa[f]), f
    assert np.allclose(ga["se"], oa["se"], rtol=0, atol=1e-6)
    for i in range(len(ga)):
        gs = gpu_res["seqs"][int(ga[i]["seq_off"]):int(ga[i]["seq_off"]) + int(ga[i]["seq_len"])].tobytes()
        os_ = ora["seqs"][int(oa[i]["seq_off"]):int(oa[i]["seq_off"]) + int(oa[i]["seq_len"])].tobytes()
        assert gs == os_, "allele %d sequence differs" % i


def run_both(gpu, oracle, batch, **kw):
    P = abi.default_params(**kw)
    ora = oracle.assemble_batch(P, batch)
    res = gpu.assemble(P, batch)
    compare(res, ora, batch)
    st = gpu.assemble_stats()
    os_ = ora["stats"][0]
    for f in ("edit_tasks", "edit_cells", "edit_seq_bytes", "affine_tasks", "affine_cells", "affine_seq_bytes", "allele_bytes", "algorithmic_bytes"):
        assert int(st[f]) == int(os_[f]), f
    return res, st


def test_config0_hifi_500bp(gpu, oracle):
    """BASELINE config 0: 100 regions x 500 bp, 10x HiFi spanning reads."""
    b = synth.make_batch(**synth.CONFIGS[0])
    res, st = run_both(gpu, oracle, b)
    assert int(st["n_regions_ok"]) == 100


def test_hifi_with_partial_reads(gpu, oracle):
    b = synth.make_batch(60, len_range=(200, 900), n_reads=14, err="hifi", frac_partial=0.3, seed=3, reads_range=(1, 20))
    run_both(gpu, oracle, b)


def test_ont_kb(gpu, oracle):
    """config-1-like slice: 1-3 kb ONT, 30 reads."""
    b = synth.make_batch(24, len_range=(1000, 3000), n_reads=30, err="ont", seed=4)
    run_both(gpu, oracle, b)


def test_realign_flanks(gpu, oracle):
    """config-2-like slice: -r given, soft-clipped flanks rescued by local re-alignment."""
    b = synth.make_batch(30, len_range=(300, 900), n_reads=16, err="hifi", realign=True, seed=6)
    res, st = run_both(gpu, oracle, b, realign=1)
    assert int(st["affine_tasks"]) > 0


def test_haps_mode(gpu, oracle):
    b = synth.make_batch(30, len_range=(300, 800), n_reads=12, err="hifi", haps=True, frac_partial=0.2, seed=8)
    # drop the tags of a third of the reads: they become "invalid" and are re-assigned by similarity
    rng = np.random.default_rng(8)
    drop = rng.random(len(b["reads"])) < 0.33
    b["reads"]["ps"][drop] = -1
    b["reads"]["hp"][drop] = -1
    run_both(gpu, oracle, b, ignore_haps=0)


def test_edge_regions(gpu, oracle):
    """Empty region, no spanning reads, too many reads (max_cov), single read, two reads, max_alleles 1 and 0."""
    b = synth.make_batch(8, len_range=(150, 300), n_reads=9, err="hifi", seed=9, frac_partial=0.1)
    reads, regions = b["reads"].copy(), b["regions"].copy()
    regions[0]["n_reads"] = 0
    r1 = regions[1]
    reads["spanning_r"][r1["first_read"]:r1["first_read"] + r1["n_reads"]] = 0
    regions[2]["n_reads"] = 1
    regions[3]["n_reads"] = 2
    b2 = dict(b, reads=reads, regions=regions)
    run_both(gpu, oracle, b2)
    run_both(gpu, oracle, b2, max_cov=5)
    run_both(gpu, oracle, b2, max_alleles=1)
    run_both(gpu, oracle, b2, max_alleles=0)
    run_both(gpu, oracle, b2, max_alleles=3)


def test_batch_composition_independence(gpu, oracle):
    """Results do not depend on how regions are batched (static BED split): shards == whole."""
    b = synth.make_batch(20, len_range=(200, 600), n_reads=10, err="hifi", seed=10)
    P = abi.default_params()
    whole = gpu.assemble(P, b)
    parts = [gpu.assemble(P, b, region_range=synth.shard_bounds(20, 3, r)) for r in range(3)]
    seqs_whole = [whole["seqs"][int(a["seq_off"]):int(a["seq_off"]) + int(a["seq_len"])].tobytes() for a in whole["alleles"]]
    seqs_parts = []
    for p in parts:
        seqs_parts += [p["seqs"][int(a["seq_off"]):int(a["seq_off"]) + int(a["seq_len"])].tobytes() for a in p["alleles"]]
    assert seqs_whole == seqs_parts
    assert np.array_equal(np.concatenate([p["regions"]["fc"] for p in parts]), whole["regions"]["fc"])


def test_device_results_gather_matches_collect(gpu):
    """The multi-GPU path forwards the library's device-resident result buffers (otg_assemble_device_results) through
    parallel.gather_records; with one rank over RCCL it must reproduce what otg_assemble_collect copies to the host."""
    import os
    import torch
    import torch.distributed as dist
    from otter_amd import parallel
    batch = synth.make_batch(12, len_range=(200, 600), n_reads=10, err="ont", seed=33)
    P = abi.default_params()
    gpu.assemble_submit(P, batch)
    gpu.assemble_run()
    host = gpu.assemble_collect()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dev = gpu.assemble_device_results()
        g = parallel.gather_records(dev, dist, 0, 1, torch.device("cuda", 0))
    finally:
        dist.destroy_process_group()
    assert len(g["alleles"]) == len(host["alleles"])
    assert g["alleles"].tobytes() == host["alleles"].tobytes()
    assert g["regions"].tobytes() == host["regions"].tobytes()
    n = len(g["seqs"])
    assert g["seqs"].tobytes() == host["seqs"][:n].tobytes()


def test_emitted_text_matches_oracle(gpu, oracle):
    """End of the path: the SAM / FASTA text of the GPU pipeline's allele records (otg_emit_alleles) against the oracle's
    emit of the oracle's records — the wire format the reference's parity diff is taken on."""
    import otter_amd
    batch = synth.make_batch(24, len_range=(300, 900), n_reads=14, err="ont", seed=35)
    P = abi.default_params()
    ora = oracle.assemble_batch(P, batch)
    res = gpu.assemble(P, batch)
    beds, carena = abi.make_beds([("chr%d" % (1 + i % 5), 10_000 * i + 7, 10_000 * i + 507) for i in range(len(batch["regions"]))])
    for rg, fa in (("", False), ("s1", False), ("s1", True)):
        assert otter_amd.emit_alleles(beds, carena, res, rg, fa) == oracle.emit_alleles(beds, carena, ora, rg, fa)


def test_realign_only_reads(gpu, oracle):
    """`--reads-only -r`: local_realignment alone (otg_assemble_realign + otg_assemble_collect_reads) against the oracle's restatement:
    rescued reads are trimmed at the rescued end and flagged spanning, every other descriptor is untouched; the emitted read records
    (otg_emit_reads on the collected descriptors) are identical."""
    import otter_amd
    from otter_amd import synth
    b = synth.make_batch(40, len_range=(300, 900), n_reads=16, err="hifi", realign=True, seed=8)
    P = abi.default_params(realign=1)
    got = gpu.realign_reads(P, b)
    exp = oracle.realign_batch(P, b)
    for f in ("seq_off", "seq_len", "spanning_l", "spanning_r", "ps", "hp", "ccoord_first", "ccoord_second"):
        assert np.array_equal(got[f], exp[f]), f
    changed = (got["seq_len"] != b["reads"]["seq_len"]) | (got["seq_off"] != b["reads"]["seq_off"])
    assert changed.sum() > 10 and (got["spanning_l"][changed] == 1).all() and (got["spanning_r"][changed] == 1).all()
    regions = [("chr1", 1000 * i, 1000 * i + 500) for i in range(len(b["regions"]))]
    beds, carena = abi.make_beds(regions)
    t_gpu = otter_amd.emit_reads(beds, carena, {**b, "reads": got}, read_group="rg", fasta=False, max_cov=200)
    t_ora = otter_amd.emit_reads(beds, carena, {**b, "reads": exp}, read_group="rg", fasta=False, max_cov=200)
    assert t_gpu == t_ora and t_gpu.count(b"\n") == len(got)
    # without -r nothing moves
    same = gpu.realign_reads(abi.default_params(), b)
    assert np.array_equal(same["seq_len"], b["reads"]["seq_len"]) and np.array_equal(same["spanning_l"], b["reads"]["spanning_l"])


def test_full_size_config1_properties(gpu, oracle):
    """BASELINE configs[1] at full size (10 000 regions x 30 ONT reads of 1-5 kb, the bench workload), checked through what
    does not need the oracle on every region: two runs agree bit for bit; a shard run alone equals the same regions inside the whole
    batch; coverage / label / length invariants hold for every region; and three 16-region windows at the start, middle and end
    equal the oracle record for record.  The batch is synth.config_batch(1) — the very bytes bench.py times — and its first 1 250 regions
    must equal the committed oracle digests (tests/golden/digest_c1.npz) although they are computed inside the 10 000-region batch here
    and were computed four regions at a time by the oracle: results do not depend on batch composition."""
    import threading
    import digests
    from test_gpu_digests import load_digest
    N = 10000
    b = synth.config_batch(1)
    assert len(b["regions"]) == N == synth.CONFIGS[1]["n_regions"]
    P = abi.default_params()
    r1 = gpu.assemble(P, b)
    want = load_digest(1)
    nd = int(want["n_regions"][0])
    assert digests.compare(digests.digest(r1, b, 0, nd), want, "configs[1] inside the full batch")[0] == nd
    r2 = gpu.assemble(P, b)
    for k in ("regions", "alleles", "labels"):
        assert r1[k].tobytes() == r2[k].tobytes(), k
    nseq = int(r1["alleles"]["seq_len"].astype(np.int64).sum())
    assert r1["seqs"][:nseq].tobytes() == r2["seqs"][:nseq].tobytes()

    def alleles_of(res, lo, hi, base=0):
        out = []
        for r in range(lo, hi):
            g = res["regions"][r - base]
            for a in res["alleles"][int(g["first_allele"]):int(g["first_allele"]) + int(g["n_alleles"])]:
                out.append((r, int(a["label"]), int(a["scov"]), int(a["acov"]), int(a["tcov"]), float(a["se"]), int(a["ic"]),
                            res["seqs"][int(a["seq_off"]):int(a["seq_off"]) + int(a["seq_len"])].tobytes()))
        return out
    part = gpu.assemble(P, b, region_range=(4000, 4064))
    assert alleles_of(part, 4000, 4064, base=4000) == alleles_of(r1, 4000, 4064)       # a shard's result arrays start at its first region

    reg, al = r1["regions"], r1["alleles"]
    ok = reg["status"] == 0
    assert ok.sum() > 0.99 * N
    assert ((reg["fc"][ok] >= 1) & (reg["fc"][ok] <= P.max_alleles) & (reg["n_alleles"][ok] == reg["fc"][ok])).all()
    assert ((al["scov"] <= al["acov"]) & (al["acov"] <= al["tcov"]) & (al["tcov"] <= 30) & (al["scov"] >= 1)).all()
    lens = b["reads"]["seq_len"].astype(np.int64)
    first = b["regions"]["first_read"].astype(np.int64)
    rmin = np.minimum.reduceat(lens, first); rmax = np.maximum.reduceat(lens, first)
    ar = al["region"].astype(np.int64)
    assert ((al["seq_len"] >= rmin[ar] // 2) & (al["seq_len"] <= rmax[ar] * 3 // 2)).all()
    assert np.isin(r1["seqs"][:nseq], np.frombuffer(b"ACGTN", dtype=np.uint8)).all()
    fc_per_read = np.repeat(reg["fc"], b["regions"]["n_reads"])
    assert ((r1["labels"] >= -1) & (r1["labels"] < np.maximum(fc_per_read, 1))).all()

    windows = [(0, 16), (5000, 5016), (N - 16, N)]
    got = [None] * len(windows)

    def work(i):
        got[i] = oracle.assemble_batch(P, b, region_range=windows[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(windows))]
    [t.start() for t in th]; [t.join() for t in th]
    for (lo, hi), ora in zip(windows, got):
        assert alleles_of(ora, lo, hi) == alleles_of(r1, lo, hi), (lo, hi)
        f0, f1 = int(first[lo]), int(first[hi - 1] + b["regions"]["n_reads"][hi - 1])
        assert np.array_equal(ora["labels"][f0:f1], r1["labels"][f0:f1])


def test_stale_cigar_member(gpu, oracle):
    """The case the reference handles by NOT aligning (src/analignments.cpp:267-273): a labelled member that is longer than the
    representative and spans neither side reaches PPOA::insert_alignment with whatever op string the aligner object still holds.
    Here (oracle and GPU alike, DESIGN.md §3 'waived') that is the op string of the previous member of the same allele, or the empty
    string for the allele's first member.  The reference's aligner would instead still hold the op string of its last call — possibly
    from the previous allele or the previous region of the worker thread, i.e. dependent on the thread split — so only the scoped form
    is deterministic.  Both forms of the case are built (member first in its allele / not first) and must match the oracle."""
    b = synth.make_batch(12, len_range=(300, 600), n_reads=12, err="hifi", seed=77, frac_partial=0.0)
    reads, regions = b["reads"].copy(), b["regions"]
    picked = []
    for r in range(len(regions)):
        f = int(regions[r]["first_read"])
        i = f + (0 if r % 2 == 0 else 5)
        # 12 more bases (they are the next read's first bases in the arena): longer than every other read of the region
        reads["seq_len"][i] += 12
        reads["spanning_l"][i] = 0
        reads["spanning_r"][i] = 0
        picked.append(i)
    b2 = dict(b, reads=reads)
    res, _ = run_both(gpu, oracle, b2)
    lab = res["labels"][picked]
    assert (lab >= 0).sum() >= 8                         # re-assigned by similarity, so they are members of an allele graph
    # and the case is live: some of them sit in alleles built by POA (more than two reads) and are longer than the representative
    live = 0
    for j, i in enumerate(picked):
        if lab[j] < 0:
            continue
        g = res["regions"][j]
        a = res["alleles"][int(g["first_allele"]) + int(lab[j])]
        if int(a["acov"]) > 2 and int(reads["seq_len"][i]) > int(a["seq_len"]) - 8:
            live += 1
    assert live >= 4


def test_reads_with_n_bases(gpu, oracle):
    """Reads that carry bases outside ACGT ('N', lower case): their gap-affine alignments cannot use the 2-bit packed LDS tiers and take the
    byte-compare tiers inside the pipeline; edit distances compare raw bytes (N == N, case-sensitive, as WFA2 does).  Same records as the oracle."""
    b = synth.make_batch(24, len_range=(400, 1500), n_reads=14, err="ont", seed=91, frac_partial=0.15)
    arena = b["arena"].copy()
    rng = np.random.default_rng(91)
    n = arena.size - 64
    hit = rng.random(n) < 0.004
    arena[:n][hit] = ord("N")
    low = rng.random(n) < 0.002
    arena[:n][low] = np.where(arena[:n][low] == ord("N"), ord("N"), arena[:n][low] | 0x20)
    b2 = dict(b, arena=arena)
    res, st = run_both(gpu, oracle, b2)
    nseq = int(res["alleles"]["seq_len"].astype(np.int64).sum())
    assert (res["seqs"][:nseq] == ord("N")).sum() > 0 or int(st["affine_tasks"]) > 50
