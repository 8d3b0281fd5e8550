"""CPU: host-side logic around the kernels — synthetic generator, static BED split, batch packing."""
import numpy as np
from otter_amd import abi, synth


def test_shard_bounds_mirror_parallelize_loop():
    """BS::thread_pool::parallelize_loop (src/BS_thread_pool.hpp:183-198): block = total/num, last takes the remainder."""
    for total, world in [(10, 3), (100, 8), (7, 8), (8, 8), (100000, 8), (1, 1), (5, 2)]:
        bounds = [synth.shard_bounds(total, world, r) for r in range(world)]
        covered = []
        for a, b in bounds:
            covered += list(range(a, b))
        assert covered == list(range(total))
        if total >= world:
            block = total // world
            assert all(b - a == block for a, b in bounds[:-1])
            assert bounds[-1][1] - bounds[-1][0] == total - block * (world - 1)


def test_synth_is_seeded_and_well_formed():
    a = synth.make_batch(5, len_range=(100, 300), n_reads=8, err="ont", seed=11)
    b = synth.make_batch(5, len_range=(100, 300), n_reads=8, err="ont", seed=11)
    assert np.array_equal(a["arena"], b["arena"]) and np.array_equal(a["reads"], b["reads"])
    c = synth.make_batch(5, len_range=(100, 300), n_reads=8, err="ont", seed=12)
    assert not np.array_equal(a["arena"][:200], c["arena"][:200])
    rd, rg = a["reads"], a["regions"]
    assert (rg["n_reads"] == 8).all() and rg["first_read"].tolist() == [0, 8, 16, 24, 32]
    assert ((rd["seq_off"] + rd["seq_len"]) <= a["arena"].size - 64).all()
    assert set(np.unique(a["arena"][:-64])) <= set(b"ACGT")
    sp = (rd["spanning_l"] == 1) & (rd["spanning_r"] == 1)
    assert 0.6 < sp.mean() <= 1.0


def test_realign_batches_carry_flanks_and_clips():
    a = synth.make_batch(6, len_range=(200, 400), n_reads=12, err="hifi", realign=True, seed=3)
    assert (a["regions"]["flank_l_len"] == 101).all() and (a["regions"]["flank_r_len"] == 101).all()
    clipped = a["reads"]["ccoord_first"] > 0
    assert clipped.any() and (a["reads"]["spanning_l"][clipped] == 0).all()


def test_pack_and_tasks():
    arena, off, ln = abi.pack_seqs([b"ACGT", b"", b"TT"])
    assert arena[:6].tobytes() == b"ACGTTT" and off.tolist() == [0, 4, 4] and ln.tolist() == [4, 0, 2]
    t = abi.make_tasks([(0, 4, 4, 2, None), (0, 4, 4, 2, (1, 2, 3, 4))])
    assert t[0]["endsfree"] == 0 and t[1]["endsfree"] == 1 and t[1]["text_end_free"] == 4


def test_dispatcher_batch_plan():
    """otg_assemble_batch_plan (the cut of a shard into batches, include/otter_gpu.h): every region once and in order whatever the size; a
    requested size is kept; the library's own plan starts small (the device idles while the first batch is read), reaches full batches
    of 2048 on a large shard and ends on two equal halves (the two contexts of a device run out of work together)."""
    import otter_amd
    for n in (0, 1, 7, 255, 256, 1025, 2048, 4096, 10000, 12500, 100000, 123457):
        for req in (0, 1, 3, 1000, 2048, 5000):
            plan = otter_amd.assemble_batch_plan(n, req)
            assert sum(plan) == n and all(p > 0 for p in plan), (n, req, plan)
            if req:
                assert plan == [req] * (n // req) + ([n % req] if n % req else []), (n, req)
            else:
                assert max(plan, default=0) <= 2048
    plan = otter_amd.assemble_batch_plan(100000, 0)
    assert plan[:3] == [256, 512, 1024] and plan[3] == 2048 and plan.count(2048) >= 40
    assert abs(plan[-1] - plan[-2]) <= 1 and plan[-1] <= 2048 * 5 // 8 + 1 and len(plan) <= 60
    assert otter_amd.assemble_batch_plan(10000, 0) == [256, 512, 1024, 2048, 2048, 2048, 1032, 1032]
    assert otter_amd.assemble_batch_plan(1000, 0) == [1000]
