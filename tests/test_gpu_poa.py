"""GPU parity: POA consensus kernel vs the CPU oracle (and the reference's own PPOA when oracle/_ref is built)."""
import json
import os
import numpy as np
import pytest
from helpers import build_poa_batch, random_poa_specs, pair_tasks

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _check(gpu, oracle, specs):
    sarena, carena, members, graphs = build_poa_batch(specs)
    exp = oracle.poa_consensus_batch(sarena, carena, members, graphs)
    got = gpu.poa_consensus_batch(sarena, carena, members, graphs)
    assert got == exp
    if oracle.ref() is not None:
        assert got == oracle.poa_consensus_batch(sarena, carena, members, graphs, which="ref")
    return got


def test_poa_reference_kats_on_gpu(gpu, oracle):
    """The reference's 4 known-answer tests (test/ppoa_test.cpp:39-105): GPU affine WFA op strings -> GPU POA."""
    kats = json.load(open(os.path.join(GOLD, "ppoa_kats.json")))
    for kat in kats:
        seqs = [s.encode() for s in kat["sequences"]]
        arena, tasks = pair_tasks([(seqs[0], s) for s in seqs])
        _, cigs = gpu.affine_align_batch(arena, tasks)
        n = len(seqs)
        spec = (seqs[0], [(seqs[i], cigs[i], True, True) for i in range(n)], np.float32(n * 0.4), np.float32(0.3))
        got = _check(gpu, oracle, [spec])
        assert got[0].decode() == kat["expected"]


def test_poa_random_small(gpu, oracle):
    rng = np.random.default_rng(41)
    _check(gpu, oracle, random_poa_specs(rng, oracle, 200, 1, 120, err=0.1))


def test_poa_random_ont_kb(gpu, oracle):
    rng = np.random.default_rng(42)
    _check(gpu, oracle, random_poa_specs(rng, oracle, 24, 800, 2500, err=0.07))


def test_poa_hifi(gpu, oracle):
    rng = np.random.default_rng(43)
    _check(gpu, oracle, random_poa_specs(rng, oracle, 60, 300, 700, err=0.002))


def test_poa_op_string_fuzz(gpu, oracle):
    """Op strings that are NOT alignment output: random valid op sequences (M/X/D count = backbone length, M/X/I count =
    read length) built from runs whose lengths straddle the 64-op chunks the wave kernel reads, starting with any op,
    with long gap runs, consecutive insertions and deletions, for spanning and non-spanning members."""
    from helpers import rand_seq
    rng = np.random.default_rng(44)
    specs = []
    for g in range(120):
        B = int(rng.integers(2, 400)) if g % 4 else int(rng.choice([2, 3, 11, 63, 64, 65, 127, 128, 129, 192]))
        bb = rand_seq(rng, B)
        mem = []
        for _ in range(int(rng.integers(1, 12))):
            ops = []
            ref = 0
            while ref < B:
                kind = rng.choice(["M", "M", "M", "X", "D", "I"], p=[0.3, 0.3, 0.15, 0.1, 0.08, 0.07])
                run = int(rng.choice([1, 1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 100]))
                if kind in "MXD":
                    run = min(run, B - ref)
                    ref += run
                ops.append(kind * run)
            if rng.random() < 0.3:
                ops.append("I" * int(rng.integers(1, 70)))
            cig = "".join(ops).encode()
            tlen = cig.count(b"M") + cig.count(b"X") + cig.count(b"I")
            mem.append((rand_seq(rng, max(tlen, 1))[:tlen] if tlen else b"", cig, bool(rng.random() < 0.8), bool(rng.random() < 0.8)))
        n = len(mem)
        c = np.float32(n * 0.4) if n >= 4 else np.float32(1.0)
        specs.append((bb, mem, c, np.float32(0.3)))
    _check(gpu, oracle, specs)


def test_poa_lds_capacity_mix(gpu, oracle):
    """The LDS kernel sizes its block for the optimistic need of 97 % of the batch: a crowd of small, clean windows sets a small
    block; graphs of the same backbone length whose members disagree everywhere (random target bases: alt nodes and edges close
    to the worst-case bound) outgrow it mid-way and must be redone by the global-memory kernel; a few long backbones never fit.
    With 40 members a window also passes 255 edges per node list.  All must equal the oracle."""
    from helpers import rand_seq, mutate
    rng = np.random.default_rng(45)
    specs = []
    for g in range(400):
        B = int(rng.integers(120, 180))
        bb = rand_seq(rng, B)
        kind = 0 if g % 20 else (1 + (g // 20) % 3)
        mem = []
        nmem = 40 if kind == 1 else int(rng.integers(8, 30))
        if kind == 3:
            B = int(rng.integers(2500, 4000)); bb = rand_seq(rng, B); nmem = 6
        for _ in range(nmem):
            ops, ref = [], 0
            noisy = kind in (1, 2)
            while ref < B:
                k = rng.choice(["M", "X", "D", "I"], p=[0.55, 0.25, 0.05, 0.15] if noisy else [0.93, 0.03, 0.02, 0.02])
                run = int(rng.integers(1, 4)) if k != "M" else int(rng.integers(1, 6 if noisy else 40))
                if k in "MXD":
                    run = min(run, B - ref); ref += run
                ops.append(k * run)
            cig = "".join(ops).encode()
            tlen = cig.count(b"M") + cig.count(b"X") + cig.count(b"I")
            mem.append((rand_seq(rng, max(tlen, 1))[:tlen], cig, bool(rng.random() < 0.9), bool(rng.random() < 0.9)))
        n = len(mem)
        specs.append((bb, mem, np.float32(n * 0.4), np.float32(0.3)))
    _check(gpu, oracle, specs)


def test_poa_crowded_anchors_and_long_jumps(gpu, oracle):
    """Shapes the second-generation path handles apart: many members leaving the backbone at the SAME node with long, different
    insertions (one anchor with hundreds of subtree edges: swept in pieces of 64), shared insertion prefixes (subtrees that branch),
    deletions longer than the 128-node window of backbone weights (landings written straight to memory), members that start off
    the backbone (start-node subtrees), and 300-op runs without an 'M' (serial threading in the middle of an op string).  Large
    backbones so that the graphs stay out of LDS."""
    from helpers import rand_seq
    rng = np.random.default_rng(46)
    specs = []
    for g in range(12):
        B = int(rng.integers(2600, 3400))
        bb = rand_seq(rng, B)
        pos = int(rng.integers(200, B - 900))
        shared = rand_seq(rng, 40)
        mem = []
        nmem = int(rng.integers(6, 16))
        for m in range(nmem):
            kind = (g + m) % 6
            ops = []
            if kind == 5:                       # starts with insertions / mismatches: start nodes and their subtrees
                ops.append("I" * int(rng.integers(1, 30)))
                ops.append("X" * int(rng.integers(1, 4)))
                ref = ops[-1].count("X")
            else:
                ref = 0
            def M(n):
                return "M" * n
            ops.append(M(pos - ref)); ref = pos
            if kind in (0, 1, 5):               # a long insertion at `pos`: own bases (0), a shared prefix then own bases (1)
                ins = int(rng.integers(70, 320))
                ops.append("I" * ins)
            elif kind == 2:                     # a deletion beyond the window, then a mismatch run
                d = int(rng.integers(140, 420))
                ops.append("D" * d); ref += d
                ops.append("X" * 3); ref += 3
            elif kind == 3:                     # 300 ops without an 'M'
                for _ in range(60):
                    ops.append("X" * 2 + "I" * 2 + "D" * 1); ref += 3
            rest = B - ref
            while rest > 0:                     # the rest: ONT-like noise
                k = rng.choice(["M", "X", "D", "I"], p=[0.9, 0.04, 0.03, 0.03])
                run = int(rng.integers(1, 40)) if k == "M" else int(rng.integers(1, 3))
                if k in "MXD":
                    run = min(run, rest); rest -= run
                ops.append(k * run)
            cig = "".join(ops).encode()
            tlen = cig.count(b"M") + cig.count(b"X") + cig.count(b"I")
            seq = bytearray(rand_seq(rng, tlen))
            if kind == 1:                       # the first 40 inserted bases are the same in every member of this kind
                t0 = cig[:cig.index(b"I")].count(b"M") + cig[:cig.index(b"I")].count(b"X")
                seq[t0:t0 + len(shared)] = shared
            mem.append((bytes(seq), cig, bool(kind != 4 or rng.random() < 0.5), bool(rng.random() < 0.9)))
        n = len(mem)
        c = np.float32(n * 0.4) if n >= 4 else np.float32(1.0)
        specs.append((bb, mem, c, np.float32(0.3)))
    _check(gpu, oracle, specs)


def test_poa_shared_and_private_insertions(gpu, oracle):
    """Long stretches of X / I ops on graphs in global memory — the lane-parallel paths of the wave-uniform code (csrc/poa.hip: a stretch that
    makes a fresh chain, a stretch that retraces a chain an earlier member made): members of one allele share insertions at the head of the read,
    inside it and next to the end of the backbone (where nodes become ending nodes), some copy them exactly (a retraced chain), some diverge in the
    middle of one (retrace, then a fresh chain), some carry substitutions inside the stretch, lengths straddle the 64-op windows; spanning and
    non-spanning members.  Reference: PPOA::insert_alignment (src/anppoa.hpp:112-241)."""
    from helpers import rand_seq
    rng = np.random.default_rng(46)
    other = {65: b"CGT", 67: b"AGT", 71: b"ACT", 84: b"ACG"}
    specs = []
    for g in range(36):
        B = int(rng.integers(700, 1900))
        bb = rand_seq(rng, B)
        templates = []
        for _ in range(int(rng.integers(1, 4))):
            ev = {}
            if rng.random() < 0.8:
                ev[0] = ("I", rand_seq(rng, int(rng.choice([3, 4, 5, 63, 64, 65, 130, 300]))))
            for _ in range(int(rng.integers(2, 9))):
                pos = int(rng.integers(1, B - 1))
                kind = rng.choice(["I", "X", "D", "IX"])
                if kind == "I":
                    ev[pos] = ("I", rand_seq(rng, int(rng.choice([4, 40, 70, 129, 200]))))
                elif kind == "X":
                    n = int(min(rng.choice([1, 5, 70]), B - pos))
                    ev[pos] = ("X", bytes(other[bb[pos + j]][int(rng.integers(0, 3))] for j in range(n)))
                elif kind == "D":
                    ev[pos] = ("D", int(min(rng.choice([1, 6, 80]), B - pos)))
                else:   # an insertion with substitutions on both sides of it
                    ev[pos] = ("IX", rand_seq(rng, int(rng.choice([8, 66, 140]))))
            if rng.random() < 0.6:
                ev[B - int(rng.integers(0, 12))] = ("I", rand_seq(rng, int(rng.choice([2, 9, 70]))))     # at / next to the end of the backbone
            templates.append(ev)
        mem = []
        for _ in range(int(rng.integers(4, 14))):
            ev = dict(templates[int(rng.integers(0, len(templates)))])
            for pos in list(ev):
                kind, val = ev[pos]
                if kind in ("I", "IX") and rng.random() < 0.35:
                    val = bytearray(val)
                    cut = int(rng.integers(0, len(val)))
                    if rng.random() < 0.5:
                        val = val[:max(cut, 1)]                                  # a shorter copy: retraces a prefix of the chain
                    else:
                        val[cut] = other[val[cut]][0]                             # diverges in the middle: retrace, then a fresh chain
                    ev[pos] = (kind, bytes(val))
            ops, seq, ref = [], bytearray(), 0
            while ref <= B:
                e = ev.get(ref)
                if e and e[0] in ("I", "IX"):
                    if e[0] == "IX" and ref < B:
                        ops.append("X"); seq.append(other[bb[ref]][0]); ref += 1
                        if ref > B:
                            break
                    ops.append("I" * len(e[1])); seq += e[1]
                    if e[0] == "IX" and ref < B:
                        ops.append("X"); seq.append(other[bb[ref]][1]); ref += 1
                        continue
                if ref >= B:
                    break
                if e and e[0] == "X":
                    n = min(len(e[1]), B - ref)
                    ops.append("X" * n); seq += e[1][:n]; ref += n
                elif e and e[0] == "D":
                    n = min(e[1], B - ref)
                    ops.append("D" * n); ref += n
                else:
                    ops.append("M"); seq.append(bb[ref]); ref += 1
            cig = "".join(ops).encode()
            assert cig.count(b"M") + cig.count(b"X") + cig.count(b"D") == B and cig.count(b"M") + cig.count(b"X") + cig.count(b"I") == len(seq)
            mem.append((bytes(seq), cig, bool(rng.random() < 0.85), bool(rng.random() < 0.85)))
        n = len(mem)
        specs.append((bb, mem, np.float32(n * 0.4) if n >= 4 else np.float32(1.0), np.float32(0.3)))
    _check(gpu, oracle, specs)
