"""GPU parity: batched edit-distance WFA kernel vs the CPU oracle (bit-exact integer scores)."""
import numpy as np
import pytest
from helpers import rand_seq, mutate, tr_seq, pair_tasks

pytestmark = pytest.mark.gpu


def _mixed_pairs(rng, n, lmax):
    pairs, forms = [], []
    for i in range(n):
        L = int(rng.integers(0, lmax))
        a = tr_seq(rng, L) if i % 2 else rand_seq(rng, L)
        mode = i % 5
        if mode == 4:
            b = rand_seq(rng, int(rng.integers(0, lmax)))
        else:
            b = mutate(rng, a, [0.002, 0.07, 0.15, 0.3][mode])
        if len(b) > len(a):
            a, b = b, a
        f = None
        if i % 3 == 0:
            d = len(a) - len(b)
            f = [(0, d, 0, 0), (d, 0, 0, 0), (d // 2, d // 2, 0, 0)][(i // 3) % 3]
        pairs.append((a, b))
        forms.append(f)
    return pairs, forms


def test_edit_small_mixed(gpu, oracle):
    rng = np.random.default_rng(11)
    pairs, forms = _mixed_pairs(rng, 600, 300)
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"ACGT", b"ACGT"), (b"N" * 70, b"N" * 70), (b"acgt", b"ACGT")]
    forms += [None] * 6
    arena, tasks = pair_tasks(pairs, forms)
    got, cells = gpu.edit_distance_batch(arena, tasks, want_cells=True)
    exp, ecells = oracle.edit_distance_batch(arena, tasks, want_cells=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(cells, ecells)


def test_edit_long_ont(gpu, oracle):
    rng = np.random.default_rng(12)
    pairs = []
    for i in range(48):
        L = int(rng.integers(1000, 6000))
        a = tr_seq(rng, L)
        a = mutate(rng, a, 0.07)
        b = mutate(rng, a, 0.07 if i % 2 else 0.25)
        if len(b) > len(a):
            a, b = b, a
        pairs.append((a, b))
    arena, tasks = pair_tasks(pairs)
    got = gpu.edit_distance_batch(arena, tasks)
    exp = oracle.edit_distance_batch(arena, tasks)
    assert np.array_equal(got, exp)


def test_edit_capacity_tiers(gpu, oracle):
    """Pairs whose wavefront outgrows the 2048- and 16384-diagonal LDS tiers (unrelated sequences)."""
    rng = np.random.default_rng(13)
    pairs = [(rand_seq(rng, 3000), rand_seq(rng, 2500)),       # s ~ 1500 -> tier 2
             (rand_seq(rng, 20000), rand_seq(rng, 9000)),       # |k| > 8192 -> tier 3 (global wavefront)
             (rand_seq(rng, 40), rand_seq(rng, 30))]
    arena, tasks = pair_tasks(pairs)
    got = gpu.edit_distance_batch(arena, tasks)
    exp = oracle.edit_distance_batch(arena, tasks)
    assert np.array_equal(got, exp)


def test_edit_properties_full_size(gpu):
    """Size-independent properties at bench-like sizes: d(a,a)=0, symmetry, triangle inequality,
    length-difference lower bound, and d(a, a with k substitutions) <= k."""
    rng = np.random.default_rng(14)
    L = 10000
    a = mutate(rng, tr_seq(rng, L), 0.07)
    b = mutate(rng, a, 0.07)
    c = mutate(rng, b, 0.07)
    sub = bytearray(a)
    for p in rng.choice(len(a), 50, replace=False):
        sub[p] = b"ACGT"[(b"ACGT".index(bytes([sub[p]])) + 1) % 4]
    sub = bytes(sub)
    def big(x, y):
        return (x, y) if len(x) >= len(y) else (y, x)
    arena, tasks = pair_tasks([(a, a), big(a, b), big(b, a), big(b, c), big(a, c), big(a, sub)])
    d = gpu.edit_distance_batch(arena, tasks)
    assert d[0] == 0
    assert d[1] == d[2]
    assert d[4] <= d[1] + d[3]
    assert d[1] >= abs(len(a) - len(b))
    assert 0 < d[5] <= 50


def test_edit_bitparallel_tiers(gpu, oracle):
    """Pairs that leave the score-capped wavefront tier for the banded bit-parallel tiers (1, 2, 4 blocks per lane)
    and back to the wavefront tiers: large length differences, ends-free forms, N runs, unsupported alphabets."""
    rng = np.random.default_rng(15)
    pairs, forms = [], []
    for i in range(120):
        L = int(rng.integers(300, 9000))
        a = mutate(rng, tr_seq(rng, L), 0.07)
        frac = [0.0, 0.1, 0.3, 0.6][i % 4]
        cut = int(len(a) * (1 - frac))
        b = mutate(rng, a[:cut] if i % 2 else a[len(a) - cut:], [0.07, 0.2][(i // 4) % 2])
        if i % 9 == 0:
            a = a[:50] + b"N" * 37 + a[50:]
        if i % 13 == 0:
            b = b[:20] + b"nnacgt" + b[20:]          # lower case in the text only: matches nothing
        if i % 17 == 0:
            a = a[:30] + b"NRY" + a[30:]            # three non-ACGT symbols in the pattern: wavefront fallback
        f = None
        if len(b) > len(a):
            a, b = b, a
        if i % 3 == 0:
            d = len(a) - len(b)
            f = [(0, d, 0, 0), (d, 0, 0, 0), (d // 2, d // 2, 0, 0)][(i // 3) % 3]
        pairs.append((a, b)); forms.append(f)
    pairs += [(b"A" * 5000, b"A" * 4000), (b"ACGT" * 1000, b"TGCA" * 900), (rand_seq(rng, 700), rand_seq(rng, 650))]
    forms += [None, None, None]
    arena, tasks = pair_tasks(pairs, forms)
    got, cells = gpu.edit_distance_batch(arena, tasks, want_cells=True)
    exp, ecells = oracle.edit_distance_batch(arena, tasks, want_cells=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(cells, ecells)


def test_edit_text_longer_than_pattern(gpu, oracle):
    rng = np.random.default_rng(16)
    pairs = []
    for i in range(40):
        a = mutate(rng, tr_seq(rng, int(rng.integers(200, 3000))), 0.07)
        b = mutate(rng, a, 0.1) + rand_seq(rng, int(rng.integers(0, 400)))
        pairs.append((a, b) if len(a) <= len(b) else (b, a))
    arena, tasks = pair_tasks(pairs)
    assert np.array_equal(gpu.edit_distance_batch(arena, tasks), oracle.edit_distance_batch(arena, tasks))


def test_edit_band_edges(gpu, oracle):
    """Distances placed right at the thresholds the bit-parallel tiers can certify (K + 1 diagonals per tier: about
    456, 976, 2016 rows for 8/16/32-lane groups): two far-apart block indels drive the optimal path to one edge of the
    band and back, with and without free pattern ends.  Pairs just under a threshold must be exact in that tier,
    pairs just over it must move up a tier — either way the score equals the oracle's."""
    rng = np.random.default_rng(17)
    pairs, forms = [], []
    for T in (456, 904, 976, 1352, 1936, 2016, 2896):       # (1352 / 2896: the three-block tiers <3,8> / <3,16>)
        for trial in range(14 if T < 1000 else 8):
            L = int(rng.integers(1400, 2600)) if T < 2500 else int(rng.integers(3200, 4200))
            core = rand_seq(rng, L)
            tot = T + int(rng.integers(-8, 9))
            d1 = int(rng.integers(0, tot + 1)) if trial % 3 else tot
            d2 = tot - d1
            p_, q_ = sorted(rng.integers(100, L - 100, 2).tolist())
            if trial % 2:
                a = core[:p_] + rand_seq(rng, d1) + core[p_:]
                b = core[:q_] + rand_seq(rng, d2) + core[q_:]
            else:           # both blocks in the same sequence: the path leaves the main diagonal by d1 + d2
                a = core[:p_] + rand_seq(rng, d1) + core[p_:q_] + rand_seq(rng, d2) + core[q_:]
                b = core
            if trial % 5 == 4:
                a = mutate(rng, a, 0.01)
            if len(b) > len(a):
                a, b = b, a
            d = len(a) - len(b)
            f = [None, (0, d, 0, 0), (d, 0, 0, 0), (d // 2, d - d // 2, 0, 0), None][trial % 5]
            pairs.append((a, b))
            forms.append(f)
    arena, tasks = pair_tasks(pairs, forms)
    got = gpu.edit_distance_batch(arena, tasks)
    exp = oracle.edit_distance_batch(arena, tasks)
    bad = [(i, int(got[i]), int(exp[i])) for i in range(len(pairs)) if got[i] != exp[i]]
    assert not bad, bad[:10]


def test_edit_threshold_sweep(gpu, oracle):
    """Systematic sweep across the certification thresholds: one block deletion of D bases early / late / split in two,
    for every even D from 36 below to 12 above each threshold.  With the band at its full width a lane switches from
    one superblock to its next without a spare step, which is where a stale text group once slipped in."""
    rng = np.random.default_rng(18)
    pairs, meta = [], []
    for T, L in ((456, 1100), (904, 1400), (904, 2100), (976, 1500), (1936, 2500), (2016, 2600)):
        core = rand_seq(rng, L)
        p_, q_ = 300, L - 300
        for D in range(T - 36, T + 13, 2):
            for split in (0.0, 0.3, 1.0):
                d1 = int(D * split)
                d2 = D - d1
                pairs.append((core[:p_] + rand_seq(rng, d1) + core[p_:q_] + rand_seq(rng, d2) + core[q_:], core))
                a2, b2 = core[:p_] + rand_seq(rng, d1) + core[p_:], core[:q_] + rand_seq(rng, d2) + core[q_:]
                if len(b2) > len(a2):
                    a2, b2 = b2, a2
                pairs.append((a2, b2))
                meta += [(T, L, D, split, "one"), (T, L, D, split, "two")]
    arena, tasks = pair_tasks(pairs)
    got = gpu.edit_distance_batch(arena, tasks)
    exp = oracle.edit_distance_batch(arena, tasks)
    bad = [(meta[i], int(got[i]), int(exp[i])) for i in range(len(pairs)) if got[i] != exp[i]]
    assert not bad, bad[:10]


def test_edit_misleading_samples(gpu, oracle):
    """The tier choice comes from two 64-base samples (first and last 64 bases of the pattern against the ends of the text).  Pairs
    built to mislead it: identical ends around a divergent middle (looks near-identical), garbage ends around an identical middle
    (looks hopeless), one good and one bad end, ends shifted by an indel run inside the sampled window, non-ACGT bytes in the
    samples, lengths around the 160-base minimum of the sampler, one-sided free ends.  Every route must end in the exact distance."""
    rng = np.random.default_rng(77)
    pairs, forms = [], []
    for i in range(240):
        L = int(rng.integers(150, 2600)) if i % 6 else int(rng.choice([158, 159, 160, 161, 192, 255, 256, 257]))
        core = rand_seq(rng, L)
        kind = i % 8
        a = core
        if kind == 0:      # clean ends, divergent middle
            b = core[:100] + mutate(rng, core[100:-100], 0.25) + core[-100:] if L > 260 else mutate(rng, core, 0.2)
        elif kind == 1:    # garbage ends, identical middle
            b = rand_seq(rng, 90) + core[90:-90] + rand_seq(rng, 90) if L > 260 else rand_seq(rng, L)
        elif kind == 2:    # good left, bad right
            b = core[:-120] + rand_seq(rng, 120) if L > 260 else core
        elif kind == 3:    # bad left, good right
            b = rand_seq(rng, 120) + core[120:] if L > 260 else core
        elif kind == 4:    # indel run inside the sampled windows
            g = int(rng.integers(5, 40))
            b = core[:20] + rand_seq(rng, g) + core[20:-30] + core[-30 + min(g, 25):]
        elif kind == 5:    # non-ACGT bytes in the samples
            bb = bytearray(mutate(rng, core, 0.05))
            for p in (3, 17, 40, len(bb) - 5, len(bb) - 33):
                bb[p] = b"NnRy"[int(rng.integers(0, 4))]
            b = bytes(bb)
        elif kind == 6:    # homopolymer / short-period ends (every shift matches)
            b = b"A" * 70 + mutate(rng, core[70:-70], 0.1) + b"CA" * 35 if L > 200 else core
            a = b"A" * 80 + core[70:-70] + b"CA" * 30 if L > 200 else core
        else:
            b = mutate(rng, core, [0.01, 0.08, 0.3][(i // 8) % 3])
        if len(b) > len(a):
            a, b = b, a
        f = None
        if i % 5 == 0:
            d = len(a) - len(b)
            f = [(0, d, 0, 0), (d, 0, 0, 0)][(i // 5) % 2]
        pairs.append((a, b))
        forms.append(f)
    arena, tasks = pair_tasks(pairs, forms)
    gs = gpu.edit_distance_batch(arena, tasks)
    es = oracle.edit_distance_batch(arena, tasks)
    assert np.array_equal(gs, es), np.nonzero(gs != es)[0][:10]
