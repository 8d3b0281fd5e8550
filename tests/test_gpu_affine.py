"""GPU parity: batched gap-affine WFA kernel (score + op string) vs the CPU oracle — bit-exact."""
import numpy as np
import pytest
from helpers import rand_seq, mutate, tr_seq, pair_tasks

pytestmark = pytest.mark.gpu


def _pairs(rng, n, lmax, endsfree_every=3):
    pairs, forms = [], []
    for i in range(n):
        L = int(rng.integers(0, lmax))
        a = tr_seq(rng, L) if i % 2 else rand_seq(rng, L)
        mode = i % 5
        b = rand_seq(rng, int(rng.integers(0, lmax))) if mode == 4 else mutate(rng, a, [0.002, 0.07, 0.15, 0.3][mode])
        f = None
        if endsfree_every and i % endsfree_every == 0:
            d = len(a) - len(b)
            if d >= 0:
                f = [(0, d, 0, 0), (d, 0, 0, 0), (d // 2, d // 2, 0, 0)][(i // 3) % 3]
            else:
                f = [(0, 0, 0, -d), (0, 0, -d, 0)][(i // 3) % 2]
        pairs.append((a, b))
        forms.append(f)
    return pairs, forms


def test_affine_small_mixed(gpu, oracle):
    rng = np.random.default_rng(21)
    pairs, forms = _pairs(rng, 500, 260)
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"ACGT", b"ACGT"), (b"ACTGGA", b"ACCGA")]
    forms += [None] * 5
    arena, tasks = pair_tasks(pairs, forms)
    gs, gc, gcells = gpu.affine_align_batch(arena, tasks, want_cells=True)
    es, ec, ecells = oracle.affine_align_batch(arena, tasks, want_cells=True)
    assert np.array_equal(gs, es), [(i, int(gs[i]), int(es[i]), len(pairs[i][0]), len(pairs[i][1])) for i in np.flatnonzero(gs != es)[:8]]
    bad = [(i, len(pairs[i][0]), len(pairs[i][1])) for i in range(len(pairs)) if gc[i] != ec[i]]
    assert not bad, bad[:8]
    assert np.array_equal(gcells, ecells)


def test_affine_other_penalties(gpu, oracle):
    rng = np.random.default_rng(22)
    pairs, forms = _pairs(rng, 120, 200)
    arena, tasks = pair_tasks(pairs, forms)
    for (x, o, e) in [(1, 0, 1), (3, 5, 1), (2, 4, 2), (6, 2, 3)]:
        gs, gc = gpu.affine_align_batch(arena, tasks, x, o, e)
        es, ec = oracle.affine_align_batch(arena, tasks, x, o, e)
        assert np.array_equal(gs, es), (x, o, e)
        assert gc == ec, (x, o, e)


def test_affine_long_ont(gpu, oracle):
    rng = np.random.default_rng(23)
    pairs = []
    for i in range(24):
        L = int(rng.integers(1000, 4000))
        a = mutate(rng, tr_seq(rng, L), 0.07)
        b = mutate(rng, a, 0.07 if i % 3 else 0.2)
        pairs.append((a, b))
    arena, tasks = pair_tasks(pairs)
    gs, gc = gpu.affine_align_batch(arena, tasks)
    es, ec = oracle.affine_align_batch(arena, tasks)
    assert np.array_equal(gs, es), [(i, int(gs[i]), int(es[i]), len(pairs[i][0]), len(pairs[i][1])) for i in np.flatnonzero(gs != es)[:8]]
    bad = [(i, len(pairs[i][0]), len(pairs[i][1])) for i in range(len(pairs)) if gc[i] != ec[i]]
    assert not bad, bad[:8]


def test_affine_cigar_valid_full_size(gpu, oracle):
    """Size-independent property at 10 kb: the op string consumes both sequences, M/X agree with the bytes and
    re-scoring it reproduces the reported penalty (oracle.cigar_score is a plain re-scorer)."""
    rng = np.random.default_rng(24)
    a = mutate(rng, tr_seq(rng, 10000), 0.07)
    b = mutate(rng, a, 0.07)
    arena, tasks = pair_tasks([(a, b), (a, a)])
    gs, gc = gpu.affine_align_batch(arena, tasks)
    assert gs[1] == 0 and gc[1] == b"M" * len(a)
    assert oracle.cigar_score(a, b, gc[0]) == gs[0]
    assert gc[0].count(b"M") + gc[0].count(b"X") + gc[0].count(b"D") == len(a)
    assert gc[0].count(b"M") + gc[0].count(b"X") + gc[0].count(b"I") == len(b)


def test_affine_bounded_tandem_repeats(gpu, oracle):
    """The default penalties run a banded score-bound pass first and the exact pass only inside the cells that can
    still reach the end within that bound.  Cases built to stress it: tandem repeats with copy-number changes
    (the optimal path jumps diagonals, the band may lose it -> loose bound), long end gaps, wide free ends."""
    rng = np.random.default_rng(25)
    pairs, forms = [], []
    for i in range(160):
        m = int(rng.integers(2, 40))
        motif = rand_seq(rng, m)
        n = int(rng.integers(300, 1600)) // m + 1
        a = rand_seq(rng, 60) + motif * n + rand_seq(rng, 60)
        dn = int(rng.integers(0, max(2, 200 // m)))
        n2 = max(1, n + (dn if i % 2 else -dn))
        b = a[:60] + motif * n2 + a[-60:]
        a = mutate(rng, a, [0.01, 0.05, 0.1][i % 3])
        b = mutate(rng, b, [0.01, 0.05, 0.1][(i // 3) % 3])
        f = None
        k = i % 8
        if k == 1: f = (0, len(a) // 2, 0, 0)          # pattern end free
        elif k == 2: f = (0, 0, 0, len(b) // 2)        # text end free
        elif k == 3: f = (len(a) // 3, 0, 0, 0)        # pattern begin free
        elif k == 4: f = (0, 0, len(b) // 3, 0)        # text begin free
        elif k == 5: f = (7, 9, 5, 3)
        pairs.append((a, b))
        forms.append(f)
    arena, tasks = pair_tasks(pairs, forms)
    gs, gc, gcells = gpu.affine_align_batch(arena, tasks, want_cells=True)
    es, ec, ecells = oracle.affine_align_batch(arena, tasks, want_cells=True)
    assert np.array_equal(gs, es)
    bad = [i for i in range(len(pairs)) if gc[i] != ec[i]]
    assert not bad, bad[:10]
    assert np.array_equal(gcells, ecells)


def test_affine_non_acgt_bytes(gpu, oracle):
    """Bytes outside ACGT (N, lower case, IUPAC) cannot be packed to 2 bits: those alignments must take the
    byte-compare tiers and still match the oracle op for op."""
    rng = np.random.default_rng(26)
    pairs, forms = [], []
    for i in range(60):
        L = int(rng.integers(200, 1200))
        a = bytearray(mutate(rng, tr_seq(rng, L) if i % 2 else rand_seq(rng, L), 0.03))
        b = bytearray(mutate(rng, bytes(a), 0.08))
        for s_ in (a, b) if i % 3 else (a,):
            for _ in range(int(rng.integers(1, 6))):
                s_[int(rng.integers(0, len(s_)))] = b"NnacgtRY"[int(rng.integers(0, 8))]
        pairs.append((bytes(a), bytes(b)))
        forms.append(None if i % 4 else (0, 0, 0, 0))
    arena, tasks = pair_tasks(pairs, forms)
    gs, gc, gcells = gpu.affine_align_batch(arena, tasks, want_cells=True)
    es, ec, ecells = oracle.affine_align_batch(arena, tasks, want_cells=True)
    assert np.array_equal(gs, es), [(i, int(gs[i]), int(es[i]), len(pairs[i][0]), len(pairs[i][1])) for i in np.flatnonzero(gs != es)[:8]]
    bad = [(i, len(pairs[i][0]), len(pairs[i][1])) for i in range(len(pairs)) if gc[i] != ec[i]]
    assert not bad, bad[:8]
    assert np.array_equal(gcells, ecells)


def test_affine_lds_tier_admission_sweep(gpu, oracle):
    """Alignments whose score bound sits right at the admission limit of each tier (LDS tiers: windows of 1024 / 1472 / 2048 / 4096
    diagonals; register tiers: 1024 / 1536 / 2048 / 4096 / 8192): one long gap (reduced score = gap length + 3) or two gaps in opposite directions, early and late
    in the sequence.  Whether such an alignment is admitted to a tier or passed on, op string and score must match."""
    rng = np.random.default_rng(27)
    pairs = []
    L = 900
    core = rand_seq(rng, L)
    for cap in (1024, 1472, 1536, 2048, 4096, 8192):
        for G in list(range(cap - 26, cap + 5, 3)):
            for pos in (120, L - 120):
                a = core[:pos] + rand_seq(rng, G) + core[pos:]
                pairs.append((a, core) if (G // 3) % 2 else (core, a))
        # out and back: an insertion of g1 and, far away, a deletion of g2 (the path visits diagonal +g1, ends on g1 - g2)
        for g1, g2 in ((cap // 2 - 8, cap // 2 - 9), (cap // 2 + 2, cap // 2 - 20)):
            a = core[:200] + rand_seq(rng, g1) + core[200:]
            b = core[:700] + rand_seq(rng, g2) + core[700:]
            pairs.append((a, b))
    arena, tasks = pair_tasks(pairs)
    gs, gc = gpu.affine_align_batch(arena, tasks)
    es, ec = oracle.affine_align_batch(arena, tasks)
    assert np.array_equal(gs, es)
    bad = [i for i in range(len(pairs)) if gc[i] != ec[i]]
    assert not bad, bad[:10]


def test_affine_packed_sequence_capacity_sweep(gpu, oracle):
    """The LDS and register tiers hold both sequences packed to 2 bits per base in a fixed slice (2304 / 2688 / 3072 / 6144 bytes; 4096 /
    4608 / 6144 / 8192 / 12288 bytes): near-identical pairs whose combined length crosses each capacity (about 9.1, 10.6, 12.1, 16.3, 18.3,
    24.4, 32.6 and 49 kb) must be admitted or passed on cleanly."""
    rng = np.random.default_rng(28)
    base = rand_seq(rng, 24800)
    pairs = []
    for tot in (list(range(9000, 9260, 20)) + list(range(10520, 10780, 20)) + list(range(12060, 12330, 20)) + list(range(16240, 16440, 20)) +
                list(range(18300, 18460, 20)) + list(range(24400, 24620, 20)) + list(range(32600, 32800, 20)) + list(range(48960, 49200, 30))):
        la = tot // 2 + int(rng.integers(-40, 41))
        lb = tot - la
        a = bytearray(base[:la]); b = bytearray(base[:lb])
        for s_ in (a, b):
            for _ in range(3):
                i = int(rng.integers(0, len(s_)))
                s_[i] = b"ACGT"[(b"ACGT".index(s_[i]) + 1) % 4]
        pairs.append((bytes(a), bytes(b)))
    arena, tasks = pair_tasks(pairs)
    gs, gc = gpu.affine_align_batch(arena, tasks)
    es, ec = oracle.affine_align_batch(arena, tasks)
    assert np.array_equal(gs, es), [(i, int(gs[i]), int(es[i]), len(pairs[i][0]), len(pairs[i][1])) for i in np.flatnonzero(gs != es)[:8]]
    bad = [(i, len(pairs[i][0]), len(pairs[i][1])) for i in range(len(pairs)) if gc[i] != ec[i]]
    assert not bad, bad[:8]


def test_affine_probe_boundaries(gpu, oracle):
    """Match runs whose lengths sit on the probe boundaries of the exact tiers (32 bases per probe, a second probe in the same slot visit,
    the queue beyond 64 bases): a noisy stretch sets the score bound — and with it the tier: windows of 1024 / 1536 / 2048 / 4096 / 8192 —, the
    rest of the pair is exact runs of 31 ... 130 and a few hundred bases between single substitutions, insertions and deletions."""
    rng = np.random.default_rng(29)
    runs = [31, 32, 33, 47, 62, 63, 64, 65, 66, 95, 96, 97, 127, 128, 129, 130, 257, 400]
    pairs = []
    for noisy_len in (1400, 2600, 4300, 6000, 12000):
        for rep in range(6 if noisy_len < 12000 else 2):
            base = rand_seq(rng, noisy_len)
            a = bytearray(mutate(rng, base, 0.07)); b = bytearray(mutate(rng, base, 0.07))
            order = list(rng.permutation(len(runs))) * 2
            for j, ri in enumerate(order):
                r = rand_seq(rng, runs[ri])
                a += r; b += r
                kind = (j + rep) % 3
                if kind == 0:
                    a += b"A"; b += b"C"
                elif kind == 1:
                    a += rand_seq(rng, 1 + j % 3)
                else:
                    b += rand_seq(rng, 1 + j % 2)
            pairs.append((bytes(a), bytes(b)) if rep % 2 else (bytes(b), bytes(a)))
    arena, tasks = pair_tasks(pairs)
    gs, gc = gpu.affine_align_batch(arena, tasks)
    es, ec = oracle.affine_align_batch(arena, tasks)
    assert np.array_equal(gs, es), [(i, int(gs[i]), int(es[i])) for i in np.flatnonzero(gs != es)[:8]]
    bad = [(i, len(pairs[i][0]), len(pairs[i][1])) for i in range(len(pairs)) if gc[i] != ec[i]]
    assert not bad, bad[:8]


def test_affine_wide_free_begin(gpu, oracle):
    """Right-spanning partial reads against their allele's representative (rapid_consensus, src/analignments.cpp:261-292): the read is a suffix of
    the pattern and the pattern's begin is free over hundreds to thousands of bases, so score 0 has that many start diagonals.  (The register tiers
    used to push every one of them through a 512-entry queue and hand the alignment to the HBM-row tier when they did not fit.)  Both orientations,
    free ends on the text side too, and the two-sided form."""
    rng = np.random.default_rng(29)
    pairs, forms = [], []
    for i in range(40):
        L = int(rng.integers(1800, 4700))
        a = mutate(rng, tr_seq(rng, L), 0.07)
        cut = int(rng.integers(400, L - 700))
        b = mutate(rng, a[cut:], 0.07)
        k = i % 5
        if k == 0: pairs.append((a, b)); forms.append((cut, 0, 0, 0))                       # pattern begin free = exactly the missing prefix
        elif k == 1: pairs.append((a, b)); forms.append((cut + 150, 0, 0, 0))               # ... with slack
        elif k == 2: pairs.append((b, a)); forms.append((0, 0, cut + 40, 0))                # the same on the text side
        elif k == 3: pairs.append((a, mutate(rng, a[:L - cut], 0.07))); forms.append((0, cut + 25, 0, 0))     # a prefix: pattern end free
        else: pairs.append((a, b)); forms.append((cut + 60, 0, 30, 0))                      # both begins free
    arena, tasks = pair_tasks(pairs, forms)
    gs, gc, gcells = gpu.affine_align_batch(arena, tasks, want_cells=True)
    es, ec, ecells = oracle.affine_align_batch(arena, tasks, want_cells=True)
    assert np.array_equal(gs, es), [(i, int(gs[i]), int(es[i])) for i in np.flatnonzero(gs != es)[:8]]
    bad = [i for i in range(len(pairs)) if gc[i] != ec[i]]
    assert not bad, bad[:10]
    assert np.array_equal(gcells, ecells)
