"""CPU: the oracle's WFA restatement against the mathematical pin (O(nm) DP) — edit and gap-affine scores,
end-to-end and ends-free; op strings must re-score to the optimum and consume both sequences."""
import numpy as np
from helpers import rand_seq, mutate, tr_seq, pair_tasks


def _cases(rng, n, lmax):
    for i in range(n):
        L = int(rng.integers(0, lmax))
        a = tr_seq(rng, L) if i % 2 else rand_seq(rng, L)
        b = rand_seq(rng, int(rng.integers(0, lmax))) if i % 5 == 4 else mutate(rng, a, [0.01, 0.07, 0.2, 0.35][i % 4])
        f = None
        if i % 2 == 0:
            f = tuple(int(min(x, m)) for x, m in zip(rng.integers(0, 15, 4), (len(a), len(a), len(b), len(b))))
        yield a, b, f


def test_edit_scores_match_dp(oracle):
    rng = np.random.default_rng(101)
    for a, b, f in _cases(rng, 400, 90):
        arena, tasks = pair_tasks([(a, b)], [f])
        assert oracle.edit_distance_batch(arena, tasks)[0] == oracle.dp_edit(a, b, f)


def test_affine_scores_and_cigars(oracle):
    rng = np.random.default_rng(102)
    for a, b, f in _cases(rng, 300, 80):
        arena, tasks = pair_tasks([(a, b)], [f])
        s, c = oracle.affine_align_batch(arena, tasks)
        opt = oracle.dp_affine(a, b, form=f)
        assert s[0] == opt
        assert oracle.cigar_score(a, b, c[0], form=f) == opt
        assert set(c[0]) <= set(b"MXID")


def test_affine_other_penalties(oracle):
    rng = np.random.default_rng(103)
    for (x, o, e) in [(1, 0, 1), (3, 5, 1), (2, 4, 2), (6, 2, 3)]:
        for a, b, f in _cases(rng, 60, 60):
            arena, tasks = pair_tasks([(a, b)], [f])
            s, c = oracle.affine_align_batch(arena, tasks, x, o, e)
            assert s[0] == oracle.dp_affine(a, b, x, o, e, form=f)
            assert oracle.cigar_score(a, b, c[0], x, o, e, form=f) == s[0]


def test_known_small_examples(oracle):
    arena, tasks = pair_tasks([(b"ACTGGA", b"ACAGGA"), (b"ACTGGA", b"ACCGA"), (b"", b"ACGT"), (b"AAAA", b"AAAA")])
    assert oracle.edit_distance_batch(arena, tasks).tolist() == [1, 2, 4, 0]
    s, c = oracle.affine_align_batch(arena, tasks)
    assert s.tolist() == [4, 12, 14, 0]
    assert c[0] == b"MMXMMM" and c[2] == b"IIII" and c[3] == b"MMMM"


def test_cells_formula(oracle):
    """W_p of SURVEY.md §8d for end-to-end edit alignments: sum_{t<=s} min(2t+1, a+b+1) clipped to [-a, b]."""
    rng = np.random.default_rng(104)
    for _ in range(50):
        a = rand_seq(rng, int(rng.integers(1, 60))); b = mutate(rng, a, 0.2)
        arena, tasks = pair_tasks([(a, b)])
        s, cells = oracle.edit_distance_batch(arena, tasks, want_cells=True)
        w = sum(min(t, len(b)) - max(-t, -len(a)) + 1 for t in range(int(s[0]) + 1))
        assert int(cells[0]) == w


def test_op_strings_have_a_second_witness(oracle):
    """The op-string rule twice, from two data structures: the oracle's wavefront aligner (piggy-back provenance over furthest-reaching
    offsets) and an O(nm) Gotoh dynamic programme whose backtrace applies the priorities that rule implies
    cell by cell (oracle/otter_oracle.cpp: gotoh_witness — mismatch > close a deletion > close an insertion > continue along equal bases;
    extend >= open) must return the SAME op string, not merely the same score, on 10^4 pairs: tandem-repeat and random sequences of up to
    220 bp at four divergences, end-to-end and the ends-free forms otter uses (src/analignments.cpp:268-279: free pattern end / begin /
    both halves, free text end / begin).  What stays unpinned is only whether WFA2-lib itself follows the recalled rule (4 KATs)."""
    import ctypes as C
    from otter_amd import abi
    L = oracle.lib()
    rng = np.random.default_rng(105)
    pairs, forms = [], []
    for i in range(10000):
        n = int(rng.integers(0, 220))
        a = tr_seq(rng, n) if i % 3 else rand_seq(rng, n)
        b = mutate(rng, a, [0.01, 0.07, 0.15, 0.3][i % 4]) if i % 17 else rand_seq(rng, int(rng.integers(0, 60)))
        f = None
        if i % 3 == 0:
            d = len(a) - len(b)
            f = [(0, d, 0, 0), (d, 0, 0, 0), (d // 2, d // 2, 0, 0)][(i // 3) % 3] if d >= 0 else [(0, 0, 0, -d), (0, 0, -d, 0)][(i // 3) % 2]
        pairs.append((a, b)); forms.append(f)
    arena, tasks = pair_tasks(pairs, forms)
    scores, cigs = oracle.affine_align_batch(arena, tasks)
    out = C.create_string_buffer(2048)
    ln = C.c_int(0)
    n_gap = n_free = 0
    for i, (a, b) in enumerate(pairs):
        pa, tb = np.frombuffer(a, dtype=np.uint8), np.frombuffer(b, dtype=np.uint8)
        ff = (1,) + tuple(forms[i]) if forms[i] else (0, 0, 0, 0, 0)
        s = L.oto_gotoh_align(abi.ptr(pa) if len(a) else None, len(a), abi.ptr(tb) if len(b) else None, len(b), 4, 6, 2, *ff, out, 2048, C.byref(ln))
        assert s == scores[i], (i, s, int(scores[i]))
        assert out.raw[:ln.value] == cigs[i], (i, len(a), len(b), forms[i])
        n_gap += b"I" in cigs[i] or b"D" in cigs[i]
        n_free += forms[i] is not None
    assert n_gap > 5000 and n_free > 3000
