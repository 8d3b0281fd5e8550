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
