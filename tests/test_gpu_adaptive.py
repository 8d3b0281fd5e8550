"""GPU parity of the adaptive aligner mode — wf_heuristic_wfadaptive(min_wavefront_length, max_distance_threshold, steps_between_cutoffs), the
mode the reference's aligners run in if its WFA2-lib build defaults to it (src/assemble.cpp:49-50 never calls setHeuristic*) —
against the oracle's adaptive mode: edit scores and wavefront cells, gap-affine scores + op strings + cells, and the whole pipeline."""
import numpy as np
import pytest
from helpers import rand_seq, mutate, tr_seq, pair_tasks
from otter_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def adaptive(gpu, oracle):
    """both sides in wfadaptive(10, 50, 1); restored to exact afterwards (the session's context and the oracle's process-wide switch)"""
    def set_both(on, a=10, b=50, c=1):
        gpu.set_heuristic(abi.OTG_HEURISTIC_WFADAPTIVE if on else abi.OTG_HEURISTIC_NONE, a, b, c)
        oracle.set_heuristic(1 if on else 0, a, b, c)
    set_both(True)
    yield set_both
    set_both(False)


def _tr_pair(rng, L, err, dl=0.0, partial=None):
    """two reads of a tandem-repeat locus with flanks (the bench's shape): motif copies differ by dl x L between the two"""
    m = int(rng.integers(2, 7))
    motif = rand_seq(rng, m)
    fl, fr = rand_seq(rng, 60), rand_seq(rng, 60)
    a = fl + (motif * (L // m + 1))[:L] + fr
    Lb = max(m, int(L * (1.0 - dl)))
    b = fl + (motif * (Lb // m + 1))[:Lb] + fr
    a, b = mutate(rng, a, err), mutate(rng, b, err)
    if partial == "l":
        b = b[:int(len(b) * rng.uniform(0.4, 0.9))]
    elif partial == "r":
        b = b[int(len(b) * rng.uniform(0.1, 0.6)):]
    return a, b


def _forms(a, b, kind):
    d = len(a) - len(b)
    if kind == 0 or d < 0:
        return None
    return [(0, d, 0, 0), (d, 0, 0, 0), (d // 2, d // 2, 0, 0)][kind - 1]


def _pairs(rng, n, lmin, lmax):
    pairs, forms = [], []
    for i in range(n):
        L = int(rng.integers(lmin, lmax))
        kind = i % 8
        if kind < 4:
            a, b = _tr_pair(rng, L, [0.002, 0.07, 0.07, 0.12][kind], dl=[0.0, 0.0, 0.2, 0.05][kind])
            f = 0
        elif kind < 7:
            a, b = _tr_pair(rng, L, 0.07, dl=0.1 if i % 3 == 0 else 0.0, partial="l" if kind == 4 else "r")
            f = {4: 1, 5: 2, 6: 3}[kind]
        else:
            a, b = rand_seq(rng, L), mutate(rng, rand_seq(rng, L), 0.1)      # unrelated: the wavefront is never cut much
            f = 0
        if len(b) > len(a):
            a, b = b, a
        pairs.append((a, b))
        forms.append(_forms(a, b, f))
    return pairs, forms


def test_adaptive_differs_from_exact_somewhere(gpu, oracle, adaptive):
    """the test bed is meaningful: on these inputs the adaptive oracle does not return the exact scores everywhere"""
    rng = np.random.default_rng(40)
    pairs, forms = _pairs(rng, 64, 800, 3000)
    arena, tasks = pair_tasks(pairs, forms)
    ad = oracle.edit_distance_batch(arena, tasks)
    adaptive(False)
    ex = oracle.edit_distance_batch(arena, tasks)
    adaptive(True)
    assert (ad >= ex).all() and (ad > ex).any()


def test_adaptive_edit_small(gpu, oracle, adaptive):
    rng = np.random.default_rng(41)
    pairs, forms = _pairs(rng, 400, 5, 400)
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"ACGT", b"ACGT"), (b"N" * 70, b"N" * 70), (b"acgt", b"ACGT")]
    forms += [None] * 6
    arena, tasks = pair_tasks(pairs, forms)
    got, cells = gpu.edit_distance_batch(arena, tasks, want_cells=True)
    exp, ecells = oracle.edit_distance_batch(arena, tasks, want_cells=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(cells, ecells)


def test_adaptive_edit_long(gpu, oracle, adaptive):
    rng = np.random.default_rng(42)
    pairs, forms = _pairs(rng, 160, 1000, 6000)
    arena, tasks = pair_tasks(pairs, forms)
    got, cells = gpu.edit_distance_batch(arena, tasks, want_cells=True)
    exp, ecells = oracle.edit_distance_batch(arena, tasks, want_cells=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(cells, ecells)


@pytest.mark.parametrize("params", [(10, 50, 1), (1, 0, 1), (10, 50, 3), (64, 20, 2), (10, 1000, 1), (300, 50, 1)])
def test_adaptive_edit_parameters(gpu, oracle, adaptive, params):
    adaptive(True, *params)
    rng = np.random.default_rng(43)
    pairs, forms = _pairs(rng, 96, 200, 2500)
    arena, tasks = pair_tasks(pairs, forms)
    got, cells = gpu.edit_distance_batch(arena, tasks, want_cells=True)
    exp, ecells = oracle.edit_distance_batch(arena, tasks, want_cells=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(cells, ecells)


def test_adaptive_edit_wide_and_huge(gpu, oracle, adaptive):
    """wavefronts beyond the 256- / 2048- / 16384-diagonal windows (unrelated sequences, wide free begins) and offsets beyond 16 bits"""
    rng = np.random.default_rng(44)
    a = mutate(rng, tr_seq(rng, 70000), 0.03)
    pairs = [(rand_seq(rng, 3000), rand_seq(rng, 2500)), (rand_seq(rng, 9000), rand_seq(rng, 6000)),
             (a, mutate(rng, a, 0.03)), (mutate(rng, tr_seq(rng, 20000), 0.05), mutate(rng, tr_seq(rng, 3000), 0.05))]
    forms = [None, None, None, None]
    b = mutate(rng, tr_seq(rng, 24000), 0.05)
    pairs.append((b, b[18000:])); forms.append((18000, 0, 0, 0))
    arena, tasks = pair_tasks(pairs, forms)
    got, cells = gpu.edit_distance_batch(arena, tasks, want_cells=True)
    exp, ecells = oracle.edit_distance_batch(arena, tasks, want_cells=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(cells, ecells)


@pytest.mark.parametrize("params", [(10, 50, 1), (10, 400, 1), (10, 3000, 1), (200, 50, 2)])
def test_adaptive_edit_window_and_global_row(gpu, oracle, adaptive, params):
    """the fast edit tier's storage changes: pairs that START wider than the 1024-diagonal window (free begins of thousands of bases) run their first
    scores on a global row and move in; wavefronts that outgrow the window on the way (a loose threshold never cuts much) are spilled and come back;
    pairs with bytes outside ACGT, empty and one-base sequences among them must be passed on, not spilled"""
    adaptive(True, *params)
    rng = np.random.default_rng(47)
    pairs, forms = [], []
    for i in range(40):
        L = int(rng.integers(1500, 5000))
        a = mutate(rng, tr_seq(rng, L), 0.08)
        kind = i % 5
        eb = 0.05 if (i // 5) % 2 else 0.002      # every other round nearly identical: match runs of hundreds of bases (queue, drain, wave-wide match) on the global row too
        if kind == 0:            # text = a suffix of the pattern, free begin as long as what is missing (wide start, narrow afterwards)
            cut = int(rng.integers(1200, L - 200))
            b = mutate(rng, a[cut:], eb); f = (cut, 0, 0, 0)
        elif kind == 1:          # text = a prefix, free end
            cut = int(rng.integers(200, L - 1200))
            b = mutate(rng, a[:cut], eb); f = (0, L - cut, 0, 0)
        elif kind == 2:          # both ends free, text from the middle
            x0 = int(rng.integers(600, L // 2)); x1 = int(rng.integers(L // 2 + 100, L - 600))
            b = mutate(rng, a[x0:x1], eb); f = (x0, L - x1, 0, 0)
        elif kind == 3:          # end to end, divergent: under a loose threshold the wavefront grows past the window and stays there
            b = mutate(rng, a, 0.15); f = None
        else:                    # unrelated sequences
            b = rand_seq(rng, int(rng.integers(1200, 3000))); f = None
        if f is None and len(b) > len(a):
            a, b = b, a
        pairs.append((a, b)); forms.append(f)
    n = rand_seq(rng, 2600)
    pairs += [(n[:1300] + b"N" + n[1300:], n[900:]), (n.lower(), n[1500:]), (b"", n[:1500]), (n[:1500], b""), (b"A", n[:1200]), (n, n[2000:])]
    forms += [(900, 0, 0, 0), (1500, 0, 0, 0), None, None, None, (2000, 0, 0, 0)]
    arena, tasks = pair_tasks(pairs, forms)
    got, cells = gpu.edit_distance_batch(arena, tasks, want_cells=True)
    exp, ecells = oracle.edit_distance_batch(arena, tasks, want_cells=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(cells, ecells)


def _check_affine(gpu, oracle, arena, tasks, x=4, o=6, e=2):
    got, gc, gcells = gpu.affine_align_batch(arena, tasks, x, o, e, want_cells=True)
    exp, ec, ecells = oracle.affine_align_batch(arena, tasks, x, o, e, want_cells=True)
    assert np.array_equal(got, exp)
    bad = [i for i in range(len(tasks)) if gc[i] != ec[i]]
    assert not bad, "op strings differ for tasks %s" % bad[:8]
    assert np.array_equal(gcells, ecells)


def test_adaptive_affine_small(gpu, oracle, adaptive):
    rng = np.random.default_rng(45)
    pairs, forms = _pairs(rng, 300, 5, 400)
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"ACGT", b"ACGT"), (b"ACGTNNACGT", b"ACGTNACGT")]
    forms += [None] * 5
    arena, tasks = pair_tasks(pairs, forms)
    _check_affine(gpu, oracle, arena, tasks)


def test_adaptive_affine_long(gpu, oracle, adaptive):
    rng = np.random.default_rng(46)
    pairs, forms = _pairs(rng, 96, 1000, 5000)
    arena, tasks = pair_tasks(pairs, forms)
    _check_affine(gpu, oracle, arena, tasks)


@pytest.mark.parametrize("params", [(1, 0, 1), (10, 50, 3), (64, 20, 2), (10, 1000, 1)])
def test_adaptive_affine_parameters(gpu, oracle, adaptive, params):
    adaptive(True, *params)
    rng = np.random.default_rng(47)
    pairs, forms = _pairs(rng, 64, 200, 2000)
    arena, tasks = pair_tasks(pairs, forms)
    _check_affine(gpu, oracle, arena, tasks)


def test_adaptive_affine_other_penalties_and_wide(gpu, oracle, adaptive):
    """penalties that do not reduce to (2,4,1) and wavefronts beyond the LDS windows run in the int32 tier"""
    rng = np.random.default_rng(48)
    pairs, forms = _pairs(rng, 24, 100, 900)
    arena, tasks = pair_tasks(pairs, forms)
    _check_affine(gpu, oracle, arena, tasks, 3, 5, 1)
    _check_affine(gpu, oracle, arena, tasks, 2, 3, 2)
    wide = [(rand_seq(rng, 1500), rand_seq(rng, 1200)), (rand_seq(rng, 4000), rand_seq(rng, 3000))]
    b = mutate(rng, tr_seq(rng, 9000), 0.05)
    wide.append((b, b[4000:]))
    arena, tasks = pair_tasks(wide, [None, None, (4000, 0, 0, 0)])
    _check_affine(gpu, oracle, arena, tasks)


def _compare_pipeline(res, ora):
    assert np.array_equal(res["regions"]["status"], ora["regions"]["status"])
    assert np.array_equal(res["regions"]["fc"], ora["regions"]["fc"])
    assert np.array_equal(res["regions"]["ic"], ora["regions"]["ic"])
    assert np.array_equal(res["labels"], ora["labels"])
    assert len(res["alleles"]) == len(ora["alleles"])
    for f in ("seq_len", "scov", "acov", "tcov", "ic", "ps", "hp", "region", "label"):
        assert np.array_equal(res["alleles"][f], ora["alleles"][f]), f
    n = int(ora["alleles"]["seq_len"].sum())
    assert res["seqs"][:n].tobytes() == ora["seqs"][:n].tobytes()
    assert np.allclose(res["alleles"]["se"], ora["alleles"]["se"], atol=1e-6, rtol=0)


@pytest.mark.parametrize("realign", [0, 1])
def test_adaptive_pipeline(gpu, oracle, realign):
    """otg_params.heuristic drives both aligners of the region pipeline (and local_realignment's); the context's own L1 setting is not touched"""
    b = synth.make_batch(24, len_range=(600, 2500), n_reads=24, err="ont", seed=49 + realign, frac_partial=0.2, realign=bool(realign), frac_clipped=0.3)
    P = abi.default_params(realign=realign, heuristic=abi.OTG_HEURISTIC_WFADAPTIVE)
    res = gpu.assemble(P, b)
    ora = oracle.assemble_batch(P, b)
    _compare_pipeline(res, ora)
    st, ost = gpu.assemble_stats(), ora["stats"][0]
    for f in ("edit_tasks", "edit_cells", "affine_tasks", "affine_cells"):
        assert int(st[f]) == int(ost[f]), f
    # and the exact pipeline on the same context afterwards is still the exact one
    Pe = abi.default_params(realign=realign)
    _compare_pipeline(gpu.assemble(Pe, b), oracle.assemble_batch(Pe, b))


def test_adaptive_pipeline_hifi_and_haps(gpu, oracle):
    for kw in (dict(err="hifi"), dict(err="ont", haps=True)):
        haps = kw.pop("haps", False)
        b = synth.make_batch(16, len_range=(300, 1200), n_reads=14, seed=51, frac_partial=0.25, haps=haps, **kw)
        P = abi.default_params(heuristic=abi.OTG_HEURISTIC_WFADAPTIVE, ignore_haps=0 if haps else 1)
        _compare_pipeline(gpu.assemble(P, b), oracle.assemble_batch(P, b))
