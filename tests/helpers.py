import numpy as np
from otter_amd import abi, synth

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_seq(rng, n):
    return _ACGT[rng.integers(0, 4, n)].tobytes()


def mutate(rng, s, rate, split=(0.4, 0.25, 0.35)):
    code = np.searchsorted(_ACGT, np.frombuffer(s, dtype=np.uint8)).astype(np.uint8)
    return _ACGT[synth._mutate(rng, code, rate, split)].tobytes()


def tr_seq(rng, n):
    m = int(rng.integers(2, 7))
    motif = rng.integers(0, 4, m)
    return _ACGT[np.tile(motif, n // m + 1)[:n]].tobytes()


def pair_tasks(pairs, forms=None):
    """pairs: list of (pattern_bytes, text_bytes); returns (arena, tasks)."""
    seqs = []
    for p, t in pairs:
        seqs += [p, t]
    arena, offs, lens = abi.pack_seqs(seqs)
    rows = []
    for i in range(len(pairs)):
        f = forms[i] if forms else None
        rows.append((int(offs[2 * i]), int(lens[2 * i]), int(offs[2 * i + 1]), int(lens[2 * i + 1]), f))
    return arena, abi.make_tasks(rows)
