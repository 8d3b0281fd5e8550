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


def condensed(full):
    n = full.shape[0]
    iu = np.triu_indices(n, 1)
    return np.ascontiguousarray(full[iu], dtype=np.float64)


def cluster_cases(rng, n_cases, nmax=60):
    """Synthetic condensed distance matrices + read lengths exercising every branch of otter_hclust
    (src/otterclust.cpp:118-320): n=1,2,3; unimodal; bimodal; many maxima; singleton-only; outlier repair."""
    cases = []
    for c in range(n_cases):
        kind = c % 9
        if kind == 0:
            n = [1, 2, 2, 3][(c // 9) % 4]
        else:
            n = int(rng.integers(3, nmax))
        if kind in (0, 1):      # unimodal noise
            base = rng.uniform(0.0, 0.3)
            full = base + rng.random((n, n)) * rng.uniform(0.001, 0.05)
        elif kind in (2, 3):    # two groups
            g = rng.integers(0, 2, n)
            sep = rng.uniform(0.03, 0.5)
            full = np.abs(g[:, None] - g[None, :]) * sep + rng.uniform(0.0, 0.15) + rng.random((n, n)) * rng.uniform(0.002, 0.04)
        elif kind == 4:         # several groups -> >2 maxima
            k = int(rng.integers(3, 7))
            g = rng.integers(0, k, n)
            cen = np.sort(rng.uniform(0.02, 0.9, k))
            full = np.abs(cen[g][:, None] - cen[g][None, :]) + rng.random((n, n)) * 0.01
        elif kind == 5:         # singleton-heavy: one big group + outliers
            g = np.zeros(n, dtype=int)
            no = int(rng.integers(1, max(2, n // 5)))
            g[rng.choice(n, no, replace=False)] = np.arange(1, no + 1)
            full = (g[:, None] != g[None, :]) * rng.uniform(0.2, 0.6) + rng.random((n, n)) * 0.02 + 0.05
        elif kind == 6:         # every distance far apart -> many maxima (exercises the >16 std::sort path)
            vals = np.linspace(0.03, 0.97, n * (n - 1) // 2) if n > 1 else np.zeros(0)
            rng.shuffle(vals)
            full = np.zeros((n, n))
            full[np.triu_indices(n, 1)] = vals
        elif kind == 7:         # quantised distances (ties everywhere)
            full = np.round(rng.random((n, n)) * 0.4, 2)
        else:                   # three groups with one tiny group (outlier repair)
            g = rng.choice(3, n, p=[0.55, 0.4, 0.05])
            cen = np.array([0.0, 0.25, 0.6])
            full = np.abs(cen[g][:, None] - cen[g][None, :]) + 0.08 + rng.random((n, n)) * 0.015
        full = np.triu(full, 1)
        full = full + full.T
        lens = rng.integers(100, 490, n) if c % 2 else rng.integers(300, 3000, n)
        cases.append((condensed(full) if n > 1 else np.zeros(0), lens.astype(np.uint32)))
    return cases


def pack_cluster_cases(cases):
    dist_off, len_off, nv = [], [], []
    dpos = lpos = 0
    for d, l in cases:
        dist_off.append(dpos); len_off.append(lpos); nv.append(len(l))
        dpos += d.size; lpos += len(l)
    dist = np.concatenate([c[0] for c in cases] + [np.zeros(1)])
    lens = np.concatenate([c[1] for c in cases]).astype(np.uint32)
    return (dist, np.asarray(dist_off, dtype=np.uint64), lens, np.asarray(len_off, dtype=np.uint64), np.asarray(nv, dtype=np.uint32))


def build_poa_batch(graph_specs):
    """graph_specs: list of (backbone_bytes, [(seq_bytes, cigar_bytes, spl, spr), ...], c, t).
    Returns (seq_arena, cig_arena, members, graphs)."""
    seqs, cigs = [], []
    for bb, mem, c, t in graph_specs:
        seqs.append(bb)
        for s, cg, l, r in mem:
            seqs.append(s)
            cigs.append(cg)
    sarena, soff, slen = abi.pack_seqs(seqs)
    carena, coff, clen = abi.pack_seqs(cigs) if cigs else (np.zeros(64, dtype=np.uint8), np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
    nm = sum(len(g[1]) for g in graph_specs)
    members = np.zeros(nm, dtype=abi.poa_member_dt)
    graphs = np.zeros(len(graph_specs), dtype=abi.poa_graph_dt)
    si = ci = mi = 0
    for gi, (bb, mem, c, t) in enumerate(graph_specs):
        graphs[gi]["backbone_off"] = soff[si]; graphs[gi]["backbone_len"] = slen[si]; si += 1
        graphs[gi]["first_member"] = mi; graphs[gi]["n_members"] = len(mem)
        graphs[gi]["c"] = np.float32(c); graphs[gi]["t"] = np.float32(t)
        for s, cg, l, r in mem:
            members[mi]["seq_off"] = soff[si]; members[mi]["seq_len"] = slen[si]; si += 1
            members[mi]["cigar_off"] = coff[ci]; members[mi]["cigar_len"] = clen[ci]; ci += 1
            members[mi]["spanning_l"] = int(l); members[mi]["spanning_r"] = int(r)
            mi += 1
    return sarena, carena, members, graphs


def consensus_weights(n_reads_in_allele):
    """c, t of rapid_consensus (src/analignments.cpp:285-288): c = n*0.4 narrowed to float, 1.0 if n < 4; t = 0.3f."""
    c = np.float32(n_reads_in_allele * 0.4)
    if n_reads_in_allele < 4:
        c = np.float32(1.0)
    return c, np.float32(0.3)


def random_poa_specs(rng, oracle, n_graphs, lmin, lmax, err=0.07, partial=True):
    """Allele-like read sets aligned to their first read with the oracle's affine WFA (end2end or the
    ends-free forms rapid_consensus uses, src/analignments.cpp:266-279)."""
    specs = []
    for g in range(n_graphs):
        L = int(rng.integers(lmin, lmax))
        truth = tr_seq(rng, L) if g % 2 else rand_seq(rng, L)
        n = int(rng.integers(2, 16))
        rep = mutate(rng, truth, err)
        if len(rep) == 0:
            rep = b"A"
        pairs, forms, flags, reads = [], [], [], []
        for i in range(n):
            spl = spr = True
            if partial and rng.random() < 0.2 and L > 20:
                cut = int(rng.integers(L // 3, L - 1))
                if rng.random() < 0.5:
                    rd = mutate(rng, truth[:cut], err); spr = False
                else:
                    rd = mutate(rng, truth[L - cut:], err); spl = False
            else:
                rd = mutate(rng, truth, err)
            if len(rd) == 0:
                rd = b"C"
            d = len(rep) - len(rd)
            if (spl and spr) or d < 0:
                if d >= 0:
                    f = None
                elif spl and not spr:
                    f = (0, 0, 0, -d)
                elif spr and not spl:
                    f = (0, 0, -d, 0)
                else:
                    f = None
            else:
                f = (0, d, 0, 0) if spl else (d, 0, 0, 0)
            pairs.append((rep, rd)); forms.append(f); flags.append((spl, spr)); reads.append(rd)
        arena, tasks = pair_tasks(pairs, forms)
        _, cigs = oracle.affine_align_batch(arena, tasks)
        c, t = consensus_weights(n + 1)
        specs.append((rep, [(reads[i], cigs[i], flags[i][0], flags[i][1]) for i in range(n)], c, t))
    return specs
