"""ctypes access to the CPU oracle (oracle/libotter_oracle.so) and, when built, the reference-source
build (oracle/_ref/libotter_ref.so).  TEST INFRASTRUCTURE ONLY — never imported by otter_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np
from otter_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libotter_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libotter_ref.so")
REF_IO_SO = os.path.join(ROOT, "oracle", "_ref", "libotter_ref_io.so")

u8p, i32p, u32p, u64p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_int32, C.c_uint32, C.c_uint64, C.c_double))


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        _lib = C.CDLL(ORACLE_SO)
        _lib.oto_kde_f.restype = C.c_double
        _lib.oto_assemble_batch.restype = C.c_void_p
        _lib.oto_result_seq_bytes.restype = C.c_uint64
        _lib.oto_result_dist_len.restype = C.c_uint64
        _lib.oto_result_n_alleles.restype = C.c_uint32
        _lib.oto_medoid.restype = C.c_uint32
    return _lib


def ref():
    """Reference-source build; None when absent (e.g. /root/reference not mounted and no prebuilt)."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        _ref = C.CDLL(REF_SO)
        _ref.ref_kde_f.restype = C.c_double
        _ref.ref_medoid.restype = C.c_uint32
    return _ref


def set_heuristic(on, min_wf_len=10, max_dist=50, steps=1):
    """Process-wide heuristic of the oracle's L1 aligners (0 = exact); the pipeline entry points take theirs from otg_params."""
    lib().oto_set_heuristic(int(on), int(min_wf_len), int(max_dist), int(steps))


def _b(s):
    return np.frombuffer(s if isinstance(s, (bytes, bytearray)) else s.encode(), dtype=np.uint8)


def edit_distance_batch(arena, tasks, want_cells=False):
    n = len(tasks)
    scores = np.zeros(n, dtype=np.int32)
    cells = np.zeros(n, dtype=np.uint64)
    lib().oto_edit_distance_batch(abi.ptr(arena), C.c_uint64(arena.size), abi.ptr(tasks), C.c_uint32(n), abi.ptr(scores), abi.ptr(cells))
    return (scores, cells) if want_cells else scores


def affine_align_batch(arena, tasks, x=4, o=6, e=2, want_cells=False):
    n = len(tasks)
    scores = np.zeros(n, dtype=np.int32)
    off = np.zeros(n, dtype=np.uint64)
    ln = np.zeros(n, dtype=np.uint32)
    cap = int(sum(int(t["pattern_len"]) + int(t["text_len"]) for t in tasks)) + 64
    out = np.zeros(cap, dtype=np.uint8)
    used = C.c_uint64(0)
    cells = np.zeros(n, dtype=np.uint64)
    rc = lib().oto_affine_align_batch(abi.ptr(arena), C.c_uint64(arena.size), abi.ptr(tasks), C.c_uint32(n), x, o, e,
                                      abi.ptr(scores), abi.ptr(off), abi.ptr(ln), abi.ptr(out), C.c_uint64(cap), C.byref(used), abi.ptr(cells))
    assert rc == 0
    cigs = [out[int(off[i]):int(off[i]) + int(ln[i])].tobytes() for i in range(n)]
    return (scores, cigs, cells) if want_cells else (scores, cigs)


def dp_edit(p, t, form=None):
    p, t = _b(p), _b(t)
    f = (1,) + tuple(form) if form else (0, 0, 0, 0, 0)
    return lib().oto_dp_edit(abi.ptr(p), p.size, abi.ptr(t), t.size, *f)


def dp_affine(p, t, x=4, o=6, e=2, form=None):
    p, t = _b(p), _b(t)
    f = (1,) + tuple(form) if form else (0, 0, 0, 0, 0)
    return lib().oto_dp_affine(abi.ptr(p), p.size, abi.ptr(t), t.size, x, o, e, *f)


def cigar_score(p, t, cig, x=4, o=6, e=2, form=None):
    p, t = _b(p), _b(t)
    f = (1,) + tuple(form) if form else (0, 0, 0, 0, 0)
    return lib().oto_cigar_score(abi.ptr(p), p.size, abi.ptr(t), t.size, x, o, e, *f, C.c_char_p(cig), len(cig))


def find_clustering_dist(values, bandwidth, radius=4, dinterval=0.0025, which=None):
    L = which or lib()
    values = np.ascontiguousarray(values, dtype=np.float64)
    b = np.zeros(3)
    dens = np.zeros(512)
    nd = C.c_int(0)
    err = L.oto_find_clustering_dist(radius, C.c_double(dinterval), C.c_double(bandwidth), abi.ptr(values), C.c_uint64(values.size),
                                     abi.ptr(b), abi.ptr(dens), C.byref(nd))
    return err, b, dens[:nd.value].copy()


def kde_maximas(dens, radius=4, which="oracle"):
    L = lib() if which == "oracle" else ref()
    fn = L.oto_kde_maximas if which == "oracle" else L.ref_kde_maximas
    dens = np.ascontiguousarray(dens, dtype=np.float64)
    mi, mv = np.zeros(512, dtype=np.int32), np.zeros(512)
    ni, nv = np.zeros(512, dtype=np.int32), np.zeros(512)
    a, b = C.c_int(0), C.c_int(0)
    fn(radius, abi.ptr(dens), dens.size, abi.ptr(mi), abi.ptr(mv), C.byref(a), abi.ptr(ni), abi.ptr(nv), C.byref(b))
    return list(zip(mi[:a.value].tolist(), mv[:a.value].tolist())), list(zip(ni[:b.value].tolist(), nv[:b.value].tolist()))


def kde_f(h, values, x, which="oracle"):
    values = np.ascontiguousarray(values, dtype=np.float64)
    if which == "oracle":
        return lib().oto_kde_f(C.c_double(h), abi.ptr(values), C.c_uint64(values.size), C.c_double(x))
    return ref().ref_kde_f(C.c_double(h), abi.ptr(values), C.c_uint64(values.size), C.c_double(x))


def hclust_average(n, dist, which="oracle"):
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    merge = np.zeros(2 * (n - 1), dtype=np.int32)
    height = np.zeros(n - 1)
    if which == "oracle":
        lib().oto_hclust_average(n, abi.ptr(dist), abi.ptr(merge), abi.ptr(height))
    else:
        ref().ref_hclust_average(n, abi.ptr(dist), abi.ptr(merge), abi.ptr(height))
    return merge, height


def cutree_k(n, merge, k, which="oracle"):
    labels = np.zeros(n, dtype=np.int32)
    (lib().oto_cutree_k if which == "oracle" else ref().ref_cutree_k)(n, abi.ptr(merge), k, abi.ptr(labels))
    return labels


def cutree_cdist(n, merge, height, cdist, which="oracle"):
    labels = np.zeros(n, dtype=np.int32)
    h = np.ascontiguousarray(height, dtype=np.float64).copy()
    (lib().oto_cutree_cdist if which == "oracle" else ref().ref_cutree_cdist)(n, abi.ptr(merge), abi.ptr(h), C.c_double(cdist), abi.ptr(labels))
    return labels


def medoid(n, dist, ind, which="oracle"):
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    ind = np.ascontiguousarray(ind, dtype=np.uint32)
    fn = lib().oto_medoid if which == "oracle" else ref().ref_medoid
    return int(fn(C.c_uint32(n), abi.ptr(dist), abi.ptr(ind), C.c_uint32(ind.size)))


def cluster_batch(params, dist, dist_off, read_len, len_off, n_valid):
    nreg = len(n_valid)
    labels = np.full(int(read_len.size), -1, dtype=np.int32)
    ic = np.zeros(nreg, dtype=np.int32)
    fc = np.zeros(nreg, dtype=np.int32)
    bounds = np.full(3 * nreg, np.nan)
    rc = lib().oto_cluster_batch(C.byref(params), abi.ptr(dist), abi.ptr(dist_off), abi.ptr(read_len), abi.ptr(len_off),
                                 abi.ptr(n_valid), C.c_uint32(nreg), abi.ptr(labels), abi.ptr(ic), abi.ptr(fc), abi.ptr(bounds))
    return rc, labels, ic, fc, bounds.reshape(-1, 3)


def poa_consensus_batch(seq_arena, cig_arena, members, graphs, which="oracle"):
    ng = len(graphs)
    off = np.zeros(ng, dtype=np.uint64)
    ln = np.zeros(ng, dtype=np.uint32)
    cap = int(seq_arena.size) * 2 + 1024
    out = np.zeros(cap, dtype=np.uint8)
    used = C.c_uint64(0)
    fn = lib().oto_poa_consensus_batch if which == "oracle" else ref().ref_poa_consensus_batch
    rc = fn(abi.ptr(seq_arena), C.c_uint64(seq_arena.size), abi.ptr(cig_arena), C.c_uint64(cig_arena.size),
            abi.ptr(members), C.c_uint32(len(members)), abi.ptr(graphs), C.c_uint32(ng),
            abi.ptr(off), abi.ptr(ln), abi.ptr(out), C.c_uint64(cap), C.byref(used))
    assert rc == 0
    return [out[int(off[i]):int(off[i]) + int(ln[i])].tobytes() for i in range(ng)]


def genotype_cluster_batch(params, arena, seq_off, seq_len, first_allele, n_alleles):
    nreg = len(n_alleles)
    na = len(seq_off)
    gt, gl, gk = (np.zeros(na, dtype=np.int32) for _ in range(3))
    hsd = np.zeros(na)
    ngt = np.zeros(nreg, dtype=np.int32)
    reps = np.zeros(na, dtype=np.int32)
    lib().oto_genotype_cluster_batch(C.byref(params), abi.ptr(arena), C.c_uint64(arena.size), abi.ptr(seq_off), abi.ptr(seq_len),
                                     abi.ptr(first_allele), abi.ptr(n_alleles), C.c_uint32(nreg),
                                     abi.ptr(gt), abi.ptr(gl), abi.ptr(gk), abi.ptr(hsd), abi.ptr(ngt), abi.ptr(reps))
    return gt, gl, gk, hsd, ngt, reps


def assemble_batch(params, batch, region_range=None):
    """Run the oracle pipeline on a synth batch; returns dict of numpy arrays."""
    arena, reads, regions = batch["arena"], batch["reads"], batch["regions"]
    a, b = region_range if region_range else (0, len(regions))
    h = lib().oto_assemble_batch(C.byref(params), abi.ptr(arena), C.c_uint64(arena.size), abi.ptr(reads), C.c_uint32(len(reads)),
                                 abi.ptr(regions), C.c_uint32(len(regions)), C.c_uint32(a), C.c_uint32(b))
    h = C.c_void_p(h)
    try:
        na = lib().oto_result_n_alleles(h)
        sb = lib().oto_result_seq_bytes(h)
        dl = lib().oto_result_dist_len(h)
        res = {
            "regions": np.zeros(len(regions), dtype=abi.region_result_dt),
            "alleles": np.zeros(na, dtype=abi.allele_dt),
            "seqs": np.zeros(max(1, sb), dtype=np.uint8),
            "labels": np.zeros(len(reads), dtype=np.int32),
            "dist": np.zeros(max(1, dl)),
            "dist_off": np.zeros(len(regions) + 1, dtype=np.uint64),
            "bounds": np.zeros(3 * len(regions)),
            "stats": np.zeros(1, dtype=abi.run_stats_dt),
        }
        lib().oto_result_copy(h, abi.ptr(res["regions"]), abi.ptr(res["alleles"]), abi.ptr(res["seqs"]), abi.ptr(res["labels"]),
                              abi.ptr(res["dist"]), abi.ptr(res["dist_off"]), abi.ptr(res["bounds"]), abi.ptr(res["stats"]))
        res["bounds"] = res["bounds"].reshape(-1, 3)
    finally:
        lib().oto_assemble_free(h)
    return res


def allele_seq(res, i):
    a = res["alleles"][i]
    return res["seqs"][int(a["seq_off"]):int(a["seq_off"]) + int(a["seq_len"])].tobytes()


_ref_io = None


def ref_io():
    """The reference's own record types + emit code (oracle/_ref/libotter_ref_io.so); None when not built."""
    global _ref_io
    if _ref_io is None and os.path.exists(REF_IO_SO):
        _ref_io = C.CDLL(REF_IO_SO)
        _ref_io.ref_emit_alleles.restype = C.c_uint64
    return _ref_io


def emit_alleles(beds, chr_arena, res, read_group="", fasta=False, which="oracle"):
    """Record text of a collected batch: the oracle's restatement, or the reference's own stdout_sam / stdout_fa."""
    if which == "ref":
        f = ref_io().ref_emit_alleles
    else:
        f = lib().oto_emit_alleles
        f.restype = C.c_uint64
    args = [abi.ptr(beds), abi.ptr(chr_arena, C.c_char_p), C.c_uint32(len(beds)), abi.ptr(res["regions"]), abi.ptr(res["alleles"]),
            abi.ptr(res["seqs"]), C.c_char_p(read_group.encode()), C.c_int(1 if fasta else 0)]
    n = f(*args, None, C.c_uint64(0))
    out = np.zeros(max(1, n), dtype=np.uint8)
    f(*args, abi.ptr(out, C.c_char_p), C.c_uint64(out.size))
    return out[:n].tobytes()


def emit_sam_header(targets, read_group="", offset_l=0, offset_r=0):
    f = lib().oto_emit_sam_header
    f.restype = C.c_uint64
    names = b"".join(t[0].encode() for t in targets) + b"\0"
    off = np.cumsum([0] + [len(t[0].encode()) for t in targets[:-1]]).astype(np.uint64) if targets else np.zeros(0, np.uint64)
    ln = np.array([len(t[0].encode()) for t in targets], dtype=np.uint32)
    tl = np.array([t[1] for t in targets], dtype=np.uint64)
    out = np.zeros(64 + sum(40 + len(t[0]) for t in targets) + len(read_group), dtype=np.uint8)
    n = f(C.c_char_p(names), abi.ptr(off), abi.ptr(ln), abi.ptr(tl), C.c_uint32(len(targets)), C.c_char_p(read_group.encode()),
          C.c_int32(offset_l), C.c_int32(offset_r), abi.ptr(out, C.c_char_p), C.c_uint64(out.size))
    return out[:n].tobytes()


def _names(items):
    raw = [x.encode("latin-1") for x in items]
    off = np.cumsum([0] + [len(x) for x in raw[:-1]]).astype(np.uint64) if raw else np.zeros(0, np.uint64)
    ln = np.array([len(x) for x in raw], dtype=np.uint32)
    return b"".join(raw) + b"\0", off, ln


def emit_vcf_header(targets, samples):
    f = lib().oto_emit_vcf_header
    f.restype = C.c_uint64
    tn, toff, tln = _names([t[0] for t in targets])
    tl = np.array([t[1] for t in targets], dtype=np.uint64)
    sn, soff, sln = _names(samples)
    out = np.zeros(4096 + 64 * (len(targets) + len(samples)) + len(tn) + len(sn), dtype=np.uint8)
    n = f(C.c_char_p(tn), abi.ptr(toff), abi.ptr(tln), abi.ptr(tl), C.c_uint32(len(targets)), C.c_char_p(sn), abi.ptr(soff), abi.ptr(sln),
          C.c_uint32(len(samples)), abi.ptr(out, C.c_char_p), C.c_uint64(out.size))
    return out[:n].tobytes()


def emit_vcf_lines(beds, chr_arena, blk, n_samples, gt, hsd, n_gt, reps, offset_l, offset_r):
    f = lib().oto_emit_vcf_lines
    f.restype = C.c_uint64
    gt = np.ascontiguousarray(gt, dtype=np.int32); hsd = np.ascontiguousarray(hsd, dtype=np.float64)
    n_gt = np.ascontiguousarray(n_gt, dtype=np.int32); reps = np.ascontiguousarray(reps, dtype=np.int32)
    out = np.zeros(int(blk["alleles"]["seq_len"].sum()) * 2 + 256 * (len(beds) + 1) * (n_samples + 2), dtype=np.uint8)
    n = f(abi.ptr(beds), abi.ptr(chr_arena, C.c_char_p), C.c_uint32(len(beds)), abi.ptr(blk["first_allele"]), abi.ptr(blk["alleles"]), abi.ptr(blk["arena"]),
          C.c_uint32(n_samples), abi.ptr(gt), abi.ptr(hsd), abi.ptr(n_gt), abi.ptr(reps), C.c_int32(offset_l), C.c_int32(offset_r),
          abi.ptr(out, C.c_char_p), C.c_uint64(out.size))
    assert n <= out.size
    return out[:n].tobytes()


def realign_batch(params, batch):
    """local_realignment alone (oto_realign_batch): the read descriptors after the flank rescue."""
    out = np.zeros(len(batch["reads"]), dtype=abi.read_dt)
    lib().oto_realign_batch(C.byref(params), abi.ptr(batch["arena"]), C.c_uint64(batch["arena"].size), abi.ptr(batch["reads"]),
                            C.c_uint32(len(batch["reads"])), abi.ptr(batch["regions"]), C.c_uint32(len(batch["regions"])), abi.ptr(out))
    return out
