"""ctypes mirror of include/otter_gpu.h (struct layouts + constants).  Pure data definitions."""
import ctypes as C
import numpy as np

OTG_OK = 0
OTG_ERR_NO_DEVICE = -1
OTG_ERR_ARG = -2
OTG_ERR_HIP = -3
OTG_ERR_CAPACITY = -4
OTG_ERR_FATAL = -5

OTG_REGION_OK = 0
OTG_REGION_SKIP_MAXCOV = 1
OTG_REGION_NO_SPANNING = 2
OTG_REGION_EMPTY = 3
OTG_REGION_HAP_CONFLICT = 4
OTG_REGION_ALIGN_CAPACITY = 5


class otg_params(C.Structure):
    _fields_ = [
        ("max_alleles", C.c_int32), ("ignore_haps", C.c_int32), ("max_cov", C.c_int32), ("flank", C.c_int32),
        ("bandwidth_length", C.c_int32), ("min_cov_fraction2_l", C.c_int32),
        ("mismatch", C.c_int32), ("gap_open", C.c_int32), ("gap_ext", C.c_int32), ("realign", C.c_int32),
        ("bandwidth_short", C.c_double), ("bandwidth_long", C.c_double), ("max_error", C.c_double),
        ("min_cov_fraction", C.c_double), ("min_cov_fraction2_f", C.c_double), ("min_sim", C.c_double),
        ("gt_max_error", C.c_double), ("gt_max_cosdis", C.c_double),
        ("heuristic", C.c_int32), ("heur_min_wavefront_length", C.c_int32), ("heur_max_distance_threshold", C.c_int32),
        ("heur_steps_between_cutoffs", C.c_int32),
    ]


OTG_HEURISTIC_NONE = 0
OTG_HEURISTIC_WFADAPTIVE = 1


def default_params(**kw):
    """Reference CLI defaults (src/command_assemble.cpp:34-45, src/command_genotype.cpp:25-27)."""
    p = otg_params(max_alleles=2, ignore_haps=1, max_cov=200, flank=100, bandwidth_length=500,
                   min_cov_fraction2_l=500, mismatch=4, gap_open=6, gap_ext=2, realign=0,
                   bandwidth_short=0.01, bandwidth_long=0.015, max_error=0.01, min_cov_fraction=0.2,
                   min_cov_fraction2_f=0.1, min_sim=0.9, gt_max_error=0.025, gt_max_cosdis=0.025,
                   heuristic=OTG_HEURISTIC_NONE, heur_min_wavefront_length=10, heur_max_distance_threshold=50, heur_steps_between_cutoffs=1)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


# numpy structured dtypes with the exact C layout (align=True reproduces the C padding)
align_task_dt = np.dtype([
    ("pattern_off", "<u8"), ("text_off", "<u8"), ("pattern_len", "<u4"), ("text_len", "<u4"),
    ("pattern_begin_free", "<i4"), ("pattern_end_free", "<i4"), ("text_begin_free", "<i4"), ("text_end_free", "<i4"),
    ("endsfree", "<i4"), ("_pad", "<i4")], align=True)

read_dt = np.dtype([
    ("seq_off", "<u8"), ("seq_len", "<u4"), ("spanning_l", "u1"), ("spanning_r", "u1"), ("_pad", "<u2"),
    ("ps", "<i4"), ("hp", "<i4"), ("ccoord_first", "<i4"), ("ccoord_second", "<i4")], align=True)

region_dt = np.dtype([
    ("first_read", "<u4"), ("n_reads", "<u4"), ("flank_l_off", "<u8"), ("flank_r_off", "<u8"),
    ("flank_l_len", "<u4"), ("flank_r_len", "<u4")], align=True)

allele_dt = np.dtype([
    ("seq_off", "<u8"), ("seq_len", "<u4"), ("scov", "<i4"), ("acov", "<i4"), ("tcov", "<i4"), ("se", "<f4"),
    ("ic", "<i4"), ("ps", "<i4"), ("hp", "<i4"), ("region", "<u4"), ("label", "<i4")], align=True)

region_result_dt = np.dtype([
    ("first_allele", "<u4"), ("n_alleles", "<u4"), ("status", "<i4"), ("ic", "<i4"), ("fc", "<i4"), ("n_valid", "<i4")],
    align=True)

poa_member_dt = np.dtype([
    ("seq_off", "<u8"), ("seq_len", "<u4"), ("cigar_len", "<u4"), ("cigar_off", "<u8"),
    ("spanning_l", "u1"), ("spanning_r", "u1"), ("_pad", "u1", (6,))], align=True)

poa_graph_dt = np.dtype([
    ("backbone_off", "<u8"), ("backbone_len", "<u4"), ("first_member", "<u4"), ("n_members", "<u4"),
    ("c", "<f4"), ("t", "<f4"), ("_pad", "<u4")], align=True)

run_stats_dt = np.dtype([
    ("n_regions", "<u8"), ("n_regions_ok", "<u8"),
    ("edit_tasks", "<u8"), ("edit_cells", "<u8"), ("edit_seq_bytes", "<u8"),
    ("affine_tasks", "<u8"), ("affine_cells", "<u8"), ("affine_seq_bytes", "<u8"),
    ("allele_bytes", "<u8"), ("algorithmic_bytes", "<u8"),
    ("ms_edit", "<f8"), ("ms_cluster", "<f8"), ("ms_reassign", "<f8"), ("ms_affine", "<f8"), ("ms_poa", "<f8"),
    ("ms_realign", "<f8"), ("ms_total", "<f8"), ("ms_edit_kernel", "<f8"), ("edit_kernel_launches", "<u8"),
    ("ms_affine_kernel", "<f8"), ("affine_kernel_launches", "<u8"), ("affine_visited_cells", "<u8")], align=True)

assert align_task_dt.itemsize == 48
assert read_dt.itemsize == 32
assert region_dt.itemsize == 32
assert allele_dt.itemsize == 48

bed_dt = np.dtype([("chr_off", "<u8"), ("chr_len", "<u4"), ("start", "<i4"), ("end", "<i4"), ("reserved", "<u4")], align=True)
assert bed_dt.itemsize == 24
read_meta_dt = np.dtype([("name_off", "<u8"), ("name_len", "<u4"), ("reserved", "<u4"), ("rq", "<f8")], align=True)
assert read_meta_dt.itemsize == 24


ingest_opts_dt = np.dtype([("offset_l", "<i4"), ("offset_r", "<i4"), ("mapq", "<i4"), ("nonprimary", "<i4"), ("omit_nonspanning", "<i4"),
                           ("threads", "<i4"), ("read_quality", "<f8")], align=True)
assert ingest_opts_dt.itemsize == 32


def make_beds(regions):
    """regions: list of (chr_str, start, end) -> (otg_bed array, chr byte arena)."""
    beds = np.zeros(len(regions), dtype=bed_dt)
    arena = bytearray()
    for i, (c, st, en) in enumerate(regions):
        cb = c.encode() if isinstance(c, str) else bytes(c)
        beds[i]["chr_off"] = len(arena); beds[i]["chr_len"] = len(cb); beds[i]["start"] = np.uint32(st & 0xffffffff).astype(np.int32); beds[i]["end"] = np.uint32(en & 0xffffffff).astype(np.int32)
        arena += cb
    return beds, np.frombuffer(bytes(arena) + b"\0", dtype=np.uint8).copy()
assert region_result_dt.itemsize == 24
assert poa_member_dt.itemsize == 32
assert poa_graph_dt.itemsize == 32


def ptr(a, ctype=C.c_void_p):
    """Pointer to a numpy array's buffer (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctype)


def make_tasks(pairs):
    """pairs: iterable of (pattern_off, pattern_len, text_off, text_len[, (pbf,pef,tbf,tef)])."""
    pairs = list(pairs)
    t = np.zeros(len(pairs), dtype=align_task_dt)
    for i, q in enumerate(pairs):
        t[i]["pattern_off"], t[i]["pattern_len"], t[i]["text_off"], t[i]["text_len"] = q[0], q[1], q[2], q[3]
        if len(q) > 4 and q[4] is not None:
            t[i]["endsfree"] = 1
            (t[i]["pattern_begin_free"], t[i]["pattern_end_free"], t[i]["text_begin_free"], t[i]["text_end_free"]) = q[4]
    return t


def pack_seqs(seqs, pad=64):
    """Concatenate byte strings into one arena; returns (arena u8 array with `pad` slack bytes, offsets, lens)."""
    offs, lens, total = [], [], 0
    for s in seqs:
        offs.append(total)
        lens.append(len(s))
        total += len(s)
    arena = np.zeros(total + pad, dtype=np.uint8)
    pos = 0
    for s in seqs:
        arena[pos:pos + len(s)] = np.frombuffer(s if isinstance(s, (bytes, bytearray)) else s.encode(), dtype=np.uint8)
        pos += len(s)
    return arena, np.asarray(offs, dtype=np.uint64), np.asarray(lens, dtype=np.uint32)


class IngestOpts(C.Structure):
    _fields_ = [("offset_l", C.c_int32), ("offset_r", C.c_int32), ("mapq", C.c_int32), ("nonprimary", C.c_int32), ("omit_nonspanning", C.c_int32),
                ("threads", C.c_int32), ("read_quality", C.c_double)]


class AssembleJob(C.Structure):
    """otg_assemble_job (include/otter_gpu.h)."""
    _fields_ = [("bam_path", C.c_char_p), ("bed_path", C.c_char_p), ("fasta_path", C.c_char_p), ("read_group", C.c_char_p),
                ("is_fasta", C.c_int32), ("reads_only", C.c_int32), ("params", otg_params), ("ingest", IngestOpts),
                ("batch_regions", C.c_uint32), ("n_devices", C.c_int32), ("devices", C.POINTER(C.c_int32))]


class GenotypeJob(C.Structure):
    """otg_genotype_job (include/otter_gpu.h)."""
    _fields_ = [("bam_path", C.c_char_p), ("bed_path", C.c_char_p), ("fasta_path", C.c_char_p), ("params", otg_params),
                ("threads", C.c_int32), ("device", C.c_int32), ("batch_regions", C.c_uint32), ("reserved", C.c_uint32)]


class JobStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_regions", "n_regions_ok", "n_regions_skipped", "n_reads", "n_alleles", "input_bytes", "output_bytes")] + \
               [("n_devices", C.c_uint32), ("reserved", C.c_uint32)] + [(k, C.c_double) for k in ("ms_total", "ms_ingest", "ms_hot_path", "ms_emit")]


WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_char), C.c_uint64)
