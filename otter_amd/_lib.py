"""ctypes binding of libotter_gpu.so (the C-ABI of include/otter_gpu.h).  There is no CPU fallback:
if the HIP library is missing or no device is usable, every entry point raises."""
import ctypes as C
import os
import sys
import numpy as np
from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libotter_gpu.so")

EXPORTS = [
    "otg_params_default", "otg_create", "otg_destroy", "otg_trim", "otg_last_error", "otg_device_count", "otg_exp_variant", "otg_set_heuristic",
    "otg_edit_distance_batch", "otg_affine_align_batch", "otg_cluster_batch", "otg_poa_consensus_batch",
    "otg_genotype_cluster_batch", "otg_last_kernel_ms", "otg_assemble_submit", "otg_assemble_run", "otg_assemble_result_sizes",
    "otg_assemble_collect", "otg_assemble_device_results", "otg_assemble_stats", "otg_assemble_realign", "otg_assemble_collect_reads",
    "otg_emit_alleles", "otg_emit_sam_header",
    "otg_bam_open", "otg_bam_close", "otg_bam_n_targets", "otg_bam_target", "otg_ingest_regions",
    "otg_ingest_regions_named", "otg_emit_reads", "otg_parse_bed_file", "otg_fasta_open", "otg_fasta_close", "otg_fasta_n_seqs",
    "otg_fasta_seq", "otg_fasta_fetch", "otg_fasta_region_flanks",
    "otg_bam_sample_index", "otg_bam_sample", "otg_ingest_alleles", "otg_emit_vcf_header", "otg_emit_vcf_lines", "otg_emit_genotype_lengths", "otg_assemble_files", "otg_assemble_files_release", "otg_assemble_batch_plan", "otg_genotype_files", "otg_wgat",
    "otg_comm_unique_id", "otg_comm_create", "otg_comm_destroy", "otg_gather_sizes", "otg_gather_records",
]

_lib = None


class OtterGpuError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OtterGpuError("%s is missing: build it with `python -m otter_amd.build` (hipcc, gfx950). "
                                "otter_amd has no CPU fallback." % LIB_PATH)
        # A process that also uses PyTorch must load PyTorch's HIP runtime FIRST: torch wheels bundle their own
        # libamdhip64 under the same soname, and whichever copy is mapped first serves both; with the system copy first
        # torch.cuda finds no device (and RCCL cannot start).  The library itself does not depend on torch.
        if "torch" not in sys.modules and os.environ.get("OTG_NO_TORCH_PRELOAD") is None:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _lib = C.CDLL(LIB_PATH)
        _lib.otg_last_error.restype = C.c_char_p
        _lib.otg_last_error.argtypes = [C.c_void_p]
        _lib.otg_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        _lib.otg_destroy.argtypes = [C.c_void_p]
        _lib.otg_trim.argtypes = [C.c_void_p]
    return _lib


class Context:
    """One otg_ctx = one GPU + one stream (single-threaded, like a reference worker thread's aligner pair,
    src/assemble.cpp:45-50)."""

    def __init__(self, device=0):
        L = load()
        h = C.c_void_p()
        rc = L.otg_create(int(device), C.byref(h))
        if rc != 0:
            raise OtterGpuError("otg_create(device=%d) failed (%d): %s" % (device, rc, (L.otg_last_error(None) or b"").decode()))
        self._h = h
        self._L = L
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.otg_destroy(self._h)
            self._h = None

    def trim(self):
        """Release the aligners' scratch workspaces (otg_trim); they come back with the next call that needs them."""
        self._check(self._L.otg_trim(self._h), "otg_trim")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise OtterGpuError("%s failed (%d): %s" % (what, rc, (self._L.otg_last_error(self._h) or b"").decode()))

    @property
    def exp_variant(self):
        return self._L.otg_exp_variant(self._h)

    # ------------------------------------------------------------------ L1
    def set_heuristic(self, strategy=abi.OTG_HEURISTIC_NONE, min_wavefront_length=10, max_distance_threshold=50, steps_between_cutoffs=1):
        """Heuristic of the L1 aligner calls on this context: wfa::WFAligner::setHeuristicNone / setHeuristicWFadaptive."""
        self._check(self._L.otg_set_heuristic(self._h, int(strategy), int(min_wavefront_length), int(max_distance_threshold), int(steps_between_cutoffs)),
                    "otg_set_heuristic")

    def edit_distance_batch(self, arena, tasks, want_cells=False):
        n = len(tasks)
        scores = np.zeros(n, dtype=np.int32)
        cells = np.zeros(n, dtype=np.uint64)
        rc = self._L.otg_edit_distance_batch(self._h, abi.ptr(arena), C.c_uint64(arena.size), abi.ptr(tasks), C.c_uint32(n),
                                             abi.ptr(scores), abi.ptr(cells))
        self._check(rc, "otg_edit_distance_batch")
        return (scores, cells) if want_cells else scores

    def affine_align_batch(self, arena, tasks, x=4, o=6, e=2, want_cells=False):
        n = len(tasks)
        scores = np.zeros(n, dtype=np.int32)
        off = np.zeros(n, dtype=np.uint64)
        ln = np.zeros(n, dtype=np.uint32)
        cap = int(tasks["pattern_len"].astype(np.int64).sum() + tasks["text_len"].astype(np.int64).sum()) + 64
        out = np.zeros(cap, dtype=np.uint8)
        used = C.c_uint64(0)
        cells = np.zeros(n, dtype=np.uint64)
        rc = self._L.otg_affine_align_batch(self._h, abi.ptr(arena), C.c_uint64(arena.size), abi.ptr(tasks), C.c_uint32(n),
                                            int(x), int(o), int(e), abi.ptr(scores), abi.ptr(off), abi.ptr(ln), abi.ptr(out),
                                            C.c_uint64(cap), C.byref(used), abi.ptr(cells))
        self._check(rc, "otg_affine_align_batch")
        cigs = [out[int(off[i]):int(off[i]) + int(ln[i])].tobytes() for i in range(n)]
        return (scores, cigs, cells) if want_cells else (scores, cigs)

    # ------------------------------------------------------------------ L2
    def cluster_batch(self, params, dist, dist_off, read_len, len_off, n_valid):
        nreg = len(n_valid)
        labels = np.full(int(read_len.size), -1, dtype=np.int32)
        ic = np.zeros(nreg, dtype=np.int32)
        fc = np.zeros(nreg, dtype=np.int32)
        bounds = np.full(3 * nreg, np.nan)
        rc = self._L.otg_cluster_batch(self._h, C.byref(params), abi.ptr(dist), abi.ptr(dist_off), abi.ptr(read_len), abi.ptr(len_off),
                                       abi.ptr(n_valid), C.c_uint32(nreg), abi.ptr(labels), abi.ptr(ic), abi.ptr(fc), abi.ptr(bounds))
        self._check(rc, "otg_cluster_batch")
        return labels, ic, fc, bounds.reshape(-1, 3)

    def poa_consensus_batch(self, seq_arena, cig_arena, members, graphs):
        ng = len(graphs)
        off = np.zeros(ng, dtype=np.uint64)
        ln = np.zeros(ng, dtype=np.uint32)
        cap = int(seq_arena.size) * 2 + 1024
        out = np.zeros(cap, dtype=np.uint8)
        used = C.c_uint64(0)
        rc = self._L.otg_poa_consensus_batch(self._h, abi.ptr(seq_arena), C.c_uint64(seq_arena.size), abi.ptr(cig_arena),
                                             C.c_uint64(cig_arena.size), abi.ptr(members), C.c_uint32(len(members)),
                                             abi.ptr(graphs), C.c_uint32(ng), abi.ptr(off), abi.ptr(ln), abi.ptr(out),
                                             C.c_uint64(cap), C.byref(used))
        self._check(rc, "otg_poa_consensus_batch")
        return [out[int(off[i]):int(off[i]) + int(ln[i])].tobytes() for i in range(ng)]

    def genotype_cluster_batch(self, params, arena, seq_off, seq_len, first_allele, n_alleles):
        nreg = len(n_alleles)
        na = len(seq_off)
        gt, gl, gk = (np.zeros(na, dtype=np.int32) for _ in range(3))
        hsd = np.zeros(na)
        ngt = np.zeros(nreg, dtype=np.int32)
        reps = np.zeros(na, dtype=np.int32)
        rc = self._L.otg_genotype_cluster_batch(self._h, C.byref(params), abi.ptr(arena), C.c_uint64(arena.size), abi.ptr(seq_off),
                                                abi.ptr(seq_len), abi.ptr(first_allele), abi.ptr(n_alleles), C.c_uint32(nreg),
                                                abi.ptr(gt), abi.ptr(gl), abi.ptr(gk), abi.ptr(hsd), abi.ptr(ngt), abi.ptr(reps))
        self._check(rc, "otg_genotype_cluster_batch")
        return gt, gl, gk, hsd, ngt, reps

    def last_kernel_ms(self):
        """HIP-event time of the kernels of the latest genotype_cluster_batch (otg_last_kernel_ms)."""
        ms = C.c_double(0.0)
        self._check(self._L.otg_last_kernel_ms(self._h, C.byref(ms)), "otg_last_kernel_ms")
        return float(ms.value)

    # ------------------------------------------------------------------ L3
    def assemble_submit(self, params, batch, region_range=None):
        arena, reads, regions = batch["arena"], batch["reads"], batch["regions"]
        if region_range is not None:
            a, b = region_range
            regions = np.ascontiguousarray(regions[a:b])
        self._n_regions = len(regions)
        self._n_reads = len(reads)
        self._first_read = int(regions["first_read"].min()) if len(regions) else 0
        rc = self._L.otg_assemble_submit(self._h, C.byref(params), abi.ptr(arena), C.c_uint64(arena.size), abi.ptr(reads),
                                         C.c_uint32(len(reads)), abi.ptr(regions), C.c_uint32(len(regions)))
        self._check(rc, "otg_assemble_submit")

    def assemble_run(self):
        self._check(self._L.otg_assemble_run(self._h), "otg_assemble_run")

    def realign_reads(self, params, batch):
        """local_realignment alone (`--reads-only -r`): the read descriptors after the flank rescue (otg_assemble_realign)."""
        self.assemble_submit(params, batch)
        self._check(self._L.otg_assemble_realign(self._h), "otg_assemble_realign")
        out = np.zeros(len(batch["reads"]), dtype=abi.read_dt)
        self._check(self._L.otg_assemble_collect_reads(self._h, abi.ptr(out), C.c_uint32(len(out))), "otg_assemble_collect_reads")
        return out

    def assemble_collect(self):
        na = C.c_uint32(0)
        sb = C.c_uint64(0)
        self._check(self._L.otg_assemble_result_sizes(self._h, C.byref(na), C.byref(sb)), "otg_assemble_result_sizes")
        res = {
            "regions": np.zeros(self._n_regions, dtype=abi.region_result_dt),
            "alleles": np.zeros(na.value, dtype=abi.allele_dt),
            "seqs": np.zeros(max(1, sb.value), dtype=np.uint8),
            "labels": np.zeros(self._n_reads, dtype=np.int32),
        }
        rc = self._L.otg_assemble_collect(self._h, abi.ptr(res["regions"]), abi.ptr(res["alleles"]), C.c_uint32(na.value),
                                          abi.ptr(res["seqs"]), C.c_uint64(res["seqs"].size), abi.ptr(res["labels"]))
        self._check(rc, "otg_assemble_collect")
        return res

    def assemble_device_results(self):
        """Device-resident results of the last run as zero-copy torch uint8 tensors on this context's GPU
        (valid until the next submit / run): {"regions", "alleles", "seqs"} + counts.  Used by the multi-GPU
        gather so that allele records travel GPU -> GPU (RCCL) without a host round trip."""
        import torch
        na = C.c_uint32(0)
        sb = C.c_uint64(0)
        self._check(self._L.otg_assemble_result_sizes(self._h, C.byref(na), C.byref(sb)), "otg_assemble_result_sizes")
        pr, pa, ps = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self._L.otg_assemble_device_results(self._h, C.byref(pr), C.byref(pa), C.byref(ps)), "otg_assemble_device_results")
        dev = torch.device("cuda", self.device)

        class _Span:          # minimal __cuda_array_interface__ carrier: torch.as_tensor wraps it without copying
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None}

        def wrap(ptr, nbytes):
            if not ptr or nbytes == 0:
                return torch.zeros(0, dtype=torch.uint8, device=dev)
            return torch.as_tensor(_Span(ptr, int(nbytes)), device=dev)

        return {"regions": wrap(pr.value, self._n_regions * abi.region_result_dt.itemsize),
                "alleles": wrap(pa.value, na.value * abi.allele_dt.itemsize),
                "seqs": wrap(ps.value, sb.value), "n_alleles": int(na.value), "n_regions": int(self._n_regions)}

    def assemble_stats(self):
        st = np.zeros(1, dtype=abi.run_stats_dt)
        self._check(self._L.otg_assemble_stats(self._h, abi.ptr(st)), "otg_assemble_stats")
        return st[0]

    def assemble(self, params, batch, region_range=None):
        self.assemble_submit(params, batch, region_range)
        self.assemble_run()
        return self.assemble_collect()


class Comm:
    """An RCCL communicator of the library (otg_comm_create): one per process and GPU, for Context.gather_records_rccl."""

    def __init__(self, device, rank, world, uid):
        L = load()
        h = C.c_void_p()
        L.otg_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        rc = L.otg_comm_create(int(device), int(rank), int(world), uid, C.byref(h))
        if rc != 0:
            raise OtterGpuError("otg_comm_create failed (%d): %s" % (rc, (L.otg_last_error(None) or b"").decode()))
        self._h, self._L, self.rank, self.world = h, L, int(rank), int(world)
        L.otg_comm_destroy.argtypes = [C.c_void_p]

    @staticmethod
    def unique_id():
        L = load()
        buf = C.create_string_buffer(128)
        rc = L.otg_comm_unique_id(buf)
        if rc != 0:
            raise OtterGpuError("otg_comm_unique_id failed (%d): %s" % (rc, (L.otg_last_error(None) or b"").decode()))
        return buf.raw

    def close(self):
        if getattr(self, "_h", None):
            self._L.otg_comm_destroy(self._h)
            self._h = None

    def gather_records(self, ctx):
        """End-of-run gather of ctx's last results to rank 0 (otg_gather_sizes + otg_gather_records).  Rank 0 gets
        {"regions", "alleles", "seqs", "counts"} of the whole job in rank order; the other ranks get {"counts"}."""
        L = self._L
        counts = np.zeros(3 * self.world, dtype=np.uint64)
        L.otg_gather_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.otg_gather_records.argtypes = [C.c_void_p] * 6
        ctx._check(L.otg_gather_sizes(ctx._h, self._h, abi.ptr(counts)), "otg_gather_sizes")
        if self.rank != 0:
            ctx._check(L.otg_gather_records(ctx._h, self._h, abi.ptr(counts), None, None, None), "otg_gather_records")
            return {"counts": counts.reshape(-1, 3)}
        tot = counts.reshape(-1, 3).sum(axis=0)
        res = {"regions": np.zeros(int(tot[0]), dtype=abi.region_result_dt), "alleles": np.zeros(int(tot[1]), dtype=abi.allele_dt),
               "seqs": np.zeros(max(1, int(tot[2])), dtype=np.uint8), "counts": counts.reshape(-1, 3)}
        ctx._check(L.otg_gather_records(ctx._h, self._h, abi.ptr(counts), abi.ptr(res["regions"]), abi.ptr(res["alleles"]), abi.ptr(res["seqs"])), "otg_gather_records")
        return res


def device_count():
    return load().otg_device_count()


def emit_alleles(beds, chr_arena, res, read_group="", fasta=False):
    """Text of the allele records of a collected batch exactly as `otter assemble` prints them (otg_emit_alleles;
    src/assemble.cpp:143-149).  res: dict with "regions", "alleles", "seqs" (Context.assemble_collect)."""
    L = load()
    n = C.c_uint64(0)
    args = [abi.ptr(beds), abi.ptr(chr_arena, C.c_char_p), C.c_uint32(len(beds)), abi.ptr(res["regions"]), abi.ptr(res["alleles"]),
            abi.ptr(res["seqs"]), C.c_char_p(read_group.encode()), C.c_int(1 if fasta else 0)]
    rc = L.otg_emit_alleles(*args, None, C.c_uint64(0), C.byref(n))
    if rc not in (0, abi.OTG_ERR_CAPACITY):
        raise OtterGpuError("otg_emit_alleles failed (%d): %s" % (rc, (L.otg_last_error(None) or b"").decode()))
    out = np.zeros(max(1, n.value), dtype=np.uint8)
    rc = L.otg_emit_alleles(*args, abi.ptr(out, C.c_char_p), C.c_uint64(out.size), C.byref(n))
    if rc != 0:
        raise OtterGpuError("otg_emit_alleles failed (%d)" % rc)
    return out[:n.value].tobytes()


def emit_sam_header(targets, read_group="", offset_l=0, offset_r=0):
    """targets: list of (name, length) -> the @SQ/@RG/@PG header lines of src/assemble.cpp:167-177."""
    L = load()
    names = b"".join(t[0].encode() for t in targets) + b"\0"
    off = np.cumsum([0] + [len(t[0].encode()) for t in targets[:-1]]).astype(np.uint64) if targets else np.zeros(0, np.uint64)
    ln = np.array([len(t[0].encode()) for t in targets], dtype=np.uint32)
    tl = np.array([t[1] for t in targets], dtype=np.uint64)
    out = np.zeros(64 + sum(40 + len(t[0]) for t in targets) + len(read_group), dtype=np.uint8)
    n = C.c_uint64(0)
    rc = L.otg_emit_sam_header(C.c_char_p(names), abi.ptr(off), abi.ptr(ln), abi.ptr(tl), C.c_uint32(len(targets)), C.c_char_p(read_group.encode()),
                               C.c_int32(offset_l), C.c_int32(offset_r), abi.ptr(out, C.c_char_p), C.c_uint64(out.size), C.byref(n))
    if rc != 0:
        raise OtterGpuError("otg_emit_sam_header failed (%d)" % rc)
    return out[:n.value].tobytes()


class Bam:
    """An indexed BAM (otg_bam_open: <path> + <path>.bai) as the source of region batches (otg_ingest_regions)."""

    def __init__(self, path):
        L = load()
        h = C.c_void_p()
        rc = L.otg_bam_open(C.c_char_p(path.encode()), C.byref(h))
        if rc != 0:
            raise OtterGpuError("otg_bam_open failed (%d): %s" % (rc, (L.otg_last_error(None) or b"").decode()))
        self._h, self._L, self._path = h, L, path
        L.otg_bam_target.restype = C.c_char_p
        L.otg_bam_n_targets.restype = C.c_uint32

    def close(self):
        if getattr(self, "_h", None):
            self._L.otg_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def targets(self):
        out = []
        for i in range(self._L.otg_bam_n_targets(self._h)):
            ln = C.c_uint64(0)
            nm = self._L.otg_bam_target(self._h, C.c_uint32(i), C.byref(ln))
            out.append((nm.decode(), int(ln.value)))
        return out

    def sample_index(self):
        """SampleIndex of an allele BAM (otg_bam_sample_index): ([sample names in header order], offset_l, offset_r)."""
        n, ol, orr = C.c_uint32(0), C.c_int32(0), C.c_int32(0)
        rc = self._L.otg_bam_sample_index(self._h, C.byref(n), C.byref(ol), C.byref(orr))
        if rc != 0:
            raise OtterGpuError("otg_bam_sample_index failed (%d): %s" % (rc, (self._L.otg_last_error(None) or b"").decode()))
        self._L.otg_bam_sample.restype = C.c_char_p
        return [self._L.otg_bam_sample(self._h, C.c_uint32(i)).decode("latin-1") for i in range(n.value)], int(ol.value), int(orr.value)

    def ingest_alleles(self, regions, reference=None, threads=1):
        """The allele records of each region (otg_ingest_alleles) -> {"alleles", "first_allele", "arena"}; with reference (a Fasta)
        the reference allele is appended to every non-empty region, as `otter genotype -r` does."""
        beds, carena = regions if isinstance(regions, tuple) else abi.make_beds(regions)
        fsz = os.path.getsize(self._path)
        cap_n, cap_a = max(1024, fsz // 32), max(1 << 20, 8 * fsz)
        first = np.zeros(len(beds) + 1, dtype=np.uint32)
        while True:
            alleles = np.zeros(cap_n, dtype=abi.allele_dt)
            arena = np.zeros(cap_a, dtype=np.uint8)
            used, na = C.c_uint64(0), C.c_uint32(0)
            rc = self._L.otg_ingest_alleles(self._h, abi.ptr(beds), abi.ptr(carena, C.c_char_p), C.c_uint32(len(beds)), C.c_int32(threads),
                                            reference._h if reference is not None else None, abi.ptr(arena), C.c_uint64(cap_a), C.byref(used),
                                            abi.ptr(alleles), C.c_uint32(cap_n), C.byref(na), abi.ptr(first))
            if rc == abi.OTG_ERR_CAPACITY:
                cap_n, cap_a = max(cap_n, na.value + 16), max(cap_a, used.value + 4096)
                continue
            if rc != 0:
                raise OtterGpuError("otg_ingest_alleles failed (%d): %s" % (rc, (self._L.otg_last_error(None) or b"").decode()))
            return {"alleles": np.ascontiguousarray(alleles[:na.value]), "first_allele": first, "arena": np.ascontiguousarray(arena[:used.value + 64])}

    def ingest(self, regions, offset_l=0, offset_r=0, mapq=0, nonprimary=False, omit_nonspanning=False, read_quality=0.0, threads=1,
               names=False):
        """regions: list of (chr, start, end), or the (beds, chr_arena) pair of parse_bed_file -> batch dict {"arena", "reads",
        "regions"} for Context.assemble_submit (reference flanks, needed only with -r: Fasta.region_flanks appends them).
        names=True adds "meta" / "names" (read name and rq per read, what --reads-only prints)."""
        beds, carena = regions if isinstance(regions, tuple) else abi.make_beds(regions)
        opts = np.zeros(1, dtype=abi.ingest_opts_dt)
        opts[0]["offset_l"] = offset_l; opts[0]["offset_r"] = offset_r; opts[0]["mapq"] = mapq
        opts[0]["nonprimary"] = int(nonprimary); opts[0]["omit_nonspanning"] = int(omit_nonspanning); opts[0]["read_quality"] = read_quality; opts[0]["threads"] = threads
        regs = np.zeros(len(beds), dtype=abi.region_dt)
        # generous first guess from the file size (untouched pages cost nothing), exact retry if it was too small
        fsz = os.path.getsize(self._path)
        cap_r, cap_a = max(1024, fsz // 64), max(1 << 20, 8 * fsz)
        cap_n = max(4096, fsz // 16) if names else 0
        while True:
            reads = np.empty(cap_r, dtype=abi.read_dt)
            arena = np.empty(cap_a, dtype=np.uint8)
            meta = np.empty(cap_r, dtype=abi.read_meta_dt) if names else None
            narena = np.empty(cap_n, dtype=np.uint8) if names else None
            used, nr, nused = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
            rc = self._L.otg_ingest_regions_named(self._h, abi.ptr(beds), abi.ptr(carena, C.c_char_p), C.c_uint32(len(beds)), abi.ptr(opts),
                                                  abi.ptr(arena), C.c_uint64(cap_a), C.byref(used), abi.ptr(reads), C.c_uint32(cap_r), C.byref(nr),
                                                  abi.ptr(regs), abi.ptr(meta) if names else None, abi.ptr(narena, C.c_char_p) if names else None,
                                                  C.c_uint64(cap_n), C.byref(nused) if names else None)
            if rc == abi.OTG_ERR_CAPACITY:
                cap_r, cap_a = max(cap_r, nr.value + 16), max(cap_a, used.value + 64 * (nr.value + 2) + 4096)
                cap_n = max(cap_n, nused.value + 64) if names else 0
                continue
            if rc != 0:
                raise OtterGpuError("otg_ingest_regions failed (%d): %s" % (rc, (self._L.otg_last_error(None) or b"").decode()))
            arena[used.value:used.value + 64] = 0
            if names:
                return {"arena": np.ascontiguousarray(arena[:used.value + 64]), "reads": np.ascontiguousarray(reads[:nr.value]), "regions": regs,
                        "meta": np.ascontiguousarray(meta[:nr.value]), "names": np.ascontiguousarray(narena[:nused.value])}
            return {"arena": np.ascontiguousarray(arena[:used.value + 64]), "reads": np.ascontiguousarray(reads[:nr.value]), "regions": regs}


def _err(L):
    return (L.otg_last_error(None) or b"").decode()


def parse_bed_file(path):
    """parse_bed_file (src/anbed.cpp:65-80) through the library -> (beds, chr_arena, n_skipped); beds is an abi.bed_dt array."""
    L = load()
    cap_b, cap_c = 1024, 1 << 14
    while True:
        beds = np.zeros(cap_b, dtype=abi.bed_dt)
        carena = np.zeros(cap_c, dtype=np.uint8)
        n, used, skipped = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0)
        rc = L.otg_parse_bed_file(C.c_char_p(path.encode()), abi.ptr(beds), C.c_uint32(cap_b), C.byref(n), abi.ptr(carena, C.c_char_p),
                                  C.c_uint64(cap_c), C.byref(used), C.byref(skipped))
        if rc == abi.OTG_ERR_CAPACITY:
            cap_b, cap_c = max(cap_b, n.value), max(cap_c, used.value)
            continue
        if rc != 0:
            raise OtterGpuError("otg_parse_bed_file failed (%d): %s" % (rc, _err(L)))
        return np.ascontiguousarray(beds[:n.value]), np.ascontiguousarray(carena[:max(1, used.value)]), int(skipped.value)


def bed_tuples(beds, chr_arena):
    """(chr, start, end) per region with the coordinates as BED::toString prints them (unsigned 32-bit, src/anbed.hpp:16-17)."""
    raw = chr_arena.tobytes()
    return [(raw[int(b["chr_off"]):int(b["chr_off"]) + int(b["chr_len"])].decode("latin-1"), int(b["start"]) & 0xffffffff, int(b["end"]) & 0xffffffff)
            for b in beds]


def emit_reads(beds, chr_arena, batch, read_group="", fasta=False, max_cov=-1):
    """The --reads-only records of an ingested batch (Bam.ingest(..., names=True)) as bytes (otg_emit_reads)."""
    L = load()
    meta, names = batch.get("meta"), batch.get("names")
    cap = int(batch["reads"]["seq_len"].sum()) * 2 + 512 * (len(batch["reads"]) + 1)
    while True:
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_uint64(0)
        rc = L.otg_emit_reads(abi.ptr(beds), abi.ptr(chr_arena, C.c_char_p), C.c_uint32(len(beds)), abi.ptr(batch["regions"]),
                              abi.ptr(batch["reads"]), abi.ptr(batch["arena"]), abi.ptr(meta) if meta is not None else None,
                              abi.ptr(names, C.c_char_p) if names is not None else None, C.c_char_p(read_group.encode()),
                              C.c_int(int(fasta)), C.c_int32(max_cov), abi.ptr(out, C.c_char_p), C.c_uint64(cap), C.byref(n))
        if rc == abi.OTG_ERR_CAPACITY:
            cap = n.value
            continue
        if rc != 0:
            raise OtterGpuError("otg_emit_reads failed (%d): %s" % (rc, _err(L)))
        return out[:n.value].tobytes()


class Fasta:
    """An indexed, uncompressed FASTA (otg_fasta_open) — the source of the reference flanks of local_realignment (-r)."""

    def __init__(self, path):
        L = load()
        h = C.c_void_p()
        rc = L.otg_fasta_open(C.c_char_p(path.encode()), C.byref(h))
        if rc != 0:
            raise OtterGpuError("otg_fasta_open failed (%d): %s" % (rc, _err(L)))
        self._h, self._L = h, L
        L.otg_fasta_seq.restype = C.c_char_p
        L.otg_fasta_n_seqs.restype = C.c_uint32

    def close(self):
        if getattr(self, "_h", None):
            self._L.otg_fasta_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seqs(self):
        out = []
        for i in range(self._L.otg_fasta_n_seqs(self._h)):
            ln = C.c_int64(0)
            nm = self._L.otg_fasta_seq(self._h, C.c_uint32(i), C.byref(ln))
            out.append((nm.decode(), int(ln.value)))
        return out

    def fetch(self, chr_, beg, end_inclusive):
        c = chr_.encode()
        cap = max(16, abs(int(end_inclusive) - int(beg)) + 2)
        out = C.create_string_buffer(cap)
        n = C.c_uint64(0)
        rc = self._L.otg_fasta_fetch(self._h, C.c_char_p(c), C.c_uint32(len(c)), C.c_int32(beg), C.c_int32(end_inclusive), out,
                                     C.c_uint64(cap), C.byref(n))
        if rc != 0:
            raise OtterGpuError("otg_fasta_fetch failed (%d): %s" % (rc, _err(self._L)))
        return out.raw[:n.value]

    def region_flanks(self, beds, chr_arena, batch, flank=100, offset_l=0, offset_r=0):
        """Appends the two flanks of every region to batch["arena"] and fills the flank fields of batch["regions"] (in place)."""
        arena = batch["arena"]
        base = len(arena)
        cap = base + 2 * (int(flank) + 1) * len(beds) + 128
        big = np.zeros(cap, dtype=np.uint8)
        big[:base] = arena
        used = C.c_uint64(base)
        rc = self._L.otg_fasta_region_flanks(self._h, abi.ptr(beds), abi.ptr(chr_arena, C.c_char_p), C.c_uint32(len(beds)), C.c_int32(offset_l),
                                             C.c_int32(offset_r), C.c_int32(flank), abi.ptr(big), C.c_uint64(cap), C.byref(used),
                                             abi.ptr(batch["regions"]))
        if rc != 0:
            raise OtterGpuError("otg_fasta_region_flanks failed (%d): %s" % (rc, _err(self._L)))
        batch["arena"] = np.ascontiguousarray(big[:used.value + 64])
        return batch


def _emit(call, cap):
    L = load()
    while True:
        out = np.empty(max(cap, 16), dtype=np.uint8)
        n = C.c_uint64(0)
        rc = call(abi.ptr(out, C.c_char_p), C.c_uint64(out.size), C.byref(n))
        if rc == abi.OTG_ERR_CAPACITY:
            cap = n.value
            continue
        if rc != 0:
            raise OtterGpuError("emit failed (%d): %s" % (rc, _err(L)))
        return out[:n.value].tobytes()


def emit_vcf_header(bam):
    """The VCF header of `otter genotype -r` for an allele BAM (after Bam.sample_index())."""
    L = load()
    return _emit(lambda o, c, n: L.otg_emit_vcf_header(bam._h, o, c, n), 4096)


def emit_vcf_lines(beds, chr_arena, blk, n_samples, gt, hsd, n_gt, reps, offset_l, offset_r):
    """One VCF line per region with alleles; blk = Bam.ingest_alleles(..., reference=...), gt / hsd / n_gt / reps from
    Context.genotype_cluster_batch on the same blocks."""
    L = load()
    gt = np.ascontiguousarray(gt, dtype=np.int32); hsd = np.ascontiguousarray(hsd, dtype=np.float64)
    n_gt = np.ascontiguousarray(n_gt, dtype=np.int32); reps = np.ascontiguousarray(reps, dtype=np.int32)
    cap = int(blk["alleles"]["seq_len"].sum()) * 2 + 256 * (len(beds) + 1) * (n_samples + 2)
    return _emit(lambda o, c, n: L.otg_emit_vcf_lines(abi.ptr(beds), abi.ptr(chr_arena, C.c_char_p), C.c_uint32(len(beds)), abi.ptr(blk["first_allele"]),
                                                      abi.ptr(blk["alleles"]), abi.ptr(blk["arena"]), C.c_uint32(n_samples), abi.ptr(gt), abi.ptr(hsd),
                                                      abi.ptr(n_gt), abi.ptr(reps), C.c_int32(offset_l), C.c_int32(offset_r), o, c, n), cap)


def emit_genotype_lengths(bam, beds, chr_arena, blk, n_samples):
    """`otter genotype` without a reference: region, sample, shorter and longer allele length per line."""
    L = load()
    return _emit(lambda o, c, n: L.otg_emit_genotype_lengths(bam._h, abi.ptr(beds), abi.ptr(chr_arena, C.c_char_p), C.c_uint32(len(beds)),
                                                             abi.ptr(blk["first_allele"]), abi.ptr(blk["alleles"]), C.c_uint32(n_samples), o, c, n),
                 128 * (len(beds) + 1) * (n_samples + 1))


def genotype_blocks(blk):
    """(seq_off, seq_len, first_allele, n_alleles) arrays of an ingested allele batch, the inputs of genotype_cluster_batch."""
    a = blk["alleles"]
    first = blk["first_allele"]
    return (np.ascontiguousarray(a["seq_off"], dtype=np.uint64), np.ascontiguousarray(a["seq_len"], dtype=np.uint32),
            np.ascontiguousarray(first[:-1], dtype=np.uint32), np.ascontiguousarray(np.diff(first.astype(np.int64)), dtype=np.uint32))


def assemble_files(bam, bed, fasta=None, read_group="", is_fasta=False, reads_only=False, params=None, batch_regions=0, devices=None,
                   offset_l=1, offset_r=0, mapq=0, nonprimary=False, omit_nonspanning=False, read_quality=0.0, threads=1):
    """otg_assemble_files: `otter assemble` from files to record text (the library's dispatcher, include/otter_gpu.h).
    Returns (text bytes, stats dict)."""
    L = load()
    job = abi.AssembleJob()
    job.bam_path = bam.encode(); job.bed_path = bed.encode(); job.fasta_path = fasta.encode() if fasta else None
    job.read_group = read_group.encode(); job.is_fasta = int(is_fasta); job.reads_only = int(reads_only)
    job.params = params if params is not None else abi.default_params()
    job.ingest = abi.IngestOpts(offset_l, offset_r, mapq, int(nonprimary), int(omit_nonspanning), threads, read_quality)
    job.batch_regions = batch_regions
    devs = (C.c_int32 * len(devices))(*devices) if devices else None
    job.n_devices = len(devices) if devices else 0
    job.devices = devs
    chunks = []

    def sink(_user, data, n):
        chunks.append(C.string_at(data, n))
        return 0
    cb = abi.WRITE_FN(sink)
    st = abi.JobStats()
    L.otg_assemble_files.argtypes = [C.POINTER(abi.AssembleJob), abi.WRITE_FN, C.c_void_p, C.POINTER(abi.JobStats)]
    rc = L.otg_assemble_files(C.byref(job), cb, None, C.byref(st))
    if rc != 0:
        raise OtterGpuError("otg_assemble_files failed (%d): %s" % (rc, (L.otg_last_error(None) or b"").decode()))
    return b"".join(chunks), {k: getattr(st, k) for k, _ in abi.JobStats._fields_}


def assemble_batch_plan(n_regions, batch_regions=0):
    """Batch sizes otg_assemble_files cuts a shard of n_regions into (otg_assemble_batch_plan)."""
    L = load()
    L.otg_assemble_batch_plan.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
    n = C.c_uint32(0)
    rc = L.otg_assemble_batch_plan(n_regions, batch_regions, None, 0, C.byref(n))
    if rc not in (0, abi.OTG_ERR_CAPACITY):
        raise OtterGpuError("otg_assemble_batch_plan failed (%d)" % rc)
    out = (C.c_uint32 * max(1, n.value))()
    rc = L.otg_assemble_batch_plan(n_regions, batch_regions, out, n.value, C.byref(n))
    if rc != 0:
        raise OtterGpuError("otg_assemble_batch_plan failed (%d)" % rc)
    return [int(out[i]) for i in range(n.value)]


def genotype_files(bam, bed, fasta=None, params=None, threads=1, device=0, batch_regions=0):
    """otg_genotype_files: `otter genotype` from files to text (VCF with a reference FASTA, the two-length table without).
    Returns (text bytes, stats dict)."""
    L = load()
    job = abi.GenotypeJob()
    job.bam_path = bam.encode(); job.bed_path = bed.encode(); job.fasta_path = fasta.encode() if fasta else None
    job.params = params if params is not None else abi.default_params()
    job.threads = threads; job.device = device; job.batch_regions = batch_regions
    chunks = []

    def sink(_user, data, n):
        chunks.append(C.string_at(data, n))
        return 0
    cb = abi.WRITE_FN(sink)
    st = abi.JobStats()
    L.otg_genotype_files.argtypes = [C.POINTER(abi.GenotypeJob), abi.WRITE_FN, C.c_void_p, C.POINTER(abi.JobStats)]
    rc = L.otg_genotype_files(C.byref(job), cb, None, C.byref(st))
    if rc != 0:
        raise OtterGpuError("otg_genotype_files failed (%d): %s" % (rc, (L.otg_last_error(None) or b"").decode()))
    return b"".join(chunks), {k: getattr(st, k) for k, _ in abi.JobStats._fields_}


def assemble_files_release():
    """Frees the contexts (and their HBM workspaces) the dispatcher keeps for the next job of the process (otg_assemble_files_release)."""
    load().otg_assemble_files_release()


def wgat(bam, regions, read_group="", fasta=False, offset_l=1, offset_r=0):
    """otg_wgat: `otter wgat` on an open Bam handle; regions = list of (chr, start, end) or the (beds, chr_arena) pair.  Returns the text."""
    L = load()
    beds, carena = regions if isinstance(regions, tuple) else abi.make_beds(regions)
    chunks = []

    def sink(_user, data, n):
        chunks.append(C.string_at(data, n))
        return 0
    cb = abi.WRITE_FN(sink)
    nrec = C.c_uint64(0)
    L.otg_wgat.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_int, C.c_int32, C.c_int32, abi.WRITE_FN, C.c_void_p, C.POINTER(C.c_uint64)]
    rc = L.otg_wgat(bam._h, abi.ptr(beds), abi.ptr(carena, C.c_char_p), C.c_uint32(len(beds)), read_group.encode(), int(fasta), offset_l, offset_r, cb, None, C.byref(nrec))
    if rc != 0:
        raise OtterGpuError("otg_wgat failed (%d): %s" % (rc, _err(L)))
    return b"".join(chunks), int(nrec.value)
