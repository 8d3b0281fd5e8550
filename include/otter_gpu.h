/*
 * otter_gpu.h — C-ABI of the MI355X-native drop-in for otter's per-region hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers + sizes, and replaces a named
 * interface of the reference (holstegelab/otter @ 2024_10_08, paths relative to the
 * reference root).  The library behind it (libotter_gpu.so) is hand-written HIP for gfx950;
 * there is NO CPU fallback inside: when no HIP device is usable every call returns
 * OTG_ERR_NO_DEVICE and otg_last_error() says why.
 *
 * Layers
 *   L1  batched aligners        — replace wfa::WFAligner (WFA2-lib, absent submodule) as used at
 *                                 src/analignments.cpp:25,31,37,70-71,88-97,268-280
 *   L2  per-region operators    — replace DistMatrix/otter_hclust/PPOA/anallele_cluster
 *                                 (src/andistmat.cpp, src/otterclust.cpp:20-320,463-527, src/anppoa.hpp)
 *   L3  region-batch pipeline   — replaces the five calls inside the region loop of
 *                                 assemble_process (src/assemble.cpp:74,126,129,137,141)
 *
 * Ownership: the caller owns every input buffer until the call returns (the library copies
 * to HBM); output buffers are caller-allocated.  No C++ types or exceptions cross the ABI.
 * Threading: an otg_ctx is single-threaded (one per host worker / GPU); distinct contexts are
 * independent.  Results are independent of batch composition.
 */
#ifndef OTTER_GPU_H
#define OTTER_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status codes */
#define OTG_OK               0
#define OTG_ERR_NO_DEVICE   -1   /* no usable HIP device / kernels missing                 */
#define OTG_ERR_ARG         -2   /* bad argument (null pointer, inconsistent sizes)        */
#define OTG_ERR_HIP         -3   /* a HIP runtime call failed (message in otg_last_error)  */
#define OTG_ERR_CAPACITY    -4   /* caller-provided output buffer too small                */
#define OTG_ERR_FATAL       -5   /* a condition on which the reference exit(1)s            */

/* per-region status, mirrors the reference's warnings / omissions (src/assemble.cpp:71,94,120) */
#define OTG_REGION_OK            0
#define OTG_REGION_SKIP_MAXCOV   1   /* reads > max_cov: "[WARNING] Skipping region with abnormal coverage" */
#define OTG_REGION_NO_SPANNING   2   /* "[WARNING] No spanning reads"                                         */
#define OTG_REGION_EMPTY         3   /* no reads at all                                                       */
#define OTG_REGION_HAP_CONFLICT  4   /* "ERROR: conflicting haplotag information" (reference exit(1)s)       */
#define OTG_REGION_ALIGN_CAPACITY 5  /* a gap-affine alignment of this region outgrew the device workspaces: no records for THIS region, the rest of
                                        the batch is unaffected (the reference has no length cap, src/assemble.cpp:51-154; here the last-resort tier
                                        holds one provenance byte per wavefront cell and its slab is a share of the device's memory)             */

typedef struct otg_ctx otg_ctx;

/* ---------------------------------------------------------------- parameters
 * Defaults = reference CLI defaults (src/command_assemble.cpp:34-45, src/command_genotype.cpp:25-27). */
typedef struct otg_params {
  int32_t max_alleles;          /* -a  2                                  */
  int32_t ignore_haps;          /* !--haps  => 1                          */
  int32_t max_cov;              /* -c  200                                */
  int32_t flank;                /* -f  100                                */
  int32_t bandwidth_length;     /* -h  second field, 500                  */
  int32_t min_cov_fraction2_l;  /* -A  first field, 500                   */
  int32_t mismatch;             /* WFAlignerGapAffine(4,6,2): 4           */
  int32_t gap_open;             /*                            6           */
  int32_t gap_ext;              /*                            2           */
  int32_t realign;              /* 1 iff -r <reference> was given (flanks present in the batch) */
  double  bandwidth_short;      /* -h  0.01                               */
  double  bandwidth_long;       /* -h  0.015                              */
  double  max_error;            /* -e  0.01                               */
  double  min_cov_fraction;     /* -F  0.2                                */
  double  min_cov_fraction2_f;  /* -A  0.1                                */
  double  min_sim;              /* -s  0.9                                */
  double  gt_max_error;         /* genotype -e 0.025                      */
  double  gt_max_cosdis;        /* genotype -c 0.025                      */
  /* Heuristic of BOTH aligners of the region pipeline (see otg_set_heuristic).  The reference never calls setHeuristic*
   * (src/assemble.cpp:49-50): its aligners run WFA2-lib's default.  0 = OTG_HEURISTIC_NONE (exact; the default here),
   * 1 = OTG_HEURISTIC_WFADAPTIVE with the three numbers below (WFA2-lib's own default values: 10, 50, 1).              */
  int32_t heuristic;
  int32_t heur_min_wavefront_length;
  int32_t heur_max_distance_threshold;
  int32_t heur_steps_between_cutoffs;
} otg_params;

void otg_params_default(otg_params* p);

/* ---------------------------------------------------------------- context */
int         otg_create(int device, otg_ctx** out);
void        otg_destroy(otg_ctx* ctx);
/* Gives the aligners' scratch workspaces (provenance slabs, row tables: tens of GB once long reads have been seen) back to the device;
 * the next call that needs them allocates them again.  Resident batches and results are untouched.  For hosts that share a device
 * between contexts or processes; the reference has no counterpart (its aligners own host memory, src/assemble.cpp:45-50).          */
int         otg_trim(otg_ctx* ctx);
const char* otg_last_error(otg_ctx* ctx);     /* ctx may be NULL: last global error           */
int         otg_device_count(void);           /* number of visible HIP devices (0 if none)    */
/* Which libm exp() rounding variant the device KDE mirrors (1 = glibc FMA build, 0 = non-FMA).
 * Chosen at otg_create by probing the host libm so cluster labels match the reference
 * running on this host (SURVEY.md §7.2 "FP determinism"). */
int         otg_exp_variant(otg_ctx* ctx);

/* Heuristic of the L1 aligner calls on this context — wfa::WFAligner::setHeuristicNone() /
 * setHeuristicWFadaptive(min_wavefront_length, max_distance_threshold, steps_between_cutoffs) (bindings/cpp/WFAligner.hpp:107-122 of
 * WFA2-lib).  OTG_HEURISTIC_NONE: exact alignment (default).  OTG_HEURISTIC_WFADAPTIVE: WFA2-lib's
 * adaptive wavefront reduction — after each score, diagonals whose remaining distance to the end exceeds the best one by more than
 * max_distance_threshold are dropped from both ends of the wavefront (scores >= the exact ones; op strings of a valid, possibly
 * sub-optimal alignment).  The region pipeline takes its setting from otg_params instead.                                        */
#define OTG_HEURISTIC_NONE        0
#define OTG_HEURISTIC_WFADAPTIVE  1
int otg_set_heuristic(otg_ctx* ctx, int strategy, int min_wavefront_length, int max_distance_threshold, int steps_between_cutoffs);

/* ================================================================= L1: batched aligners
 * One task = one wfa::WFAligner call.  Sequences live in one byte arena (raw bytes, compared
 * raw: 'N'=='N', case-sensitive, as WFA2 does).
 *   all four *_free == 0 and endsfree == 0  -> alignEnd2End(pattern, text)
 *   endsfree == 1 -> alignEndsFree(pattern, pattern_begin_free, pattern_end_free,
 *                                  text, text_begin_free, text_end_free)                      */
typedef struct otg_align_task {
  uint64_t pattern_off;
  uint64_t text_off;
  uint32_t pattern_len;
  uint32_t text_len;
  int32_t  pattern_begin_free;
  int32_t  pattern_end_free;
  int32_t  text_begin_free;
  int32_t  text_end_free;
  int32_t  endsfree;
  int32_t  _pad;             /* reserved, 0 */
} otg_align_task;

/* Replaces WFAlignerEdit(Score, MemoryMed)::alignEnd2End/alignEndsFree + getAlignmentScore()
 * (src/assemble.cpp:49, src/analignments.cpp:70-71,88-97).  scores_out[i] = unit-cost edit distance
 * (>= 0).  cells_out (nullable): wavefront cells W_p evaluated.                */
int otg_edit_distance_batch(otg_ctx* ctx,
                            const uint8_t* seq_arena, uint64_t arena_bytes,
                            const otg_align_task* tasks, uint32_t n_tasks,
                            int32_t* scores_out, uint64_t* cells_out);

/* Replaces WFAlignerGapAffine(x,o,e, Alignment, MemoryMed)::alignEnd2End/alignEndsFree +
 * getAlignmentCigar() (src/assemble.cpp:50, src/analignments.cpp:25,31,37,268-280).
 * scores_out[i] = gap-affine penalty (>= 0; WFA2 reports its negative).  The op string of task i
 * (alphabet M X I D, one char per column, free end gaps explicit) is written at
 * cigar_arena + cigar_off_out[i], length cigar_len_out[i].  Returns OTG_ERR_CAPACITY (and the
 * needed size in *cigar_bytes_used) if cigar_capacity is too small.                            */
int otg_affine_align_batch(otg_ctx* ctx,
                           const uint8_t* seq_arena, uint64_t arena_bytes,
                           const otg_align_task* tasks, uint32_t n_tasks,
                           int32_t mismatch, int32_t gap_open, int32_t gap_ext,
                           int32_t* scores_out,
                           uint64_t* cigar_off_out, uint32_t* cigar_len_out,
                           uint8_t* cigar_arena, uint64_t cigar_capacity, uint64_t* cigar_bytes_used,
                           uint64_t* cells_out);

/* ================================================================= L2: per-region operators */

/* Replaces otter_hclust (src/otterclust.cpp:118-320) incl. otter_find_clustering_dist (:20-116),
 * KDE (src/ankde.cpp), hclust_fast / cutree_cdist / cutree_k (include/hclust-cpp/fastcluster.cpp).
 * Region r has n_valid[r] valid reads, its condensed FP64 matrix (DistMatrix layout,
 * src/andistmat.cpp:20) starts at dist + dist_off[r], its read lengths at read_len + len_off[r].
 * Outputs: labels (same indexing as read_len), ic/fc per region, bounds = 3 doubles per region
 * (dist0, dist1, cut0; NaN when the KDE was not evaluated).                                    */
int otg_cluster_batch(otg_ctx* ctx, const otg_params* params,
                      const double* dist, const uint64_t* dist_off,
                      const uint32_t* read_len, const uint64_t* len_off,
                      const uint32_t* n_valid, uint32_t n_regions,
                      int32_t* labels_out, int32_t* ic_out, int32_t* fc_out, double* bounds_out);

/* Replaces PPOA (src/anppoa.hpp:64-380) as driven by rapid_consensus (src/analignments.cpp:261-292):
 * graph g has backbone = task `backbone`, then members inserted in order; each member is a
 * sequence + its op string + spanning flags.  c/t are the adjust_weights arguments.
 * Output consensus strings in out_arena (same off/len convention as cigars).                   */
typedef struct otg_poa_member {
  uint64_t seq_off;   uint32_t seq_len;   uint32_t cigar_len;
  uint64_t cigar_off;
  uint8_t  spanning_l, spanning_r; uint8_t _pad[6];
} otg_poa_member;

typedef struct otg_poa_graph {
  uint64_t backbone_off; uint32_t backbone_len;
  uint32_t first_member; uint32_t n_members;
  float    c, t; uint32_t _pad;
} otg_poa_graph;

int otg_poa_consensus_batch(otg_ctx* ctx,
                            const uint8_t* seq_arena, uint64_t arena_bytes,
                            const uint8_t* cigar_arena, uint64_t cigar_bytes,
                            const otg_poa_member* members, uint32_t n_members,
                            const otg_poa_graph* graphs, uint32_t n_graphs,
                            uint64_t* out_off, uint32_t* out_len,
                            uint8_t* out_arena, uint64_t out_capacity, uint64_t* out_bytes_used);

/* Replaces anallele_cluster (src/otterclust.cpp:463-527) for `otter genotype`.
 * Region r has n_alleles[r] allele sequences described by (seq_off,seq_len) entries starting at
 * allele index first_allele[r].  Outputs per allele: gt, gt_l, gt_k, hsd; per region: n_gt and
 * representative allele indices (region-local) written at reps_out + first_allele[r].           */
int otg_genotype_cluster_batch(otg_ctx* ctx, const otg_params* params,
                               const uint8_t* seq_arena, uint64_t arena_bytes,
                               const uint64_t* seq_off, const uint32_t* seq_len,
                               const uint32_t* first_allele, const uint32_t* n_alleles, uint32_t n_regions,
                               int32_t* gt_out, int32_t* gt_l_out, int32_t* gt_k_out, double* hsd_out,
                               int32_t* n_gt_out, int32_t* reps_out);

/* HIP-event time (ms, events on the context's own stream) of the device kernels of the latest otg_genotype_cluster_batch on this context:
 * what a roofline figure divides by; host copies excluded.  The reference has no counterpart (measurement hook).        */
int otg_last_kernel_ms(otg_ctx* ctx, double* ms);

/* ================================================================= L3: region-batch pipeline
 * SoA image of std::vector<ANREAD> (src/anseqs.hpp:56-76) for a batch of regions.               */
typedef struct otg_read {
  uint64_t seq_off;          /* into seq_arena                                   */
  uint32_t seq_len;
  uint8_t  spanning_l;       /* ANREAD::is_spanning_l                            */
  uint8_t  spanning_r;       /* ANREAD::is_spanning_r                            */
  uint16_t _pad;
  int32_t  ps, hp;           /* HAPLOTAG (-1 = undefined)                        */
  int32_t  ccoord_first;     /* ANREAD::ccoords                                  */
  int32_t  ccoord_second;
} otg_read;

typedef struct otg_region {
  uint32_t first_read;       /* index into reads[]                               */
  uint32_t n_reads;
  uint64_t flank_l_off;      /* reference flank [start-flank,start], upper-cased  */
  uint64_t flank_r_off;      /* reference flank [end,end+flank]                   */
  uint32_t flank_l_len;      /* 0 when -r not given                              */
  uint32_t flank_r_len;
} otg_region;

/* image of ANALLELE (src/anseqs.hpp:40-54) */
typedef struct otg_allele {
  uint64_t seq_off;          /* into the output arena                            */
  uint32_t seq_len;
  int32_t  scov, acov, tcov;
  float    se;
  int32_t  ic;
  int32_t  ps, hp;
  uint32_t region;           /* batch-local region index                         */
  int32_t  label;            /* allele index within the region (0..fc-1)         */
} otg_allele;

typedef struct otg_region_result {
  uint32_t first_allele;
  uint32_t n_alleles;        /* = fc for OK regions, else 0                      */
  int32_t  status;           /* OTG_REGION_*                                     */
  int32_t  ic, fc;
  int32_t  n_valid;
} otg_region_result;

/* ---------------------------------------------------------------------------------------------
 * Record emit (the first "next" row after the hot path): the text `otter assemble`
 * prints for the allele records of a batch — reference: the emit loop src/assemble.cpp:143-149 ->
 * ANALLELE::stdout_sam / stdout_fa (src/anseqs.cpp:42-63), BED::toScString (src/anbed.cpp:17-20),
 * header lines src/assemble.cpp:167-177.  Host-side formatting (no device work), byte-identical to the
 * reference: regions in batch order (= BED order; the reference's own order with -t 1), alleles in label
 * order; `se` printed the way `std::cout << float` prints it.
 * ------------------------------------------------------------------------------------------- */
typedef struct otg_bed {
  uint64_t chr_off;          /* chromosome name: chr_arena + chr_off, chr_len bytes (no terminator needed) */
  uint32_t chr_len;
  int32_t  start, end;       /* the UN-modified BED coordinates (local_bed, src/assemble.cpp:53)           */
  uint32_t reserved;
} otg_bed;

/* Writes the record lines into `out` (capacity out_capacity) and their total length into *out_len; returns
 * OTG_ERR_CAPACITY (with *out_len = bytes needed) when the buffer is too small.  is_fasta: 0 = SAM lines
 * (name `chr:start-end_l`, read group tag when read_group is non-empty), 1 = FASTA (`>rg#chr:start-end#l#...`). */
int otg_emit_alleles(const otg_bed* beds, const char* chr_arena, uint32_t n_regions,
                     const otg_region_result* regions, const otg_allele* alleles, const uint8_t* seqs,
                     const char* read_group, int is_fasta, char* out, uint64_t out_capacity, uint64_t* out_len);
/* The three kinds of SAM header lines of src/assemble.cpp:167-177: @SQ per target, @RG, @PG with the offsets. */
int otg_emit_sam_header(const char* name_arena, const uint64_t* name_off, const uint32_t* name_len,
                        const uint64_t* target_len, uint32_t n_targets, const char* read_group,
                        int32_t offset_l, int32_t offset_r, char* out, uint64_t out_capacity, uint64_t* out_len);

/* ---------------------------------------------------------------------------------------------
 * BAM/BAI region ingest: the reads of each BED region as `parse_anreads` delivers them to the hot
 * path (src/anseqs.cpp:244-460, called at src/assemble.cpp:55-65), straight into the region batch of
 * otg_assemble_submit.  Host code (zlib + stdio); the index is `<bam>.bai` (src/anbamfilehelper.cpp:20).
 * ------------------------------------------------------------------------------------------- */
typedef struct otg_bam otg_bam;
typedef struct otg_fasta otg_fasta;     /* indexed FASTA handle, see otg_fasta_open below */
typedef struct otg_ingest_opts {
  int32_t offset_l, offset_r;   /* --offset: the query region is [start - offset_l, end + offset_r] (src/assemble.cpp:55-57) */
  int32_t mapq;                 /* --mapq                                                     */
  int32_t nonprimary;           /* --non-primary: keep secondary / supplementary alignments   */
  int32_t omit_nonspanning;     /* --omit-nonspanning                                         */
  int32_t threads;              /* host threads for the ingest (regions are split into contiguous slices); <= 1: one */
  double  read_quality;         /* --read-quality (tag rq)                                    */
} otg_ingest_opts;
int  otg_bam_open(const char* bam_path, otg_bam** out);
void otg_bam_close(otg_bam* bam);
uint32_t otg_bam_n_targets(const otg_bam* bam);
const char* otg_bam_target(const otg_bam* bam, uint32_t i, uint64_t* length);   /* name (owned by the handle) + length: @SQ lines */
/* Appends the reads of n_regions BED regions to arena / reads (in-out counters *arena_used, *n_reads) and fills
 * regions[i].first_read / n_reads (flank fields zeroed).  OTG_ERR_CAPACITY: buffers too small, the counters hold the
 * needed totals.  Reads come in file order; filters, sub-sequence, spanning flags and clip coordinates as the reference. */
int  otg_ingest_regions(otg_bam* bam, const otg_bed* beds, const char* chr_arena, uint32_t n_regions,
                        const otg_ingest_opts* opts, uint8_t* arena, uint64_t arena_capacity, uint64_t* arena_used,
                        otg_read* reads, uint32_t reads_capacity, uint32_t* n_reads, otg_region* regions);

/* The same, also returning what `--reads-only` prints per read: ANREAD::name and ANREAD::rq (src/anseqs.cpp:447,250-251).
 * meta[i] belongs to reads[i]; names are appended to name_arena (in-out counter *name_used, no terminators).  meta == NULL:
 * exactly otg_ingest_regions. */
typedef struct otg_read_meta {
  uint64_t name_off;         /* into name_arena                                   */
  uint32_t name_len;
  uint32_t reserved;
  double   rq;               /* tag rq, 0 when absent                             */
} otg_read_meta;
int  otg_ingest_regions_named(otg_bam* bam, const otg_bed* beds, const char* chr_arena, uint32_t n_regions,
                              const otg_ingest_opts* opts, uint8_t* arena, uint64_t arena_capacity, uint64_t* arena_used,
                              otg_read* reads, uint32_t reads_capacity, uint32_t* n_reads, otg_region* regions,
                              otg_read_meta* meta, char* name_arena, uint64_t name_capacity, uint64_t* name_used);
/* `otter assemble --reads-only`: the read records of each region instead of alleles (src/assemble.cpp:82-89 ->
 * ANREAD::stdout_sam / stdout_fa, src/anseqs.cpp:83-106).  Regions with more than max_cov reads print nothing
 * (src/assemble.cpp:69; max_cov < 0: no limit).  Same buffer protocol as otg_emit_alleles. */
int  otg_emit_reads(const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const otg_region* regions,
                    const otg_read* reads, const uint8_t* seq_arena, const otg_read_meta* meta, const char* name_arena,
                    const char* read_group, int is_fasta, int32_t max_cov, char* out, uint64_t out_capacity, uint64_t* out_len);

/* ---------------------------------------------------------------------------------------------
 * `otter genotype` ingest: the allele BAM written by `otter assemble`.
 * ------------------------------------------------------------------------------------------- */
/* SampleIndex::init (src/anbamdb.cpp:10-63): sample names = the `@RG ID:` values in header order (everything after "ID:",
 * as the reference takes it), offsets from `@PG ID:otter OF:l,r` (defaults 1, 0).  Errors as the reference's exit(1) cases. */
int  otg_bam_sample_index(otg_bam* bam, uint32_t* n_samples, int32_t* offset_l, int32_t* offset_r);
const char* otg_bam_sample(const otg_bam* bam, uint32_t i);      /* valid after otg_bam_sample_index */
/* parse_analleles (src/anseqs.cpp:462-524) per BED region: the records whose `ta` tag equals "chr:start-end", in file order, as
 * otg_allele records (seq, sc/ac/tc, se, ic, PS/HP; .region = region index, .label = SAMPLE index of the RG tag; an unknown
 * read group is the reference's exit(1): OTG_ERR_ARG).  With a FASTA handle the reference allele of genotype_process
 * (src/genotype.cpp:92-101: bases [start - offset_l, end + offset_r - 1], sample index = n_samples) is appended to every
 * non-empty region.  first_allele has n_regions + 1 entries (in-out counter *n_alleles is the running total). */
int  otg_ingest_alleles(otg_bam* bam, const otg_bed* beds, const char* chr_arena, uint32_t n_regions, int32_t threads,
                        const otg_fasta* reference, uint8_t* arena, uint64_t arena_capacity, uint64_t* arena_used,
                        otg_allele* alleles, uint32_t alleles_capacity, uint32_t* n_alleles, uint32_t* first_allele);

/* `otter genotype` record emit.  The header: output_vcf_header (src/genotype.cpp:16-40; contigs = BAM targets,
 * one column per sample of otg_bam_sample_index).  The lines: per region with alleles, genotype numbers re-centred on the reference
 * allele (src/genotype.cpp:139-150) and output_vcf_line (:43-78) — alleles / first_allele as otg_ingest_alleles delivers them with a
 * reference (reference allele last in every region, .label = sample index, n_samples = number of real samples); gt / hsd / n_gt /
 * reps exactly as otg_genotype_cluster_batch returns them (per allele, per region, region-local representative indices at
 * reps + first_allele[r]).  Without a reference `otter genotype` prints the shorter and longer allele length per sample instead
 * (src/genotype.cpp:112-121): otg_emit_genotype_lengths.  Same buffer protocol as otg_emit_alleles. */
int  otg_emit_vcf_header(const otg_bam* bam, char* out, uint64_t out_capacity, uint64_t* out_len);
int  otg_emit_vcf_lines(const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const uint32_t* first_allele,
                        const otg_allele* alleles, const uint8_t* seqs, uint32_t n_samples, const int32_t* gt, const double* hsd,
                        const int32_t* n_gt, const int32_t* reps, int32_t offset_l, int32_t offset_r,
                        char* out, uint64_t out_capacity, uint64_t* out_len);
int  otg_emit_genotype_lengths(const otg_bam* bam, const otg_bed* beds, const char* chr_arena, uint32_t n_regions,
                               const uint32_t* first_allele, const otg_allele* alleles, uint32_t n_samples,
                               char* out, uint64_t out_capacity, uint64_t* out_len);

/* ---------------------------------------------------------------------------------------------
 * The two text inputs beside the BAM: the BED file of regions and the indexed FASTA the reference
 * flanks of local_realignment come from.  Host code.
 * ------------------------------------------------------------------------------------------- */
/* parse_bed_file (src/anbed.cpp:23-80): tab-separated `chr start end [...]` or single-column `chr:start-end` lines;
 * '#' lines, empty lines and lines the reference calls ambiguous are skipped (*n_skipped counts the latter two kinds,
 * nullable); coordinates go through the same unsigned-32-bit conversion.  A coordinate that is not a number makes the
 * reference terminate: OTG_ERR_ARG here.  OTG_ERR_CAPACITY: *n_beds / *chr_used hold the needed sizes. */
int  otg_parse_bed_file(const char* path, otg_bed* beds, uint32_t beds_capacity, uint32_t* n_beds, char* chr_arena,
                        uint64_t chr_capacity, uint64_t* chr_used, uint32_t* n_skipped);
/* FaidxInstance (src/anfahelper.cpp:6-20) over an uncompressed FASTA with its `.fai` (read when present, otherwise built
 * as src/faidx.c:64-133 does and written beside the file when possible).  Handles are read-only after open: one may be
 * shared by threads. */
int  otg_fasta_open(const char* fasta_path, otg_fasta** out);
void otg_fasta_close(otg_fasta* fa);
uint32_t otg_fasta_n_seqs(const otg_fasta* fa);
const char* otg_fasta_seq(const otg_fasta* fa, uint32_t i, int64_t* length);
/* FaidxInstance::fetch: bases [beg, end_inclusive] (0-based), clamped into the contig as faidx_fetch_seq clamps them
 * (src/faidx.c:418-445), upper-cased; an unknown contig gives length 0. */
int  otg_fasta_fetch(const otg_fasta* fa, const char* chr, uint32_t chr_len, int32_t beg, int32_t end_inclusive,
                     char* out, uint64_t out_capacity, uint64_t* out_len);
/* The two flanks of every region, [start - offset_l - flank, start - offset_l] and [end + offset_r, end + offset_r + flank]
 * (src/analignments.cpp:22,28 on mod_bed, src/assemble.cpp:55-57), appended to the region batch's arena (in-out counter
 * *arena_used); fills regions[i].flank_{l,r}_{off,len} and leaves the read fields alone. */
int  otg_fasta_region_flanks(const otg_fasta* fa, const otg_bed* beds, const char* chr_arena, uint32_t n_regions,
                             int32_t offset_l, int32_t offset_r, int32_t flank, uint8_t* arena, uint64_t arena_capacity,
                             uint64_t* arena_used, otg_region* regions);

/* A sink for record text: called with consecutive pieces of the output; a non-zero return aborts the call. */
typedef int (*otg_write_fn)(void* user, const char* data, uint64_t len);

/* ---------------------------------------------------------------------------------------------
 * `otter wgat`: the BED regions sliced out of whole-genome assembly alignments, as allele records `otter genotype`
 * reads — wgat() / wga_bam_genotyper_process (src/wgat.cpp:31-179) with get_op_intervals (src/opinterval.cpp:12-34).  Host code.
 * Per BAM target in header order, per alignment in file order, per overlapping region in the order the reference's interval tree reports
 * them: the query interval under [start - offset_l, end + offset_r] from the CIGAR operations that overlap it; alignments clipped inside
 * the interval are skipped (the reference's warning).  Record lines as ANALLELE::stdout_sam / stdout_fa print them (name
 * `contig#chr:start-end_index`, tc / ac / sc = 1, sp:A:b); SAM output starts with the @SQ / @RG / @PG OF: lines.  Order = the reference's
 * with -t 1.  *n_records (nullable) = records written. */
int otg_wgat(otg_bam* bam, const otg_bed* beds, const char* chr_arena, uint32_t n_beds, const char* read_group, int is_fasta,
             int32_t offset_l, int32_t offset_r, otg_write_fn write, void* user, uint64_t* n_records);

/* Workload statistics of the last otg_assemble_run (for the roofline figure). */
typedef struct otg_run_stats {
  uint64_t n_regions, n_regions_ok;
  uint64_t edit_tasks,   edit_cells,   edit_seq_bytes;
  uint64_t affine_tasks, affine_cells, affine_seq_bytes;
  uint64_t allele_bytes;
  uint64_t algorithmic_bytes;         /* Σ (a+b) + 4·W (+ W/2 for CIGAR scope) + Σ(len+40) */
  double   ms_edit, ms_cluster, ms_reassign, ms_affine, ms_poa, ms_realign, ms_total;
  double   ms_edit_kernel;            /* HIP-event time of the edit WFA kernel launches (tier 1, summed)   */
  uint64_t edit_kernel_launches;
  double   ms_affine_kernel;          /* HIP-event time of the gap-affine WFA kernel launches (tier 1)     */
  uint64_t affine_kernel_launches;
  uint64_t affine_visited_cells;      /* (score, diagonal) cells the exact gap-affine kernels actually evaluated (pruned wavefronts) */
} otg_run_stats;

/* Upload a batch (H2D).  After it returns the inputs are resident in HBM.                       */
int otg_assemble_submit(otg_ctx* ctx, const otg_params* params,
                        const uint8_t* seq_arena, uint64_t arena_bytes,
                        const otg_read* reads, uint32_t n_reads,
                        const otg_region* regions, uint32_t n_regions);
/* Run the whole hot path on the resident batch: [local_realignment] -> fill_dist_matrix ->
 * otter_hclust -> invalid_reassignment -> rapid_consensus.  Results stay resident.             */
int otg_assemble_run(otg_ctx* ctx);
/* `otter assemble --reads-only -r`: run local_realignment alone on the resident batch (src/assemble.cpp:72-89 stops there), then read
 * the read descriptors back: a rescued read has its seq_off / seq_len trimmed (src/analignments.cpp:53-54) and both spanning flags
 * set; everything else is as submitted.  otg_emit_reads on them prints what the reference prints.  n_reads = the submitted count. */
int otg_assemble_realign(otg_ctx* ctx);
int otg_assemble_collect_reads(otg_ctx* ctx, otg_read* reads_out, uint32_t n_reads);
/* Sizes needed by otg_assemble_collect for the last run.                                        */
int otg_assemble_result_sizes(otg_ctx* ctx, uint32_t* n_alleles, uint64_t* seq_bytes);
/* Device-resident results of the last run, for callers that forward them without a host round trip
 * (one process per GPU: the end-of-run gather of allele records to rank 0, north_star / src/assemble.cpp:143-149
 * in the single-process reference).  Pointers are HBM addresses on the context's device, valid until the next
 * otg_assemble_submit / otg_assemble_run / otg_destroy; sizes: n_regions records, and the two figures of
 * otg_assemble_result_sizes.  The library's stream is synchronised before returning.             */
int otg_assemble_device_results(otg_ctx* ctx, const otg_region_result** d_regions, const otg_allele** d_alleles,
                                const uint8_t** d_seqs);
/* ---------------------------------------------------------------------------------------------
 * One process per GPU: the end-of-run gather of the allele records to rank 0 over RCCL (xGMI) — north_star's multi-GPU form; the
 * single-process reference prints under a mutex instead (src/assemble.cpp:143-149).  Rank r owns the r-th contiguous BED shard
 * (src/BS_thread_pool.hpp:183-198), so rank order is BED order.  librccl is loaded at run time by the first call.
 *   otg_comm_unique_id   rank 0 makes the 128-byte id; the host hands it to the other ranks (file, socket, environment: its business);
 *   otg_comm_create      every rank, with its device, its rank and the world size (collective: returns when all ranks have called it);
 *   otg_gather_sizes     after otg_assemble_run: counts_out[3 * r + {0, 1, 2}] = regions, allele records, sequence bytes of rank r (every rank);
 *   otg_gather_records   rank 0 passes buffers for the totals and receives regions / alleles / sequences of all ranks in rank order,
 *                        region / allele / sequence indices rebased to job-wide ones; the other ranks pass NULL.  Records travel
 *                        device -> device; the one device-to-host copy happens on rank 0.
 * ------------------------------------------------------------------------------------------- */
#define OTG_COMM_ID_BYTES 128
typedef struct otg_comm otg_comm;
int  otg_comm_unique_id(uint8_t* id_out);
int  otg_comm_create(int device, int rank, int world, const uint8_t* id, otg_comm** out);
void otg_comm_destroy(otg_comm* comm);
int  otg_gather_sizes(otg_ctx* ctx, otg_comm* comm, uint64_t* counts_out);
int  otg_gather_records(otg_ctx* ctx, otg_comm* comm, const uint64_t* counts, otg_region_result* regions_out, otg_allele* alleles_out, uint8_t* seqs_out);

/* D2H of the results.  labels_out (nullable) gets the final per-read label (-1 unassigned).     */
int otg_assemble_collect(otg_ctx* ctx,
                         otg_region_result* region_out,
                         otg_allele* alleles_out, uint32_t allele_capacity,
                         uint8_t* seq_out, uint64_t seq_capacity,
                         int32_t* labels_out);
int otg_assemble_stats(otg_ctx* ctx, otg_run_stats* out);

/* ---------------------------------------------------------------------------------------------
 * The dispatcher: `otter assemble` from files to record text in one call — the role of assemble() /
 * assemble_process() (src/assemble.cpp:39-179) over BS::thread_pool::parallelize_loop (src/BS_thread_pool.hpp:175-200).
 * The BED list is split into one contiguous shard per device (the reference's static split, GPUs for threads); every shard is cut
 * into bounded batches of `batch_regions` regions that are ingested on host threads, run through the hot path on the device (two
 * contexts per device: the upload of a batch overlaps the kernels of the previous one) and emitted, the three stages concurrently.
 * Text reaches `write` strictly in BED order (SAM header first unless is_fasta), whatever the batch size or the number of devices.
 * Host memory is bounded by the batch size.  A non-zero return of `write` aborts the job.
 * ------------------------------------------------------------------------------------------- */
typedef struct otg_assemble_job {
  const char* bam_path;          /* <BAM> (its index is <BAM>.bai)                                   */
  const char* bed_path;          /* -b                                                               */
  const char* fasta_path;        /* -r (NULL or "": no local re-alignment)                           */
  const char* read_group;        /* -R sample name                                                   */
  int32_t     is_fasta;          /* --fasta                                                          */
  int32_t     reads_only;        /* --reads-only                                                     */
  otg_params  params;            /* heuristics (realign is set from fasta_path)                      */
  otg_ingest_opts ingest;        /* --offset, --mapq, --non-primary, --omit-nonspanning, --read-quality; .threads = -t (host ingest threads) */
  uint32_t    batch_regions;     /* regions per batch; 0 = the library's plan: 256, 512, 1024 first, then
                                  * batches of 2048, the last stretch in two equal halves                */
  int32_t     n_devices;         /* 0: device 0 only                                                 */
  const int32_t* devices;        /* HIP device ordinals, one shard each                              */
} otg_assemble_job;
typedef struct otg_job_stats {
  uint64_t n_regions, n_regions_ok, n_regions_skipped, n_reads, n_alleles, input_bytes, output_bytes;
  uint32_t n_devices, reserved;
  double   ms_total;             /* wall                                                             */
  double   ms_ingest, ms_hot_path, ms_emit;   /* busy time of the three stages, summed over their threads (they overlap) */
} otg_job_stats;
/* (otg_write_fn is declared above, with otg_wgat) */
int otg_assemble_files(const otg_assemble_job* job, otg_write_fn write, void* user, otg_job_stats* stats);
/* `otter genotype` from files to text in one call — genotype() / genotype_process() (src/genotype.cpp:69-192): BED regions in bounded
 * batches through allele ingest (otg_ingest_alleles on `threads` host threads), anallele_cluster on the device (otg_genotype_cluster_batch)
 * and the VCF text (header first; otg_emit_vcf_lines), in BED order.  Without a reference FASTA the reference prints region, sample and the
 * two allele lengths instead (src/genotype.cpp:112-121): otg_emit_genotype_lengths, no device work.  stats: n_reads counts allele records. */
typedef struct otg_genotype_job {
  const char* bam_path;          /* the allele BAM `otter assemble` wrote (its index is <BAM>.bai)  */
  const char* bed_path;          /* -b                                                               */
  const char* fasta_path;        /* -r (NULL or "": the length table instead of VCF)                 */
  otg_params  params;            /* gt_max_error (-e), gt_max_cosdis (-c)                            */
  int32_t     threads;           /* -t: host threads of the allele ingest                            */
  int32_t     device;            /* HIP device ordinal                                               */
  uint32_t    batch_regions;     /* regions per batch, 0 = 1024                                      */
  uint32_t    reserved;
} otg_genotype_job;
int otg_genotype_files(const otg_genotype_job* job, otg_write_fn write, void* user, otg_job_stats* stats);
/* The dispatcher keeps its per-device contexts (and their HBM workspaces) for the next job of the process; this frees them. */
void otg_assemble_files_release(void);

/* The batch sizes otg_assemble_files cuts a shard of `n_regions` regions into (batch_regions as in the job; 0 = the library's plan).
 * Writes at most `capacity` sizes, *n_batches = how many there are; OTG_ERR_CAPACITY when they do not fit. */
int otg_assemble_batch_plan(uint32_t n_regions, uint32_t batch_regions, uint32_t* sizes, uint32_t capacity, uint32_t* n_batches);

#ifdef __cplusplus
}
#endif
#endif /* OTTER_GPU_H */
