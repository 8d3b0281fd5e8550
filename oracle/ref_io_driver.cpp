// ref_io_driver.cpp — C entry points over the REFERENCE's own record types and emit code, compiled from the sources
// where they lie (/root/reference/src/anseqs.cpp, anbed.cpp + the vendored htslib-lite C files they include; recipe in
// oracle/Makefile -> oracle/_ref/libotter_ref_io.so).  Test infrastructure: pins oracle/otter_oracle.cpp's restatement
// of the emit row and generates tests/golden/emit_ref.json.
#include <cstdint>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include "anseqs.hpp"
#include "anbed.hpp"
#include "anbamfilehelper.hpp"
#include "anfahelper.hpp"
#include "anbamdb.hpp"
#include "otter_opts.hpp"
#include "wgat.hpp"
#include "sam.h"
#include "faidx.h"
#include <vector>
#include "../include/otter_gpu.h"

extern "C" {

// The emit loop of assemble_process (src/assemble.cpp:143-149) on caller-provided records.
uint64_t ref_emit_alleles(const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const otg_region_result* regions,
                          const otg_allele* alleles, const uint8_t* seqs, const char* read_group, int is_fasta, char* out, uint64_t cap)
{
  std::ostringstream os;
  std::streambuf* old = std::cout.rdbuf(os.rdbuf());
  const std::string rg = read_group ? read_group : "";
  for (uint32_t r = 0; r < n_regions; ++r) {
    BED local_bed;
    local_bed.chr = std::string(chr_arena + beds[r].chr_off, beds[r].chr_len);
    local_bed.start = beds[r].start;
    local_bed.end = beds[r].end;
    for (uint32_t l = 0; l < regions[r].n_alleles; ++l) {
      const otg_allele& A = alleles[regions[r].first_allele + l];
      ANALLELE al(std::string((const char*)seqs + A.seq_off, A.seq_len), A.scov, A.acov, A.tcov, A.se, A.ic, A.hp, A.ps);
      al.hpt.ps = A.ps; al.hpt.hp = A.hp;
      if (is_fasta) al.stdout_fa(rg, local_bed.toScString() + '#' + std::to_string(l));
      else al.stdout_sam(local_bed.toScString() + "_" + std::to_string(l), local_bed.chr, local_bed.start, local_bed.end, rg);
    }
  }
  std::cout.rdbuf(old);
  const std::string t = os.str();
  if (out && cap) memcpy(out, t.data(), t.size() < cap ? t.size() : cap);
  return t.size();
}


// ---- the reference's own ingest as a fixture generator and end-to-end driver ------------

// SAM text -> BAM + BAI with the htslib-lite the reference vendors (sam_parse1 / bam_write1 / bam_index_build).
int ref_sam_to_bam(const char* sam_path, const char* bam_path)
{
  samFile* in = sam_open(sam_path, "r", nullptr);
  if (!in) return -1;
  bam_hdr_t* h = sam_hdr_read(in);
  if (!h) return -2;
  BGZF* out = bgzf_open(bam_path, "w");
  if (!out) return -3;
  if (bam_hdr_write(out, h) < 0) return -4;
  bam1_t* b = bam_init1();
  int n = 0;
  while (sam_read1(in, h, b) >= 0) { if (bam_write1(out, b) < 0) return -5; ++n; }
  bam_destroy1(b);
  bgzf_close(out);
  bam_hdr_destroy(h);
  sam_close(in);
  if (bam_index_build(bam_path, 0) < 0) return -6;
  return n;
}

struct RefIngest { BamInstance bam; FaidxInstance fa; bool has_fa; };

void* ref_ingest_open(const char* bam_path, const char* fasta_path)
{
  RefIngest* r = new RefIngest();
  r->bam.init(bam_path, true);
  r->has_fa = fasta_path && fasta_path[0];
  if (r->has_fa) r->fa.init(fasta_path);
  return r;
}
void ref_ingest_close(void* hnd) { RefIngest* r = (RefIngest*)hnd; r->bam.destroy(); if (r->has_fa) r->fa.destroy(); delete r; }

// parse_anreads (src/anseqs.cpp:439-460) for one BED region, exactly as assemble_process calls it (src/assemble.cpp:55-65):
// the query region is the BED region widened by the offsets.  Appends the reads to caller buffers in the layout of
// include/otter_gpu.h (otg_read + byte arena); returns the number of reads, or -(needed) style negatives on overflow.
int64_t ref_ingest_region(void* hnd, const char* chr, int start, int end, int offset_l, int offset_r, int mapq, int nonprimary,
                          double read_quality, int omitnonspanning, otg_read* reads, uint64_t reads_cap, uint8_t* arena,
                          uint64_t arena_cap, uint64_t* arena_used)
{
  RefIngest* r = (RefIngest*)hnd;
  OtterOpts params{};
  params.mapq = mapq; params.nonprimary = nonprimary != 0; params.read_quality = read_quality; params.omitnonspanning = omitnonspanning != 0;
  BED mod_bed;
  mod_bed.chr = chr; mod_bed.start = start - offset_l; mod_bed.end = end + offset_r;
  std::vector<ANREAD> block;
  parse_anreads(params, mod_bed, r->bam, block);
  if (block.size() > reads_cap) return -1;
  uint64_t used = *arena_used;
  for (size_t i = 0; i < block.size(); ++i) {
    const ANREAD& a = block[i];
    if (used + a.seq.size() + 64 > arena_cap) return -2;
    memset(&reads[i], 0, sizeof(otg_read));
    reads[i].seq_off = used; reads[i].seq_len = (uint32_t)a.seq.size();
    reads[i].spanning_l = a.is_spanning_l; reads[i].spanning_r = a.is_spanning_r;
    reads[i].ps = a.hpt.ps; reads[i].hp = a.hpt.hp;
    reads[i].ccoord_first = a.ccoords.first; reads[i].ccoord_second = a.ccoords.second;
    memcpy(arena + used, a.seq.data(), a.seq.size());
    used += a.seq.size();
  }
  *arena_used = used;
  return (int64_t)block.size();
}

// the two reference flanks local_realignment fetches (src/analignments.cpp:22,28 via FaidxInstance::fetch): 0-based inclusive on both
// ends, i.e. flank + 1 bases, upper-cased by the helper.  Returns the length written (0 when no FASTA was opened).
int ref_fetch(void* hnd, const char* chr, int beg, int end_incl, char* out, int cap)
{
  RefIngest* r = (RefIngest*)hnd;
  if (!r->has_fa) return 0;
  std::string s;
  r->fa.fetch(chr, beg, end_incl, s);
  const int n = (int)s.size() < cap ? (int)s.size() : cap;
  memcpy(out, s.data(), n);
  return (int)s.size();
}

// `otter assemble --reads-only` for one region: parse_anreads on the widened region, then ANREAD::stdout_fa / stdout_sam with the
// un-widened BED (src/assemble.cpp:55-65,82-89; regions above max_cov print nothing, :69).  Returns the text length.
uint64_t ref_reads_only(void* hnd, const char* chr, int start, int end, int offset_l, int offset_r, int mapq, int nonprimary,
                        double read_quality, int omitnonspanning, int max_cov, const char* read_group, int is_fasta, char* out, uint64_t cap)
{
  RefIngest* r = (RefIngest*)hnd;
  OtterOpts params{};
  params.mapq = mapq; params.nonprimary = nonprimary != 0; params.read_quality = read_quality; params.omitnonspanning = omitnonspanning != 0;
  BED local_bed;
  local_bed.chr = chr; local_bed.start = start; local_bed.end = end;
  BED mod_bed = local_bed;
  mod_bed.start -= offset_l; mod_bed.end += offset_r;
  std::vector<ANREAD> block;
  parse_anreads(params, mod_bed, r->bam, block);
  std::ostringstream os;
  std::streambuf* old = std::cout.rdbuf(os.rdbuf());
  const std::string rg = read_group ? read_group : "";
  if (!((int)block.size() > max_cov)) {
    for (const auto& read : block) {
      if (is_fasta) read.stdout_fa(local_bed.toScString());
      else read.stdout_sam(local_bed.chr, local_bed.start, local_bed.end, rg);
    }
  }
  std::cout.rdbuf(old);
  const std::string t = os.str();
  if (out && cap) memcpy(out, t.data(), t.size() < cap ? t.size() : cap);
  return t.size();
}

// parse_bed_file (src/anbed.cpp:65-80) -> "chr\tstart\tend\n" per accepted region (BED::toString), stderr chatter dropped.
// Returns the text length, or -1 when the reference's parser threw (std::stoul on a non-number).
int64_t ref_parse_bed_file(const char* path, char* out, uint64_t cap)
{
  std::vector<BED> v;
  std::ostringstream sink;
  std::streambuf* old = std::cerr.rdbuf(sink.rdbuf());
  bool threw = false;
  try { parse_bed_file(path, v); } catch (...) { threw = true; }
  std::cerr.rdbuf(old);
  if (threw) return -1;
  std::string t;
  for (const BED& b : v) t += b.toString() + "\n";
  if (out && cap) memcpy(out, t.data(), t.size() < cap ? t.size() : cap);
  return (int64_t)t.size();
}

// SampleIndex::init (src/anbamdb.cpp:42-63) -> "offset_l\toffset_r\n" then one sample name per line
int64_t ref_sample_index(const char* bam_path, char* out, uint64_t cap)
{
  SampleIndex si;
  si.init(bam_path);
  std::string t = std::to_string(si.offset_l) + "\t" + std::to_string(si.offset_r) + "\n";
  for (const auto& nm : si.index2sample) t += nm + "\n";
  if (out && cap) memcpy(out, t.data(), t.size() < cap ? t.size() : cap);
  return (int64_t)t.size();
}

// parse_analleles (src/anseqs.cpp:513-524) for one region + the reference allele genotype_process appends (src/genotype.cpp:92-101),
// in the layout of otg_ingest_alleles.  Returns the number of alleles, -1 / -2 on overflow.
int64_t ref_ingest_alleles(void* hnd, const char* bam_path, const char* chr, uint32_t start, uint32_t end, uint32_t region_index, otg_allele* alleles,
                           uint64_t alleles_cap, uint8_t* arena, uint64_t arena_cap, uint64_t* arena_used)
{
  RefIngest* r = (RefIngest*)hnd;
  SampleIndex si;
  si.init(bam_path);
  OtterOpts params{};
  BED region;
  region.chr = chr; region.start = start; region.end = end;
  std::vector<ANALLELE> block;
  std::vector<int> samples;
  parse_analleles(params, r->bam, region, si.sample2index, block, samples);
  if (!block.empty() && r->has_fa) {
    std::string refseq;
    r->fa.fetch(region.chr, region.start - si.offset_l, region.end + si.offset_r - 1, refseq);
    samples.emplace_back((int)si.index2sample.size());
    block.emplace_back(refseq);
  }
  if (block.size() > alleles_cap) return -1;
  uint64_t used = *arena_used;
  for (size_t i = 0; i < block.size(); ++i) {
    const ANALLELE& a = block[i];
    if (used + a.seq.size() + 64 > arena_cap) return -2;
    memset(&alleles[i], 0, sizeof(otg_allele));
    alleles[i].seq_off = used; alleles[i].seq_len = (uint32_t)a.seq.size();
    alleles[i].scov = a.scov; alleles[i].acov = a.acov; alleles[i].tcov = a.tcov; alleles[i].se = a.se; alleles[i].ic = a.ic;
    alleles[i].ps = a.hpt.ps; alleles[i].hp = a.hpt.hp; alleles[i].region = region_index; alleles[i].label = samples[i];
    memcpy(arena + used, a.seq.data(), a.seq.size());
    used += a.seq.size();
  }
  *arena_used = used;
  return (int64_t)block.size();
}

// `otter wgat` (src/wgat.cpp:157-179) as the reference runs it with -t 1: stdout captured (warnings on stderr dropped).
uint64_t ref_wgat(const char* bam, const char* bed, const char* read_group, int is_fasta, int offset_l, int offset_r, char* out, uint64_t cap)
{
  OtterOpts params;
  params.read_group = read_group ? read_group : "";
  params.is_fa = is_fasta != 0;
  params.offset_l = (uint32_t)offset_l; params.offset_r = (uint32_t)offset_r;
  params.threads = 1;
  std::ostringstream os, sink;
  std::streambuf* old = std::cout.rdbuf(os.rdbuf());
  std::streambuf* olde = std::cerr.rdbuf(sink.rdbuf());
  wgat(params, bam, bed);
  std::cout.rdbuf(old);
  std::cerr.rdbuf(olde);
  const std::string t = os.str();
  if (out && t.size() <= cap) memcpy(out, t.data(), t.size());
  return t.size();
}

} // extern "C"
