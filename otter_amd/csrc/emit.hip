// emit.hip — record emit of `otter assemble`: SAM / FASTA lines of the allele records and the SAM
// header, byte-identical to the reference (BED coordinates are unsigned 32-bit there, src/anbed.hpp:16-17: the region name built by
// toScString prints them unsigned, the POS column and the ta tag take them as int — same bits, two spellings above 2^31)
// — (src/assemble.cpp:143-149,167-177; ANALLELE::stdout_sam / stdout_fa
// src/anseqs.cpp:42-63; BED::toScString src/anbed.cpp:17-20).  Host-side formatting only; lives in the library so
// that a caller of the C-ABI gets the wire format the parity diff is taken on without re-implementing it.
#include "otg_common.hpp"
#include <cstdio>
#include <algorithm>
#include <vector>

namespace {

struct Sink {
  char* out; uint64_t cap; uint64_t len;
  void put(const char* p, size_t n) { if (len + n <= cap && out) memcpy(out + len, p, n); len += n; }
  void put(const char* z) { put(z, strlen(z)); }
  void ch(char c) { put(&c, 1); }
  void i64(long long v) { char b[32]; const int n = snprintf(b, sizeof b, "%lld", v); put(b, (size_t)n); }
  void u64(unsigned long long v) { char b[32]; const int n = snprintf(b, sizeof b, "%llu", v); put(b, (size_t)n); }
  // `std::cout << float`: default float field, precision 6 == printf %g of the value widened to double
  void dbl(double v) { char b[48]; const int n = snprintf(b, sizeof b, "%g", v); put(b, (size_t)n); }
  void flt(float v) { char b[48]; const int n = snprintf(b, sizeof b, "%g", (double)v); put(b, (size_t)n); }
  void fill(char c, size_t n) { if (len + n <= cap && out) memset(out + len, c, n); len += n; }
};

} // namespace

extern "C" {

int otg_emit_alleles(const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const otg_region_result* regions,
                     const otg_allele* alleles, const uint8_t* seqs, const char* read_group, int is_fasta,
                     char* out, uint64_t out_capacity, uint64_t* out_len)
{
  if ((n_regions && (!beds || !regions)) || !out_len) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_alleles: null argument");
  const char* rg = read_group ? read_group : "";
  Sink s{out, out_capacity, 0};
  for (uint32_t r = 0; r < n_regions; ++r) {
    const otg_bed& b = beds[r];
    const char* chr = chr_arena + b.chr_off;
    for (uint32_t a = 0; a < regions[r].n_alleles; ++a) {
      const otg_allele& A = alleles[regions[r].first_allele + a];
      const char* seq = (const char*)seqs + A.seq_off;
      if (is_fasta) {
        // stdout_fa(name = read group, region = toScString() + '#' + l)  (src/assemble.cpp:146, src/anseqs.cpp:56-63)
        s.ch('>'); s.put(rg); s.ch('#'); s.put(chr, b.chr_len); s.ch(':'); s.u64((uint32_t)b.start); s.ch('-'); s.u64((uint32_t)b.end); s.ch('#'); s.i64(a);
        s.put("#tc:i:"); s.i64(A.tcov); s.put("#ac:i:"); s.i64(A.acov); s.put("#sc:i:"); s.i64(A.scov);
        if (A.ps >= 0) { s.put("#PS:i:"); s.i64(A.ps); }
        if (A.hp >= 0) { s.put("#HP:i:"); s.i64(A.hp); }
        s.ch('\n'); s.put(seq, A.seq_len); s.ch('\n');
      } else {
        // stdout_sam(name = toScString() + "_" + l, chr, start, end, rg)  (src/assemble.cpp:147, src/anseqs.cpp:42-54)
        s.put(chr, b.chr_len); s.ch(':'); s.u64((uint32_t)b.start); s.ch('-'); s.u64((uint32_t)b.end); s.ch('_'); s.i64(a);
        s.put("\t0\t"); s.put(chr, b.chr_len); s.ch('\t'); s.i64(b.start); s.put("\t0\t"); s.u64(A.seq_len); s.put("M\t*\t0\t0\t");
        s.put(seq, A.seq_len); s.ch('\t'); s.fill('!', A.seq_len);
        if (rg[0]) { s.put("\tRG:Z:"); s.put(rg); }
        s.put("\tta:Z:"); s.put(chr, b.chr_len); s.ch(':'); s.i64(b.start); s.ch('-'); s.i64(b.end);
        s.put("\ttc:i:"); s.i64(A.tcov); s.put("\tac:i:"); s.i64(A.acov); s.put("\tsc:i:"); s.i64(A.scov);
        s.put("\tic:i:"); s.i64(A.ic); s.put("\tse:f:"); s.flt(A.se);
        if (A.ps >= 0) { s.put("\tPS:i:"); s.i64(A.ps); }
        if (A.hp >= 0) { s.put("\tHP:i:"); s.i64(A.hp); }
        s.ch('\n');
      }
    }
  }
  *out_len = s.len;
  if (s.len > out_capacity || (!out && s.len)) return OTG_ERR_CAPACITY;
  return OTG_OK;
}

// `otter assemble --reads-only`: the reads of each region as they entered the hot path (src/assemble.cpp:82-89 ->
// ANREAD::stdout_sam / stdout_fa, src/anseqs.cpp:83-106)
int otg_emit_reads(const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const otg_region* regions, const otg_read* reads,
                   const uint8_t* seq_arena, const otg_read_meta* meta, const char* name_arena, const char* read_group, int is_fasta,
                   int32_t max_cov, char* out, uint64_t out_capacity, uint64_t* out_len)
{
  if ((n_regions && (!beds || !regions)) || !out_len) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_reads: null argument");
  const char* rg = read_group ? read_group : "";
  Sink s{out, out_capacity, 0};
  for (uint32_t r = 0; r < n_regions; ++r) {
    const otg_bed& b = beds[r];
    const char* chr = chr_arena + b.chr_off;
    if (max_cov >= 0 && regions[r].n_reads > (uint32_t)max_cov) continue;     // "abnormal coverage" regions print nothing (:69)
    for (uint32_t k = 0; k < regions[r].n_reads; ++k) {
      const uint32_t i = regions[r].first_read + k;
      const otg_read& R = reads[i];
      const char* seq = (const char*)seq_arena + R.seq_off;
      const char sp = R.spanning_l && R.spanning_r ? 'b' : R.spanning_l ? 'l' : R.spanning_r ? 'r' : 'n';
      if (is_fasta) {
        s.ch('>'); if (meta) s.put(name_arena + meta[i].name_off, meta[i].name_len);
        s.ch('#'); s.put(chr, b.chr_len); s.ch(':'); s.u64((uint32_t)b.start); s.ch('-'); s.u64((uint32_t)b.end);
        s.put("#sp:A:"); s.ch(sp);
        if (R.ps >= 0) { s.put("#PS:i:"); s.i64(R.ps); }
        if (R.hp >= 0) { s.put("#HP:i:"); s.i64(R.hp); }
        s.ch('\n'); s.put(seq, R.seq_len); s.ch('\n');
      } else {
        if (meta) s.put(name_arena + meta[i].name_off, meta[i].name_len);
        s.put("\t0\t"); s.put(chr, b.chr_len); s.ch('\t'); s.i64(b.start); s.put("\t0\t"); s.u64(R.seq_len); s.put("M\t*\t0\t0\t");
        s.put(seq, R.seq_len); s.ch('\t'); s.fill('!', R.seq_len);
        if (rg[0]) { s.put("\tRG:Z:"); s.put(rg); }
        s.put("\tta:Z:"); s.put(chr, b.chr_len); s.ch(':'); s.i64(b.start); s.ch('-'); s.i64(b.end);
        s.put("\tsp:A:"); s.ch(sp);
        if (R.ps >= 0) { s.put("\tPS:i:"); s.i64(R.ps); }
        if (R.hp >= 0) { s.put("\tHP:i:"); s.i64(R.hp); }
        s.put("\trq:f:"); s.dbl(meta ? meta[i].rq : 0.0);
        s.ch('\n');
      }
    }
  }
  *out_len = s.len;
  if (s.len > out_capacity || (!out && s.len)) return OTG_ERR_CAPACITY;
  return OTG_OK;
}

int otg_emit_sam_header(const char* name_arena, const uint64_t* name_off, const uint32_t* name_len, const uint64_t* target_len,
                        uint32_t n_targets, const char* read_group, int32_t offset_l, int32_t offset_r,
                        char* out, uint64_t out_capacity, uint64_t* out_len)
{
  if ((n_targets && (!name_arena || !name_off || !name_len || !target_len)) || !out_len) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_sam_header: null argument");
  Sink s{out, out_capacity, 0};
  for (uint32_t i = 0; i < n_targets; ++i) { s.put("@SQ\tSN:"); s.put(name_arena + name_off[i], name_len[i]); s.put("\tLN:"); s.u64(target_len[i]); s.ch('\n'); }
  s.put("@RG\tID:"); s.put(read_group ? read_group : ""); s.ch('\n');
  s.put("@PG\tID:otter\tOF:"); s.i64(offset_l); s.ch(','); s.i64(offset_r); s.ch('\n');
  *out_len = s.len;
  if (s.len > out_capacity || (!out && s.len)) return OTG_ERR_CAPACITY;
  return OTG_OK;
}

// ---- `otter genotype`: VCF header (output_vcf_header, src/genotype.cpp:16-40), one VCF line per region (genotype_process
// src/genotype.cpp:103-157 -> output_vcf_line :43-78) and the length table printed without a reference (:112-121)
int otg_emit_vcf_header(const otg_bam* bam, char* out, uint64_t out_capacity, uint64_t* out_len)
{
  if (!bam || !out_len) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_vcf_header: null argument");
  Sink s{out, out_capacity, 0};
  s.put("##fileformat=VCFv4.2\n");
  for (uint32_t i = 0; i < otg_bam_n_targets(bam); ++i) {
    uint64_t len = 0;
    const char* nm = otg_bam_target(bam, i, &len);
    s.put("##contig=<ID="); s.put(nm); s.put(",length="); s.u64(len); s.put(">\n");
  }
  s.put("##INFO=<ID=HSD,Number=R,Type=Float,Description=\"Hill-Shannon Diversity Metric\">\n"
        "##ALT=<ID=DEL,Description=\"Deletion\">\n"
        "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
        "##FORMAT=<ID=PS,Number=1,Type=Integer,Description=\"Phase Set\">\n"
        "##FORMAT=<ID=HP,Number=1,Type=Integer,Description=\"Haplotype Identifier\">\n"
        "##FORMAT=<ID=TC,Number=1,Type=Integer,Description=\"Total Coverage of Region\">\n"
        "##FORMAT=<ID=AC,Number=2,Type=Integer,Description=\"Total Coverage For Each Allele\">\n"
        "##FORMAT=<ID=SC,Number=2,Type=Integer,Description=\"Total Coverage of Spanning Reads For Each Allele\">\n"
        "##FORMAT=<ID=SE,Number=2,Type=Float,Description=\"Standard Mean Error of Spanning Reads For Each Allele\">\n");
  s.put("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");
  for (uint32_t i = 0;; ++i) { const char* nm = otg_bam_sample(bam, i); if (!nm) break; s.ch('\t'); s.put(nm); }
  s.ch('\n');
  *out_len = s.len;
  if (s.len > out_capacity || (!out && s.len)) return OTG_ERR_CAPACITY;
  return OTG_OK;
}

int otg_emit_vcf_lines(const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const uint32_t* first_allele,
                       const otg_allele* alleles, const uint8_t* seqs, uint32_t n_samples, const int32_t* gt, const double* hsd,
                       const int32_t* n_gt, const int32_t* reps, int32_t offset_l, int32_t offset_r, char* out, uint64_t out_capacity,
                       uint64_t* out_len)
{
  (void)offset_r;
  if ((n_regions && (!beds || !first_allele || !gt || !hsd || !n_gt || !reps)) || !out_len) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_vcf_lines: null argument");
  Sink s{out, out_capacity, 0};
  std::vector<int> first(n_samples + 1), second(n_samples + 1), g2, rc;
  for (uint32_t r = 0; r < n_regions; ++r) {
    const uint32_t a0 = first_allele[r], na = first_allele[r + 1] - a0;
    if (na == 0) continue;                                        // "[WARNING] no alleles found": no line
    const otg_allele* A = alleles + a0;
    const int ref_i = (int)na - 1;                                // the reference allele was appended last (src/genotype.cpp:96-99)
    if ((uint32_t)A[ref_i].label != n_samples) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_vcf_lines: region %u has no reference allele in last place", r);
    // first / last allele of every sample (:103-110)
    std::fill(first.begin(), first.end(), -1); std::fill(second.begin(), second.end(), -1);
    for (int i = 0; i < (int)na; ++i) {
      const int sm = A[i].label;
      if (sm < 0 || (uint32_t)sm > n_samples) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_vcf_lines: sample index %d out of range", sm);
      if (first[(size_t)sm] < 0) { first[(size_t)sm] = i; second[(size_t)sm] = i; }
      else if (i < first[(size_t)sm]) first[(size_t)sm] = i;
      else if (i > second[(size_t)sm]) second[(size_t)sm] = i;
    }
    // genotype numbers re-centred on the reference allele (:139-150)
    const int ngt = n_gt[r];
    const int32_t* rp = reps + a0;
    const int ref_gt = gt[a0 + (uint32_t)ref_i];
    rc.assign(rp, rp + ngt);
    for (int i = 0; i < ngt; ++i) { if (i == 0) rc[0] = ref_i; else if (i <= ref_gt) rc[(size_t)i] = rp[i - 1]; }
    g2.assign(gt + a0, gt + a0 + na);
    for (uint32_t i = 0; i < na; ++i) { if (g2[i] == ref_gt) g2[i] = 0; else if (g2[i] < ref_gt) ++g2[i]; }
    // output_vcf_line (:43-78)
    const otg_bed& b = beds[r];
    const char* chr = chr_arena + b.chr_off;
    s.put(chr, b.chr_len); s.ch('\t'); s.u64((uint32_t)(1u + (uint32_t)b.start - (uint32_t)offset_l)); s.ch('\t');
    s.put(chr, b.chr_len); s.ch(':'); s.u64((uint32_t)b.start); s.ch('-'); s.u64((uint32_t)b.end); s.ch('\t');
    s.put((const char*)seqs + A[ref_i].seq_off, A[ref_i].seq_len); s.ch('\t');
    if (ngt == 1) s.ch('.');
    else for (int i = 1; i < ngt; ++i) {
      if (i > 1) s.ch(',');
      const otg_allele& al = A[rc[(size_t)i]];
      if (al.seq_len == 1 && seqs[al.seq_off] == 'N') s.put("<DEL>"); else s.put((const char*)seqs + al.seq_off, al.seq_len);
    }
    s.put("\t.\t.\tHSD=");
    for (int i = 0; i < ngt; ++i) { if (i > 0) s.ch(','); s.dbl(hsd[a0 + (uint32_t)rc[(size_t)i]]); }
    s.put("\tGT:PS:HP:TC:AC:SC:SE");
    for (uint32_t sm = 0; sm < n_samples; ++sm) {
      if (first[sm] < 0) { s.put("\t./.:.:.:.:.:.:."); continue; }
      const otg_allele& a1 = A[first[sm]]; const otg_allele& a2 = A[second[sm]];
      s.ch('\t'); s.i64(g2[(size_t)first[sm]]); s.ch('/'); s.i64(g2[(size_t)second[sm]]); s.ch(':'); s.i64(a1.ps); s.ch(':'); s.i64(a1.hp); s.ch(':'); s.i64(a1.tcov);
      s.ch(':'); s.i64(a1.acov); s.ch(','); s.i64(a2.acov); s.ch(':'); s.i64(a1.scov); s.ch(','); s.i64(a2.scov); s.ch(':'); s.flt(a1.se); s.ch(','); s.flt(a2.se);
    }
    s.ch('\n');
  }
  *out_len = s.len;
  if (s.len > out_capacity || (!out && s.len)) return OTG_ERR_CAPACITY;
  return OTG_OK;
}

int otg_emit_genotype_lengths(const otg_bam* bam, const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const uint32_t* first_allele,
                              const otg_allele* alleles, uint32_t n_samples, char* out, uint64_t out_capacity, uint64_t* out_len)
{
  if (!bam || (n_regions && (!beds || !first_allele)) || !out_len) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_genotype_lengths: null argument");
  Sink s{out, out_capacity, 0};
  std::vector<int> first(n_samples + 1), second(n_samples + 1);
  for (uint32_t r = 0; r < n_regions; ++r) {
    const uint32_t a0 = first_allele[r], na = first_allele[r + 1] - a0;
    if (na == 0) continue;
    const otg_allele* A = alleles + a0;
    std::fill(first.begin(), first.end(), -1); std::fill(second.begin(), second.end(), -1);
    for (int i = 0; i < (int)na; ++i) {
      const int sm = A[i].label;
      if (sm < 0 || (uint32_t)sm > n_samples) return otg_fail(nullptr, OTG_ERR_ARG, "otg_emit_genotype_lengths: sample index %d out of range", sm);
      if (first[(size_t)sm] < 0) { first[(size_t)sm] = i; second[(size_t)sm] = i; }
      else if (i > second[(size_t)sm]) second[(size_t)sm] = i;
    }
    const otg_bed& b = beds[r];
    for (uint32_t sm = 0; sm < n_samples; ++sm) {
      if (first[sm] < 0) continue;
      const int a1 = (int)A[first[sm]].seq_len, a2 = (int)A[second[sm]].seq_len;
      const char* nm = otg_bam_sample(bam, sm);
      s.put(chr_arena + b.chr_off, b.chr_len); s.ch(':'); s.u64((uint32_t)b.start); s.ch('-'); s.u64((uint32_t)b.end); s.ch('\t'); s.put(nm ? nm : "");
      s.ch('\t'); s.i64(a1 < a2 ? a1 : a2); s.ch('\t'); s.i64(a1 > a2 ? a1 : a2); s.ch('\n');
    }
  }
  *out_len = s.len;
  if (s.len > out_capacity || (!out && s.len)) return OTG_ERR_CAPACITY;
  return OTG_OK;
}

} // extern "C"
