// fuzz_host_io — drives the host-side readers of libotter_gpu (BAM / BAI ingest, allele ingest, wgat, BED parser, FASTA index + fetch) over
// possibly corrupt files; built with -fsanitize=address,undefined from the same sources (tests/test_host_sanitizers.py).  Exit code 0 = every
// call came back (with OTG_OK or an error code) and the sanitizers saw nothing; a sanitizer report aborts the process.
//   fuzz_host_io <bam> <bed> [<fasta>]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/otter_gpu.h"

static int sink(void*, const char*, uint64_t) { return 0; }

int main(int argc, char** argv)
{
  if (argc < 3) return 2;
  const char* bam_path = argv[1]; const char* bed_path = argv[2]; const char* fa_path = argc > 3 ? argv[3] : nullptr;
  int errors = 0;
  unsigned long long seen_reads = 0, seen_alleles = 0, seen_wgat = 0;
  std::vector<otg_bed> beds; std::vector<char> chr;
  {
    uint32_t n = 0, sk = 0; uint64_t cu = 0;
    int rc = otg_parse_bed_file(bed_path, nullptr, 0, &n, nullptr, 0, &cu, &sk);
    if (rc == OTG_OK || rc == OTG_ERR_CAPACITY) {
      beds.resize(n + 1); chr.resize(cu + 16);
      rc = otg_parse_bed_file(bed_path, beds.data(), (uint32_t)beds.size(), &n, chr.data(), chr.size(), &cu, &sk);
      beds.resize(rc == OTG_OK ? n : 0);
    } else { beds.clear(); ++errors; }
  }
  otg_fasta* fa = nullptr;
  if (fa_path && otg_fasta_open(fa_path, &fa) != OTG_OK) { fa = nullptr; ++errors; }
  if (fa) {
    char buf[512]; uint64_t len = 0;
    for (uint32_t i = 0; i < otg_fasta_n_seqs(fa); ++i) { int64_t l = 0; const char* nm = otg_fasta_seq(fa, i, &l); (void)otg_fasta_fetch(fa, nm, (uint32_t)strlen(nm), (int32_t)(l > 100 ? l - 100 : 0), (int32_t)(l + 50), buf, sizeof buf, &len); }
  }
  otg_bam* bam = nullptr;
  if (otg_bam_open(bam_path, &bam) != OTG_OK) { if (fa) otg_fasta_close(fa); printf("open failed (%s), errors %d\n", otg_last_error(nullptr), errors + 1); return 0; }
  const uint32_t nb = (uint32_t)beds.size();
  for (int threads = 1; threads <= 3; threads += 2) {
    otg_ingest_opts o; memset(&o, 0, sizeof o); o.offset_l = 1; o.threads = threads;
    std::vector<uint8_t> arena(64u << 20); std::vector<otg_read> reads(1u << 18); std::vector<otg_region> regs(nb + 1);
    std::vector<otg_read_meta> meta(reads.size()); std::vector<char> names(16u << 20);
    uint64_t used = 0, nused = 0; uint32_t nr = 0;
    if (otg_ingest_regions_named(bam, beds.data(), chr.data(), nb, &o, arena.data(), arena.size(), &used, reads.data(), (uint32_t)reads.size(), &nr, regs.data(), meta.data(), names.data(),
                                 names.size(), &nused) != OTG_OK) ++errors;
    else if (fa) { uint64_t u2 = used; if (otg_fasta_region_flanks(fa, beds.data(), chr.data(), nb, 1, 0, 100, arena.data(), arena.size(), &u2, regs.data()) != OTG_OK) ++errors; }
    seen_reads += nr;
  }
  {
    uint32_t ns = 0; int32_t ol = 0, orr = 0;
    if (otg_bam_sample_index(bam, &ns, &ol, &orr) == OTG_OK) {
      std::vector<uint8_t> arena(64u << 20); std::vector<otg_allele> al(1u << 18); std::vector<uint32_t> first(nb + 2);
      uint64_t used = 0; uint32_t na = 0;
      if (otg_ingest_alleles(bam, beds.data(), chr.data(), nb, 2, fa, arena.data(), arena.size(), &used, al.data(), (uint32_t)al.size(), &na, first.data()) != OTG_OK) ++errors;
      seen_alleles += na;
    } else ++errors;
  }
  { uint64_t n = 0; if (otg_wgat(bam, beds.data(), chr.data(), nb, "rg", 0, 1, 0, sink, nullptr, &n) != OTG_OK) ++errors; seen_wgat += n; }
  otg_bam_close(bam);
  if (fa) otg_fasta_close(fa);
  printf("done, %d calls returned an error code; reads %llu, alleles %llu, wgat records %llu\n", errors, seen_reads, seen_alleles, seen_wgat);
  return 0;
}
