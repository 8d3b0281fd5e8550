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

} // extern "C"
