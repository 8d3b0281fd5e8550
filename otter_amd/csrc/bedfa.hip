// bedfa.hip — the two small text inputs either side of the BAM ingest of `otter assemble`:
//   * BED file -> region list, as parse_bed_file / parse_bed / parse_sc_bed read it (src/anbed.cpp:23-80);
//   * indexed FASTA -> the two reference flanks local_realignment aligns clipped read ends to
//     (src/analignments.cpp:22,28 via FaidxInstance::fetch src/anfahelper.cpp:8-18 -> faidx_fetch_seq src/faidx.c:418-445;
//     index text format src/faidx.c:135-176, index build src/faidx.c:64-133).
// Host code only (stdio + pread), no device work; lives in the library so that the C-ABI covers `otter assemble` from the
// BED / BAM / FASTA files to the emitted records.
#include "otg_common.hpp"
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>

namespace {

// std::stoul as the reference calls it (base 10): leading white space and a sign are accepted, parsing stops at the first
// other byte; no digits or a value beyond unsigned long throw there (the reference then terminates) -> false here.
bool stoul_like(const std::string& s, unsigned long* v)
{
  const char* p = s.c_str();
  char* e = nullptr;
  errno = 0;
  const unsigned long x = strtoul(p, &e, 10);
  if (e == p || errno == ERANGE) return false;
  *v = x;
  return true;
}

// std::getline(stream, value, delim) token walk: a trailing delimiter yields no empty last token, an empty string no token
template <class F> void split_like_getline(const std::string& s, char delim, F&& f)
{
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find(delim, i);
    if (j == std::string::npos) { f(s.substr(i)); return; }
    f(s.substr(i, j - i));
    i = j + 1;
  }
}

struct FaiEntry { int64_t len; uint64_t offset; int32_t line_blen, line_len; };

} // namespace

struct otg_fasta {
  int fd = -1;
  std::unordered_map<std::string, FaiEntry> idx;
  std::vector<std::string> order;
};

namespace {

void fai_insert(otg_fasta* f, const std::string& name, int64_t len, int32_t line_len, int32_t line_blen, uint64_t offset)
{
  // a repeated name overwrites the earlier record (kh_put returns the existing slot, src/faidx.c:47-61); lengths are `int` there
  auto it = f->idx.find(name);
  if (it != f->idx.end()) { it->second = FaiEntry{(int64_t)(int)len, offset, line_blen, line_len}; return; }
  f->idx.emplace(name, FaiEntry{(int64_t)(int)len, offset, line_blen, line_len});
  f->order.push_back(name);
}

// index a plain FASTA the way fai_build_core does (src/faidx.c:64-133): per record the base count, the byte offset of the first base,
// bases per line and bytes per line; ragged lines are allowed only at the end of a record
bool fai_build(otg_fasta* f, FILE* fp, std::string& err)
{
  std::string name;
  int64_t len = -1;
  int32_t line_len = -1, line_blen = -1;
  int state = 0;
  uint64_t offset = 0, pos = 0;
  int ci;
  auto get = [&]() { ci = fgetc(fp); if (ci != EOF) ++pos; return ci != EOF; };
  while (get()) {
    char c = (char)ci;
    if (c == '\n') {
      if (state == 1) { offset = pos; continue; }
      else if ((state == 0 && len < 0) || state == 2) continue;
    }
    if (c == '>') {
      if (len >= 0) fai_insert(f, name, len, line_len, line_blen, offset);
      name.clear();
      bool more;
      while ((more = get()) && !isspace((unsigned char)ci)) name.push_back((char)ci);
      if (!more) { err = "the last FASTA entry has no sequence"; return false; }
      if ((char)ci != '\n') while (get() && (char)ci != '\n') {}
      state = 1; len = 0;
      offset = pos;
    } else {
      if (state == 3) { err = "inlined empty line in sequence '" + name + "'"; return false; }
      if (state == 2) state = 3;
      int32_t l1 = 0, l2 = 0;
      bool more;
      do { ++l1; if (isgraph((unsigned char)ci)) ++l2; } while ((more = get()) && (char)ci != '\n');
      if (state == 3 && l2) { err = "different line length in sequence '" + name + "'"; return false; }
      ++l1; len += l2;
      if (state == 1) { line_len = l1; line_blen = l2; state = 0; }
      else if (state == 0) { if (l1 != line_len || l2 != line_blen) state = 2; }
    }
  }
  if (len < 0) { err = "no FASTA record"; return false; }
  fai_insert(f, name, len, line_len, line_blen, offset);
  return true;
}

// faidx_fetch_seq (src/faidx.c:418-445): both ends inclusive, clamped into the contig; bytes that are not printable are skipped
int64_t fa_fetch(const otg_fasta* f, const std::string& chr, int beg, int end, std::string& seq)
{
  seq.clear();
  auto it = f->idx.find(chr);
  if (it == f->idx.end()) return 0;
  const FaiEntry& v = it->second;
  if (v.len <= 0 || v.line_blen <= 0) return 0;
  if (end < beg) beg = end;
  if (beg < 0) beg = 0; else if (v.len <= beg) beg = (int)(v.len - 1);
  if (end < 0) end = 0; else if (v.len <= end) end = (int)(v.len - 1);
  const int64_t want = (int64_t)end - beg + 1;
  uint64_t pos = v.offset + (uint64_t)(beg / v.line_blen) * (uint64_t)v.line_len + (uint64_t)(beg % v.line_blen);
  char buf[4096];
  while ((int64_t)seq.size() < want) {
    const ssize_t got = pread(f->fd, buf, sizeof buf, (off_t)pos);
    if (got <= 0) break;
    pos += (uint64_t)got;
    for (ssize_t i = 0; i < got && (int64_t)seq.size() < want; ++i)
      if (isgraph((unsigned char)buf[i])) seq.push_back((char)toupper((unsigned char)buf[i]));     // upper-cased by FaidxInstance::fetch
  }
  return (int64_t)seq.size();
}

} // namespace

extern "C" {

int otg_parse_bed_file(const char* path, otg_bed* beds, uint32_t beds_capacity, uint32_t* n_beds, char* chr_arena,
                       uint64_t chr_capacity, uint64_t* chr_used, uint32_t* n_skipped)
{
  if (!path || !n_beds || !chr_used) return otg_fail(nullptr, OTG_ERR_ARG, "otg_parse_bed_file: null argument");
  FILE* fp = fopen(path, "rb");
  // an unreadable file is an empty list in the reference (std::ifstream, no check); reported here instead
  if (!fp) return otg_fail(nullptr, OTG_ERR_ARG, "otg_parse_bed_file: cannot open %s", path);
  uint32_t n = 0, skipped = 0;
  uint64_t used = 0;
  bool overflow = false;
  std::string line;
  std::vector<std::string> cols;
  int rc = OTG_OK;
  long lineno = 0;
  auto handle = [&]() {
    ++lineno;
    if (line.empty()) { ++skipped; return; }                 // "[WARNING] Skipping empty BED line"
    if (line[0] == '#') return;
    cols.clear();
    split_like_getline(line, '\t', [&](std::string&& v) { cols.emplace_back(std::move(v)); });
    std::string chr;
    long long start = -1, end = -1;
    if (cols.size() == 1) {
      // single column "chr:start-end" (parse_sc_bed): field 0 up to ':' is the name, field 1 is split on '-'
      int index = 0;
      bool bad = false;
      split_like_getline(cols[0], ':', [&](std::string&& v) {
        if (index == 0) chr = v;
        else if (index == 1) {
          int index2 = 0;
          split_like_getline(v, '-', [&](std::string&& v2) {
            unsigned long x = 0;
            if (index2 < 2) { if (!stoul_like(v2, &x)) bad = true; else if (index2 == 0) start = (int)(uint32_t)x; else end = (int)(uint32_t)x; }
            ++index2;
          });
        }
        ++index;
      });
      if (bad) { rc = OTG_ERR_ARG; return; }
      if (chr.empty() || start < 0 || end < 0) { ++skipped; return; }        // "Skipping ambiguous multi-BED line"
    } else if (cols.size() < 3) { ++skipped; return; }                       // "Skipping ambiguous BED line"
    else {
      unsigned long a = 0, b = 0;
      if (!stoul_like(cols[1], &a) || !stoul_like(cols[2], &b)) { rc = OTG_ERR_ARG; return; }
      chr = cols[0]; start = (int)(uint32_t)a; end = (int)(uint32_t)b;
    }
    if (n < beds_capacity && beds && chr_arena && used + chr.size() <= chr_capacity) {
      memset(&beds[n], 0, sizeof(otg_bed));
      beds[n].chr_off = used; beds[n].chr_len = (uint32_t)chr.size(); beds[n].start = (int32_t)start; beds[n].end = (int32_t)end;
      memcpy(chr_arena + used, chr.data(), chr.size());
    } else overflow = true;
    ++n; used += chr.size();
  };
  int ci;
  bool any = false;
  line.clear();
  while ((ci = fgetc(fp)) != EOF) {
    any = true;
    if (ci == '\n') { handle(); line.clear(); any = false; if (rc != OTG_OK) break; }
    else line.push_back((char)ci);
  }
  if (rc == OTG_OK && any) handle();                          // last line without a newline
  fclose(fp);
  if (rc != OTG_OK) return otg_fail(nullptr, rc, "otg_parse_bed_file: %s line %ld: coordinate is not a number (the reference terminates here)", path, lineno);
  *n_beds = n; *chr_used = used;
  if (n_skipped) *n_skipped = skipped;
  return overflow ? OTG_ERR_CAPACITY : OTG_OK;
}

int otg_fasta_open(const char* path, otg_fasta** out)
{
  if (!path || !out) return otg_fail(nullptr, OTG_ERR_ARG, "otg_fasta_open: null argument");
  *out = nullptr;
  otg_fasta* f = new otg_fasta();
  f->fd = open(path, O_RDONLY);
  if (f->fd < 0) { delete f; return otg_fail(nullptr, OTG_ERR_ARG, "otg_fasta_open: cannot open %s", path); }
  unsigned char magic[2] = {0, 0};
  if (pread(f->fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
    close(f->fd); delete f;
    return otg_fail(nullptr, OTG_ERR_ARG, "otg_fasta_open: %s is compressed (RAZF / gzip FASTA is not supported, decompress it)", path);
  }
  const std::string fai = std::string(path) + ".fai";
  FILE* fp = fopen(fai.c_str(), "rb");
  if (fp) {
    // fai_read (src/faidx.c:151-176): name = the line up to the first non-printable byte, then "%d%lld%d%d"
    std::vector<char> buf(0x10000);
    while (fgets(buf.data(), (int)buf.size(), fp)) {
      char* p = buf.data();
      while (*p && isgraph((unsigned char)*p)) ++p;
      const bool had_rest = *p != 0;
      *p = 0;
      int len = 0, line_len = 0, line_blen = 0; long long offset = 0;
      if (had_rest) sscanf(p + 1, "%d%lld%d%d", &len, &offset, &line_blen, &line_len);
      fai_insert(f, buf.data(), len, line_len, line_blen, (uint64_t)offset);
    }
    fclose(fp);
  } else {
    // no index yet: build it (fai_load -> fai_build, src/faidx.c:178-225) and leave `<fasta>.fai` beside the file when that is writable
    FILE* fa = fopen(path, "rb");
    std::string err;
    const bool ok = fa && fai_build(f, fa, err);
    if (fa) fclose(fa);
    if (!ok) { close(f->fd); delete f; return otg_fail(nullptr, OTG_ERR_ARG, "otg_fasta_open: cannot index %s: %s", path, err.c_str()); }
    if (FILE* w = fopen(fai.c_str(), "wb")) {
      for (const std::string& nm : f->order) { const FaiEntry& x = f->idx[nm]; fprintf(w, "%s\t%d\t%lld\t%d\t%d\n", nm.c_str(), (int)x.len, (long long)x.offset, (int)x.line_blen, (int)x.line_len); }
      fclose(w);
    }
  }
  *out = f;
  return OTG_OK;
}

void otg_fasta_close(otg_fasta* f) { if (f) { if (f->fd >= 0) close(f->fd); delete f; } }

uint32_t otg_fasta_n_seqs(const otg_fasta* f) { return f ? (uint32_t)f->order.size() : 0; }

const char* otg_fasta_seq(const otg_fasta* f, uint32_t i, int64_t* length)
{
  if (!f || i >= f->order.size()) return nullptr;
  if (length) *length = f->idx.at(f->order[i]).len;
  return f->order[i].c_str();
}

int otg_fasta_fetch(const otg_fasta* f, const char* chr, uint32_t chr_len, int32_t beg, int32_t end_inclusive, char* out,
                    uint64_t out_capacity, uint64_t* out_len)
{
  if (!f || !chr || !out_len) return otg_fail(nullptr, OTG_ERR_ARG, "otg_fasta_fetch: null argument");
  std::string s;
  fa_fetch(f, std::string(chr, chr_len), beg, end_inclusive, s);
  *out_len = s.size();
  if (s.size() > out_capacity || (!out && !s.empty())) return OTG_ERR_CAPACITY;
  if (!s.empty()) memcpy(out, s.data(), s.size());
  return OTG_OK;
}

int otg_fasta_region_flanks(const otg_fasta* f, const otg_bed* beds, const char* chr_arena, uint32_t n_regions, int32_t offset_l,
                            int32_t offset_r, int32_t flank, uint8_t* arena, uint64_t arena_capacity, uint64_t* arena_used,
                            otg_region* regions)
{
  if (!f || (n_regions && (!beds || !regions)) || !arena_used) return otg_fail(nullptr, OTG_ERR_ARG, "otg_fasta_region_flanks: null argument");
  uint64_t used = *arena_used;
  bool overflow = false;
  std::string l, r;
  for (uint32_t g = 0; g < n_regions; ++g) {
    // mod_bed (src/assemble.cpp:55-57); flanks [start - flank, start] and [end, end + flank], both ends inclusive
    const int start = beds[g].start - offset_l, end = beds[g].end + offset_r;
    const std::string chr(chr_arena + beds[g].chr_off, beds[g].chr_len);
    fa_fetch(f, chr, start - flank, start, l);
    fa_fetch(f, chr, end, end + flank, r);
    if (!overflow && arena && used + l.size() + r.size() + 64 <= arena_capacity) {
      regions[g].flank_l_off = used; regions[g].flank_l_len = (uint32_t)l.size();
      if (!l.empty()) memcpy(arena + used, l.data(), l.size());
      regions[g].flank_r_off = used + l.size(); regions[g].flank_r_len = (uint32_t)r.size();
      if (!r.empty()) memcpy(arena + used + l.size(), r.data(), r.size());
    } else overflow = true;
    used += l.size() + r.size();
  }
  *arena_used = used;
  return overflow ? OTG_ERR_CAPACITY : OTG_OK;
}

} // extern "C"
