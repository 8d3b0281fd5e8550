// ingest.hip — BAM/BAI region ingest (the "next" row in front of the hot path): for every BED region the
// reads overlapping it, cut to the region and flagged, as `parse_anreads` hands them to the five hot-path calls
// (reference: src/anseqs.cpp:244-460 — parse_standard_auxs, get_breakpoints, parse_alignment, parse_anreads — called at
// src/assemble.cpp:55-65 on the offset-widened region; file formats: BGZF / BAM / BAI as read by the htslib-lite the
// reference vendors, src/bgzf.c, src/sam.c, src/hts.c:690-870).  Host code (north_star keeps ingest on the host); written
// from the format definitions, not from those sources: a BGZF block reader on zlib, a BAI loader, the standard
// bin + linear-index region query, and a restatement of the reference's CIGAR walk with all of its quirks.
// Output goes straight into the region batch of the L3 pipeline (otg_read / otg_region + byte arena).
#include "otg_common.hpp"
#include <cerrno>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <string>
#include <thread>
#include <memory>
#include <unordered_map>
#include <vector>

namespace {

static std::atomic<unsigned long long> g_blocks_inflated{0};      // OTG_DEBUG statistics

struct Bgzf {
  FILE* fp = nullptr;
  uint64_t block_address = 0;      // file offset of the current (or next, when nothing is loaded) block
  uint32_t block_csize = 0;        // compressed size of the loaded block (0 = nothing loaded)
  uint32_t block_length = 0;       // its uncompressed size
  uint32_t block_offset = 0;
  std::vector<uint8_t> cbuf;
  // Inflated blocks are kept in a small LRU set: consecutive regions of a BED file query overlapping runs of blocks (a region's
  // chunk starts at the 16 kb window of the linear index, i.e. inside the records of the regions before it), so without it every
  // block is inflated about three times.
  struct Slot { uint64_t address = ~0ull; uint32_t csize = 0, length = 0; uint64_t stamp = 0; std::vector<uint8_t> data; };
  static constexpr int NSLOT = 32;
  Slot slots[NSLOT];
  int cur = 0;                     // slot of the loaded block
  uint64_t clock_ = 0;
  bool eof = false;
  z_stream zs;
  bool zs_ready = false;
  bool failed = false;           // the last load() met a malformed / truncated / non-inflatable block (as opposed to the end of the file)

  bool open(const char* path) { fp = fopen(path, "rb"); return fp != nullptr; }
  void close() { if (fp) fclose(fp); fp = nullptr; if (zs_ready) { inflateEnd(&zs); zs_ready = false; } }
  // loads the block at block_address; false on EOF / malformed data
  bool load() {
    block_csize = block_length = block_offset = 0;
    for (int i = 0; i < NSLOT; ++i) if (slots[i].address == block_address && slots[i].csize) {
      cur = i; slots[i].stamp = ++clock_; block_csize = slots[i].csize; block_length = slots[i].length;
      return true;
    }
    int victim = 0;
    for (int i = 1; i < NSLOT; ++i) if (slots[i].stamp < slots[victim].stamp) victim = i;
    Slot& S = slots[victim];
    S.address = ~0ull; S.csize = 0;
    std::vector<uint8_t>& ubuf = S.data;
    if (fseeko(fp, (off_t)block_address, SEEK_SET) != 0) return false;
    uint8_t h[12];
    const size_t hg = fread(h, 1, 12, fp);
    if (hg != 12) { eof = true; if (hg != 0) failed = true; return false; }      // 0 bytes: clean end of file; 1-11: truncated block header
    failed = true;                                                                 // cleared once the block has been inflated
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
    const uint32_t xlen = h[10] | (h[11] << 8);
    uint8_t extra[256];
    if (xlen > sizeof extra) return false;
    if (xlen && fread(extra, 1, xlen, fp) != xlen) return false;
    int bsize = -1;
    for (uint32_t i = 0; i + 4 <= xlen;) {
      const uint32_t slen = extra[i + 2] | (extra[i + 3] << 8);
      if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = extra[i + 4] | (extra[i + 5] << 8);
      i += 4 + slen;
    }
    if (bsize < 0) return false;
    const uint32_t total = (uint32_t)bsize + 1;
    if (total < 12 + xlen + 8) return false;
    const uint32_t clen = total - 12 - xlen - 8;
    cbuf.resize(clen + 8);
    if (fread(cbuf.data(), 1, clen + 8, fp) != clen + 8) return false;
    const uint8_t* t = cbuf.data() + clen;
    const uint32_t isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
    ubuf.resize(isize ? isize : 1);
    if (isize) {
      if (!zs_ready) { memset(&zs, 0, sizeof zs); if (inflateInit2(&zs, -15) != Z_OK) return false; zs_ready = true; }
      else if (inflateReset(&zs) != Z_OK) return false;            // one inflate state per reader, not one per block
      zs.next_in = cbuf.data(); zs.avail_in = clen; zs.next_out = ubuf.data(); zs.avail_out = isize;
      const int rc = inflate(&zs, Z_FINISH);
      if (rc != Z_STREAM_END || zs.total_out != isize) return false;
    }
    failed = false;
    block_csize = total; block_length = isize;
    S.address = block_address; S.csize = total; S.length = isize; S.stamp = ++clock_; cur = victim;
    g_blocks_inflated.fetch_add(1, std::memory_order_relaxed);
    return true;
  }
  // like bgzf_read: returns bytes read (< n at end of file)
  size_t read(void* dst, size_t n) {
    size_t got = 0;
    uint8_t* out = (uint8_t*)dst;
    while (got < n) {
      if (block_offset >= block_length) {
        if (block_csize) block_address += block_csize;          // move past the consumed block
        if (!load()) break;
        if (block_length == 0) { if (eof) break; continue; }     // empty block (e.g. the EOF marker): try the next one
      }
      const size_t k = std::min<size_t>(n - got, block_length - block_offset);
      memcpy(out + got, slots[cur].data.data() + block_offset, k);
      got += k; block_offset += (uint32_t)k;
    }
    if (block_csize && block_offset == block_length) {           // bgzf_tell semantics: a fully consumed block points at the next one
      block_address += block_csize; block_csize = block_length = block_offset = 0;
    }
    return got;
  }
  bool seek(uint64_t voff) {
    eof = false;
    if (block_csize && block_address == (voff >> 16)) { block_offset = (uint32_t)(voff & 0xffff); return block_offset <= block_length; }   // already inflated
    block_address = voff >> 16;
    if (!load()) return false;
    block_offset = (uint32_t)(voff & 0xffff);
    return block_offset <= block_length;
  }
  uint64_t tell() const { return (block_address << 16) | (block_offset & 0xffffu); }
};

struct RefIndex {
  std::unordered_map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
  std::vector<uint64_t> linear;
};

} // namespace

struct otg_bam {
  Bgzf fp;
  std::vector<std::string> names;
  std::vector<uint32_t> lengths;
  std::unordered_map<std::string, int> name2id;
  std::vector<RefIndex> idx;
  std::vector<uint8_t> rec;        // one decoded record
  std::string err;
  std::string path;
  std::string text;                // SAM header text of the BAM
  bool samples_parsed = false;     // SampleIndex (src/anbamdb.cpp): read groups in header order, offsets from `@PG ID:otter OF:`
  std::vector<std::string> index2sample;
  std::unordered_map<std::string, int> sample2index;
  int offset_l = 1, offset_r = 0;
};

namespace {

inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

bool load_bai(otg_bam* b, const std::string& path)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { b->err = "cannot open index " + path; return false; }
  auto rd = [&](void* p, size_t n) { return fread(p, 1, n, f) == n; };
  char magic[4]; int32_t n_ref = 0;
  bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0;
  if (ok) b->idx.resize((size_t)n_ref);
  for (int32_t r = 0; ok && r < n_ref; ++r) {
    int32_t n_bin = 0;
    ok = rd(&n_bin, 4);
    for (int32_t i = 0; ok && i < n_bin; ++i) {
      uint32_t bin; int32_t n_chunk;
      ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0;
      if (!ok) break;
      std::vector<std::pair<uint64_t, uint64_t>> ch((size_t)n_chunk);
      for (int32_t c = 0; ok && c < n_chunk; ++c) ok = rd(&ch[c].first, 8) && rd(&ch[c].second, 8);
      b->idx[r].bins[bin] = std::move(ch);
    }
    int32_t n_intv = 0;
    ok = ok && rd(&n_intv, 4) && n_intv >= 0;
    if (ok) {
      b->idx[r].linear.resize((size_t)n_intv);
      if (n_intv) ok = rd(b->idx[r].linear.data(), (size_t)n_intv * 8);
      for (int32_t j = 1; j < n_intv; ++j) if (b->idx[r].linear[j] == 0) b->idx[r].linear[j] = b->idx[r].linear[j - 1];
    }
  }
  fclose(f);
  if (!ok) b->err = "malformed BAI " + path;
  return ok;
}

// candidate bins of [beg, end) in the 6-level UCSC binning scheme (min_shift 14, 5 levels below the root)
void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>& out)
{
  if (beg >= end) return;
  if (end > (1LL << 29)) end = 1LL << 29;
  --end;
  int s = 29, t = 0;
  for (int l = 0; l <= 5; ++l) {
    const int64_t b = t + (beg >> s), e = t + (end >> s);
    for (int64_t i = b; i <= e; ++i) out.push_back((uint32_t)i);
    t += 1 << (3 * l);
    s -= 3;
  }
}

struct Rec {
  int32_t tid, pos, l_seq;
  uint32_t mapq, flag, n_cigar;
  const uint8_t* cigar;        // n_cigar 32-bit words at any alignment: only ever read through rd32 (cig_op / cig_len)
  const uint8_t* seq;
  const uint8_t* aux;
  const uint8_t* aux_end;
  const char* name;            // NUL-terminated query name inside the record
  uint32_t l_name;             // its length without the terminator
};

inline int cig_op(const uint8_t* c, uint32_t i) { return (int)(rd32(c + 4 * i) & 0xf); }
inline int cig_len(const uint8_t* c, uint32_t i) { return (int)(rd32(c + 4 * i) >> 4); }

const uint8_t* aux_get(const Rec& r, char t0, char t1);

// returns 1 record decoded, 0 end of file, -1 error
int read_record(otg_bam* b, Rec* r)
{
  int32_t block_len = 0;
  const size_t g = b->fp.read(&block_len, 4);
  if (g == 0) return b->fp.failed ? -1 : 0;                    // a block that does not inflate is an error, not the end of the file
  if (g != 4 || block_len < 32 || block_len > (1 << 29)) return -1;
  b->rec.resize((size_t)block_len);
  if (b->fp.read(b->rec.data(), (size_t)block_len) != (size_t)block_len) return -1;
  const uint8_t* p = b->rec.data();
  r->tid = (int32_t)rd32(p); r->pos = (int32_t)rd32(p + 4);
  const uint32_t bmn = rd32(p + 8), fnc = rd32(p + 12);
  const uint32_t l_name = bmn & 0xff;
  r->mapq = (bmn >> 8) & 0xff; r->flag = fnc >> 16; r->n_cigar = fnc & 0xffff;
  r->l_seq = (int32_t)rd32(p + 16);
  const size_t need = 32 + (size_t)l_name + 4 * (size_t)r->n_cigar + ((size_t)r->l_seq + 1) / 2 + (size_t)r->l_seq;
  if (r->l_seq < 0 || need > (size_t)block_len) return -1;
  const uint8_t* q = p + 32 + l_name;
  r->name = (const char*)p + 32;
  r->l_name = l_name ? (uint32_t)strnlen((const char*)p + 32, l_name) : 0;
  r->cigar = q;
  r->seq = q + 4 * (size_t)r->n_cigar;
  r->aux = r->seq + ((size_t)r->l_seq + 1) / 2 + (size_t)r->l_seq;
  r->aux_end = p + block_len;
  // Alignments with more than 65535 CIGAR operations (ultra-long reads) carry a placeholder `<l_seq>S<rlen>N` and the real CIGAR in the
  // tag CG:B,I; the reference's bam_read1 moves it into place (bam_tag2cigar, src/sam.c:243-285).  Same test, the ops are read in place.
  if (r->n_cigar != 0 && r->tid >= 0 && r->pos >= 0 && cig_op(r->cigar, 0) == 4 && cig_len(r->cigar, 0) == r->l_seq) {
    const uint8_t* cg = aux_get(*r, 'C', 'G');
    if (cg && cg[0] == 'B' && cg[1] == 'I') {
      const uint32_t n = rd32(cg + 2);
      // bam_tag2cigar's own condition: a CG array shorter than the placeholder, or absurdly long, leaves the placeholder in place
      if (n >= r->n_cigar && n < (1u << 29)) { r->cigar = cg + 6; r->n_cigar = n; }      // (bounds were checked by aux_get)
    }
  }
  return 1;
}

// reference length of the CIGAR (ops that consume the reference: M D N = X)
int cigar_rlen(const Rec& r)
{
  int l = 0;
  const uint8_t* c = r.cigar;
  for (uint32_t k = 0; k < r.n_cigar; ++k) { const int op = cig_op(c, k); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l += cig_len(c, k); }
  return l;
}

// pointer to the value (type byte first) of an aux tag, or null (bam_aux_get).  Every value is checked to lie inside the record
// (a string needs its terminator there), so the readers below never look past aux_end; a malformed tail ends the scan.
const uint8_t* aux_get(const Rec& r, char t0, char t1)
{
  const uint8_t* s = r.aux;
  const uint8_t* const end = r.aux_end;
  while (s + 3 <= end) {
    const bool hit = s[0] == (uint8_t)t0 && s[1] == (uint8_t)t1;
    const uint8_t* val = s + 2;
    s += 2;
    const int type = *s++;
    size_t sz = 0;
    switch (type) {
      case 'A': case 'c': case 'C': sz = 1; break;
      case 's': case 'S': sz = 2; break;
      case 'i': case 'I': case 'f': sz = 4; break;
      case 'd': sz = 8; break;
      case 'Z': case 'H': { const uint8_t* z = (const uint8_t*)memchr(s, 0, (size_t)(end - s)); if (!z) return nullptr; sz = (size_t)(z - s) + 1; break; }
      case 'B': {
        if (s + 5 > end) return nullptr;
        const int sub = *s; const uint32_t n = rd32(s + 1);
        const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        sz = 5 + es * (size_t)n; break;
      }
      default: return nullptr;
    }
    if (sz > (size_t)(end - s)) return nullptr;
    if (hit) return val;
    s += sz;
  }
  return nullptr;
}
// bam_aux2Z: the string of a Z / H value (empty for any other type; the terminator is inside the record, see aux_get)
std::string aux2str(const uint8_t* a) { return (a && (a[0] == 'Z' || a[0] == 'H')) ? std::string((const char*)a + 1) : std::string(); }
int32_t aux2i(const uint8_t* s)
{
  const int type = *s++;
  if (type == 'c') return (int32_t)(int8_t)s[0];
  if (type == 'C') return (int32_t)s[0];
  if (type == 's') { int16_t v; memcpy(&v, s, 2); return v; }
  if (type == 'S') { uint16_t v; memcpy(&v, s, 2); return v; }
  if (type == 'i' || type == 'I') { int32_t v; memcpy(&v, s, 4); return v; }
  return 0;
}
double aux2f(const uint8_t* s)
{
  const int type = *s++;
  if (type == 'd') { double v; memcpy(&v, s, 8); return v; }
  if (type == 'f') { float v; memcpy(&v, s, 4); return (double)v; }
  return 0.0;
}

struct ParseMsg { bool successful = true, spanning_l = true, spanning_r = true; int c_first = -1, c_second = -1; };

// get_breakpoints (src/anseqs.cpp:286-408), restated with its quirks; q_first / q_second = the extracted query interval
void get_breakpoints(int start, int end, const Rec& r, ParseMsg& msg, bool& have, int& q_first, int& q_second)
{
  bool clipped_l = false, clipped_r = false;
  int qstart_dist = -1;
  int leftmost_q = -1, rightmost_q = -1, leftmost_r = -1, rightmost_r = -1;
  int qstart_q = -1, qend_q = -1;
  uint32_t qstart_cigar_i = 0, qend_cigar_i = 0;
  const uint8_t* cg = r.cigar;
  int rpos = r.pos, qpos = 0;
  for (uint32_t i = 0; i < r.n_cigar; ++i) {
    const int op = cig_op(cg, i), ol = cig_len(cg, i);
    if (op == 5 || op == 4) {                      // H, S
      if (i == 0) clipped_l = true;
      if (i == r.n_cigar - 1) clipped_r = true;
      if (op == 4) qpos += ol;
    } else if (op == 0 || op == 7 || op == 8) {    // M, =, X
      // The reference visits every base of the op (:301-322).  Reference positions only grow along the CIGAR, so what its
      // per-base updates leave behind is: leftmost = first aligned base, rightmost = last aligned base, the query start = the
      // FIRST aligned base at or after `start` (later ones are farther), the query end = the LAST aligned base at or before
      // `end` (every closer one overwrites) — computed per op here.
      if (ol > 0) {
        if (leftmost_q == -1) { leftmost_q = qpos; leftmost_r = rpos; }
        rightmost_q = qpos + ol - 1; rightmost_r = rpos + ol - 1;
        if (qstart_dist < 0) {
          long long j0 = (long long)start - rpos; if (j0 < 0) j0 = 0;
          if (j0 < ol) { qstart_dist = (int)(rpos + j0 - start); qstart_q = qpos + (int)j0; qstart_cigar_i = i; }
        }
        long long j1 = (long long)end - rpos;
        if (j1 >= 0) {
          if (j1 > ol - 1) j1 = ol - 1;
          qend_q = qpos + (int)j1; qend_cigar_i = i;
        }
        rpos += ol; qpos += ol;
      }
    } else if (op == 1) qpos += ol;                // I
    else if (op == 2) rpos += ol;                  // D   (N, P: ignored, as in the reference)
  }
  if (rightmost_r < start || leftmost_r > end) {
    qstart_q = qend_q = -1;
    msg.successful = false; msg.spanning_l = false; msg.spanning_r = false;
  } else if (qstart_q > -1 && qend_q > -1 && qstart_q > qend_q) {     // region deleted in the read
    qstart_q = qend_q = -1;
    msg.successful = true; msg.spanning_l = true; msg.spanning_r = true;
  } else {
    msg.c_first = qstart_q; msg.c_second = qend_q;
    if (leftmost_r > start && clipped_l && qstart_cigar_i == 1) {
      while (qstart_q > 0 && qstart_cigar_i > 0) {
        const int op = cig_op(cg, qstart_cigar_i - 1), ol = cig_len(cg, qstart_cigar_i - 1);
        if (op == 2) --qstart_cigar_i;
        else if (op == 5 || op == 4 || op == 1) { qstart_q -= ol; --qstart_cigar_i; }
        else break;
      }
    }
    if (rightmost_r < end && clipped_r && qend_cigar_i == r.n_cigar - 1) {
      while (qend_q < r.l_seq - 1 && qend_cigar_i < r.n_cigar) {
        const int op = cig_op(cg, qend_cigar_i - 1), ol = cig_len(cg, qend_cigar_i - 1);
        if (op == 2) ++qend_cigar_i;
        else if (op == 5 || op == 4 || op == 1) { qend_q += ol; ++qend_cigar_i; }
        else break;
      }
    }
    msg.spanning_l = leftmost_q >= 0 && leftmost_r <= start;
    msg.spanning_r = rightmost_q >= 0 && rightmost_r >= end;
    msg.successful = true;
  }
  have = false;
  if (msg.successful) {
    have = true;
    if (msg.spanning_l && msg.spanning_r) { q_first = qstart_q; q_second = qend_q; }
    else if (msg.spanning_l) { q_first = qstart_q; q_second = r.l_seq; }
    else if (msg.spanning_r) { q_first = 0; q_second = qend_q; }
    else { q_first = 0; q_second = r.l_seq; }
  }
}

} // namespace

extern "C" {

int otg_bam_open(const char* bam_path, otg_bam** out)
{
  if (!bam_path || !out) return otg_fail(nullptr, OTG_ERR_ARG, "otg_bam_open: null argument");
  otg_bam* b = new otg_bam();
  auto bail = [&](const std::string& m) { const int rc = otg_fail(nullptr, OTG_ERR_ARG, "otg_bam_open(%s): %s", bam_path, m.c_str()); delete b; return rc; };
  if (!b->fp.open(bam_path)) return bail("cannot open");
  char magic[4]; int32_t l_text = 0, n_ref = 0;
  if (b->fp.read(magic, 4) != 4 || memcmp(magic, "BAM\1", 4) != 0) { b->fp.close(); return bail("not a BAM file"); }
  if (b->fp.read(&l_text, 4) != 4 || l_text < 0) { b->fp.close(); return bail("truncated header"); }
  std::vector<char> text((size_t)l_text + 1);
  if (l_text && b->fp.read(text.data(), (size_t)l_text) != (size_t)l_text) { b->fp.close(); return bail("truncated header text"); }
  b->text.assign(text.data(), (size_t)l_text);
  if (b->fp.read(&n_ref, 4) != 4 || n_ref < 0) { b->fp.close(); return bail("truncated reference list"); }
  for (int32_t i = 0; i < n_ref; ++i) {
    int32_t l_name = 0; uint32_t l_ref = 0;
    if (b->fp.read(&l_name, 4) != 4 || l_name <= 0) { b->fp.close(); return bail("bad reference name"); }
    std::string nm((size_t)l_name, '\0');
    if (b->fp.read(&nm[0], (size_t)l_name) != (size_t)l_name || b->fp.read(&l_ref, 4) != 4) { b->fp.close(); return bail("truncated reference entry"); }
    nm.resize(strlen(nm.c_str()));
    b->name2id.emplace(nm, i);
    b->names.push_back(nm); b->lengths.push_back(l_ref);
  }
  if (!load_bai(b, std::string(bam_path) + ".bai")) { const std::string e = b->err; b->fp.close(); return bail(e); }   // src/anbamfilehelper.cpp:20
  b->path = bam_path;
  *out = b;
  return OTG_OK;
}

void otg_bam_close(otg_bam* b) { if (b) { b->fp.close(); delete b; } }

uint32_t otg_bam_n_targets(const otg_bam* b) { return b ? (uint32_t)b->names.size() : 0; }
const char* otg_bam_target(const otg_bam* b, uint32_t i, uint64_t* length)
{
  if (!b || i >= b->names.size()) return nullptr;
  if (length) *length = b->lengths[i];
  return b->names[i].c_str();
}

} // extern "C"

// Every record htslib's iterator would return for [qbeg, qend) on target tid (bins of the region, linear-index lower bound,
// overlap filter; src/hts.c:690-870), in file order, on the private reader `local`.  fn(record) is called for each.
template <class F>
static int scan_region(const otg_bam* b, otg_bam& local, int tid, long long qbeg, long long qend, std::vector<uint32_t>& bins,
                       std::vector<std::pair<uint64_t, uint64_t>>& chunks, std::string& err, F&& fn)
{
  const RefIndex& ri = b->idx[(size_t)tid];
  uint64_t min_off = 0;
  if (!ri.linear.empty()) { const size_t w = (size_t)(qbeg >> 14); min_off = ri.linear[w < ri.linear.size() ? w : ri.linear.size() - 1]; }
  bins.clear(); chunks.clear();
  reg2bins(qbeg, qend, bins);
  for (uint32_t bn : bins) { auto f = ri.bins.find(bn); if (f != ri.bins.end()) for (auto& c : f->second) if (c.second > min_off) chunks.push_back(c); }
  std::sort(chunks.begin(), chunks.end());
  size_t m = 0;
  for (size_t i = 0; i < chunks.size(); ++i) {              // merge overlapping / adjacent chunks
    if (m && chunks[i].first <= chunks[m - 1].second) chunks[m - 1].second = std::max(chunks[m - 1].second, chunks[i].second);
    else chunks[m++] = chunks[i];
  }
  chunks.resize(m);
  bool finished = false;
  for (size_t ci = 0; ci < chunks.size() && !finished; ++ci) {
    if (!local.fp.seek(chunks[ci].first)) { err = "cannot seek in BAM"; return OTG_ERR_ARG; }
    while (local.fp.tell() < chunks[ci].second) {
      Rec r;
      const int rc = read_record(&local, &r);
      if (rc == 0) { finished = true; break; }
      if (rc < 0) { err = "malformed BAM record"; return OTG_ERR_ARG; }
      const long long rend = (long long)r.pos + (r.n_cigar ? cigar_rlen(r) : 1);
      if (r.tid != tid || r.pos >= qend) { finished = true; break; }
      if (!(rend > qbeg && qend > r.pos)) continue;
      fn(r);
    }
  }
  return OTG_OK;
}

// one contiguous slice of regions on its own file handle: reads + bytes into private vectors, regions[].first_read relative
static int ingest_slice(const otg_bam* b, const char* path, const otg_bed* beds, const char* chr_arena, uint32_t g0, uint32_t g1,
                        const otg_ingest_opts* opts, std::vector<otg_read>& reads, std::vector<uint8_t>& arena, otg_region* regions, std::string& err,
                        std::vector<otg_read_meta>* meta, std::string* names)
{
  static const char nt16[] = "=ACMGRSVTWYHKDBN";
  otg_bam local;                               // private BGZF reader + record buffer; the index is read through `b`
  if (!local.fp.open(path)) { err = "cannot reopen BAM"; return OTG_ERR_ARG; }
  std::vector<uint32_t> bins;
  std::vector<std::pair<uint64_t, uint64_t>> chunks;
  std::string seq;
  int rc_out = OTG_OK;
  for (uint32_t g = g0; g < g1 && rc_out == OTG_OK; ++g) {
    memset(&regions[g], 0, sizeof(otg_region));
    const uint32_t first = (uint32_t)reads.size();
    regions[g].first_read = first;
    // the query region is the BED region widened by the offsets (src/assemble.cpp:55-57), passed to htslib as "chr:start-end":
    // the start is read 1-based (beg = start - 1, src/hts.c:813).  BED coordinates are unsigned in the reference, so a start
    // below the offset wraps; the wrapped number parses to a negative int, which htslib clamps to 0, while the CIGAR walk
    // sees the (negative) int — exactly what the plain int arithmetic here gives.
    const int start = beds[g].start - opts->offset_l, end = beds[g].end + opts->offset_r;
    const std::string chr(chr_arena + beds[g].chr_off, beds[g].chr_len);
    auto it = b->name2id.find(chr);
    long long qbeg = (long long)start - 1; if (qbeg < 0) qbeg = 0;
    const long long qend = end;
    if (it == b->name2id.end() || end < 0 || qbeg > qend) { regions[g].n_reads = 0; continue; }
    rc_out = scan_region(b, local, it->second, qbeg, qend, bins, chunks, err, [&](const Rec& r) {
      // ---- parse_anreads (src/anseqs.cpp:444-457)
      if (!((int)r.mapq >= opts->mapq && (opts->nonprimary || !(r.flag & 0x100u || r.flag & 0x800u)))) return;
      ParseMsg msg; bool have = false; int q_first = 0, q_second = 0;
      get_breakpoints(start, end, r, msg, have, q_first, q_second);
      if (!msg.successful) return;
      seq.clear();
      if (q_first == -1 || r.l_seq < (q_second - q_first)) seq = "N";          // parse_alignment :421-432
      else {
        const int l_sub = q_second - q_first, l_og = msg.c_second - msg.c_first;
        msg.c_first = msg.c_first - q_first;
        msg.c_second = msg.c_first + l_og;
        seq.resize(l_sub > 0 ? (size_t)l_sub : 0);
        for (int i = 0; i < l_sub; ++i) { const int qi = i + q_first; seq[(size_t)i] = nt16[(r.seq[qi >> 1] >> ((~qi & 1) << 2)) & 0xf]; }
        if (seq.empty()) seq = "N";
      }
      if (opts->omit_nonspanning && !(msg.spanning_l && msg.spanning_r)) return;
      int32_t hp = -1, ps = -1; double rq = 0.0;
      if (const uint8_t* a = aux_get(r, 'H', 'P')) hp = aux2i(a);
      if (const uint8_t* a = aux_get(r, 'P', 'S')) ps = aux2i(a);
      if (const uint8_t* a = aux_get(r, 'r', 'q')) rq = aux2f(a);
      if (!(rq >= opts->read_quality)) return;
      otg_read o;
      memset(&o, 0, sizeof(o));
      o.seq_off = arena.size(); o.seq_len = (uint32_t)seq.size();
      o.spanning_l = msg.spanning_l ? 1 : 0; o.spanning_r = msg.spanning_r ? 1 : 0;
      o.ps = ps; o.hp = hp; o.ccoord_first = msg.c_first; o.ccoord_second = msg.c_second;
      arena.insert(arena.end(), seq.begin(), seq.end());
      reads.push_back(o);
      if (meta) {                                                              // ANREAD::name / ANREAD::rq for the reads-only records
        otg_read_meta mm;
        memset(&mm, 0, sizeof(mm));
        mm.name_off = names->size(); mm.name_len = r.l_name; mm.rq = rq;
        names->append(r.name, r.l_name);
        meta->push_back(mm);
      }
    });
    regions[g].n_reads = (uint32_t)reads.size() - first;
  }
  local.fp.close();
  return rc_out;
}

// SampleIndex::init / _init (src/anbamdb.cpp:10-63) on the header text
static int parse_sample_index(otg_bam* b, std::string& err)
{
  if (b->samples_parsed) return OTG_OK;
  b->index2sample.clear(); b->sample2index.clear(); b->offset_l = 1; b->offset_r = 0;
  bool bad = false;
  auto one = [&](const std::string& line) {
    if (line.substr(0, 2) == "RG") {
      if (line.size() > 3 && line.substr(3, 2) == "ID") b->index2sample.emplace_back(line.size() >= 6 ? line.substr(6) : std::string());
    } else if (line.substr(0, 2) == "PG") {
      if (line.size() >= 15 && line.substr(0, 15) == "PG\tID:otter\tOF:") {
        const std::string input = line.substr(15);
        std::vector<std::string> columns;
        size_t i = 0;
        while (i < input.size()) { size_t j = input.find(',', i); if (j == std::string::npos) { columns.push_back(input.substr(i)); break; } columns.push_back(input.substr(i, j - i)); i = j + 1; }
        auto to_i = [&](const std::string& v, int* o) { char* e = nullptr; errno = 0; const long x = strtol(v.c_str(), &e, 10); if (e == v.c_str() || errno == ERANGE || x > 2147483647L || x < -2147483648L) { bad = true; return; } *o = (int)x; };
        if (columns.size() == 1) { to_i(columns[0], &b->offset_l); b->offset_r = b->offset_l; }
        else if (columns.size() == 2) { to_i(columns[0], &b->offset_l); to_i(columns[1], &b->offset_r); }
        else bad = true;
      }
    }
  };
  std::string tag;
  for (char c : b->text) {
    if (c != '@' && c != '\n') tag += c;
    else if (!tag.empty()) { one(tag); tag.clear(); }
  }
  if (!tag.empty()) one(tag);
  if (bad) { err = "cannot parse the offsets in the @PG ID:otter OF: header line"; return OTG_ERR_ARG; }
  if (b->index2sample.empty()) { err = "no sample name (@RG ID:) in the BAM header"; return OTG_ERR_ARG; }
  for (int i = 0; i < (int)b->index2sample.size(); ++i) b->sample2index[b->index2sample[i]] = i;
  b->samples_parsed = true;
  return OTG_OK;
}

// parse_analleles / parse_anallele (src/anseqs.cpp:462-524) for a slice of regions, plus the reference allele genotype_process appends
// (src/genotype.cpp:92-101).  alleles[].region = region index, alleles[].label = sample index.
static int alleles_slice(const otg_bam* b, const char* path, const otg_bed* beds, const char* chr_arena, uint32_t g0, uint32_t g1,
                         const otg_fasta* fa, std::vector<otg_allele>& alleles, std::vector<uint8_t>& arena, uint32_t* n_per_region, std::string& err)
{
  static const char nt16[] = "=ACMGRSVTWYHKDBN";
  otg_bam local;
  if (!local.fp.open(path)) { err = "cannot reopen BAM"; return OTG_ERR_ARG; }
  std::vector<uint32_t> bins;
  std::vector<std::pair<uint64_t, uint64_t>> chunks;
  int rc_out = OTG_OK;
  const int refindex = (int)b->index2sample.size();
  for (uint32_t g = g0; g < g1 && rc_out == OTG_OK; ++g) {
    const size_t first = alleles.size();
    n_per_region[g] = 0;
    const std::string chr(chr_arena + beds[g].chr_off, beds[g].chr_len);
    const std::string target = chr + ":" + std::to_string((uint32_t)beds[g].start) + "-" + std::to_string((uint32_t)beds[g].end);   // toScString
    auto it = b->name2id.find(chr);
    // the region string goes through hts_parse_reg: numbers are parsed into int (the unsigned spelling of a wrapped value overflows to garbage
    // there; such regions are not supported here), the start is 1-based
    const long long s0 = (long long)(uint32_t)beds[g].start, e0 = (long long)(uint32_t)beds[g].end;
    long long qbeg = s0 - 1; if (qbeg < 0) qbeg = 0;
    const long long qend = e0;
    if (it == b->name2id.end() || s0 > 2147483647LL || e0 > 2147483647LL || qbeg > qend) continue;
    int cb_rc = OTG_OK;
    const int sc_rc = scan_region(b, local, it->second, qbeg, qend, bins, chunks, err, [&](const Rec& r) {
      if (cb_rc != OTG_OK) return;
      const uint8_t* a = aux_get(r, 't', 'a');
      std::string parsed = aux2str(a);
      if (parsed != target) return;
      a = aux_get(r, 'R', 'G');
      const std::string sample = aux2str(a);
      auto si = b->sample2index.find(sample);
      if (si == b->sample2index.end()) { err = "unrecognized sample name (read group): " + sample; cb_rc = OTG_ERR_ARG; return; }
      otg_allele o;
      memset(&o, 0, sizeof(o));
      o.tcov = 1; o.acov = 1; o.scov = 1; o.ps = -1; o.hp = -1; o.se = 0.0f; o.ic = 1;
      if ((a = aux_get(r, 't', 'c'))) o.tcov = aux2i(a);
      if ((a = aux_get(r, 'a', 'c'))) o.acov = aux2i(a);
      if ((a = aux_get(r, 's', 'c'))) o.scov = aux2i(a);
      if ((a = aux_get(r, 'P', 'S'))) o.ps = aux2i(a);
      if ((a = aux_get(r, 'H', 'P'))) o.hp = aux2i(a);
      if ((a = aux_get(r, 's', 'e'))) o.se = (float)aux2f(a);
      if ((a = aux_get(r, 'i', 'c'))) o.ic = aux2i(a);
      const uint32_t l_qseq = r.l_seq > 0 ? (uint32_t)r.l_seq : 1u;
      o.seq_off = arena.size(); o.seq_len = l_qseq; o.region = g; o.label = si->second;
      const size_t at = arena.size();
      arena.resize(at + l_qseq, (uint8_t)'N');
      for (int i = 0; i < r.l_seq; ++i) arena[at + (size_t)i] = (uint8_t)nt16[(r.seq[i >> 1] >> ((~i & 1) << 2)) & 0xf];
      alleles.push_back(o);
    });
    rc_out = sc_rc != OTG_OK ? sc_rc : cb_rc;
    if (rc_out != OTG_OK) break;
    if (fa && alleles.size() > first) {
      // anallele_block.emplace_back(refseq): ANALLELE(seq) = coverage 1/1/1, se 0, ic 1, no haplotag; sample = the internal reference sample
      const char* ref = nullptr; uint64_t ref_len = 0;
      std::vector<char> buf;
      const int fb = (int)(uint32_t)beds[g].start - b->offset_l, fe = (int)(uint32_t)beds[g].end + b->offset_r - 1;
      uint64_t need = 0;
      buf.resize((size_t)(fe >= fb ? (long long)fe - fb + 2 : 2) + 16);
      if (otg_fasta_fetch(fa, chr.data(), (uint32_t)chr.size(), fb, fe, buf.data(), buf.size(), &need) != OTG_OK) {
        buf.resize((size_t)need + 16);
        if (otg_fasta_fetch(fa, chr.data(), (uint32_t)chr.size(), fb, fe, buf.data(), buf.size(), &need) != OTG_OK) { err = "cannot fetch the reference allele"; rc_out = OTG_ERR_ARG; break; }
      }
      ref = buf.data(); ref_len = need;
      otg_allele o;
      memset(&o, 0, sizeof(o));
      o.tcov = 1; o.acov = 1; o.scov = 1; o.ps = -1; o.hp = -1; o.se = 0.0f; o.ic = 1;
      o.seq_off = arena.size(); o.seq_len = (uint32_t)ref_len; o.region = g; o.label = refindex;
      arena.insert(arena.end(), (const uint8_t*)ref, (const uint8_t*)ref + ref_len);
      alleles.push_back(o);
    }
    n_per_region[g] = (uint32_t)(alleles.size() - first);
  }
  local.fp.close();
  return rc_out;
}

extern "C" {

int otg_ingest_regions_named(otg_bam* b, const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const otg_ingest_opts* opts,
                             uint8_t* arena, uint64_t arena_capacity, uint64_t* arena_used, otg_read* reads, uint32_t reads_capacity,
                             uint32_t* n_reads, otg_region* regions, otg_read_meta* meta, char* name_arena, uint64_t name_capacity,
                             uint64_t* name_used)
{
  if (!b || (n_regions && (!beds || !regions)) || !opts || !arena_used || !n_reads) return otg_fail(nullptr, OTG_ERR_ARG, "otg_ingest_regions: null argument");
  const bool want_meta = meta != nullptr;
  if (want_meta && !name_used) return otg_fail(nullptr, OTG_ERR_ARG, "otg_ingest_regions_named: name_used is null");
  // regions are independent: contiguous slices on separate threads (each with its own file handle, as the reference's pool
  // threads have, src/assemble.cpp:45-46), merged in region order afterwards
  uint32_t T = opts->threads > 1 ? (uint32_t)opts->threads : 1u;
  if (T > n_regions) T = n_regions ? n_regions : 1;
  std::vector<std::vector<otg_read>> R(T);
  std::vector<std::vector<uint8_t>> A(T);
  std::vector<std::vector<otg_read_meta>> M(T);
  std::vector<std::string> N(T);
  std::vector<int> rcs(T, OTG_OK);
  std::vector<std::string> errs(T);
  auto work = [&](uint32_t t) {
    const uint32_t g0 = (uint32_t)((uint64_t)n_regions * t / T), g1 = (uint32_t)((uint64_t)n_regions * (t + 1) / T);
    try {
      rcs[t] = ingest_slice(b, b->path.c_str(), beds, chr_arena, g0, g1, opts, R[t], A[t], regions, errs[t],
                            want_meta ? &M[t] : nullptr, want_meta ? &N[t] : nullptr);
    } catch (const std::exception& e) { rcs[t] = OTG_ERR_ARG; errs[t] = std::string("exception while reading the BAM: ") + e.what(); }   // nothing may unwind through the C ABI / a std::thread
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (uint32_t t = 0; t < T; ++t) if (rcs[t] != OTG_OK) return otg_fail(nullptr, rcs[t], "otg_ingest_regions: %s", errs[t].c_str());
  uint64_t used = *arena_used; uint32_t nr = *n_reads;
  uint64_t nused = want_meta ? *name_used : 0;
  bool overflow = false;
  for (uint32_t t = 0; t < T; ++t) {
    const uint32_t g0 = (uint32_t)((uint64_t)n_regions * t / T), g1 = (uint32_t)((uint64_t)n_regions * (t + 1) / T);
    for (uint32_t g = g0; g < g1; ++g) regions[g].first_read += nr;
    const bool fits = (uint64_t)nr + R[t].size() <= reads_capacity && used + A[t].size() + 64 <= arena_capacity &&
                      (!want_meta || (name_arena && nused + N[t].size() <= name_capacity));
    if (fits && !overflow) {
      for (size_t i = 0; i < R[t].size(); ++i) { reads[nr + i] = R[t][i]; reads[nr + i].seq_off += used; }
      if (!A[t].empty()) memcpy(arena + used, A[t].data(), A[t].size());
      if (want_meta) {
        for (size_t i = 0; i < M[t].size(); ++i) { meta[nr + i] = M[t][i]; meta[nr + i].name_off += nused; }
        if (!N[t].empty()) memcpy(name_arena + nused, N[t].data(), N[t].size());
      }
    } else overflow = true;
    nr += (uint32_t)R[t].size(); used += A[t].size(); nused += N[t].size();
  }
  *arena_used = used; *n_reads = nr;
  if (want_meta) *name_used = nused;
  if (getenv("OTG_DEBUG")) fprintf(stderr, "[otg] ingest: %llu BGZF blocks inflated so far, %u reads kept in this call\n", g_blocks_inflated.load(), nr);
  return overflow ? OTG_ERR_CAPACITY : OTG_OK;
}

int otg_ingest_regions(otg_bam* b, const otg_bed* beds, const char* chr_arena, uint32_t n_regions, const otg_ingest_opts* opts,
                       uint8_t* arena, uint64_t arena_capacity, uint64_t* arena_used, otg_read* reads, uint32_t reads_capacity,
                       uint32_t* n_reads, otg_region* regions)
{
  return otg_ingest_regions_named(b, beds, chr_arena, n_regions, opts, arena, arena_capacity, arena_used, reads, reads_capacity,
                                  n_reads, regions, nullptr, nullptr, 0, nullptr);
}

int otg_bam_sample_index(otg_bam* b, uint32_t* n_samples, int32_t* offset_l, int32_t* offset_r)
{
  if (!b) return otg_fail(nullptr, OTG_ERR_ARG, "otg_bam_sample_index: null argument");
  std::string err;
  const int rc = parse_sample_index(b, err);
  if (rc != OTG_OK) return otg_fail(nullptr, rc, "otg_bam_sample_index(%s): %s", b->path.c_str(), err.c_str());
  if (n_samples) *n_samples = (uint32_t)b->index2sample.size();
  if (offset_l) *offset_l = b->offset_l;
  if (offset_r) *offset_r = b->offset_r;
  return OTG_OK;
}

const char* otg_bam_sample(const otg_bam* b, uint32_t i)
{
  if (!b || !b->samples_parsed || i >= b->index2sample.size()) return nullptr;
  return b->index2sample[i].c_str();
}

int otg_ingest_alleles(otg_bam* b, const otg_bed* beds, const char* chr_arena, uint32_t n_regions, int32_t threads, const otg_fasta* fa,
                       uint8_t* arena, uint64_t arena_capacity, uint64_t* arena_used, otg_allele* alleles, uint32_t alleles_capacity,
                       uint32_t* n_alleles, uint32_t* first_allele)
{
  if (!b || (n_regions && (!beds || !first_allele)) || !arena_used || !n_alleles) return otg_fail(nullptr, OTG_ERR_ARG, "otg_ingest_alleles: null argument");
  {
    std::string err;
    const int rc = parse_sample_index(b, err);
    if (rc != OTG_OK) return otg_fail(nullptr, rc, "otg_ingest_alleles(%s): %s", b->path.c_str(), err.c_str());
  }
  uint32_t T = threads > 1 ? (uint32_t)threads : 1u;
  if (T > n_regions) T = n_regions ? n_regions : 1;
  std::vector<std::vector<otg_allele>> R(T);
  std::vector<std::vector<uint8_t>> A(T);
  std::vector<uint32_t> per(n_regions, 0u);
  std::vector<int> rcs(T, OTG_OK);
  std::vector<std::string> errs(T);
  auto work = [&](uint32_t t) {
    const uint32_t g0 = (uint32_t)((uint64_t)n_regions * t / T), g1 = (uint32_t)((uint64_t)n_regions * (t + 1) / T);
    try { rcs[t] = alleles_slice(b, b->path.c_str(), beds, chr_arena, g0, g1, fa, R[t], A[t], per.data(), errs[t]); }
    catch (const std::exception& e) { rcs[t] = OTG_ERR_ARG; errs[t] = std::string("exception while reading the BAM: ") + e.what(); }
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (uint32_t t = 0; t < T; ++t) if (rcs[t] != OTG_OK) return otg_fail(nullptr, rcs[t], "otg_ingest_alleles: %s", errs[t].c_str());
  uint64_t used = *arena_used; uint32_t na = *n_alleles;
  bool overflow = false;
  for (uint32_t t = 0; t < T; ++t) {
    const bool fits = (uint64_t)na + R[t].size() <= alleles_capacity && used + A[t].size() + 64 <= arena_capacity && alleles && arena;
    if (fits && !overflow) {
      for (size_t i = 0; i < R[t].size(); ++i) { alleles[na + i] = R[t][i]; alleles[na + i].seq_off += used; }
      if (!A[t].empty()) memcpy(arena + used, A[t].data(), A[t].size());
    } else overflow = true;
    na += (uint32_t)R[t].size(); used += A[t].size();
  }
  uint32_t acc = *n_alleles;
  for (uint32_t g = 0; g < n_regions; ++g) { first_allele[g] = acc; acc += per[g]; }
  first_allele[n_regions] = acc;
  *arena_used = used; *n_alleles = na;
  return overflow ? OTG_ERR_CAPACITY : OTG_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// `otter wgat`: slice the BED regions out of whole-genome assembly alignments — wga_bam_genotyper_process
// (src/wgat.cpp:31-124) with get_op_intervals (src/opinterval.cpp:12-34).  Host code, no device work.
//
// The reference finds overlaps with the interval tree vendored in its src/interval_tree.h (E. Garrison's intervaltree, MIT licence:
// "Copyright (c) 2011 Erik Garrison").  The ORDER in which that tree reports overlaps decides the order of the emitted records and, through an
// unstable std::sort on equal (start, stop) keys, which CIGAR operation counts as the first of a region — so the tree's construction
// (centre = (min start + max stop) / 2, leaf buckets below 64 intervals, depth 16) and visit order (node, left, right) are restated
// here; std::sort is libstdc++'s on both sides.
namespace {

struct Iv { int start, stop, value; };

struct ITree {
  std::vector<Iv> ivs;
  std::unique_ptr<ITree> left, right;
  int center = 0;
  ITree() = default;
  ITree(std::vector<Iv>&& v, size_t depth = 16, size_t minbucket = 64, size_t maxbucket = 512, int leftextent = 0, int rightextent = 0)
  {
    --depth;
    auto by_start = [](const Iv& a, const Iv& b) { return a.start < b.start; };
    auto by_stop = [](const Iv& a, const Iv& b) { return a.stop < b.stop; };
    if (!v.empty()) {
      const auto mm_stop = std::minmax_element(v.begin(), v.end(), by_stop);
      const auto mm_start = std::minmax_element(v.begin(), v.end(), by_start);
      center = (mm_start.first->start + mm_stop.second->stop) / 2;
    }
    if (leftextent == 0 && rightextent == 0) std::sort(v.begin(), v.end(), by_start);
    if (depth == 0 || (v.size() < minbucket && v.size() < maxbucket)) {
      std::sort(v.begin(), v.end(), by_start);
      ivs = std::move(v);
      return;
    }
    int leftp, rightp;
    if (leftextent || rightextent) { leftp = leftextent; rightp = rightextent; }
    else { leftp = v.front().start; rightp = std::max_element(v.begin(), v.end(), by_stop)->stop; }
    std::vector<Iv> lefts, rights;
    for (const Iv& x : v) {
      if (x.stop < center) lefts.push_back(x);
      else if (x.start > center) rights.push_back(x);
      else ivs.push_back(x);
    }
    if (!lefts.empty()) left.reset(new ITree(std::move(lefts), depth, minbucket, maxbucket, leftp, center));
    if (!rights.empty()) right.reset(new ITree(std::move(rights), depth, minbucket, maxbucket, center, rightp));
  }
  void overlapping(int start, int stop, std::vector<Iv>& out) const
  {
    if (!ivs.empty() && !(stop < ivs.front().start)) for (const Iv& x : ivs) if (x.stop >= start && x.start <= stop) out.push_back(x);
    if (left && start <= center) left->overlapping(start, stop, out);
    if (right && stop >= center) right->overlapping(start, stop, out);
  }
};

struct SinkW {            // buffered writer over the caller's callback
  otg_write_fn write; void* user; std::string buf; bool failed = false; uint64_t total = 0;
  void put(const std::string& s) { buf += s; if (buf.size() > (1u << 20)) flush(); }
  void flush() { if (!buf.empty() && !failed) { if (write(user, buf.data(), buf.size()) != 0) failed = true; total += buf.size(); } buf.clear(); }
};

} // namespace

extern "C" int otg_wgat(otg_bam* b, const otg_bed* beds, const char* chr_arena, uint32_t n_beds, const char* read_group, int is_fasta,
                        int32_t offset_l, int32_t offset_r, otg_write_fn write, void* user, uint64_t* n_records)
{
  if (!b || (n_beds && (!beds || !chr_arena)) || !write) return otg_fail(nullptr, OTG_ERR_ARG, "otg_wgat: null argument");
  const std::string rg = read_group ? read_group : "";
  SinkW out{write, user};
  if (!is_fasta) {        // src/wgat.cpp:166-174
    for (size_t i = 0; i < b->names.size(); ++i) out.put("@SQ\tSN:" + b->names[i] + "\tLN:" + std::to_string(b->lengths[i]) + "\n");
    out.put("@RG\tID:" + rg + "\n");
    out.put("@PG\tID:otter\tOF:" + std::to_string((uint32_t)offset_l) + "," + std::to_string((uint32_t)offset_r) + "\n");
  }
  // construct_bed_interval_tree (src/wgat.cpp:19-29): every region of every chromosome in ONE tree, extended by the offsets (uint32 arithmetic)
  std::vector<Iv> bed_iv;
  std::vector<std::string> bed_chr(n_beds), bed_sc(n_beds);
  for (uint32_t i = 0; i < n_beds; ++i) {
    const int start = (int)((uint32_t)beds[i].start - (uint32_t)offset_l), end = (int)((uint32_t)beds[i].end + (uint32_t)offset_r);
    bed_iv.push_back({std::min(start, end), std::max(start, end), (int)i});
    bed_chr[i].assign(chr_arena + beds[i].chr_off, beds[i].chr_len);
    bed_sc[i] = bed_chr[i] + ":" + std::to_string((uint32_t)beds[i].start) + "-" + std::to_string((uint32_t)beds[i].end);       // BED::toScString
  }
  const ITree bed_tree(std::move(bed_iv));
  static const char nt16[] = "=ACMGRSVTWYHKDBN";
  otg_bam local;
  if (!local.fp.open(b->path.c_str())) return otg_fail(nullptr, OTG_ERR_ARG, "otg_wgat: cannot reopen %s", b->path.c_str());
  std::vector<uint32_t> bins;
  std::vector<std::pair<uint64_t, uint64_t>> chunks;
  std::string err;
  uint64_t nrec = 0;
  int rc_all = OTG_OK;
  for (size_t tid = 0; tid < b->names.size() && rc_all == OTG_OK; ++tid) {
    int alignment_index = 0;
    std::vector<Iv> hits, ops_hit;
    // bam_itr_querys("chr:1-len"): every alignment that overlaps [0, len)
    const int rc = scan_region(b, local, (int)tid, 0, (long long)b->lengths[tid], bins, chunks, err, [&](const Rec& r) {
      if (r.l_seq <= 0) return;
      const uint8_t* cg = r.cigar;
      const int ref_end = r.pos + cigar_rlen(r);
      hits.clear();
      bed_tree.overlapping(r.pos, ref_end, hits);
      size_t keep = 0;
      for (const Iv& h : hits) if (bed_chr[(size_t)h.value] == b->names[tid]) hits[keep++] = h;
      hits.resize(keep);
      if (!hits.empty()) {
        const std::string name(r.name, r.l_name);
        // get_op_intervals (src/opinterval.cpp:12-34): one reference / query interval per CIGAR operation
        std::vector<Iv> op_iv; std::vector<int> q_start, q_end, q_op;
        int rpos = r.pos, qpos = 0;
        for (uint32_t i = 0; i < r.n_cigar; ++i) {
          const int op = cig_op(cg, i), ol = cig_len(cg, i);
          int rn = rpos, qn = qpos;
          if (op == 4) qn += ol;
          else if (op == 0 || op == 7 || op == 8) { rn += ol; qn += ol; }
          else if (op == 1) qn += ol;
          else if (op == 2) rn += ol;
          op_iv.push_back({rpos, rn, (int)i}); q_start.push_back(qpos); q_end.push_back(qn); q_op.push_back(op);
          rpos = rn; qpos = qn;
        }
        const ITree op_tree(std::move(op_iv));
        for (const Iv& ov : hits) {
          ops_hit.clear();
          op_tree.overlapping(ov.start, ov.stop, ops_hit);
          std::sort(ops_hit.begin(), ops_hit.end(), [](const Iv& x, const Iv& y) { if (x.start == y.start) return x.stop < y.stop; else return x.start < y.start; });
          bool clipped_l = false, clipped_r = false;
          int query_start = 0, query_end = 0;
          for (int i = 0; i < (int)ops_hit.size(); ++i) {
            const Iv& o = ops_hit[(size_t)i];
            const int qs = q_start[(size_t)o.value], qe = q_end[(size_t)o.value], op = q_op[(size_t)o.value];
            if (op == 4 || op == 5) {
              if (i == 0) { clipped_l = true; query_start = qe; }
              else { clipped_r = true; query_end = qs; }
            } else {
              if (i == 0) {
                if (op == 2) { if (o.start <= ov.start && o.stop >= ov.stop) break; else query_start = qs; }
                else query_start = qs + (ov.start - o.start);
              }
              if (i + 1 == (int)ops_hit.size()) {
                if (op == 2) query_end = qe;
                else query_end = qe - (o.stop - ov.stop);
              }
            }
          }
          if (clipped_l || clipped_r) continue;            // "[WARNING] skipping non-spanning whole-genome alignment"
          const long long len = (long long)query_end - query_start;
          // An alignment that begins or ends INSIDE the interval without a clip gives the reference a query interval outside the sequence
          // (negative start / end past l_seq): it then reads bytes before or behind the packed sequence — undefined output.  Such an alignment
          // does not span the region; it is skipped here like the clipped ones.
          if (len < 0 || query_start < 0 || query_end > r.l_seq) continue;
          std::string seq((size_t)(len == 0 ? 1 : len), 'N');
          for (int i = query_start; i < query_end; ++i) seq[(size_t)(i - query_start)] = nt16[(r.seq[i >> 1] >> ((~i & 1) << 2)) & 0xf];
          const otg_bed& lb = beds[(size_t)ov.value];
          const std::string& sc = bed_sc[(size_t)ov.value];
          std::string line;
          if (is_fasta) {
            // stdout_fa(read_group, name#region#index, is_read, 1, 1): tc / ac / sc = 1, sp:A:b (src/anseqs.cpp:58-64)
            line = ">" + rg + "#" + name + "#" + sc + "#" + std::to_string(alignment_index) + "#tc:i:1#ac:i:1#sc:i:1#sp:A:b\n" + seq + "\n";
          } else {
            const std::string chr = bed_chr[(size_t)ov.value];
            line = name + "#" + sc + "_" + std::to_string(alignment_index) + "\t0\t" + chr + "\t" + std::to_string(lb.start) + "\t0\t" + std::to_string(seq.size()) + "M\t*\t0\t0\t" +
                   seq + "\t" + std::string(seq.size(), '!');
            if (!rg.empty()) line += "\tRG:Z:" + rg;
            line += "\tta:Z:" + chr + ":" + std::to_string(lb.start) + "-" + std::to_string(lb.end) + "\ttc:i:1\tac:i:1\tsc:i:1\tsp:A:b\tic:i:1\tse:f:0\n";
          }
          out.put(line);
          ++nrec;
        }
      }
      ++alignment_index;
    });
    if (rc != OTG_OK) { rc_all = rc; break; }
  }
  local.fp.close();
  out.flush();
  if (n_records) *n_records = nrec;
  if (rc_all != OTG_OK) return otg_fail(nullptr, rc_all, "otg_wgat: %s", err.c_str());
  if (out.failed) return otg_fail(nullptr, OTG_ERR_ARG, "otg_wgat: the writer failed");
  return OTG_OK;
}
